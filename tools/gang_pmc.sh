# Counters for the large-batch scan with gangs of 1, 2 and 4 workgroups per tile sequence (1,024 queries x
# top-100 over 21,015,324 x 768 fp16 rows): clock + MFMA busy + wave states, HBM fetch bytes, L2 hit rate.
# Separate rocprofv3 passes per counter group.  usage: bash tools/gang_pmc.sh [outdir-name] [Q] [K]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-gang_pmc}
Q=${2:-2048}
K=${3:-100}
mkdir -p $O
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so  # ARTMIPS_GANG exists only in the diagnostic build
for G in 1 2 4 8; do
  export ARTMIPS_GANG=$G
  timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $O/sq_g$G -o r --output-format csv -- python3 $R/tools/run_search.py 21015324 $Q $K > $O/sq_g$G.log 2>&1 || exit 1
  tail -1 $O/sq_g$G.log
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_g$G -o r --output-format csv -- python3 $R/tools/run_search.py 21015324 $Q $K > $O/fetch_g$G.log 2>&1 || exit 1
  timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/l2_g$G -o r --output-format csv -- python3 $R/tools/run_search.py 21015324 $Q $K > $O/l2_g$G.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $O/sq_g1 $O/sq_g2 $O/sq_g4 $O/sq_g8 $O/fetch_g1 $O/fetch_g2 $O/fetch_g4 $O/fetch_g8 $O/l2_g1 $O/l2_g2 $O/l2_g4 $O/l2_g8 > $O/summary.json
cat $O/summary.json
