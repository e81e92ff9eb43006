# round 3: do the members of a gang drift apart because of their survivors?  L2 hit rate and HBM fetch bytes of the
# 2,048 x top-1000 gang scan as shipped and with the threshold at +inf (no survivor, nothing else changed)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/drift_pmc
mkdir -p $O
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
for A in 0 2048; do
  export ARTMIPS_ABLATE=$A
  timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/l2_a$A -o r --output-format csv -- python3 $R/tools/run_search.py 21015324 2048 1000 bf16 > $O/l2_a$A.log 2>&1 || exit 1
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_a$A -o r --output-format csv -- python3 $R/tools/run_search.py 21015324 2048 1000 bf16 > $O/fetch_a$A.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $O/l2_a0 $O/l2_a2048 $O/fetch_a0 $O/fetch_a2048 > $O/summary.json
cat $O/summary.json
