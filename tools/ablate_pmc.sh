# Clock, MFMA busy and wave states of the large-batch scan (gang of 4, 1,024 queries x top-100 over 21M rows)
# with parts of the loop switched off (ARTMIPS_ABLATE bits: 1 no refill DMA, 2 no barrier, 8 no LDS reads,
# 16 no filter, 32 reads never waited for).  usage: bash tools/ablate_pmc.sh [outdir-name] "0 8 16 ..."
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-ablate_pmc}
mkdir -p $O
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
dirs=""
for A in ${2:-0 1 8 16 24 27}; do
  export ARTMIPS_ABLATE=$A
  timeout -s KILL 150 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS --kernel-trace -d $O/a$A -o r --output-format csv -- python3 $R/tools/run_search.py 21015324 ${3:-1024} ${4:-100} > $O/a$A.log 2>&1 || exit 1
  dirs="$dirs $O/a$A"
done
python3 $R/tools/pmc_summary.py $dirs > $O/summary.json
python3 - $O/summary.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for k, v in d.items():
    print(k, "ms %.2f clock %.3f GHz mfma_busy %.3f wait_any %.3f wait_inst %.3f active %.3f" % (
        v["launch_ms"], v.get("clock_ghz", 0), v.get("mfma_busy_frac_of_simd_cycles", 0), v.get("sq_wait_any_frac", 0),
        v.get("sq_wait_inst_any_frac", 0), v.get("sq_active_inst_any_frac", 0)),
        "lds_wait_inst %.3f lds_active %.3f" % (v["counters_per_launch"].get("SQ_WAIT_INST_LDS", 0) / v["counters_per_launch"]["SQ_WAVE_CYCLES"],
                                                v["counters_per_launch"].get("SQ_ACTIVE_INST_LDS", 0) / v["counters_per_launch"]["SQ_WAVE_CYCLES"]))
PY
