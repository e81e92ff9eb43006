cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so  # the knobs below exist only in the diagnostic build
for cfg in "4 4 0" "4 4 1" "4 4 4" "8 2 0" "8 2 1"; do
  set -- $cfg
  export ARTMIPS_NW=$1 ARTMIPS_MAX_NQ=$2 ARTMIPS_ABLATE=$3
  d=$R/gpurun_out/clk_$1_$2_$3
  timeout -s KILL 120 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d $d -o r --output-format csv -- python3 $R/tools/run_search.py 21015324 1024 100 > $d.log 2>&1 || exit 1
  tail -1 $d.log
done
