# Prices the phases of select_kernel by difference: the diagnostic build returns after phase N
# (ARTMIPS_SELECT_STOP: 1 offsets, 2 gather, 3 sample radix-select, 4 score filter + id lookup, 5 sort,
# 0 everything) and rocprofv3's kernel stats give the kernel's average duration in each setting.
# usage: bash tools/select_phases.sh [rows] [outdir-name]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ROWS=${1:-2626916}
O=$R/gpurun_out/${2:-select_phases}
mkdir -p $O
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
for S in ${3:-1 2 6 3 4 5 7 0}; do
  export ARTMIPS_SELECT_STOP=$S
  timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/s$S -o r --output-format csv -- python3 $R/bench.py --rows $ROWS --steps 60 --warmup 5 --no-cpu-baseline --settle-seconds 0 > $O/s$S.json 2> $O/s$S.log || { [ $S != 0 ] && echo "(stop $S: bench's own result check fails, as it must)"; } || exit 1
  python3 - $O/s$S/r_kernel_stats.csv $S <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "select_kernel" in n or "kth_kernel" in n or "Li1ELi2ELi1ELi1ELi4E" in n:
        print("stop", sys.argv[2], n[:60], "calls", r["Calls"], "avg us %.2f" % (float(r["AverageNs"]) / 1e3), "min us %.2f" % (float(r["MinNs"]) / 1e3))
PY
done
