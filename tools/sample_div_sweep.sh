# Step time against the pass-0 sample divisor (diagnostic build, ARTMIPS_SAMPLE_DIV) on one shard size.
# usage: bash tools/sample_div_sweep.sh ROWS "DIV DIV ..." [steps] [topk] [queries]   (DIV 0 = the fitted value)
R=$GRAFT_REPO_ROOT
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
for rep in 1 2; do
for D in $2; do
  if [ $D = 0 ]; then unset ARTMIPS_SAMPLE_DIV; else export ARTMIPS_SAMPLE_DIV=$D; fi
  timeout -k 10 200 python3 $R/bench.py --rows $1 --topk ${4:-32} --queries ${5:-64} --steps ${3:-300} --warmup 20 --no-cpu-baseline --settle-seconds 1 > /tmp/sd.json 2> /tmp/sd.err || exit 1
  python3 -c "
import json; d=json.load(open('/tmp/sd.json')); print('rows $1 k ${4:-32} q ${5:-64} sample_div', '$D', 'ms/step %.4f' % d['ms_per_step'], 'scan %.4f' % d['roofline']['kernel_ms_avg'], 'cand/query %.0f' % d['candidates_per_query'])"
done
done
