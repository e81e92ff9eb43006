# round 3, call 29: the constant of the pass-0 sample size fit (0.45, fitted in round 1 when a survivor cost more):
# 0.45 against 0.7 / 1.0 / 1.4 on small shards and on the large-batch configs (diagnostic build, ARTMIPS_SAMPLE_C100)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r29
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
for rep in 1 2; do
for C in 45 70 100 140; do
  export ARTMIPS_SAMPLE_C100=$C
  echo "== c $C: 1/8 shard | 1/2 shard | 8.8M rows 64 q | 16 q top-100 1/8 shard | 2048 q top-1000 | 2048 q top-100 | 2048 q top-1000 8.8M" | tee -a $O/ab.txt
  for ROWS in 2626916 10507662 8841823; do
    timeout -k 10 200 python3 bench.py --rows $ROWS --steps 300 --warmup 20 --no-cpu-baseline --settle-seconds 1 > /tmp/sd.json 2> /tmp/sd.err
    python3 -c "
import json; d=json.load(open('/tmp/sd.json')); print('  rows $ROWS ms/step %.4f scan %.4f cand/query %.0f' % (d['ms_per_step'], d['roofline']['kernel_ms_avg'], d['candidates_per_query']))" | tee -a $O/ab.txt
  done
  timeout -k 10 200 python3 bench.py --rows 2626916 --topk 100 --queries 16 --steps 300 --warmup 20 --no-cpu-baseline --settle-seconds 1 > /tmp/sd.json 2> /tmp/sd.err
  python3 -c "
import json; d=json.load(open('/tmp/sd.json')); print('  16 q top-100 ms/step %.4f cand/query %.0f' % (d['ms_per_step'], d['candidates_per_query']))" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 8841823 2048 1000 bf16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
done
done
