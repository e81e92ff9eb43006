# round 3, call 14: phased 8-wave loop, refill pieces inside the lone-wave MFMA stretches (diag) against the
# bursts at the two ends of a stage (diag_burst), alternating processes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r14b
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
for rep in 1 2; do
for L in _burst ""; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag$L.so
  echo "== diag$L k1000 (8x2) | k100 forced to 8x2 | Q 300 k1000 (8x2, no gang)" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  ARTMIPS_NW=88 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 300 1000 bf16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
done
done
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_stamps.so
ARTMIPS_STAMPS=1 timeout -k 10 200 python -u tools/run_search.py 21015324 2048 1000 bf16 > $O/stamps_k1000.txt 2>&1 && grep "stamps\|scan" $O/stamps_k1000.txt | tail -9
