# round 3, call 9: the phased 8-wave loop (two waves of a SIMD do their non-MFMA work at opposite ends of a
# stage) against the lock-step one (build -DARTMIPS_PHASED=0), alternating processes on one box
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r9d
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
export ARTMIPS_DIAGNOSTICS=1
for rep in 1 2; do
for L in _p0 ""; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag$L.so
  echo "== diag$L k1000 (8x2 gangs)" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
  echo "== diag$L k100 forced to 8x2 gangs" | tee -a $O/ab.txt
  ARTMIPS_NW=88 GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
done
done
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_stamps.so
ARTMIPS_STAMPS=1 timeout -k 10 200 python -u tools/run_search.py 21015324 2048 1000 bf16 > $O/stamps_k1000.txt 2>&1 && grep "stamps\|scan" $O/stamps_k1000.txt | tail -9
