# round 3, call 21: leaner survivor slow path, step 2 (branch-free tile formula, a full list takes entries in its last slot) against step 1
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r21d
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
for rep in 1 2; do
for L in _old ""; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag$L.so
  echo "== diag$L k1000 | k300 | k100 | 300 q k1000" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 300 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 300 1000 bf16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
done
done
