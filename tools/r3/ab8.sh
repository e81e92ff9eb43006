# wave-flushed staging (STG=2) in the 4 x 4 gang shape at top-1000: parity, then against direct list writes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5h
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_search_gpu.py tests/test_search2_gpu.py tests/test_search3_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
for rep in 1 2; do
  echo "== 8x2 direct | 4x4 coop staging | 4x4 direct (k1000), then k100 4x4" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
  ARTMIPS_NW=44 GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
  ARTMIPS_NW=44 ARTMIPS_COOP_STAGING=0 GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
  ARTMIPS_COOP_STAGING=0 GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
done
ARTMIPS_NW=44 timeout -k 10 300 python -u tools/refine_check.py > $O/refine_check.txt 2>&1 || true
tail -5 $O/refine_check.txt
