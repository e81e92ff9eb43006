# lean survivor slow path (v3) against the build before it (v1) and the round-2 build; parity first
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5g
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_search_gpu.py tests/test_search2_gpu.py tests/test_search3_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
export ARTMIPS_DIAGNOSTICS=1
for rep in 1 2; do
for L in libartmips_diag libartmips_diag_v1 libartmips_diag_r02; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/$L.so
  echo "== $L k100 | k1000 (8x2) | k1000 as 4x4" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
  ARTMIPS_NW=44 GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
done
done
