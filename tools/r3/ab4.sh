# round 3, call 4: parity on the shipped build, stamps, then this build against the round-2 build
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5d
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_search_gpu.py tests/test_search2_gpu.py tests/test_search3_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
ARTMIPS_STAMPS=1 timeout -k 10 200 python -u tools/run_search.py 21015324 2048 100 f16 > $O/stamps_k100.txt 2>&1 && grep "stamps" $O/stamps_k100.txt | tail -4
ARTMIPS_STAMPS=1 timeout -k 10 200 python -u tools/run_search.py 21015324 2048 1000 bf16 > $O/stamps_k1000.txt 2>&1 && grep "stamps" $O/stamps_k1000.txt | tail -8
for rep in 1 2; do
for L in libartmips_diag libartmips_diag_r02; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/$L.so
  echo "== $L k100" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
  echo "== $L k1000" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
done
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
echo "== q128 k32, q300 k100 (HBM-bound multi-group shapes), new then r02" | tee -a $O/ab.txt
for L in libartmips_diag libartmips_diag_r02; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/$L.so
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 128 32 f16 9 base=0 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 300 100 f16 9 base=0 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
done
