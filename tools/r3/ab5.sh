# round 3, call 5: build variants against each other, alternating processes on one box
# v0 = grouped DMA + one-branch filter (three-instruction asm pieces) + first tile walker (commit 8fe0c65)
# v1 = one-statement filter pre-check, tile formula on the slow path only, simpler walker
# v2 = v1 + row-fragment reads of chunk C+2 between the MFMAs of chunk C
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5e
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1
for rep in 1 2; do
for L in r02 v0 v1 v2; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag_$L.so
  echo "== $L k100" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
  echo "== $L k1000" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
done
done
for L in r02 v0 v1 v2; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag_$L.so
  echo "== $L q128 k32 / q300 k100" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 128 32 f16 9 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
  GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 300 100 f16 9 base=0 2>&1 | grep "base" | tee -a $O/ab.txt
done
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_stamps.so
ARTMIPS_STAMPS=1 timeout -k 10 200 python -u tools/run_search.py 21015324 2048 100 f16 > $O/stamps_k100.txt 2>&1 && grep "stamps" $O/stamps_k100.txt | tail -4
