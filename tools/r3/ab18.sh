# round 3, call 18: gang shape at small k once more, now with the rendezvous in both shapes: default dispatch (4 x 4 for
# k <= 128) against 8 x 2 forced (ARTMIPS_NW=88), alternating processes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r18b
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
for rep in 1 2; do
for Q in 3610 2048 1024; do
for K in 100 32; do
  echo "== Q $Q k $K 4x4" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 $Q $K f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  echo "== Q $Q k $K 8x2" | tee -a $O/ab.txt
  ARTMIPS_NW=88 timeout -k 10 200 python -u tools/gang_ab.py 21015324 $Q $K f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
done
done
done
