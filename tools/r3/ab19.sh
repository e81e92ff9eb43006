# round 3, call 19: where does 8 x 2 take over from 4 x 4?  k = 200 / 300 / 500 / 1000, default dispatch (8 x 2 above
# k = 128) against 4 x 4 forced (ARTMIPS_NW=44), alternating processes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r19
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
for rep in 1 2; do
for K in 200 300 500 1000; do
  echo "== k $K 8x2 (default)" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 $K f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  echo "== k $K 4x4 forced" | tee -a $O/ab.txt
  ARTMIPS_NW=44 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 $K f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
done
done
