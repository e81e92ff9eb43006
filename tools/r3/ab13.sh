# round 3, call 13: cache policy bits on the gang kernels' refill DMA (default | sc1 | sc0 | sc0 sc1), alternating
# processes; and how many candidates the filter keeps
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r13
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
for rep in 1 2; do
for L in "" _sc1 _sc0 _sc0sc1; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag$L.so
  echo "== diag$L k1000 (8x2) | k100 (4x4)" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "rows\|ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "rows\|ablate" | tee -a $O/ab.txt
done
done
