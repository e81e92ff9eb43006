# round 3, call 23: the SHIPPED builds (no diagnostics code) before and after the slow-path work, alternating processes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r23
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
for rep in 1 2; do
for L in head new; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_ship_$L.so
  echo "== shipped build $L: k1000 | k300 | k100 | k32 | 300 q k1000 | headline" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 300 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 32 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 300 1000 bf16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 64 32 f16 15 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
done
done
