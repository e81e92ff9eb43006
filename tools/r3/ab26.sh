# round 3, call 26: gang launches that stage their survivors in LDS (few survivors per lane) have no room for the
# rendezvous: staged and unsynchronised (default) against direct list writes + rendezvous (ARTMIPS_STAGE_LDS=0)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r26
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
for rep in 1 2; do
for S in 1 0; do
  export ARTMIPS_STAGE_LDS=$S
  echo "== stage_lds $S: 512 q k100 | 512 q k32 | 2048 q k10 | 1024 q k10" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 512 100 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 512 32 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 10 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 1024 10 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
done
done
