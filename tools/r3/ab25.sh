# round 3, call 25: the rendezvous' lead for smaller gangs (more gangs per XCD share the 4 MiB of L2)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r25
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
echo "== gang 4, top-1000 (8x2)" | tee -a $O/ab.txt
timeout -k 10 300 python -u tools/gang_ab.py 21015324 1024 1000 bf16 5 default=0 off=8192 l6=0//6 l8=0//8 l10=0//10 l12=0//12 2>&1 | grep "ablate" | tee -a $O/ab.txt
echo "== gang 2, top-1000 (8x2)" | tee -a $O/ab.txt
timeout -k 10 300 python -u tools/gang_ab.py 21015324 512 1000 bf16 5 default=0 off=8192 l4=0//4 l6=0//6 l8=0//8 2>&1 | grep "ablate" | tee -a $O/ab.txt
echo "== gang 4, top-100 (4x4, period 8)" | tee -a $O/ab.txt
timeout -k 10 300 python -u tools/gang_ab.py 21015324 1024 100 f16 5 default=0 off=8192 l12=0//12 l16=0//16 l20=0//20 2>&1 | grep "ablate" | tee -a $O/ab.txt
