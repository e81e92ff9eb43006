# round 3, call 11: L2 prefetch of the gang's tiles (wave 0 of every member touches its share of the lines of the
# tile PF_DIST stages ahead of the refill).  4096 = prefetch off, (d << 16) = distance d
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r11
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
echo "== diag k1000 (8x2 gangs, phased)" | tee -a $O/ab.txt
timeout -k 10 300 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 pf6=0 nopf=4096 pf2=131072 pf12=786432 pf24=1572864 2>&1 | grep "ablate" | tee -a $O/ab.txt
echo "== diag k100 (4x4 gangs)" | tee -a $O/ab.txt
timeout -k 10 300 python -u tools/gang_ab.py 21015324 2048 100 f16 5 pf6=0 nopf=4096 pf2=131072 pf12=786432 pf24=1572864 2>&1 | grep "ablate" | tee -a $O/ab.txt
echo "== diag k100 forced to 8x2 gangs" | tee -a $O/ab.txt
ARTMIPS_NW=88 timeout -k 10 300 python -u tools/gang_ab.py 21015324 2048 100 f16 5 pf6=0 nopf=4096 2>&1 | grep "ablate" | tee -a $O/ab.txt
