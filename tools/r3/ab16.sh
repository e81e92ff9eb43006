# round 3, call 16: gang rendezvous (members of a gang pause when somebody is more than LEAD stages behind):
# on (default lead 16, and 8 / 32 / 64) against off (ablate 8192) inside one process; then L2 hit rate and HBM bytes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r16
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests -x -q -m gpu --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -2 $O/tests.log
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
echo "== k1000 (8x2 phased), gangs of 8" | tee -a $O/ab.txt
timeout -k 10 300 python -u tools/gang_ab.py 21015324 2048 1000 bf16 7 sync16=0 nosync=8192 sync8=0//8 sync32=0//32 sync64=0//64 2>&1 | grep "ablate" | tee -a $O/ab.txt
echo "== k1000, 8.8M rows" | tee -a $O/ab.txt
timeout -k 10 300 python -u tools/gang_ab.py 8841823 2048 1000 bf16 7 sync16=0 nosync=8192 2>&1 | grep "ablate" | tee -a $O/ab.txt
echo "== k300, gangs of 4 (1024 queries)" | tee -a $O/ab.txt
timeout -k 10 300 python -u tools/gang_ab.py 21015324 1024 300 f16 7 sync16=0 nosync=8192 2>&1 | grep "ablate" | tee -a $O/ab.txt
cd /tmp && export TMPDIR=/tmp
for A in 0 8192; do
  export ARTMIPS_ABLATE=$A
  timeout -s KILL 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/l2_a$A -o r --output-format csv -- python3 $R/tools/run_search.py 21015324 2048 1000 bf16 > $O/l2_a$A.log 2>&1 || exit 1
  timeout -s KILL 150 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/fetch_a$A -o r --output-format csv -- python3 $R/tools/run_search.py 21015324 2048 1000 bf16 > $O/fetch_a$A.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $O/l2_a0 $O/l2_a8192 $O/fetch_a0 $O/fetch_a8192 > $O/summary.json
grep "hit_rate\|corrected\|launch_ms" $O/summary.json
