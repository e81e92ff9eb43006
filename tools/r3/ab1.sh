# round 3, call 1: parity of the grouped DMA issue on every kernel shape, then A/B of the variants
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5a
mkdir -p $O
cd $R
timeout -k 10 400 python -u -m pytest tests/test_search_gpu.py tests/test_search2_gpu.py -m gpu -x -q --timeout 120 --timeout-method thread > $O/tests.log 2>&1 || { tail -30 $O/tests.log; exit 1; }
tail -3 $O/tests.log
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 7 new=0 piecewise=64 late=128 nobranch=256 late_nobranch=384 > $O/ab_k100.txt 2>&1 && cat $O/ab_k100.txt
timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 7 new=0 piecewise=64 nobranch=256 > $O/ab_k1000.txt 2>&1 && cat $O/ab_k1000.txt
timeout -k 10 200 python -u tools/gang_ab.py 21015324 64 32 f16 15 new=0 piecewise=64 > $O/ab_headline.txt 2>&1 && cat $O/ab_headline.txt
