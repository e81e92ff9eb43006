set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5w
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_search3_gpu.py tests/test_search_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
