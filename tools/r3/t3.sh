# round 3: full GPU suite on the final build
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5v
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
