# round 3: the whole GPU suite on the current build, then the default bench line
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5t
mkdir -p $O
cd $R
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python bench.py > $O/bench.json 2> $O/bench.err && cat $O/bench.json
