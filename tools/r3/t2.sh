# round 3: streaming refresh + fused token gather tests, then their bench lines
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5u
mkdir -p $O
cd $R
timeout -k 10 600 python -u -m pytest tests/test_refresh_gpu.py tests/test_token_store_gpu.py -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1 || { tail -40 $O/tests.log; exit 1; }
tail -3 $O/tests.log
timeout -k 10 300 python -u tools/token_bench.py > $O/token_bench.json 2> $O/token_bench.err && cat $O/token_bench.json
timeout -k 10 400 python -u tools/upload_bench.py > $O/upload_bench.jsonl 2> $O/upload_bench.err && cat $O/upload_bench.jsonl
