# Round-end evidence for the shipped scan kernel: the default bench line, kernel-trace stats of the
# same command, then HBM traffic counters in separate passes (FETCH_SIZE, WRITE_SIZE).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/final
mkdir -p $O
sleep 10
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt -o r --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.log || exit 1
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --settle-seconds 0 > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --settle-seconds 0 > $O/write.log 2>&1 || exit 1
cat $O/bench.json
