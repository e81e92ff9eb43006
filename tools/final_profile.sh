# Round-end evidence for the shipped kernels: the default bench line, kernel-trace stats of the same
# command, HBM traffic counters in separate passes (FETCH_SIZE, WRITE_SIZE), the 1/8-shard SPMD step
# and the large-batch configs.  usage: bash tools/final_profile.sh [outdir-name]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-final}
mkdir -p $O
sleep 10
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
echo "bench done"
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt -o r --output-format csv -- python3 $R/bench.py --no-cpu-baseline > $O/bench_under_rocprof.json 2> $O/kt.log || exit 1
echo "kernel trace done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --settle-seconds 0 > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --settle-seconds 0 > $O/write.log 2>&1 || exit 1
python3 $R/tools/make_traffic.py $(ls $O/fetch/*counter_collection.csv | head -1) $(ls $O/write/*counter_collection.csv | head -1) 21015324 f16 ${1:-final} > $O/traffic.json || exit 1
echo "pmc done"
timeout -k 10 200 python3 $R/bench.py --rows 2626916 --force-spmd --steps 400 --warmup 20 --no-cpu-baseline --settle-seconds 2 > $O/spmd_eighth.json 2> $O/spmd.err || exit 1
timeout -k 10 200 python3 $R/bench.py --rows 2626916 --steps 400 --warmup 20 --no-cpu-baseline --settle-seconds 2 > $O/local_eighth.json 2>> $O/spmd.err || exit 1
echo "eighth shard done"
timeout -k 10 500 python3 $R/tools/config_bench.py c3,c4,c5,c5f32 > $O/config_bench.jsonl 2> $O/config.err || exit 1
cat $O/bench.json
