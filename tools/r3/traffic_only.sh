# the two PMC passes of tools/final_profile.sh on their own (after a change to the kernel sources that does not change
# the code, e.g. a comment: profiles/traffic.json records the sources' sha256), then one bench line that carries the figure
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-traffic_only}
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --settle-seconds 0 > $O/fetch.log 2>&1 || exit 1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o r --output-format csv -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --settle-seconds 0 > $O/write.log 2>&1 || exit 1
python3 $R/tools/make_traffic.py $(ls $O/fetch/*counter_collection.csv | head -1) $(ls $O/write/*counter_collection.csv | head -1) 21015324 f16 r03_zz > $O/traffic.json || exit 1
cp $O/traffic.json $R/profiles/traffic.json
sleep 8
timeout -k 10 300 python3 $R/bench.py > $O/bench.json 2> $O/bench.err || exit 1
cut -c1-160 $O/bench.json
