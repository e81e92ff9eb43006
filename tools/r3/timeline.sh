# GPU timeline of the 1/8-shard step (local and SPMD form): rocprofv3 --kernel-trace, then tools/step_timeline.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/tl3
mkdir -p $O
timeout -s KILL 200 rocprofv3 --kernel-trace -d $O/local -o r --output-format csv -- python3 $R/bench.py --rows 2626916 --steps 60 --warmup 10 --no-cpu-baseline --settle-seconds 2 > $O/local.json 2> $O/local.err || exit 1
python3 $R/tools/step_timeline.py $O/local 2 > $O/local_timeline.txt
timeout -s KILL 200 rocprofv3 --kernel-trace -d $O/spmd -o r --output-format csv -- python3 $R/bench.py --rows 2626916 --force-spmd --steps 60 --warmup 10 --no-cpu-baseline --settle-seconds 2 > $O/spmd.json 2> $O/spmd.err || exit 1
python3 $R/tools/step_timeline.py $O/spmd 2 > $O/spmd_timeline.txt
cat $O/local_timeline.txt $O/spmd_timeline.txt
