# round 3: the gang rendezvous on a GPU that two processes use at once (the members of a gang are then not all
# resident together): searches side by side must take about twice a lone one, not more, and never hang.
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/shared
mkdir -p $O
rm -f $O/go1 $O/go2
cd $R
touch $O/go1
timeout -k 10 200 python -u tools/r3/shared_gpu.py 8841823 2048 1000 40 $O/go1 2>&1 | tail -1
timeout -k 10 300 python -u tools/r3/shared_gpu.py 8841823 2048 1000 40 $O/go2 > $O/a.txt 2>&1 &
PA=$!
timeout -k 10 300 python -u tools/r3/shared_gpu.py 8841823 2048 1000 40 $O/go2 > $O/b.txt 2>&1 &
PB=$!
sleep 25
touch $O/go2
wait $PA
wait $PB
tail -1 $O/a.txt
tail -1 $O/b.txt
