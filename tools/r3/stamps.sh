# in-kernel clock stamps of one iteration of the gang scan (diagnostic build, ARTMIPS_STAMPS=1)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5s
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
export ARTMIPS_STAMPS=1
timeout -k 10 200 python -u tools/run_search.py 21015324 2048 100 f16 > $O/k100.txt 2>&1 && grep "stamps\|scan" $O/k100.txt
timeout -k 10 200 python -u tools/run_search.py 21015324 2048 1000 bf16 > $O/k1000.txt 2>&1 && grep "stamps\|scan" $O/k1000.txt
