# in-kernel clock stamps of one iteration of the gang scan.  Needs the stamps build:
#   bash art_amd/csrc/build.sh --stamps   (-> art_amd/lib/libartmips_stamps.so; the product never loads it)
# prints, per wave of one workgroup, the shader cycles since the start of the iteration at the points listed in
# profiles/r03_c_stamps.txt
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/stamps
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_stamps.so
export ARTMIPS_STAMPS=1
timeout -k 10 200 python -u tools/run_search.py 21015324 2048 100 f16 > $O/k100.txt 2>&1 && grep "stamps\|scan" $O/k100.txt | tail -5
timeout -k 10 200 python -u tools/run_search.py 21015324 2048 1000 bf16 > $O/k1000.txt 2>&1 && grep "stamps\|scan" $O/k1000.txt | tail -9
