# round 3, call 2: early-read loop form (ablate 512) on the gang shapes and on HBM-bound multi-group shapes
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5b
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 7 base=0 late=128 early=512 > $O/ab_k100.txt 2>&1 && cat $O/ab_k100.txt
timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 7 base=0 early=512 > $O/ab_k1000.txt 2>&1 && cat $O/ab_k1000.txt
ARTMIPS_NW=44 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 7 base=0 late=128 early=512 > $O/ab_k1000_4x4.txt 2>&1 && cat $O/ab_k1000_4x4.txt
timeout -k 10 200 python -u tools/gang_ab.py 21015324 128 32 f16 11 base=0 piecewise=64 nobranch=256 early=512 > $O/ab_q128.txt 2>&1 && cat $O/ab_q128.txt
timeout -k 10 200 python -u tools/gang_ab.py 21015324 300 100 f16 9 base=0 piecewise=64 early=512 > $O/ab_q300.txt 2>&1 && cat $O/ab_q300.txt
