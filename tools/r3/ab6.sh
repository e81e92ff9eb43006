# what the filter costs in the gang shapes at top-100 and top-1000 (ARTMIPS_ABLATE=16: no filter at all)
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r5f
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
export GANG_AB_NOCHECK=1
GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 nofilter=16 noprecheck=256 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 nofilter=16 noprecheck=256 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
ARTMIPS_NW=44 GANG_AB_SETTLE=3 timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 1000 bf16 5 base=0 nofilter=16 2>&1 | grep -v amdgpu.ids | tee -a $O/ab.txt
