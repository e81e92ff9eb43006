# round 3, call 10: what does the refill DMA cost the phased 8-wave loop, and why?  (diagnostic build, results
# meaningless): 2048 no survivors (threshold +inf, nothing else changed), +1 no refill at all, +512 refill
# issued but its landing never waited for, +1024 every refill re-reads one L2-resident tile
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r10b
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_NOCHECK=1 GANG_AB_SETTLE=3
export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag.so
echo "== diag k1000 (8x2 gangs, phased)" | tee -a $O/ab.txt
timeout -k 10 300 python -u tools/gang_ab.py 21015324 2048 1000 bf16 3 base=0 nosurv=2048 nosurv_nodma=2049 nosurv_nowaitvm=2560 nosurv_l2=3072 nosurv_l2_nowaitvm=3584 2>&1 | grep "ablate" | tee -a $O/ab.txt
echo "== diag k100 (4x4 gangs)" | tee -a $O/ab.txt
timeout -k 10 300 python -u tools/gang_ab.py 21015324 2048 100 f16 3 base=0 nosurv=2048 nosurv_nodma=2049 nosurv_nowaitvm=2560 nosurv_l2=3072 nosurv_l2_nowaitvm=3584 2>&1 | grep "ablate" | tee -a $O/ab.txt
