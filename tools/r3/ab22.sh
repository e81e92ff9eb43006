# round 3, call 22: which of the two slow-path changes of step 2 costs the 4 x 4 shape its 1.4 %?  old = step 1;
# diag = both; t0c1 = old tile formula + clamped store index; t1c0 = branch-free tile formula + room test
set -e
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r22
mkdir -p $O
cd $R
export ARTMIPS_DIAGNOSTICS=1 GANG_AB_SETTLE=3
for rep in 1 2; do
for L in _old "" _t0c1 _t1c0; do
  export ARTMIPS_LIBRARY=$R/art_amd/lib/libartmips_diag$L.so
  echo "== diag$L k100 | k32" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 100 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
  timeout -k 10 200 python -u tools/gang_ab.py 21015324 2048 32 f16 5 base=0 2>&1 | grep "ablate" | tee -a $O/ab.txt
done
done
