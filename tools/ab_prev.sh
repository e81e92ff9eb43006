cd /tmp
for rep in 1 2; do
for L in prev diag; do
  echo "== $L k=100"; ARTMIPS_LIBRARY=$GRAFT_REPO_ROOT/art_amd/lib/libartmips_$L.so timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/scan_sweep.py --queries 2048 --topk 100 --rounds 3 --envs --variants "ARTMIPS_GANG=8" 2>&1 | grep "variant\|MISMATCH" || exit 1
  echo "== $L k=1000"; ARTMIPS_LIBRARY=$GRAFT_REPO_ROOT/art_amd/lib/libartmips_$L.so timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/scan_sweep.py --queries 2048 --topk 1000 --rounds 3 --envs --variants "ARTMIPS_GANG=8" 2>&1 | grep "variant\|MISMATCH" || exit 1
done
done
