# A/B of two builds of the library on ONE box: art_amd/lib/libartmips_prev.so (built by hand from an
# earlier commit) against libartmips_diag.so, alternating processes.  usage: bash tools/ab_prev.sh Q K [rows]
cd /tmp
for rep in 1 2 3; do
for L in prev diag; do
  echo "== $L"; ARTMIPS_DIAGNOSTICS=1 ARTMIPS_LIBRARY=$GRAFT_REPO_ROOT/art_amd/lib/libartmips_$L.so timeout -k 10 200 python3 $GRAFT_REPO_ROOT/tools/scan_sweep.py --rows ${3:-21015324} --queries ${1:-64} --topk ${2:-32} --rounds 8 --envs --variants "ARTMIPS_GANG=8" 2>&1 | grep "variant\|MISMATCH" || exit 1
done
done
