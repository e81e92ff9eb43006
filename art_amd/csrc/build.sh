#!/bin/bash
# Build libartmips.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# usage: build.sh [--asm DIR]   (--asm keeps the ISA and resource remarks in DIR)
#        build.sh --stamps      (only libartmips_stamps.so: the diagnostic build with in-kernel clock stamps,
#                                tools/r3/stamps.sh; never loaded by the product or the tests)
set -euo pipefail
here="$(cd "$(dirname "$0")" && pwd)"
out="$here/../lib"
mkdir -p "$out"
flags="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared"
if [ "${1:-}" = "--stamps" ]; then
  hipcc $flags -DARTMIPS_DIAG -DARTMIPS_STAMPS_BUILD -o "$out/libartmips_stamps.so" "$here/artmips.hip"
  echo "built $out/libartmips_stamps.so"
  exit 0
fi
# the diagnostic build: the only one that reads the ARTMIPS_* environment knobs (tools/)
hipcc $flags -DARTMIPS_DIAG -o "$out/libartmips_diag.so" "$here/artmips.hip" &
diag_pid=$!
hipcc $flags -o "$out/libarttok.so" "$here/arttok.hip" &
tok_pid=$!
if [ "${1:-}" = "--asm" ]; then
  tmp="${2:-/tmp/artmips_asm}"
  mkdir -p "$tmp"
  (cd "$tmp" && hipcc $flags -save-temps -Rpass-analysis=kernel-resource-usage \
      -o "$out/libartmips.so" "$here/artmips.hip" 2> "$tmp/resource_usage.txt")
else
  hipcc $flags -o "$out/libartmips.so" "$here/artmips.hip"
fi
wait $diag_pid
wait $tok_pid
echo "built $out/libartmips.so $out/libartmips_diag.so $out/libarttok.so"
