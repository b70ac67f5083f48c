#!/bin/bash
# Build libdvae_hip.so for gfx950 (cross-compiles without a GPU).  Usage: csrc/build.sh [extra hipcc flags]
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libdvae_hip.so"
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics \
  -fvisibility=hidden -Wall -Wno-unused-function \
  "$HERE/gemm.hip" "$HERE/lstm.hip" "$HERE/lstm_pers.hip" "$HERE/bn.hip" "$HERE/elem.hip" "$HERE/frontend.hip" "$HERE/prof.hip" "$HERE/repack.hip" \
  -o "$OUT" "$@"
echo "built $OUT"
