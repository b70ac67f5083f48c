#!/bin/bash
# Build the gfx950 library (cross-compiles without a GPU).
#   csrc/build.sh [extra hipcc flags]        -> ../libdvae_hip.so   the product: one deterministic kernel dispatch
#   csrc/build.sh dev [extra hipcc flags]    -> ../libdvae_dev.so   + probe kernels, environment tile knobs, in-kernel
#                                                                     timelines (include/dvae_hip_dev.h; scripts/ only)
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libdvae_hip.so"
DEFS=()
if [ "${1:-}" = "dev" ]; then
  shift
  OUT="$HERE/../libdvae_dev.so"
  DEFS=(-DDVAE_DEV -DDVAE_PERS_TS)
fi
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
"$HIPCC" --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics \
  -fvisibility=hidden -Wall -Wno-unused-function "${DEFS[@]}" \
  "$HERE/gemm.hip" "$HERE/lstm.hip" "$HERE/lstm_pers.hip" "$HERE/bn.hip" "$HERE/elem.hip" "$HERE/frontend.hip" "$HERE/prof.hip" "$HERE/repack.hip" \
  -o "$OUT" "$@"
echo "built $OUT"
