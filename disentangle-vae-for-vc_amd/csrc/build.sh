#!/bin/bash
# Build the gfx950 library (cross-compiles without a GPU).  Each .hip is compiled to an object of its own, all in
# parallel, and only when it (or a header) is newer than its object; then one link.
#   csrc/build.sh [extra hipcc flags]        -> ../libdvae_hip.so   the product: one deterministic kernel dispatch
#   csrc/build.sh dev [extra hipcc flags]    -> ../libdvae_dev.so   + probe kernels, environment tile knobs, in-kernel
#                                                                     timelines (include/dvae_hip_dev.h; scripts/ only)
# FORCE=1 recompiles everything.
set -euo pipefail
HERE="$(cd "$(dirname "$0")" && pwd)"
OUT="$HERE/../libdvae_hip.so"
OBJ="$HERE/build/hip"
DEFS=()
if [ "${1:-}" = "dev" ]; then
  shift
  OUT="$HERE/../libdvae_dev.so"
  OBJ="$HERE/build/dev"
  DEFS=(-DDVAE_DEV -DDVAE_PERS_TS)
fi
HIPCC="${HIPCC:-/opt/rocm/bin/hipcc}"
FLAGS=(--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fvisibility=hidden -Wall -Wno-unused-function
       "${DEFS[@]}" "$@")
mkdir -p "$OBJ"
# a change of flags invalidates every object
SIG="$(printf '%s ' "${FLAGS[@]}" | sha256sum | cut -c1-16)"
if [ "$(cat "$OBJ/.flags" 2>/dev/null || true)" != "$SIG" ] || [ "${FORCE:-0}" = "1" ]; then
  rm -f "$OBJ"/*.o
  echo "$SIG" > "$OBJ/.flags"
fi
HDR_NEWEST="$(ls -t "$HERE"/*.h "$HERE"/../../include/*.h | head -1)"
pids=()
objs=()
compiled=()
for f in gemm gemm256 lstm lstm_pers bn elem frontend prof repack; do
  o="$OBJ/$f.o"
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$HERE/$f.hip" -nt "$o" ] || [ "$HDR_NEWEST" -nt "$o" ]; then
    "$HIPCC" "${FLAGS[@]}" -c "$HERE/$f.hip" -o "$o" &
    pids+=($!)
    compiled+=("$f")
  fi
done
rc=0
for p in "${pids[@]:-}"; do
  [ -n "$p" ] && { wait "$p" || rc=1; }
done
[ $rc -eq 0 ] || { echo "compile failed" >&2; exit 1; }
"$HIPCC" --offload-arch=gfx950 -shared -fPIC -fvisibility=hidden "${objs[@]}" -o "$OUT"
# what this run actually did (the driver's build check reads it: "hipcc -c" commands run vs objects reused)
echo "hipcc: compiled ${#compiled[@]} of ${#objs[@]} objects for gfx950 (${compiled[*]:-none}; the rest reused, newer than their sources), 1 link"
echo "built $OUT"
