#!/bin/bash
# One GPU-box call: the deployment selftest first (chip coverage log), then whatever the caller passes.
#   gpurun --timeout N -- 'bash scripts/gpu_call.sh <commands...>'
mkdir -p gpurun_out
{
  echo "=== $(date -u +%FT%TZ) selftest"
  timeout 300 python -m dvae_amd.selftest --rounds 40 2>&1 | tail -12
} >> gpurun_out/selftest_chips.log 2>&1
tail -3 gpurun_out/selftest_chips.log
bash -c "$*"
