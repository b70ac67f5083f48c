#!/bin/bash
# One GPU-box call: the deployment selftest first (chip coverage log), then whatever the caller passes.
#   gpurun --timeout N -- 'bash scripts/gpu_call.sh <commands...>'
# Every call writes its own log (gpurun merges gpurun_out/ back file by file: one shared name would keep the last call only);
# scripts/selftest_summary.py turns them into profiles/selftest_chips.md.
mkdir -p gpurun_out/selftest
LOG=gpurun_out/selftest/$(date -u +%Y%m%dT%H%M%S)_$RANDOM.log
{
  echo "=== $(date -u +%FT%TZ) selftest"
  timeout 300 python -m dvae_amd.selftest --rounds 40 2>&1 | tail -12
} > "$LOG" 2>&1
tail -3 "$LOG"
bash -c "$*"
