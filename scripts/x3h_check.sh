#!/bin/bash
# GPU box: the 8-unit fp32x3 forward kernel at H = 512 (lstm_pers_fwd_x3h) — parity + stress through the product library,
# then us per frame against the 16-unit kernel (dev build, DVAE_PERS_X3_H8=0).
mkdir -p gpurun_out
{
  timeout 1500 python -m pytest tests/test_hip_lstm_pers.py -q -x -k "fp32x3 or selftest or uneven or epoch" 2>&1 | tail -8
  for h8 in 0 1 0 1; do
    echo "--- DVAE_PERS_X3_H8=$h8"
    DVAE_LIB_PATH=$PWD/disentangle-vae-for-vc_amd/libdvae_dev.so DVAE_PERS_X3_H8=$h8 LSTM_MODE=2 LSTM_PERS=1 \
      timeout 300 python scripts/lstm_rec_bench.py 512 0 10 2>&1 | tail -3
  done
  for t in 256 512; do
    for h8 in 0 1; do
      echo "--- T=$t DVAE_PERS_X3_H8=$h8"
      DVAE_LIB_PATH=$PWD/disentangle-vae-for-vc_amd/libdvae_dev.so DVAE_PERS_X3_H8=$h8 LSTM_MODE=2 LSTM_PERS=1 LSTM_T=$t \
        timeout 300 python scripts/lstm_rec_bench.py 512 0 5 2>&1 | tail -3
    done
  done
  timeout 600 python bench.py 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'], d['roofline']['frac'])"
} > gpurun_out/x3h_check.log 2>&1
tail -60 gpurun_out/x3h_check.log
