#!/bin/bash
# GPU box: XCD-local hand-off (default) against the write-through form (DVAE_PERS_XCD_LOCAL=0): us per frame of the recurrence
# launches that have it, then the hand-off tests (mixed geometries on one workspace: local and write-through launches alternate).
{
  for rep in 1 2; do
    for loc in 0 1; do
      echo "--- DVAE_PERS_XCD_LOCAL=$loc"
      DVAE_PERS_XCD_LOCAL=$loc LSTM_MODE=1 LSTM_S16=1 LSTM_PERS=1 timeout 300 python scripts/lstm_rec_bench.py 1024 0 10 2>&1 | tail -1
      DVAE_PERS_XCD_LOCAL=$loc LSTM_MODE=1 LSTM_S16=1 LSTM_PERS=1 timeout 300 python scripts/lstm_rec_bench.py 512 0 10 2>&1 | tail -1
      DVAE_PERS_XCD_LOCAL=$loc LSTM_MODE=1 LSTM_S16=1 LSTM_PERS=1 LSTM_N=256 timeout 300 python scripts/lstm_rec_bench.py 1024 0 10 2>&1 | tail -1
      DVAE_PERS_XCD_LOCAL=$loc LSTM_MODE=1 LSTM_S16=1 LSTM_PERS=1 LSTM_N=256 timeout 300 python scripts/lstm_rec_bench.py 512 0 10 2>&1 | tail -1
      DVAE_PERS_XCD_LOCAL=$loc LSTM_MODE=2 LSTM_PERS=1 timeout 300 python scripts/lstm_rec_bench.py 512 0 10 2>&1 | tail -1
    done
  done
  echo "--- hand-off tests"
  timeout 900 python -m pytest tests/test_hip_lstm_pers.py -q 2>&1 | tail -8
} > gpurun_out/local_probe.log 2>&1
cat gpurun_out/local_probe.log
