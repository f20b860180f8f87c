cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_lstm.py -m gpu -q -x -k "not 330k" 2>&1 | tail -2
timeout 120 python scripts/gpu_lstm_time.py 2000 2>&1 | tail -2
CMX_LSTM_PERBYTE=1 timeout 120 python scripts/gpu_lstm_time.py 2000 2>&1 | tail -1
