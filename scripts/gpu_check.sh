#!/bin/bash
# state check: GPU parity tests, LSTM stage timing, bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/pytest_gpu.txt
timeout 300 python scripts/gpu_lstm_time.py 2000 2>&1 | grep -v amdgpu.ids | tee gpurun_out/lstm_time.txt
timeout 600 python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err; cat gpurun_out/bench.json
