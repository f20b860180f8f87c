#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_ctxmodels.py -x -q 2>&1 | tail -25 | tee gpurun_out/pytest_ctx.txt
timeout 300 python scripts/gpu_ctx_time.py 65536 2>&1 | grep -v amdgpu.ids | tee gpurun_out/ctx_time.txt
