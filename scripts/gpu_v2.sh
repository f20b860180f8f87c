#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
timeout 600 python -m pytest tests -q -m gpu -x 2>&1 | tail -15
timeout 120 python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids
