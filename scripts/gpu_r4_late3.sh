#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  date
  timeout 300 python scripts/gpu_late_time.py text_2k_nofull 2048 2>&1 | grep -v amdgpu.ids
  timeout 600 python -m pytest tests/test_gpu_late.py -q -p no:cacheprovider -k "replay_text_96 or pretrained or brackets" 2>&1 | grep -E "MISMATCH|^OK|passed|failed|rror" | cut -c1-600
  date
} > gpurun_out/r4_late3.log 2>&1
cat gpurun_out/r4_late3.log
