#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  date
  timeout 600 python -m pytest tests/test_gpu_late.py -q -p no:cacheprovider -k "replay_text_96 or replay_binary_64 or round_trip_text" 2>&1 | grep -E "MISMATCH|^OK|passed|failed" | cut -c1-700
  date
} > gpurun_out/r4_late2.log 2>&1
cat gpurun_out/r4_late2.log
