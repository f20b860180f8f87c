#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  date
  timeout 600 python -m pytest tests/test_gpu_dropin.py -q -p no:cacheprovider -k "decodes_empty_and_tiny or decodes_the_reference" 2>&1 | tail -25 | cut -c1-400
  date
} > gpurun_out/r4_late3.log 2>&1
cat gpurun_out/r4_late3.log
