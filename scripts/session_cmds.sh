#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/final7; mkdir -p $O
( time timeout 85 python -m pytest tests/test_gpu_dropin.py -q -k "decodes_the_reference or decodes_empty or decodes_mixed or decodes_12k" 2>&1 | grep -v amdgpu | tail -4 ) > $O/pytest_dec.txt 2>&1
cat $O/pytest_dec.txt
