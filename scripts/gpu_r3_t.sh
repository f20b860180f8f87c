#!/bin/bash
# Round-3 session T: the co-residency accounting test
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3t; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_pipeline.py -m gpu -q -x -k "refused or bad_args" 2>&1 | tail -6 ) | tee $O/pytest.txt
