#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5h; mkdir -p $O
( timeout 600 python -m pytest tests/test_zgpu_stage_fxcm.py -q -x -p no:cacheprovider 2>&1 | tail -3 ) | tee $O/fxcm_tests.txt
( export CMX_FXCM_PROFILE=1; timeout 200 python scripts/gpu_fxcm_time.py 16 2>&1 | grep -v "amdgpu.ids\|bpos" ) | tee $O/fxcm_time.txt
