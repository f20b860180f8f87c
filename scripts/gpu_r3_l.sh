#!/bin/bash
# Round-3 session L: fxcm role M's clocks by bit position; stage parity
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3l; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_zgpu_stage_fxcm.py -m gpu -q -x 2>&1 | tail -6 ) | tee $O/pytest.txt
CMX_FXCM_PROFILE=1 timeout 300 python scripts/gpu_fxcm_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/fxcm_roles_phases.txt
timeout 300 python scripts/gpu_fxcm_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/fxcm_time.txt
