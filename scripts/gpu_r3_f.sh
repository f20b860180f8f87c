#!/bin/bash
# Round-3 session F: phase timers of the fxcm roles, the paq8 family kernel and the paq8 mixer.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3f; mkdir -p $O
export TMPDIR=/tmp
CMX_FXCM_PROFILE=1 timeout 300 python scripts/gpu_fxcm_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/fxcm_roles_phases.txt
CMX_P8MIX_PROFILE=1 timeout 300 python scripts/gpu_p8stage_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/p8_phases.txt
