#!/bin/bash
# Round-3 session N: paq8 family -- clocks by bit position and phase, how often it leaves the common path
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3n; mkdir -p $O
export TMPDIR=/tmp
CMX_P8FAM_PROFILE=1 timeout 300 python scripts/gpu_p8stage_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/p8_fam_phases.txt
