#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
echo "== normal"; python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids
echo "== dbg=1 (no row swaps after t=0)"; CMX_MIXNET_DBG=1 python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids
