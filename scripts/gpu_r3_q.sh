#!/bin/bash
# Round-3 session Q: several full-ensemble streams on one GPU (throughput mode) with the round-3 kernels
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3q; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python scripts/gpu_multistream_engine.py 1,2 131072 2>&1 | grep -v amdgpu.ids | tee $O/multistream_engine.txt
CMX_MIXNET_SPEC=0 timeout 600 python scripts/gpu_multistream_engine.py 2,3 131072 2>&1 | grep -v amdgpu.ids | tee -a $O/multistream_engine.txt
