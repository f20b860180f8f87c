#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5l; mkdir -p $O
( export CMX_CROWD=lstm,mixnet,fxcm,paq8; timeout 300 python scripts/gpu_contention.py 2>&1 | grep -v amdgpu.ids | tee $O/crowd_default.txt )
( export CMX_CROWD=lstm,mixnet,fxcm,paq8 CMX_MIXNET_XCD=7; timeout 300 python scripts/gpu_contention.py 2>&1 | grep -v amdgpu.ids | tee $O/crowd_xcd7.txt )
