#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5e; mkdir -p $O
( export CMX_MIXNET_XCD=7; timeout 300 python scripts/gpu_contention.py lstm,mixnet 2>&1 | grep -v amdgpu.ids | tee $O/contention_xcd7_avoid.txt )
( export CMX_MIXNET_XCD=7 CMX_LSTM_AVOID_XCD=9; timeout 300 python scripts/gpu_contention.py lstm,mixnet 2>&1 | grep -v amdgpu.ids | tee $O/contention_xcd7_noavoid.txt )
( export CMX_LSTM_AVOID_XCD=7; timeout 300 python scripts/gpu_contention.py lstm,mixnet 2>&1 | grep -v amdgpu.ids | tee $O/contention_avoid_only.txt )
