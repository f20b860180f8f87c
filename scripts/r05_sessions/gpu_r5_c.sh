#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5c; mkdir -p $O
for v in "CMX_MIXNET_SEG16=1" "CMX_MIXNET_SEG16=2" "CMX_MIXNET_SEG16=1 CMX_MIXNET_XCD=0"; do
  echo "== $v" | tee -a $O/mixnet_variants.txt
  ( export $v; timeout 120 python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids ) | tee -a $O/mixnet_variants.txt
done
