#!/bin/bash
# Round 5, session w: which handle sequence makes the one-XCD placement abort
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5w; mkdir -p $O
for seq in x xx dx ddx; do
  echo "== sequence $seq" | tee -a $O/xcd_fault.txt
  timeout 60 python scripts/gpu_xcd_fault.py $seq 2>&1 | grep -v "amdgpu.ids\|^  File\|Extension modules\|^$" | head -8 | cut -c1-200 | tee -a $O/xcd_fault.txt
done
