#!/bin/bash
# Round 5, session x: the one-XCD placement after the mode-bit fix (the CU-mask flag had shared bit 22 with the XCD number's field)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5x; mkdir -p $O
timeout 100 python -m pytest tests/test_gpu_mixnet.py -v -p no:cacheprovider -k "round5 and XCD" 2>&1 | grep -v "amdgpu.ids\|Extension modules\|^  File" | tail -6 | cut -c1-200 | tee $O/pytest_xcd.txt
timeout 60 python scripts/gpu_xcd_fault.py dxx 2>&1 | grep -v "amdgpu.ids\|^  File\|Extension modules\|^$" | head -6 | cut -c1-200 | tee $O/xcd_seq.txt
