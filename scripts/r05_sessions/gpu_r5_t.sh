#!/bin/bash
# Round 5, session t: which of the new GPU tests aborts the process (session s lost the head of the output)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5t; mkdir -p $O
timeout 200 python -m pytest tests/test_zgpu_p8stage.py -v -p no:cacheprovider -k "counter_wraps" 2>&1 | grep -v "amdgpu.ids\|Extension modules" | head -40 | cut -c1-300 | tee $O/pytest_counter.txt
timeout 300 python -m pytest tests/test_gpu_mixnet.py -v -p no:cacheprovider -k "round5" 2>&1 | grep -v "amdgpu.ids\|Extension modules" | head -60 | cut -c1-300 | tee $O/pytest_variants.txt
timeout 200 python -m pytest tests/test_gpu_dropin.py -v -p no:cacheprovider -k "jpeg_is_byte" 2>&1 | grep -v "amdgpu.ids\|Extension modules" | tail -8 | cut -c1-300 | tee $O/pytest_jpeg.txt
