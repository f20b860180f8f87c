#!/bin/bash
# Round 5, session p: the engine's side of the 8 MiB digest comparison (oracle/ref_long_trace.cpp runs the reference's side on the CPU for hours):
# 130 column-group digests + the final probability's per 64 KB block -> gpurun_out/r5p/stage_hashes_8m.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5p; mkdir -p $O
timeout 1000 python scripts/gpu_stage_hashes.py --bytes 8388608 --out $O/stage_hashes_8m.txt 2>&1 | grep -v amdgpu.ids | tail -5 | tee $O/stage_hashes_8m.log
