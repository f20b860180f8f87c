#!/bin/bash
# Round 5, session r: dry run of the detail comparison (scripts/gpu_stage_hashes.py --detail-block) on block 0 against the reference's rows of the first
# 6000 bytes (one value doctored to exercise the report)
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5r; mkdir -p $O
timeout 300 python scripts/gpu_stage_hashes.py --bytes 8388608 --detail-block 0 --detail-dir tmp_longref --out $O/detail_dry.txt 2>&1 | grep -v amdgpu.ids | tail -20 | cut -c1-1200 | tee $O/detail_dry.log
