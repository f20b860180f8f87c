#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/xfd
hipcc --offload-arch=gfx950 -O3 -o /tmp/xfd scripts/ubench/xcd_flag_data.hip 2>/dev/null
timeout 30 /tmp/xfd 14 1 > gpurun_out/xfd/xcd_flag_data.txt 2>&1
cat gpurun_out/xfd/xcd_flag_data.txt
