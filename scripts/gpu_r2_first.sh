#!/bin/bash
# Round 2, first GPU session: time the fxcm stage and the paq8 building blocks (never timed before), bench with fxcm on device.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 300 python scripts/gpu_fxcm_time.py 32 > gpurun_out/fxcm_time.txt 2>&1
CMX_FXCM_SERIAL_MAPS=1 timeout 300 python scripts/gpu_fxcm_time.py 32 >> gpurun_out/fxcm_time.txt 2>&1; cat gpurun_out/fxcm_time.txt
timeout 500 python scripts/gpu_p8blocks_time.py 1024 > gpurun_out/p8blocks_time.txt 2>&1; cat gpurun_out/p8blocks_time.txt
timeout 400 python bench.py --fxcm-device --no-cpu-baseline > gpurun_out/bench_fxcm.json 2> gpurun_out/bench_fxcm.err; tail -c 1800 gpurun_out/bench_fxcm.json
