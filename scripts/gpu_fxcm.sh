#!/bin/bash
# First GPU session of the fxcm stage (written after round 1's GPU budget was spent): parity, timing A/B, bench with the stage, kernel stats.
#   gpurun --timeout 1500 -- 'bash scripts/gpu_fxcm.sh'
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zgpu_stage_fxcm.py -q > gpurun_out/fxcm_tests.log 2>&1; tail -3 gpurun_out/fxcm_tests.log
timeout 600 python -m pytest tests/test_zgpu_p8cm2.py tests/test_zgpu_p8cm.py tests/test_zgpu_p8dmc.py tests/test_zgpu_p8match.py -q > gpurun_out/p8_blocks_tests.log 2>&1; tail -3 gpurun_out/p8_blocks_tests.log
timeout 300 python scripts/gpu_fxcm_time.py 64 > gpurun_out/fxcm_time.txt 2>&1
CMX_FXCM_SERIAL_MAPS=1 timeout 300 python scripts/gpu_fxcm_time.py 64 >> gpurun_out/fxcm_time.txt 2>&1; cat gpurun_out/fxcm_time.txt
timeout 600 python bench.py --fxcm-device --no-cpu-baseline > gpurun_out/bench_fxcm.json 2> gpurun_out/bench_fxcm.err; tail -c 1500 gpurun_out/bench_fxcm.json
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_fxcm" -- python "$OLDPWD/bench.py" --fxcm-device --no-cpu-baseline --steps 16 > /dev/null 2>&1)
find gpurun_out/prof_fxcm -name '*kernel_stats.csv' | head -1 | xargs -r head -12
