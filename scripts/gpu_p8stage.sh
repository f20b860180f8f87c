#!/bin/bash
# paq8 stage on the GPU: parity tests, then per-role timing from a kernel trace of a 4 KB run.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
R=$(pwd)
timeout 1200 python -m pytest tests/test_zgpu_p8stage.py -x -q > gpurun_out/p8stage_tests.log 2>&1; tail -15 gpurun_out/p8stage_tests.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_p8stage -- python $R/scripts/gpu_p8stage_time.py > $R/gpurun_out/p8stage_time.txt 2>&1)
tail -3 gpurun_out/p8stage_time.txt; find gpurun_out/prof_p8stage -name '*kernel_stats.csv' | head -1 | xargs -r head -12
