#!/bin/bash
# The whole predictor on the device: engine compressor parity (no reference model objects) and a short full-ensemble bench.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -k engine > gpurun_out/engine_tests.log 2>&1; tail -8 gpurun_out/engine_tests.log
timeout 900 python bench.py --payload-bytes 65536 --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/bench_64k.json 2> gpurun_out/bench_64k.err; tail -c 2500 gpurun_out/bench_64k.json; tail -5 gpurun_out/bench_64k.err
