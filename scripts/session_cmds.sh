CMX_SKIP_TESTS=1 bash scripts/gpu_measure.sh r06
