#!/bin/bash
# Stage parity + timing + the full-ensemble bench on the 64 KB fixture.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_zgpu_stage_fxcm.py tests/test_zgpu_p8stage.py -x -q > gpurun_out/stage_tests.log 2>&1; tail -3 gpurun_out/stage_tests.log
python scripts/gpu_fxcm_time.py 16 2>&1 | grep "fxcm stage"
python scripts/gpu_p8stage_time.py 8 2>&1 | grep "paq8 stage"
timeout 900 python bench.py --payload-bytes 65536 --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/bench_64k.json 2> gpurun_out/bench_64k.err; python - <<'PY'
import json
try:
    j = json.load(open('gpurun_out/bench_64k.json'))
    print(j['value'], j['stage_us_per_bit'], j['verified']['identical_to_reference_file'])
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/bench_64k.err').read()[-1500:])
PY
