#!/bin/bash
# paq8 stage on the GPU: parity tests, per-role timing from a kernel trace, then the full-ensemble bench on the 64 KB fixture.
cd "$(dirname "$0")/.." && mkdir -p gpurun_out && export TMPDIR=/tmp
R=$(pwd)
timeout 1200 python -m pytest tests/test_zgpu_p8stage.py -x -q > gpurun_out/p8stage_tests.log 2>&1; tail -6 gpurun_out/p8stage_tests.log
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_p8stage -- python $R/scripts/gpu_p8stage_time.py 8 > $R/gpurun_out/p8stage_time.txt 2>&1)
grep "paq8 stage" gpurun_out/p8stage_time.txt; find gpurun_out/prof_p8stage -name '*kernel_stats.csv' | head -1 | xargs -r head -8 | cut -c1-150
timeout 900 python bench.py --payload-bytes 65536 --steps 8 --warmup 1 --no-cpu-baseline > gpurun_out/bench_64k.json 2> gpurun_out/bench_64k.err; python - <<'PY'
import json
try:
    j = json.load(open('gpurun_out/bench_64k.json'))
    print(j['value'], j['stage_us_per_bit'], j['verified'])
except Exception as e:
    print('bench failed', e); print(open('gpurun_out/bench_64k.err').read()[-1500:])
PY
