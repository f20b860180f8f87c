#!/bin/bash
# Round 5, session u: the test files that touch this round's last changes (paq8 family refill, tolerance switch order, late_stop, the variants test) on the
# final build, and the 256 KB bench stream's SHA
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5u; mkdir -p $O
( time timeout 330 python -m pytest tests/test_zgpu_p8stage.py tests/test_gpu_mixnet.py tests/test_gpu_late.py tests/test_gpu_pipeline.py -q -x -p no:cacheprovider 2>&1 | grep -v amdgpu.ids | tail -8 ) 2>&1 | tee $O/pytest_subset.txt
timeout 120 python bench.py --payload-bytes 262144 --steps 5 --warmup 1 --no-cpu-baseline > $O/bench_256k.json 2> $O/bench_256k.err
python -c "import json; d=json.load(open('$O/bench_256k.json')); print(d['value'], d['stage_us_per_bit'], d['verified'])" | cut -c1-600 | tee $O/bench_256k.txt
