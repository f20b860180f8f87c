#!/bin/bash
# Round-3 session M: after fxcm role M's free-running wavefronts and the four-workgroup paq8 mixer -- 128 KB and 1 MiB bench, drop-in tests
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3m; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python bench.py --payload-bytes 131072 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_128k.json 2> $O/bench_128k.err
python - <<PY
import json
d = json.load(open("$O/bench_128k.json"))
print(round(d["value"]), "B/s", d["verified"]["sha256"][:16], d["verified"]["output_bytes"], {k: round(x, 2) for k, x in d["stage_us_per_bit"].items() if k != "note"})
print({k: round(x, 2) for k, x in d["paq8_role_us_per_bit"].items() if k != "note"})
PY
tail -2 $O/bench_128k.err
( timeout 1200 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_predictor.py -m gpu -q -x 2>&1 | tail -6 ) | tee $O/pytest_dropin.txt
