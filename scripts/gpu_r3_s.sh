#!/bin/bash
# Round-3 session S: after the exchange-tag layout fix of the four-workgroup paq8 mixer -- stage parity + 128 KB bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3s; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_zgpu_p8stage.py -m gpu -q -x 2>&1 | tail -4 ) | tee $O/pytest.txt
timeout 300 python bench.py --payload-bytes 131072 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_128k.json 2> $O/bench_128k.err
python - <<PY
import json
d = json.load(open("$O/bench_128k.json"))
print(round(d["value"]), "B/s", d["verified"]["sha256"][:16], d["verified"]["output_bytes"], {k: round(x, 2) for k, x in d["stage_us_per_bit"].items() if k != "note"})
PY
