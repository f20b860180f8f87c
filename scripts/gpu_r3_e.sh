#!/bin/bash
# Round-3 session D: spec kernel with the deeper scout lead and row-state prefetch -- parity, phase timers (gather / scout / tail), bench 128 KB.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_mixnet.py -m gpu -q -x 2>&1 | tail -6 ) | tee $O/pytest_mixnet.txt
for d in 0 4; do
  CMX_MIXNET_DBG=$d timeout 200 python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids > $O/mixnet_phases_dbg$d.txt; head -24 $O/mixnet_phases_dbg$d.txt | grep -v " 0 ticks"
done
timeout 300 python bench.py --payload-bytes 131072 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_128k.json 2> $O/bench_128k.err
python - <<PY
import json
d = json.load(open("$O/bench_128k.json"))
print(round(d["value"]), "B/s", d["verified"]["sha256"][:16], d["verified"]["output_bytes"], {k: round(x, 2) for k, x in d["stage_us_per_bit"].items() if k != "note"})
print({k: round(x, 2) for k, x in d["paq8_role_us_per_bit"].items() if k != "note"})
PY
tail -2 $O/bench_128k.err
