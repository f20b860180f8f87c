#!/bin/bash
# Round-3 session O: the mixing network's tail on two waves (layer 1 | layer 2 + SSE) -- parity, phase timers, 128 KB bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3o; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_mixnet.py -m gpu -q -x 2>&1 | tail -15 ) | tee $O/pytest_mixnet.txt
timeout 200 python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids > $O/mixnet_phases.txt; head -20 $O/mixnet_phases.txt
CMX_MIXNET_DBG=4 timeout 200 python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids > $O/mixnet_phases_tail.txt; head -22 $O/mixnet_phases_tail.txt
timeout 300 python bench.py --payload-bytes 131072 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_128k.json 2> $O/bench_128k.err
python - <<PY
import json
d = json.load(open("$O/bench_128k.json"))
print(round(d["value"]), "B/s", d["verified"]["sha256"][:16], d["verified"]["output_bytes"], {k: round(x, 2) for k, x in d["stage_us_per_bit"].items() if k != "note"})
PY
tail -2 $O/bench_128k.err
