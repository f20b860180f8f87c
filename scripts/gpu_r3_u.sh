#!/bin/bash
# Round-3 session U: the mixing network's opt-in tolerance mode (tree-sum dot products) -- deviation from strict mode, kernel time, 128 KB stream
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3u; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_mixnet.py -m gpu -q -x -s -k "tolerance or golden_text_96" 2>&1 | grep -v amdgpu.ids | tail -8 ) | tee $O/pytest_tolerance.txt
CMX_MIXNET_TOLERANCE=1 timeout 200 python scripts/gpu_prof.py 4096 2>&1 | grep -v "amdgpu.ids\|P0\|^  -" | head -12 | tee $O/mixnet_phases_tolerance.txt
CMX_MIXNET_TOLERANCE=1 timeout 300 python bench.py --payload-bytes 131072 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_128k_tolerance.json 2> $O/bench_128k_tolerance.err
python - <<PY
import json
d = json.load(open("$O/bench_128k_tolerance.json"))
print("tolerance mode:", round(d["value"]), "B/s", d["verified"]["sha256"][:16], d["verified"]["output_bytes"], "bytes (strict: e860a8d54dde990d, 43410)", {k: round(x, 2) for k, x in d["stage_us_per_bit"].items() if k != "note"})
PY
tail -2 $O/bench_128k_tolerance.err
