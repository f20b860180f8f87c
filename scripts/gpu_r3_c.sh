#!/bin/bash
# Round-3 session C: the speculative multi-workgroup mixing-network kernel -- parity tests, phase timers, A/B bench at 128 KB.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
( timeout 600 python -m pytest tests/test_gpu_mixnet.py -m gpu -q -x 2>&1 | tail -15 ) | tee $O/pytest_mixnet.txt
for v in 1 0; do
  CMX_MIXNET_SPEC=$v timeout 200 python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids > $O/mixnet_phases_spec$v.txt; head -12 $O/mixnet_phases_spec$v.txt
  CMX_MIXNET_SPEC=$v timeout 300 python bench.py --payload-bytes 131072 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_128k_spec$v.json 2> $O/bench_128k_spec$v.err
  python - <<PY
import json
d = json.load(open("$O/bench_128k_spec$v.json"))
print("spec=$v", round(d["value"]), "B/s", d["verified"]["sha256"][:16], d["verified"]["output_bytes"], {k: round(x, 2) for k, x in d["stage_us_per_bit"].items() if k != "note"})
PY
  tail -2 $O/bench_128k_spec$v.err
done
