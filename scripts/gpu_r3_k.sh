#!/bin/bash
# Round-3 session K: the paq8 mixer on four workgroups (cmx_p8s_mix4_kernel) -- parity, stage timings, 128 KB bench; the one-workgroup
# mixer (CMX_P8MIX_ONE_WG=1) beside it.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3k; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_zgpu_p8stage.py tests/test_zgpu_stage_fxcm.py -m gpu -q -x 2>&1 | tail -6 ) | tee $O/pytest.txt
timeout 300 python scripts/gpu_p8stage_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/p8_time.txt
CMX_P8MIX_ONE_WG=1 timeout 300 python scripts/gpu_p8stage_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/p8_time_one_wg.txt
timeout 300 python bench.py --payload-bytes 131072 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_128k.json 2> $O/bench_128k.err
python - <<PY
import json
d = json.load(open("$O/bench_128k.json"))
print(round(d["value"]), "B/s", d["verified"]["sha256"][:16], d["verified"]["output_bytes"], {k: round(x, 2) for k, x in d["stage_us_per_bit"].items() if k != "note"})
print({k: round(x, 2) for k, x in d["paq8_role_us_per_bit"].items() if k != "note"})
PY
tail -2 $O/bench_128k.err
