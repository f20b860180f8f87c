#!/bin/bash
# Round-3 session R: key lists with fixed positions (no scratch-memory arrays) in the paq8 family's phase 1 and fxcm's touch step, bucket
# staging without a scratch array, run-time-indexed tables of fxcm's role X as arithmetic -- parity, stage timings, 128 KB bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3r; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_zgpu_p8stage.py tests/test_zgpu_stage_fxcm.py tests/test_gpu_mixnet.py -m gpu -q -x 2>&1 | tail -6 ) | tee $O/pytest.txt
CMX_FXCM_PROFILE=1 timeout 300 python scripts/gpu_fxcm_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/fxcm_roles_phases.txt
CMX_P8FAM_PROFILE=1 timeout 300 python scripts/gpu_p8stage_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/p8_fam_phases.txt
timeout 300 python bench.py --payload-bytes 131072 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_128k.json 2> $O/bench_128k.err
python - <<PY
import json
d = json.load(open("$O/bench_128k.json"))
print(round(d["value"]), "B/s", d["verified"]["sha256"][:16], d["verified"]["output_bytes"], {k: round(x, 2) for k, x in d["stage_us_per_bit"].items() if k != "note"})
print({k: round(x, 2) for k, x in d["paq8_role_us_per_bit"].items() if k != "note"})
PY
tail -2 $O/bench_128k.err
