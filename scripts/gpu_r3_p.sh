#!/bin/bash
# Round-3 session P: the paq8 family's narrowed walk (only the contexts that share a key) -- parity (also with the fall-back forced), profile, bench
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3p; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_zgpu_p8stage.py -m gpu -q -x 2>&1 | tail -6 ) | tee $O/pytest.txt
( CMX_P8CM_SERIAL=2 timeout 900 python -m pytest tests/test_zgpu_p8stage.py -m gpu -q -x -k "hash" 2>&1 | tail -6 ) | tee $O/pytest_forced_fallback.txt
CMX_P8FAM_PROFILE=1 timeout 300 python scripts/gpu_p8stage_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/p8_fam_phases.txt
timeout 300 python bench.py --payload-bytes 131072 --steps 8 --warmup 1 --no-cpu-baseline > $O/bench_128k.json 2> $O/bench_128k.err
python - <<PY
import json
d = json.load(open("$O/bench_128k.json"))
print(round(d["value"]), "B/s", d["verified"]["sha256"][:16], d["verified"]["output_bytes"], {k: round(x, 2) for k, x in d["stage_us_per_bit"].items() if k != "note"})
print({k: round(x, 2) for k, x in d["paq8_role_us_per_bit"].items() if k != "note"})
PY
tail -2 $O/bench_128k.err
