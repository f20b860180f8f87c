#!/bin/bash
# Round-3 session H: after the prune -- drop-in / engine file tests again, phase timers of the three fxcm roles, end-to-end timing of
# cmix_engine with and without construction during preprocessing (cmx_prewarm).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3h; mkdir -p $O
export TMPDIR=/tmp
( timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_lstm.py tests/test_zgpu_stage_fxcm.py -m gpu -q -k "dropin_engine or engine_no_pre or engine_text or lstm or fxcm" --durations=5 2>&1 | tail -12 ) | tee $O/pytest.txt
CMX_FXCM_PROFILE=1 timeout 300 python scripts/gpu_fxcm_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/fxcm_roles_phases.txt
python - <<'PY' > /tmp/in128k
import sys; sys.path.insert(0, ".")
from cmix_amd import synth
sys.stdout.buffer.write(synth.enwik_like(131072, 1000, rich=True))
PY
for m in prewarm noprewarm; do
  if [ $m = noprewarm ]; then export CMIX_NO_PREWARM=1; else unset CMIX_NO_PREWARM; fi
  ( time CMIX_TIMING=1 oracle/_ref/cmix_engine -c /tmp/in128k /tmp/out_$m ) 2>&1 | grep -i "timing\|real" | sed "s/^/$m: /" | tee -a $O/engine_e2e_timing.txt
done
cmp /tmp/out_prewarm /tmp/out_noprewarm && echo "outputs identical" | tee -a $O/engine_e2e_timing.txt
( time oracle/_ref/cmix_dropin -c /tmp/in128k /tmp/out_dropin ) 2>&1 | grep real | sed "s/^/cmix_dropin (reference runner + coder, look-ahead, prewarm at start): /" | tee -a $O/engine_e2e_timing.txt
cmp /tmp/out_prewarm /tmp/out_dropin && echo "drop-in output identical" | tee -a $O/engine_e2e_timing.txt
