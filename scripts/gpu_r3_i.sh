#!/bin/bash
# Round-3 session I: after cmx_pipeline_fetch (no device-wide synchronisation per chunk) -- the engine / drop-in command lines end to end.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3i; mkdir -p $O
export TMPDIR=/tmp
python - <<'PY' > /tmp/in256k
import sys; sys.path.insert(0, ".")
from cmix_amd import synth
sys.stdout.buffer.write(synth.enwik_like(262144, 1000, rich=True))
PY
for m in prewarm noprewarm; do
  if [ $m = noprewarm ]; then export CMIX_NO_PREWARM=1; else unset CMIX_NO_PREWARM; fi
  ( time CMIX_TIMING=1 oracle/_ref/cmix_engine -c /tmp/in256k /tmp/out_$m ) 2>&1 | grep -i "timing\|real" | sed "s/^/$m: /" | tee -a $O/engine_e2e_timing.txt
done
unset CMIX_NO_PREWARM
cmp /tmp/out_prewarm /tmp/out_noprewarm && echo "outputs identical" | tee -a $O/engine_e2e_timing.txt
( time oracle/_ref/cmix_dropin -c /tmp/in256k /tmp/out_dropin ) 2>&1 | grep real | sed "s/^/cmix_dropin, 256 KB (reference runner + coder, look-ahead mode, prewarm from a static initialiser): /" | tee -a $O/engine_e2e_timing.txt
cmp /tmp/out_prewarm /tmp/out_dropin && echo "drop-in output identical" | tee -a $O/engine_e2e_timing.txt
( time oracle/_ref/cmix_lookahead -c /tmp/in256k /tmp/out_la ) 2>&1 | grep real | sed "s/^/cmix_lookahead, 256 KB (host paq8 + fxcm from the reference, device rest): /" | tee -a $O/engine_e2e_timing.txt
cmp /tmp/out_prewarm /tmp/out_la && echo "look-ahead hybrid output identical" | tee -a $O/engine_e2e_timing.txt
( timeout 900 python -m pytest tests/test_gpu_dropin.py tests/test_gpu_predictor.py -m gpu -q -k "dropin_engine or engine or lookahead" --durations=6 2>&1 | tail -12 ) | tee $O/pytest.txt
