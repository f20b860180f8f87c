#!/bin/bash
# Round-3 session B: the new tests again (no -x), with timings of configs 3 and 4.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
( time timeout 1700 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_dropin.py tests/test_zgpu_p8stage.py tests/test_zgpu_stage_fxcm.py -m gpu -q \
    -k "lookahead_mode or protocol or dropin_engine or silesia or rich or hdrs" --durations=14 2>&1 | tail -40 ) 2>&1 | tee $O/pytest_new.txt
cp gpurun_out/config3_dict_time.txt gpurun_out/config4_silesia_time.txt $O/ 2>/dev/null
