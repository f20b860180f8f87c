#!/bin/bash
# Round-3 session A: the new look-ahead mode / drop-in engine / config 3 + 4 tests, the rich-alphabet reference-hash fixtures, the
# default bench line (1 MiB rich shard, CPU reference beside it), mixing-network phase timers. Outputs under gpurun_out/r3a/.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3a; mkdir -p $O
export TMPDIR=/tmp
nproc > $O/host.txt; grep -m1 "model name" /proc/cpuinfo >> $O/host.txt; free -g | head -2 >> $O/host.txt
( time timeout 1500 python -m pytest tests/test_gpu_predictor.py tests/test_gpu_dropin.py tests/test_zgpu_p8stage.py tests/test_zgpu_stage_fxcm.py -m gpu -q -x \
    -k "lookahead_mode or protocol or dropin_engine or silesia or rich or hdrs or decode_the_reference" --durations=12 2>&1 | tail -30 ) 2>&1 | tee $O/pytest_new.txt
timeout 900 python bench.py > $O/bench_1m.json 2> $O/bench_1m.err; cut -c1-600 $O/bench_1m.json; tail -3 $O/bench_1m.err
python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids > $O/mixnet_phases.txt; head -20 $O/mixnet_phases.txt
