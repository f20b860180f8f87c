#!/bin/bash
# end-of-round validation: full GPU test suite, smoke, default bench line, rocprofv3 kernel stats of the bench command,
# then (CMX_FINAL_LONG=1) the 1 MiB shard-prefix parity run of the look-ahead compressor
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --durations=6 2>&1 | tail -14 | tee gpurun_out/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/smoke.txt
timeout 600 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o pipe -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err )
cat gpurun_out/prof_bench.json | cut -c1-300
for f in $(find gpurun_out/prof -name '*kernel_stats*.csv'); do echo == $f; grep -v "at::native" $f | head -12; done
if [ "$CMX_FINAL_LONG" = "1" ]; then
  ( time CMX_LONG=1 timeout 900 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k 1mib 2>&1 | tail -3 ) 2>&1 | tee gpurun_out/lookahead_1mib.txt
fi
