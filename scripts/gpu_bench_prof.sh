#!/bin/bash
# bench line + rocprofv3 kernel-trace stats of the same command + smoke
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python __graft_entry__.py smoke 2>&1 | tail -2
python bench.py --steps 3 --warmup 1 > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json; tail -3 gpurun_out/bench.err
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o pipe -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err )
cat gpurun_out/prof_bench.json
for f in $(find gpurun_out/prof -name '*kernel_stats*.csv'); do echo == $f; head -16 $f; done
