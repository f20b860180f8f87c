#!/bin/bash
# bench line + rocprofv3 kernel-trace stats of the same command + phase timers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/prof
export TMPDIR=/tmp
python bench.py --steps 3 --warmup 1 --chunk-bytes 1024 > gpurun_out/bench.json 2> gpurun_out/bench.err
cat gpurun_out/bench.json
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof -o mixnet -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --chunk-bytes 1024 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/prof_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/prof.err )
cat gpurun_out/prof_bench.json
find gpurun_out/prof -name '*stats*' | head; 
for f in $(find gpurun_out/prof -name '*stats*.csv'); do echo == $f; head -12 $f; done
python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids > gpurun_out/phases.txt; cat gpurun_out/phases.txt
