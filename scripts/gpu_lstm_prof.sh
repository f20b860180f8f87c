#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/prof_lstm
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_lstm -o lstm -- python $GRAFT_REPO_ROOT/scripts/gpu_lstm_time.py 1000 > $GRAFT_REPO_ROOT/gpurun_out/lstm_prof.out 2> $GRAFT_REPO_ROOT/gpurun_out/lstm_prof.err )
grep "us/byte" gpurun_out/lstm_prof.out
for f in $(find gpurun_out/prof_lstm -name '*kernel_stats*.csv'); do head -12 $f | cut -c1-150; done
