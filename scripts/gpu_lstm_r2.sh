#!/bin/bash
# round 2: the multi-workgroup LSTM kernels (lstm_block.hip): parity, then timing of the new / mixed / old paths, then a kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/lstm_r2
O=gpurun_out/lstm_r2
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
timeout 900 python -m pytest tests/test_gpu_lstm.py -x -q > $O/pytest_new.log 2>&1; echo "pytest new rc=$?" | tee -a $O/summary.txt
tail -5 $O/pytest_new.log
if ! grep -q " passed" $O/pytest_new.log || grep -q "failed" $O/pytest_new.log; then
  CMX_LSTM_BPTT_V1=1 timeout 900 python -m pytest tests/test_gpu_lstm.py -x -q -k "not 330k" > $O/pytest_fwdonly.log 2>&1; echo "pytest fwd-new/bptt-old rc=$?" | tee -a $O/summary.txt
  tail -5 $O/pytest_fwdonly.log
fi
for mode in new bpttv1 v1; do
  case $mode in new) E="";; bpttv1) E="CMX_LSTM_BPTT_V1=1";; v1) E="CMX_LSTM_V1=1";; esac
  env $E timeout 300 python scripts/gpu_lstm_time.py 4000 > $O/time_$mode.txt 2>&1
  echo "== $mode"; grep "us/byte" $O/time_$mode.txt
done | tee -a $O/summary.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o lstm -- python $GRAFT_REPO_ROOT/scripts/gpu_lstm_time.py 2000 > $GRAFT_REPO_ROOT/$O/prof.out 2> $GRAFT_REPO_ROOT/$O/prof.err )
for f in $(find $O/prof -name '*kernel_stats*.csv'); do head -12 $f | cut -c1-160; cp $f $O/lstm_kernel_stats.csv; done
