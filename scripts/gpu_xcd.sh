#!/bin/bash
# Throughput-mode experiments behind profiles/r01_multiproc.txt: XCD placement of the single-workgroup stage kernels
# (CMX_SPREAD=1 -> CMX_MIXNET_XCD / CMX_LSTM_XCD per stream), hardware-queue count, HIP streams per pipeline.
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== 8 processes, no placement"; timeout 200 python scripts/gpu_multiproc.py 8 2>/dev/null | tail -1
echo "== 8 processes, spread over XCDs"; CMX_SPREAD=1 timeout 200 python scripts/gpu_multiproc.py 8 2>/dev/null | tail -1
for q in 3 4; do
  echo "== 8 processes, GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 200 python scripts/gpu_multiproc.py 8 2>/dev/null | tail -1
done
echo "== threads, 2 HIP streams per pipeline"; CMX_PIPELINE_STREAMS=2 timeout 300 python scripts/gpu_multistream.py 1,8,11 2>/dev/null | cut -c1-200
