#!/bin/bash
# Round 5, session q: (1) throughput mode with this round's kernels -- S = 1, 2, 3 full-ensemble streams on one GPU, every file compared;
# (2) the N = 2 bench path on ONE GPU (CMX_BENCH_SAME_DEVICE=1: both ranks on device 0): barrier, per-rank verification, per-rank reference CPU baseline.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5q; mkdir -p $O
timeout 420 python scripts/gpu_multistream_engine.py 1,2,3 131072 2>&1 | grep -v amdgpu.ids | tee $O/multistream_engine.txt
CMX_BENCH_SAME_DEVICE=1 timeout 420 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
  bench.py --gpus 2 --steps 8 --warmup 1 --payload-bytes 131072 --cpu-baseline-bytes 16384 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/bench_2ranks_one_gpu.txt
