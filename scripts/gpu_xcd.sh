cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
for q in 3 4; do
  echo "== 8 processes, GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 200 python scripts/gpu_multiproc.py 8 2>/dev/null | tail -1
done
echo "== 6 processes, GPU_MAX_HW_QUEUES=4"; GPU_MAX_HW_QUEUES=4 timeout 200 python scripts/gpu_multiproc.py 6 2>/dev/null | tail -1
echo "== 12 processes, GPU_MAX_HW_QUEUES=3"; GPU_MAX_HW_QUEUES=3 timeout 200 python scripts/gpu_multiproc.py 12 2>/dev/null | tail -1
