cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
echo "== threads, 2 HIP streams per pipeline"; CMX_PIPELINE_STREAMS=2 timeout 300 python scripts/gpu_multistream.py 1,8,11 2>/dev/null | cut -c1-200
echo "== processes, 2 HIP streams per pipeline, GPU_MAX_HW_QUEUES=2"; CMX_PIPELINE_STREAMS=2 GPU_MAX_HW_QUEUES=2 timeout 300 python scripts/gpu_multiproc.py 8,12 2>/dev/null | cut -c1-230
