cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_pipeline.py -m gpu -q 2>&1 | tail -3
timeout 200 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
r=json.loads(sys.stdin.read()); print('bench', round(r['value']), r['ms_per_step'], r['stage_us_per_bit'])"
timeout 600 python -m pytest tests/test_gpu_dropin.py -m gpu -q -k 50k --durations=1 2>&1 | tail -5
