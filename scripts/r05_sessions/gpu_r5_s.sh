#!/bin/bash
# Round 5, session s: after the fix of the ContextMap family's generator ring (counter passing 2^32): the new GPU tests, then the 8 MiB run again -- is the
# file the reference binary's now?
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5s; mkdir -p $O
timeout 400 python -m pytest tests/test_zgpu_p8stage.py tests/test_gpu_mixnet.py tests/test_gpu_dropin.py -q -p no:cacheprovider -k "counter_wraps or round5 or jpeg_is_byte or rich_16k" 2>&1 | tail -6 | tee $O/pytest_new.txt
timeout 800 python scripts/gpu_long_run.py --bytes 8388608 --out $O/long_run_8m_fixed.json 2>&1 | grep -v amdgpu.ids | tail -4 | tee $O/long_run_8m_fixed.txt
