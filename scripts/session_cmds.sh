O=gpurun_out/s12; mkdir -p $O
timeout 120 scripts/ubench/bin/sgpr_chain 2>&1 | tee $O/sgpr_chain.txt
( time timeout 900 python -m pytest tests/test_gpu_late.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -8 ) > $O/late_tests.txt 2>&1; tail -6 $O/late_tests.txt
CMX_LATE_NATIVE_LOOP=1 timeout 200 python scripts/gpu_late_time.py text_2k_nofull 2>&1 | grep -v amdgpu.ids | tee $O/late_time_native.txt
timeout 200 python scripts/gpu_late_time.py text_2k_nofull 2>&1 | grep -v amdgpu.ids | tee $O/late_time_stamps.txt
