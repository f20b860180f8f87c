O=gpurun_out/s4; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_late.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 ) > $O/late_tests.txt 2>&1; tail -8 $O/late_tests.txt
timeout 200 python scripts/gpu_late_time.py text_2k_nofull 2>&1 | grep -v amdgpu.ids | tee $O/late_time_stamps.txt
CMX_LATE_NATIVE_LOOP=1 timeout 200 python scripts/gpu_late_time.py text_2k_nofull 2>&1 | grep -v amdgpu.ids | tee $O/late_time_native.txt
CMX_LATE_LSTM_PER_BYTE=1 CMX_LATE_NATIVE_LOOP=1 timeout 200 python scripts/gpu_late_time.py text_2k_nofull 2>&1 | grep -v amdgpu.ids | tee $O/late_time_native_per_byte.txt
( time timeout 900 python -m pytest tests/test_gpu_lstm.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -5 ) > $O/lstm_tests.txt 2>&1; tail -4 $O/lstm_tests.txt
