O=gpurun_out/s3; mkdir -p $O
( time timeout 900 python -m pytest tests/test_wraps_and_thresholds.py tests/test_gpu_mixnet.py -x -q -m gpu -p no:cacheprovider 2>&1 | tail -15 ) > $O/wrap_and_mixnet_tests.txt 2>&1; tail -6 $O/wrap_and_mixnet_tests.txt
( time CMX_LONG=1 timeout 1500 python -m pytest tests/test_gpu_dropin.py -x -q -m gpu -p no:cacheprovider -k "4mib" 2>&1 | tail -15 ) > $O/dropin_4mib.txt 2>&1; tail -6 $O/dropin_4mib.txt
