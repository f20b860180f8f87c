O=gpurun_out/s2; mkdir -p $O
timeout 600 python scripts/gpu_stage_hashes.py --bytes 8388608 --stop-block 25 --selfcheck --out $O/hashes_26blk.txt 2>&1 | grep -v amdgpu.ids | tee $O/selfcheck.txt
timeout 200 python scripts/gpu_late_time.py text_2k_nofull 2>&1 | grep -v amdgpu.ids | tee $O/late_time_stamps.txt
