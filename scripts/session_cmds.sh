O=gpurun_out/enwik8; mkdir -p $O
timeout 6500 python scripts/gpu_stage_hashes.py --bytes 100000000 --finish --out $O/hashes_100m_engine.txt 2>&1 | grep -v amdgpu.ids | tee $O/run_100m.log
