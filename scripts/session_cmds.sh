O=gpurun_out/s5; mkdir -p $O
timeout 1500 python scripts/gpu_stage_hashes.py --bytes 16777216 --out $O/hashes_16m_engine.txt 2>&1 | grep -v amdgpu.ids | tee $O/hashes_16m.log
