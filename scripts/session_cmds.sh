O=gpurun_out/enwik8b; mkdir -p $O
timeout 3500 python -u scripts/gpu_stage_hashes.py --bytes 100000000 --stop-block 767 --finish --out $O/hashes_100m_head768_engine.txt > $O/run.log 2>&1
tail -5 $O/run.log
