O=gpurun_out/vote; mkdir -p $O
timeout 3550 python -u scripts/gpu_mixnet_vote.py --bytes 70000000 --extra 4 --seconds 3350 --out $O/vote.json > $O/vote.log 2>&1
tail -5 $O/vote.log | cut -c1-600
