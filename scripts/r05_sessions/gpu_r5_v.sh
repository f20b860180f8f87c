#!/bin/bash
# Round 5, session v: the pipeline's sub-chunk size (bytes per submit: launches per byte, fill / drain) on the 256 KB bench stream
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5v; mkdir -p $O
for sc in 8192 4096 16384; do
  timeout 100 python bench.py --payload-bytes 262144 --steps 4 --warmup 1 --no-cpu-baseline --sub-chunk $sc > $O/bench_256k_sc$sc.json 2> $O/bench_256k_sc$sc.err
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), {k: round(v, 2) for k, v in d['stage_us_per_bit'].items() if k != 'note'}, d['verified']['sha256'][:16])" $O/bench_256k_sc$sc.json $sc | tee -a $O/subchunk.txt
done
