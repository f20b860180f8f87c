#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5i; mkdir -p $O
for v in "X=0" "CMX_MIXNET_XCD=7" "CMX_MIXNET_XCD=7 CMX_LSTM_AVOID_XCD=9"; do
  n=$(echo $v | tr ' ' '_')
  ( export $v; timeout 200 python bench.py --payload-bytes 262144 --steps 5 --warmup 1 --no-cpu-baseline > "$O/bench_256k_$n.json" 2> "$O/bench_256k_$n.err" )
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d['stage_us_per_bit'].items() if k != 'note'}, d['verified']['sha256'][:16])" "$O/bench_256k_$n.json" "$n" 2>&1 | cut -c1-400 | tee -a $O/bench_ab.txt
done
