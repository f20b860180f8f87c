#!/bin/bash
# Round 5, session d: re-runs in pieces (CMX_MIXNET_RERUN4) and the one-XCD placement with the LSTM's block kernels leaving that XCD free.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5d; mkdir -p $O
for v in "CMX_MIXNET_RERUN4=1" "CMX_MIXNET_RERUN4=1 CMX_MIXNET_XCD=7"; do
  echo "== $v" | tee -a $O/mixnet_variants.txt
  ( export $v; timeout 300 python -m pytest tests/test_gpu_mixnet.py -q -x -p no:cacheprovider 2>&1 | tail -4 ) | tee -a $O/mixnet_variants.txt
  ( export $v; timeout 120 python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids | head -9 ) | tee -a $O/mixnet_variants.txt
done
( export CMX_MIXNET_XCD=7; timeout 200 python -m pytest tests/test_gpu_lstm.py -q -x -p no:cacheprovider 2>&1 | tail -3 ) | tee $O/lstm_avoid_tests.txt
for v in "CMX_MIXNET_XCD=7" "CMX_MIXNET_XCD=7 CMX_MIXNET_RERUN4=1" "CMX_MIXNET_RERUN4=1"; do
  n=$(echo $v | tr ' ' '_')
  ( export $v; timeout 200 python bench.py --payload-bytes 262144 --steps 5 --warmup 1 --no-cpu-baseline > "$O/bench_256k_$n.json" 2> "$O/bench_256k_$n.err" )
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d['stage_us_per_bit'].items() if k != 'note'}, d['verified']['sha256'][:16])" "$O/bench_256k_$n.json" "$n" 2>&1 | cut -c1-400 | tee -a $O/bench_ab.txt
  tail -2 "$O/bench_256k_$n.err"
done
