#!/bin/bash
# Round 5, session g: fxcm role X with register-resident mixer rows (no store drain); the one-XCD mixing network with every LSTM kernel leaving that XCD.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5g; mkdir -p $O
( timeout 600 python -m pytest tests/test_zgpu_stage_fxcm.py -q -x -p no:cacheprovider 2>&1 | tail -4 ) | tee $O/fxcm_tests.txt
( export CMX_FXCM_PROFILE=1; timeout 200 python scripts/gpu_fxcm_time.py 16 2>&1 | grep -v amdgpu.ids ) | tee $O/fxcm_time.txt
for v in "X=0" "CMX_MIXNET_XCD=7"; do
  n=$(echo $v | tr ' ' '_')
  ( export $v; timeout 200 python bench.py --payload-bytes 262144 --steps 5 --warmup 1 --no-cpu-baseline > "$O/bench_256k_$n.json" 2> "$O/bench_256k_$n.err" )
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d['stage_us_per_bit'].items() if k != 'note'}, d['verified']['sha256'][:16])" "$O/bench_256k_$n.json" "$n" 2>&1 | cut -c1-400 | tee -a $O/bench_ab.txt
  tail -2 "$O/bench_256k_$n.err"
done
