#!/bin/bash
# Round 5, session a: microbenchmarks behind the mixing network's redesign (MFMA as an exact broadcast adder, the all-gather hand-off inside one XCD),
# the two cheap hand-off experiments of the round-4 review (one line per value|tag word, sleepy polls), and who-slows-whom between the stages.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5a; mkdir -p $O
export TMPDIR=/tmp
timeout 120 scripts/ubench/bin/mfma_adder > $O/mfma_adder.txt 2>&1; tail -25 $O/mfma_adder.txt
timeout 120 scripts/ubench/bin/allgather > $O/allgather.txt 2>&1; tail -26 $O/allgather.txt
for v in "" "CMX_MIXNET_PAD=1" "CMX_MIXNET_SLEEP=1" "CMX_MIXNET_PAD=1 CMX_MIXNET_SLEEP=1"; do
  echo "== mixnet alone: $v" | tee -a $O/mixnet_variants.txt
  ( export $v; timeout 120 python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids | head -9 ) | tee -a $O/mixnet_variants.txt
done
( export CMX_MIXNET_PAD=1 CMX_MIXNET_SLEEP=1; timeout 300 python -m pytest tests/test_gpu_mixnet.py -q -x -p no:cacheprovider 2>&1 | tail -3 ) | tee $O/mixnet_pad_tests.txt
timeout 300 python scripts/gpu_contention.py 2>&1 | grep -v amdgpu.ids | tee $O/contention.txt
( export CMX_LSTM_SLEEP=1 CMX_MIXNET_SLEEP=1 CMX_MIXNET_PAD=1; timeout 300 python scripts/gpu_contention.py lstm,mixnet 2>&1 | grep -v amdgpu.ids | tee $O/contention_sleepy.txt )
for v in "X=0" "CMX_MIXNET_PAD=1" "CMX_MIXNET_SLEEP=1" "CMX_LSTM_SLEEP=1"; do
  ( export $v; timeout 200 python bench.py --payload-bytes 262144 --steps 5 --warmup 1 --no-cpu-baseline > "$O/bench_256k_$v.json" 2> "$O/bench_256k_$v.err" )
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], round(d['value']), {k: (round(v, 2) if isinstance(v, float) else v) for k, v in d['stage_us_per_bit'].items() if k != 'note'}, d['verified']['identical_to_reference_file'])" "$O/bench_256k_$v.json" "$v" 2>&1 | cut -c1-300 | tee -a $O/bench_ab.txt
done
