#!/bin/bash
# One measurement session on the MI355X box (gpurun -- 'bash scripts/gpu_measure.sh [tag]'): the whole GPU suite, smoke, the default
# bench line (1 MiB rich shard, verified against the reference binary's file, CPU reference beside it), rocprofv3 kernel stats and the
# HBM-traffic counters of the bench command (one counter per pass, PMC + kernel-trace only), the decoder's time per bit.
# Outputs under gpurun_out/<tag>/; the summaries that are judged are copied to profiles/<rNN>_* by hand.
# CMX_SKIP_TESTS=1 skips the suite; CMX_SKIP_PMC=1 the counter passes; CMX_AB="VAR=1 ..." adds one 256 KB bench run per opt-in variant.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
R="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-measure}"
O=gpurun_out/$TAG; mkdir -p $O/prof $O/pmc
export TMPDIR=/tmp
if [ "$CMX_SKIP_TESTS" != "1" ]; then
( time timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=12 2>&1 | tail -40 ) > $O/pytest_gpu.txt 2>&1; tail -22 $O/pytest_gpu.txt
cp gpurun_out/decode_time.txt gpurun_out/config3_dict_time.txt gpurun_out/config4_silesia_time.txt $O/ 2>/dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
fi
timeout 600 python bench.py > $O/bench_1m.json 2> $O/bench_1m.err; cut -c1-400 $O/bench_1m.json; tail -2 $O/bench_1m.err
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof -o pipe -- python $R/bench.py --payload-bytes 262144 --steps 8 --warmup 1 --no-cpu-baseline --decode-bytes 0 > $R/$O/prof_bench_256k.json 2> $R/$O/prof.err )
for f in $(find $O/prof -name '*kernel_stats*.csv'); do grep -v "at::native\|rocclr" $f | head -30 > $O/bench_256k_kernel_stats.csv; done
cut -c1-160 $O/bench_256k_kernel_stats.csv | head -12
# A/B of opt-in kernel variants in the pipeline (256 KB, no CPU baseline): CMX_AB="CMX_MIXNET_SEG8=1 ..." (one variable assignment per variant)
for v in $CMX_AB; do
  ( export "$v"; timeout 200 python bench.py --payload-bytes 262144 --steps 5 --warmup 1 --no-cpu-baseline > "$O/bench_256k_$v.json" 2> "$O/bench_256k_$v.err" )
  python -c "import json,sys; d=json.load(open(sys.argv[1])); print(sys.argv[2], d['value'], d['stage_us_per_bit'])" "$O/bench_256k_$v.json" "$v" 2>&1 | cut -c1-300
done
if [ "$CMX_SKIP_PMC" != "1" ]; then
for c in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && timeout -k 5 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/$O/pmc -o pmc_$c -- python $R/bench.py --payload-bytes 131072 --steps 4 --warmup 1 --no-cpu-baseline --decode-bytes 0 > $R/$O/pmc_$c.out 2> $R/$O/pmc_$c.err ; echo "rocprofv3 $c rc=$?" )
done
# one occupancy pass (round-5 review 4 iv): waves launched, cycles the SQ was busy, cycles waves waited for any instruction, per kernel
( cd /tmp && timeout -k 5 300 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $R/$O/pmc -o pmc_SQ -- python $R/bench.py --payload-bytes 131072 --steps 4 --warmup 1 --no-cpu-baseline --decode-bytes 0 > $R/$O/pmc_SQ.out 2> $R/$O/pmc_SQ.err ; echo "rocprofv3 SQ rc=$?" )
python - "$O" <<'PY'
import csv, glob, collections, json, sys
O = sys.argv[1]
out = {}
for f in sorted(set(glob.glob(O + '/pmc/**/*counter_collection*.csv', recursive=True))):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '').split('(')[0][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k in agg:
        if k.startswith('cmx_') or k.startswith('_Z'):
            for c, v in agg[k].items():
                out.setdefault(k, {})[c] = {"sum": v, "launches": n[(k, c)]}
# bytes of stream the profiled command pushed through the stages: the timed stream + the warm-up engine's, as the bench line of that run reports them
b = json.load(open(O + '/pmc_FETCH_SIZE.out'))
out["_meta"] = {"stream_bytes_processed": b["config"]["stream_bytes"] + b["config"]["warmup_stream_bytes"], "command": "python bench.py --payload-bytes 131072 --steps 4 --warmup 1 --no-cpu-baseline",
                "unit": "KB summed over the launches (rocprofv3 FETCH_SIZE / WRITE_SIZE); bench.py uses 2 x FETCH_SIZE + WRITE_SIZE"}
json.dump(out, open(O + '/pmc_bench.json', 'w'), indent=1)
for k, v in out.items():
    if k != "_meta":
        print(k, {c: (round(x["sum"] / 1e3, 1), x["launches"]) for c, x in v.items()})
PY
fi
# the decoder's form: time per bit of a replayed 2 KB trace (native loop), and the device time stamps of its stages
CMX_LATE_NATIVE_LOOP=1 timeout 200 python scripts/gpu_late_time.py text_2k_nofull 2>&1 | grep -v amdgpu.ids | tee $O/late_time.txt
