#!/bin/bash
# Round-2 measurement session: the whole GPU suite, smoke, the default bench line (1 MiB, verified against the reference
# binary's file, CPU baseline beside it), rocprofv3 kernel stats and HBM-traffic counters of the bench command (one counter
# per pass, PMC + kernel-trace only), stage timings in isolation. Outputs under gpurun_out/r2/ (copied to profiles/r02_*).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r2; mkdir -p $O/prof $O/pmc
export TMPDIR=/tmp GPU_MAX_HW_QUEUES=16
( time timeout 1500 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 ) 2>&1 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
if [ "$CMX_SKIP_BENCH" != "1" ]; then
  timeout 900 python bench.py > $O/bench_1m.json 2> $O/bench_1m.err; cut -c1-400 $O/bench_1m.json; tail -2 $O/bench_1m.err
fi
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o pipe -- python $GRAFT_REPO_ROOT/bench.py --payload-bytes 262144 --steps 8 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench_256k.json 2> $GRAFT_REPO_ROOT/$O/prof.err )
for f in $(find $O/prof -name '*kernel_stats*.csv'); do grep -v "at::native\|rocclr" $f | head -24 > $O/bench_256k_kernel_stats.csv; done
cat $O/bench_256k_kernel_stats.csv | cut -c1-150
for c in $( [ "$CMX_SKIP_PMC" = 1 ] || echo FETCH_SIZE WRITE_SIZE ); do
( cd /tmp && timeout -k 5 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -o pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --payload-bytes 65536 --steps 4 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_$c.out 2> $GRAFT_REPO_ROOT/$O/pmc_$c.err ; echo "rocprofv3 $c rc=$?" )
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(set(glob.glob('gpurun_out/r2/pmc/**/*counter_collection*.csv', recursive=True))):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '').split('(')[0][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k in agg:
        if k.startswith('cmx_') or k.startswith('_Z'):
            for c, v in agg[k].items():
                out.setdefault(k, {})[c] = {"sum": v, "launches": n[(k, c)]}
json.dump(out, open('gpurun_out/r2/pmc_bench_64k.json', 'w'), indent=1)
for k, v in out.items():
    print(k, {c: (round(x["sum"] / 1e3, 1), x["launches"]) for c, x in v.items()})
PY
python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids > $O/mixnet_phases.txt; head -3 $O/mixnet_phases.txt
python scripts/gpu_fxcm_time.py 16 2>&1 | grep -i "fxcm stage" | tee $O/fxcm_time.txt
python scripts/gpu_p8stage_time.py 16 2>&1 | grep "paq8 stage" | tee $O/p8stage_time.txt
python scripts/gpu_lstm_time.py 4000 2>&1 | grep "us/byte" | tee $O/lstm_time.txt
