#!/bin/bash
# Round-3 measurement session: the whole GPU suite, smoke, the default bench line (1 MiB rich shard, verified against the reference
# binary's file, CPU reference beside it), rocprofv3 kernel stats and HBM-traffic counters of the bench command (one counter per pass,
# PMC + kernel-trace only), mixing-network phase timers and speculation statistics. Outputs under gpurun_out/r3m/ (copied to profiles/r03_*).
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3m; mkdir -p $O/prof $O/pmc
export TMPDIR=/tmp
if [ "$CMX_SKIP_TESTS" != "1" ]; then
( time timeout 1700 python -m pytest tests -m gpu -q --durations=8 2>&1 | tail -16 ) 2>&1 | tee $O/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee $O/smoke.txt
fi
timeout 900 python bench.py > $O/bench_1m.json 2> $O/bench_1m.err; cut -c1-300 $O/bench_1m.json; tail -2 $O/bench_1m.err
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof -o pipe -- python $GRAFT_REPO_ROOT/bench.py --payload-bytes 262144 --steps 8 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/prof_bench_256k.json 2> $GRAFT_REPO_ROOT/$O/prof.err )
for f in $(find $O/prof -name '*kernel_stats*.csv'); do grep -v "at::native\|rocclr" $f | head -28 > $O/bench_256k_kernel_stats.csv; done
cut -c1-160 $O/bench_256k_kernel_stats.csv
for c in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && timeout -k 5 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/$O/pmc -o pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --payload-bytes 131072 --steps 4 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$O/pmc_$c.out 2> $GRAFT_REPO_ROOT/$O/pmc_$c.err ; echo "rocprofv3 $c rc=$?" )
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(set(glob.glob('gpurun_out/r3m/pmc/**/*counter_collection*.csv', recursive=True))):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '').split('(')[0][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k in agg:
        if k.startswith('cmx_') or k.startswith('_Z'):
            for c, v in agg[k].items():
                out.setdefault(k, {})[c] = {"sum": v, "launches": n[(k, c)]}
# bytes of stream the profiled command pushed through the stages: the timed stream + the warm-up engine's, as the bench line of that run reports them
b = json.load(open('gpurun_out/r3m/pmc_FETCH_SIZE.out'))
out["_meta"] = {"stream_bytes_processed": b["config"]["stream_bytes"] + b["config"]["warmup_stream_bytes"], "command": "python bench.py --payload-bytes 131072 --steps 4 --warmup 1 --no-cpu-baseline",
                "unit": "KB summed over the launches (rocprofv3 FETCH_SIZE / WRITE_SIZE); bench.py uses 2 x FETCH_SIZE + WRITE_SIZE"}
json.dump(out, open('gpurun_out/r3m/pmc_bench.json', 'w'), indent=1)
for k, v in out.items():
    if k != "_meta":
        print(k, {c: (round(x["sum"] / 1e3, 1), x["launches"]) for c, x in v.items()})
PY
python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids > $O/mixnet_phases.txt; head -12 $O/mixnet_phases.txt
CMX_MIXNET_DBG=4 python scripts/gpu_prof.py 4096 2>&1 | grep -v amdgpu.ids > $O/mixnet_phases_tail.txt
CMX_FXCM_PROFILE=1 timeout 300 python scripts/gpu_fxcm_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/fxcm_roles_phases.txt
CMX_P8FAM_PROFILE=1 timeout 300 python scripts/gpu_p8stage_time.py 16 2>&1 | grep -v amdgpu.ids | tee $O/p8_fam_phases.txt
timeout 200 python scripts/gpu_create_time.py 2>&1 | grep -v amdgpu.ids | tee $O/create_time.txt
