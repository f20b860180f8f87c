#!/bin/bash
# HBM-traffic counters of the bench command, one counter per pass (PMC + kernel-trace only), 150 s cap each
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/pmcb
export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && timeout -k 5 150 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmcb -o pmc_$c -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmcb_$c.out 2> $GRAFT_REPO_ROOT/gpurun_out/pmcb_$c.err ; echo "rocprofv3 $c rc=$?" )
done
python - <<'PY'
import csv, glob, collections, json
out = {}
for f in sorted(set(glob.glob('gpurun_out/pmcb/**/*counter_collection*.csv', recursive=True))):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '').split('(')[0][:48]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k in agg:
        if k.startswith('cmx_'):
            for c, v in agg[k].items():
                out.setdefault(k, {})[c] = {"sum_kb": v, "launches": n[(k, c)]}
print(json.dumps(out, indent=1))
json.dump(out, open('gpurun_out/pmc_bench.json', 'w'), indent=1)
PY
