#!/bin/bash
# HBM-traffic counters of the bench command (separate pass: PMC with kernel-trace only)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
( cd /tmp && rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc -- python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc_bench.json 2> $GRAFT_REPO_ROOT/gpurun_out/pmc.err )
tail -2 gpurun_out/pmc.err
ls gpurun_out/pmc
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc/*counter_collection*.csv'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')[:40]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k in agg:
        if k.startswith('cmx_'):
            print(k, {c: (v, n[(k, c)]) for c, v in agg[k].items()})
PY
