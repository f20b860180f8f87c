#!/bin/bash
# HBM-traffic counters of the mixing-network kernel (separate pass: PMC with kernel-trace only).
# Hard 100 s cap: an earlier attempt over bench.py aborted inside rocprofv3 and then hung.
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/pmc
export TMPDIR=/tmp
# one counter per pass: FETCH_SIZE and WRITE_SIZE together exceed what one pass can collect on gfx950
for c in FETCH_SIZE WRITE_SIZE; do
( cd /tmp && timeout -k 5 100 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc -o pmc_$c -- python $GRAFT_REPO_ROOT/scripts/gpu_prof.py 4096 > $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.out 2> $GRAFT_REPO_ROOT/gpurun_out/pmc_$c.err ; echo "rocprofv3 $c rc=$?" )
done
cp gpurun_out/pmc_FETCH_SIZE.out gpurun_out/pmc.out
grep kernel gpurun_out/pmc.out | head -3
python - <<'PY'
import csv, glob, collections
for f in glob.glob('gpurun_out/pmc/**/*counter_collection*.csv', recursive=True) + glob.glob('gpurun_out/pmc/*counter_collection*.csv'):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')[:40]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); n[(k, r['Counter_Name'])] += 1
    for k in agg:
        if k.startswith('cmx_'):
            print(k, {c: (v, n[(k, c)]) for c, v in agg[k].items()})
PY
