#!/bin/bash
# Round-3 session V: the default bench (1 MiB shard) in the mixing network's tolerance mode
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
O=gpurun_out/r3v; mkdir -p $O
export TMPDIR=/tmp
CMX_MIXNET_TOLERANCE=1 timeout 600 python bench.py --no-cpu-baseline > $O/bench_1m_tolerance.json 2> $O/bench_1m_tolerance.err
cut -c1-260 $O/bench_1m_tolerance.json; tail -2 $O/bench_1m_tolerance.err
