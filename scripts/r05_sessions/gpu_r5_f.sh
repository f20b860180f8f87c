#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
R="${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/r5f; mkdir -p $O
export TMPDIR=/tmp
for v in "X=0" "CMX_MIXNET_XCD=7"; do
  mkdir -p $O/prof_$v
  ( export $v; cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$O/prof_$v -o pipe -- python $R/bench.py --payload-bytes 131072 --steps 4 --warmup 1 --no-cpu-baseline > $R/$O/bench_$v.json 2> $R/$O/prof_$v.err )
  for f in $(find $O/prof_$v -name '*kernel_stats*.csv'); do grep -v "at::native\|rocclr" $f | head -40 > $O/kernel_stats_$v.csv; done
  echo "== $v"; cut -d, -f1-4 $O/kernel_stats_$v.csv | cut -c1-120 | head -32
done
