#!/bin/bash
# round 4: the decoder's form of the engine (late-bit protocol) on the MI355X -- replay parity vs reference traces, round trips, cmix_dropin -d
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  echo "== late replay / round trips"; date
  timeout 1200 python -m pytest tests/test_gpu_late.py -q -p no:cacheprovider 2>&1 | tail -80
  echo "== cmix_dropin -d"; date
  timeout 900 python -m pytest tests/test_gpu_dropin.py -q -p no:cacheprovider -k "decodes" 2>&1 | tail -40
  echo "== tolerance test"; date
  timeout 300 python -m pytest tests/test_gpu_mixnet.py -q -s -p no:cacheprovider -k "tolerance" 2>&1 | tail -8
  date
} > gpurun_out/r4_late.log 2>&1
tail -c 3000 gpurun_out/r4_late.log
