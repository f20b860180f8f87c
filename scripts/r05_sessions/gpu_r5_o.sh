#!/bin/bash
# Round 5, session o: the 8 MiB stream through the ROUND-4 build of the library (old_r04/, built from commit ea41fae) -- is the 2-byte difference to the
# reference binary's 8 MiB file (profiles/r05_long_run_8m.json) this round's, or was it there all along (round 4 stopped at 4 MiB)?
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
O=gpurun_out/r5o; mkdir -p $O
R=$(pwd)
( cd old_r04 && timeout 900 python scripts/gpu_long_run.py --bytes 8388608 --golden $R/tests/golden --out $R/$O/long_run_8m_r04_build.json 2>&1 | grep -v amdgpu.ids | tail -3 ) | tee $O/long_run_8m_r04_build.txt
