#!/bin/bash
# rocprofv3 kernel stats of the drop-in build compressing a 2000-byte text (per-bit surface)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out/prof_dropin
export TMPDIR=/tmp
python - <<'PY'
import numpy as np
v = np.load("tests/golden/dropin_vectors.npz")
open("/tmp/in.txt", "wb").write(v["text_c_payload"].tobytes())
PY
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof_dropin" -o dropin -- "$GRAFT_REPO_ROOT/oracle/_ref/cmix_hybrid" -c /tmp/in.txt /tmp/out.cmix > "$GRAFT_REPO_ROOT/gpurun_out/prof_dropin.out" 2>&1
cd "$GRAFT_REPO_ROOT"
find gpurun_out/prof_dropin -name "*kernel_stats*" | head -1 | xargs -I{} cp {} gpurun_out/dropin_kernel_stats.csv
cat gpurun_out/dropin_kernel_stats.csv | head -30
