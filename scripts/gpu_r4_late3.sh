#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
mkdir -p gpurun_out
{
  date
  python - <<'PY'
import numpy as np, os, subprocess, time, tempfile
v = np.load("tests/golden/dropin_vectors.npz")
d = tempfile.mkdtemp()
open(d + "/in", "wb").write(v["raw_n_file"].tobytes())
t0 = time.time()
r = subprocess.run(["oracle/_ref/cmix_dropin", "-d", d + "/in", d + "/out"], env=dict(os.environ, CMX_TIMING="1"), capture_output=True, text=True, timeout=100)
print("wall %.1f s rc %d" % (time.time() - t0, r.returncode))
print("\n".join(l for l in r.stderr.replace("\r", "\n").split("\n") if "cmx timing" in l))
PY
  date
} > gpurun_out/r4_late3.log 2>&1
cat gpurun_out/r4_late3.log
