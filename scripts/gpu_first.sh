#!/bin/bash
# first GPU contact: parity tests + a quick timing of the mixnet kernel
set -x
cd "$GRAFT_REPO_ROOT" 2>/dev/null || true
mkdir -p gpurun_out
rocminfo | grep -m2 gfx
timeout 900 python -m pytest tests -q -m gpu 2>&1 | tail -25
timeout 300 python - <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from conftest import synth_mixnet_inputs
from cmix_amd import engine as E
T = 8192
probs, sel, bits = synth_mixnet_inputs(T, seed=1)
net = E.MixNet(0)
dp = torch.from_numpy(probs).cuda(); ds = torch.from_numpy((sel & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)).cuda(); db = torch.from_numpy(bits).cuda()
for rep in range(3):
    t0 = time.time(); net.run(dp, ds, db); torch.cuda.synchronize(); dt = time.time() - t0
    print('rep', rep, 'wall %.1f ms' % (dt * 1e3), 'kernel %.1f ms' % net.last_kernel_ms(), 'us/bit %.2f' % (net.last_kernel_ms() * 1e3 / T))
PY
