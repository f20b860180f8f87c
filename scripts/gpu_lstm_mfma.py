#!/usr/bin/env python3
"""The LSTM stage in tolerance mode for a counter pass: 450 bytes of a golden trace = five BPTT rounds whose weight-update contraction runs on the matrix cores
(cmx_lstm_bptt_acc_mfma). Meant to be run under `rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 --kernel-trace` (scripts/gpu_measure.sh ... or by hand);
prints the deviation from strict mode as tests/test_gpu_lstm.py does."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cmix_amd import engine as E  # noqa: E402

with np.load(os.path.join(ROOT, "tests", "golden", "text_2k_nofull.npz")) as z:
    g = {k: z[k] for k in z.files}
N = 450
res = []
for tol in (False, True):
    l = E.Lstm(g["vocab"], 0)
    l.set_tolerance(tol)
    o, _, _ = l.run(torch.from_numpy(np.ascontiguousarray(g["ppmd_probs"][1:N + 1], np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(g["stream"][:N], np.uint8)).cuda())
    torch.cuda.synchronize()
    res.append(o.cpu().numpy())
    l.close()
print("LSTM, 450 bytes, 5 BPTT rounds: tolerance vs strict max |dp| = %.3g, %.1f %% of the 256-way values bit-identical" %
      (float(np.abs(res[0].astype(np.float64) - res[1]).max()), 100 * float((res[0].view(np.uint32) == res[1].view(np.uint32)).mean())))
