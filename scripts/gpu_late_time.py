#!/usr/bin/env python3
"""Time the decoder's form of the engine (late-bit protocol) on a replayed golden trace: python scripts/gpu_late_time.py [name] [nbytes]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmix_amd import engine as E  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "text_2k_nofull"
with np.load(os.path.join(ROOT, "tests", "golden", name + ".npz")) as z:
    g = {k: z[k] for k in z.files}
nb = int(sys.argv[2]) if len(sys.argv) > 2 else len(g["stream"])
pipe = E.Pipeline(g["vocab"], 0, 4096)
pipe.enable_fxcm(None)
pipe.enable_paq8()
t0 = time.time()
pipe.late_start(0)
t_start = time.time() - t0
bits = np.ascontiguousarray(g["bits"], np.uint8)[:8 * nb]
pf = np.ascontiguousarray(g["p_final"], np.float32)
bad = 0
marks = []
t0 = time.time()
for t in range(len(bits)):
    p = np.float32(pipe.late_predict())
    bad += int(p.view(np.uint32) != pf[t].view(np.uint32))
    pipe.late_perceive(int(bits[t]))
    if (t & 1023) == 1023:
        marks.append(time.time())
dt = time.time() - t0
ms, n = pipe.late_host_ms()
pipe.late_stop()
pipe.close()
per = [round(1e6 * (b - a) / 128, 1) for a, b in zip([t0] + marks[:-1], marks)]
print("late replay %s: %d bytes in %.2f s = %.1f us/byte (%.1f us/bit); start %.2f s; p mismatches %d; us/byte per 128-byte window: %s" % (name, nb, dt, 1e6 * dt / nb, 1e6 * dt / len(bits), t_start, bad, per))
print("  host thread, us/byte:", {k: round(1000 * v / nb, 1) for k, v in ms.items()})
