#!/usr/bin/env python3
"""Time the fxcm stage on one GPU: us per bit of cmx_fxcm_chunk_kernel (HIP events around cmx_fxcm_run, host parser time
reported separately), 1 KB chunks of enwik-like text after a warm-up. CMX_FXCM_SERIAL_MAPS=1 times the one-lane-per-map
variant.     python scripts/gpu_fxcm_time.py [nchunks]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmix_amd import engine as E, synth  # noqa: E402

nchunks = int(sys.argv[1]) if len(sys.argv) > 1 else 64
C = 1024
data = np.frombuffer(synth.enwik_like(C * (nchunks + 4), 1000), np.uint8)
r = np.random.default_rng(1)
pr = torch.from_numpy(r.integers(1, 4096, 8 * C).astype(np.int16)).cuda()
ex = torch.from_numpy(r.integers(0, 256, 8 * C).astype(np.uint8)).cuda()
probs = torch.full((8 * C, 2078), 0.5, dtype=torch.float32, device="cuda")
fx = E.Fxcm(None, 0)
for i in range(4):
    fx.run(data[i * C:(i + 1) * C], pr, ex, probs)
fx.sync()
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nchunks)]
host = 0.0
for i in range(nchunks):
    ev[i][0].record()
    t0 = time.perf_counter()
    fx.run(data[(i + 4) * C:(i + 5) * C], pr, ex, probs)
    host += time.perf_counter() - t0
    ev[i][1].record()
fx.sync()
ms = [a.elapsed_time(b) for a, b in ev]
print("fxcm stage, %d x %d-byte chunks, serial_maps=%s: kernel+copy %.2f us/bit (min %.2f, max %.2f); host parse+launch %.2f us/byte" % (
    nchunks, C, os.environ.get("CMX_FXCM_SERIAL_MAPS", "0"), np.mean(ms) * 1e3 / (8 * C), np.min(ms) * 1e3 / (8 * C), np.max(ms) * 1e3 / (8 * C), host / (nchunks * C) * 1e6))

if os.environ.get("CMX_FXCM_PROFILE") == "1":
    import ctypes as C
    acc = (C.c_ulonglong * 64)()
    E.lib().cmx_fxcm_profile.argtypes = [C.c_void_p, C.c_void_p]
    if E.lib().cmx_fxcm_profile(fx.h, acc) == 0:
        nb = 8.0 * 1024 * (nchunks + 4)
        for w in range(8):
            print("wave %d clocks per bit by phase (1a work, 1a barrier, 1c, 2, 3, 4, 5):" % w, " ".join("%6.0f" % (acc[8 * w + i] / nb) for i in range(7)), "| 1c work %.0f | total %.0f" % (acc[8 * w + 7] / nb, sum(acc[8 * w + i] for i in range(8)) / nb))
