#!/usr/bin/env python3
"""Time the fxcm stage on one GPU: us per bit of cmx_fxcm_roles_kernel (HIP events around cmx_fxcm_run, host parser time
reported separately), 1 KB chunks of enwik-like text after a warm-up. CMX_FXCM_SERIAL_MAPS=1 times the one-lane-per-map
variant; CMX_FXCM_PROFILE=1 prints thread 0's clocks per phase of each of the three roles.     python scripts/gpu_fxcm_time.py [nchunks]"""
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
    acc = (C.c_ulonglong * 128)()
    E.lib().cmx_fxcm_profile.argtypes = [C.c_void_p, C.c_void_p]
    if E.lib().cmx_fxcm_profile(fx.h, acc) == 0:
        nb = 8.0 * 1024 * (nchunks + 4)
        names = {0: ("M", ["touch (bucket lists, fetches)", "wave sync (load / store drain)", "maps run", "wave sync", "row stores", "store drain + publish"]),
                 1: ("U", ["units (match models, SSCMs, run map)", "lds barrier", "row stores", "store drain + publish"]),
                 2: ("X", ["wait M/U rows", "lds barrier", "gather inputs + trainers + APM updates", "full barrier (store drain)", "phase 2 (selectors)", "phases 3-4 (dots)",
                           "phase 5 (final mixers + APM chain)", "row stores"])}
        for r in range(3):
            nm, ph = names[r]
            vals = [acc[16 * r + k] / nb for k in range(len(ph))]
            print("role %s: %6.0f clk/bit |" % (nm, sum(vals)), "  ".join("%s %.0f" % (a_, b_) for a_, b_ in zip(ph, vals)))
        print("role X per wave, clk/bit: top block", " ".join("%5.0f" % (acc[48 + w] / nb) for w in range(8)), "| phase 3", " ".join("%5.0f" % (acc[56 + w] / nb) for w in range(8)))
        print("role X wave 7 (APM), clk/bit: cell update %.0f, contexts %.0f, row fetch issue %.0f" % (acc[40] / nb, acc[41] / nb, acc[42] / nb))
        print("role M by bit position (bpos of the update = position of the NEXT bit; lookups at 0, 2, 5), clk per bit of that position:")
        for bp in range(8):
            vals = [acc[64 + 8 * bp + k] / (nb / 8) for k in range(6)]
            print("  bpos %d: %6.0f |" % (bp, sum(vals)), "  ".join("%.0f" % v for v in vals))
