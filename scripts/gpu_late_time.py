#!/usr/bin/env python3
"""Time the decoder's form of the engine (late-bit protocol) on a replayed golden trace: python scripts/gpu_late_time.py [name] [nbytes]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if os.environ.get("CMX_LATE_NATIVE_LOOP") != "1":
    os.environ["CMX_LATE_PULL"] = "1"   # the per-stage time stamps are taken relative to the RELAY's "step is on the device" stamp: the relay wave runs (round 6's default is the host push)
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
if os.environ.get("CMX_LATE_NATIVE_LOOP") == "1":   # the loop in the library: no per-call Python overhead
    t0 = time.time()
    p = pipe.late_replay(bits)
    dt = time.time() - t0
    ms, n = pipe.late_host_ms()
    print("late replay %s (native loop): %d bytes in %.3f s = %.1f us/byte (%.1f us/bit); p mismatches %d" % (name, nb, dt, 1e6 * dt / nb, 1e6 * dt / len(bits), int((p.view(np.uint32) != pf[:len(bits)].view(np.uint32)).sum())))
    print("  host thread, us/byte:", {k: round(1000 * v / nb, 1) for k, v in ms.items()})
    pipe.late_stop(); pipe.close(); sys.exit(0)
bad = 0
marks = []
import ctypes as C
L = E.lib()
L.cmx_pipeline_late_debug_times.argtypes = [C.c_void_p, C.c_void_p]
NAMES = ["ctx", "bm0", "bm1", "bm2", "fx", "p8", "cm2a", "cm2b", "cm2c", "fam", "lanes", "dmc", "brk", "lstm", "known"]
samples = {}   # bit position -> list of dicts
t0 = time.time()
for t in range(len(bits)):
    p = np.float32(pipe.late_predict())
    bad += int(p.view(np.uint32) != pf[t].view(np.uint32))
    if t >= 64 and t % 5 == 0 and (t % 4096) >= 16:   # sample: every counter stands at row t now; times relative to the relay's "step t is on the device"
        tv = np.zeros(48, np.uint32)
        if L.cmx_pipeline_late_debug_times(pipe.h, tv.ctypes.data) == 0:
            t_known = int(tv[2 * 14 + 1])
            d = {NAMES[i]: ((int(tv[2 * i + 1]) - t_known) & 0xFFFFFFFF) / 100.0 for i in range(12)}
            for k, nm in ((1, "p_out"), (2, "row_complete"), (3, "sums_there"), (4, "layer1_done"), (5, "helpers_fed"), (6, "lstm_byte_seen"), (7, "lstm_h0"), (8, "lstm_h1"), (9, "lstm_dist")):
                d[nm] = ((int(tv[30 + k]) - t_known) & 0xFFFFFFFF) / 100.0
            if t & 7:   # the LSTM's stamps belong to the byte's first bit (the step that followed the completed byte)
                for nm in ("lstm_byte_seen", "lstm_h0", "lstm_h1", "lstm_dist"):
                    d[nm] = 0.0
            if all(v < 1e5 for v in d.values()):
                samples.setdefault(t & 7, []).append(d)
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
print("  device: us from 'the step's inputs are on the device' (relay) to each stage's row / the mixing network's milestones, mean by bit position (sampled bits):")
for bp in sorted(samples):
    keys = list(samples[bp][0].keys())
    print("   bit %d (%3d samples): " % (bp, len(samples[bp])) + "  ".join("%s %.1f" % (k, np.mean([x[k] for x in samples[bp]])) for k in keys))
