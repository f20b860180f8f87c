#!/usr/bin/env python3
"""Time the paq8 stage on one GPU: us per bit over 4 x 1 KB chunks of enwik-like text after a 2 KB warm-up (device span
from HIP events on the stage's stream, and wall time including the host front end)."""
import os
import sys
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmix_amd import engine as E, synth  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 4
data = synth.enwik_like((N + 2) * 1024, 1000)
st = E.P8Stage(0)
st.run(data[:2048])
st.sync()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter()
a.record()
for i in range(2, N + 2):
    st.run(data[1024 * i:1024 * (i + 1)])
b.record()
st.sync()
print("paq8 stage %d x 1 KB: %.2f us/bit device span, %.2f us/bit wall" % (N, a.elapsed_time(b) * 1e3 / (8192 * N), (time.perf_counter() - t0) * 1e6 / (8192 * N)))

if os.environ.get("CMX_P8FAM_PROFILE"):
    import ctypes as C
    acc = (C.c_ulonglong * 128)()
    E.lib().cmx_p8stage_profile.argtypes = [C.c_void_p, C.c_void_p]
    if E.lib().cmx_p8stage_profile(st.h, acc) == 0:
        print("family kernel, thread 0, clocks per step of each bit position: per-step values | phase 1 | barrier | run or rounds | rest ; share of steps in rounds ; instances walked per step")
        tot = 0.0
        for bp in range(8):
            n = max(1, acc[80 + bp])
            v = [acc[8 * bp + k] / n for k in range(5)]
            tot += sum(v)
            print("  bp %d: %6.0f |" % (bp, sum(v)), " ".join("%6.0f" % x for x in v), "; rounds %.3f ; walks %.3f" % (acc[64 + bp] / n, acc[72 + bp] / n))
        print("  mean %.0f clk/bit" % (tot / 8))
