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

if os.environ.get("CMX_P8MIX_PROFILE"):
    import ctypes as C
    acc = (C.c_ulonglong * 56)()
    E.lib().cmx_p8stage_mix_profile.argtypes = [C.c_void_p, C.c_void_p]
    if E.lib().cmx_p8stage_mix_profile(st.h, acc) == 0:
        nb = 8192.0 * (N + 2)
        for w in range(7):
            print("mixer wave %d clocks/bit by phase (->B1 ->B2 ->B3 ->B4 tail):" % w, " ".join("%6.0f" % (acc[8 * w + i] / nb) for i in range(5)), "| total %.0f" % (sum(acc[8 * w + i] for i in range(5)) / nb))
