#!/usr/bin/env python3
"""Golden for the PPMd host stage with a 1 MB arena (reference PPMD::PPMD(25, 1, ...) stand-alone through
oracle/ref_harness.cpp): fills the whole arena within ~54 KB of text, so the allocator (free lists, splitting,
expand/shrink, rescaling) is exercised up to the brim. The reference build itself segfaults when the arena is
exhausted (its cut-off/restore path), so the trace stops 4 KB short of that point; the engine must match up to
there and then report exhaustion instead of crashing. Stored per byte: the probability of the byte that follows
and a weighted checksum of the whole distribution (full distributions would be 50 MB)."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from cmix_amd import synth
    N = 50000
    data = np.frombuffer(synth.enwik_like(60000, 5), np.uint8)
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libcmixref.so"))
    L.ref_ppmd_create.argtypes = [C.c_int, C.c_int, C.c_void_p]
    L.ref_ppmd_update.argtypes = [C.c_int, C.c_void_p]
    v = np.ones(256, np.uint8)
    L.ref_ppmd_create(25, 1, v.ctypes.data)
    out = np.empty(256, np.float32)
    w = np.arange(1, 257, dtype=np.float64)
    p_next = np.empty(N, np.float32)
    chk = np.empty(N, np.float64)
    for i in range(N):
        L.ref_ppmd_update(int(data[i]), out.ctypes.data)
        p_next[i] = out[data[i + 1]]
        chk[i] = float((out.astype(np.float64) * w).sum())
    path = os.path.join(ROOT, "tests", "golden", "ppmd_1mb_50k.npz")
    np.savez_compressed(path, stream=data, p_next=p_next, chk=chk)
    print("wrote", path, os.path.getsize(path))
