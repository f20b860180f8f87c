#!/usr/bin/env python3
"""A binary file for the drop-in test (tests/golden/dropin_binary.npz): random bytes, fixed-length records, zero / 0xFF runs, a
ramp -- nothing the reference's detector takes for text or x86 code, so `cmix -c` codes it as DEFAULT blocks -- and the `.cmix`
file the UNMODIFIED reference binary writes for it.    python tests/golden/make_dropin_binary.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def payload():
    r = np.random.default_rng(99)
    rec = b"".join(bytes([i & 255, (i >> 8) & 255, 7, 0]) + bytes(r.integers(0, 4, 20, dtype=np.uint8)) for i in range(200))   # 24-byte records
    return (bytes(r.integers(0, 256, 4000, dtype=np.uint8)) + rec + b"\x00" * 700 + b"\xff" * 300 + bytes(range(256)) * 6 +
            bytes(r.integers(128, 256, 3000, dtype=np.uint8)))


if __name__ == "__main__":
    from make_dropin_vectors import run
    p = payload()
    f = run("-c", [("in", p)])
    print(len(p), "->", len(f), "bytes")
    np.savez_compressed(os.path.join(HERE, "dropin_binary.npz"), payload=np.frombuffer(p, np.uint8), cmix_file=np.frombuffer(f, np.uint8))
