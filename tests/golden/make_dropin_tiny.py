#!/usr/bin/env python3
"""Tiny files for the drop-in test (tests/golden/dropin_tiny.npz): 0, 1, 2, 17 and 100 bytes through `cmix -c` and 0 and 1 byte through
`cmix -n` -- shorter than a block header, than a byte of LSTM history, than one BPTT block -- with the files the UNMODIFIED
reference binary writes.    python tests/golden/make_dropin_tiny.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
CASES = [("c0", "-c", b""), ("n0", "-n", b""), ("c1", "-c", b"A"), ("c2", "-c", b"ab"), ("c17", "-c", b"hello, hello, wor"), ("c100", "-c", (b"the quick brown fox. " * 5)[:100]), ("n1", "-n", b"\x00")]

if __name__ == "__main__":
    from make_dropin_vectors import run
    out = {}
    for name, mode, p in CASES:
        f = run(mode, [("in", p)])
        print(name, len(p), "->", len(f), "bytes")
        out[name + "_payload"] = np.frombuffer(p, np.uint8)
        out[name + "_file"] = np.frombuffer(f, np.uint8)
    np.savez_compressed(os.path.join(HERE, "dropin_tiny.npz"), **out)
