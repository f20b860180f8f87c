import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

GOLDEN = os.path.join(ROOT, "tests", "golden")
GOLDEN_BIG = os.path.join(ROOT, "oracle", "_ref", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name, big=False):
    path = os.path.join(GOLDEN_BIG if big else GOLDEN, name + ".npz")
    if not os.path.exists(path):
        pytest.skip(f"{path} not present")
    with np.load(path) as z:  # materialise: NpzFile re-inflates an array on every access
        return {k: z[k] for k in z.files}


def bits_equal(a, b):
    a = np.ascontiguousarray(a, np.float32).view(np.uint32)
    b = np.ascontiguousarray(b, np.float32).view(np.uint32)
    return a == b


def synth_mixnet_inputs(T, seed, n_ctx_bits=6):
    """Seeded stand-in for the upstream stages: model outputs on paq8/fxcm's k/4095 grid plus a
    few off-grid floats, and selector keys with enwik8-like locality (some mixers keyed per bit,
    some per byte, one constant)."""
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 4096, (T, 2078), dtype=np.int64)
    # concentrate half of the columns around confident predictions like real models do
    conf = rng.random((T, 2078)) < 0.5
    k = np.where(conf, np.where(rng.random((T, 2078)) < 0.5, k % 200, 4095 - (k % 200)), k)
    probs = (k.astype(np.float32) * np.float32(1.0 / 4095)).astype(np.float32)
    probs[:, 2025:2078] = rng.random((T, 53), dtype=np.float32)
    probs[:, 432:434] = 0.5
    bits = rng.integers(0, 2, T, dtype=np.uint8)
    sel = np.zeros((T, 47), np.uint64)
    byte_ix = np.arange(T) // 8
    lbc = np.ones(T, np.uint64)
    for t in range(T):
        if t % 8:
            lbc[t] = lbc[t - 1] * 2 + bits[t - 1]
    bytectx = rng.integers(0, 1 << n_ctx_bits, (T // 8 + 1, 47), dtype=np.uint64)
    for m in range(47):
        per_bit = m in (0, 1, 2, 3, 4, 5, 16, 19, 22, 23, 28, 29, 30, 43, 44, 45)
        base = bytectx[byte_ix, m] * np.uint64(1 + (m % 7) * 977)
        sel[:, m] = (base << np.uint64(8)) + lbc if per_bit else base
    sel[:, 8] = 0
    sel[:, 26] = 0
    sel[:, 27] = 0
    sel[:, 46] = 0
    return probs, sel, bits
