#!/usr/bin/env python3
"""Golden vectors for the arithmetic coder and the container header (tests/golden/coder_vectors.npz).

Two sources, both the UNMODIFIED reference:
  * its Encoder class compiled from src/coder/encoder.cpp by oracle/ref_coder.cpp (a replaying Predictor stands in
    for the real one): seeded (p, bits) sequences -> code bytes;
  * its command-line binary (oracle/_ref/cmix_O3, built by oracle/Makefile) compressing the payloads of the
    committed trace binary_64 (`-n`, no preprocessing): the whole output file, i.e. header + the code of that
    trace's p_final; and a 10 000-byte seeded payload (`-n`), of which only the 37 header bytes are kept (length
    and vocabulary bitmap, runner.cpp:34-52).

    python tests/golden/make_coder_vectors.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def seeded_cases():
    rng = np.random.default_rng(20260924)
    cases = {}
    # well-predicted bits: p close to the coded bit most of the time
    n = 20000
    bits = rng.integers(0, 2, n, dtype=np.uint8)
    conf = rng.random(n).astype(np.float32) ** 4
    p = np.where(bits == 1, 1 - 0.5 * conf, 0.5 * conf).astype(np.float32)
    cases["confident"] = (p, bits)
    # uniform p, independent bits (expands)
    cases["uniform"] = (rng.random(5000).astype(np.float32), rng.integers(0, 2, 5000, dtype=np.uint8))
    # the extremes Predict() can return: SSE/override clamp region, exact 0 / 1 / 0.5, subnormal-ish
    ext = np.array([0.0, 1.0, 0.5, 1e-4, 1 - 1e-4, 1.0 / 65534, 1 - 1.0 / 65534, 1e-7, 0.99999994, 1.5e-5,
                    0.25, 0.75], np.float32)
    p = np.tile(ext, 200)
    bits = (p >= 0.5).astype(np.uint8)
    bits[::7] ^= (p[::7] > 0) & (p[::7] < 1)   # surprises wherever they are codable
    cases["extremes"] = (p, bits)
    # float values whose product with 65534 sits next to an integer (rounding of the float product and sum matters)
    k = rng.integers(0, 65534, 4000)
    base = (k.astype(np.float64) / 65534).astype(np.float32)
    p = np.nextafter(base, np.where(rng.random(4000) < 0.5, 0, 1).astype(np.float32)).astype(np.float32)
    p = np.clip(p, 0, 1)
    cases["grid_edges"] = (p, rng.integers(0, 2, 4000, dtype=np.uint8))
    cases["empty"] = (np.zeros(0, np.float32), np.zeros(0, np.uint8))
    cases["one_bit"] = (np.array([0.9], np.float32), np.array([1], np.uint8))
    return cases


def run_binary(mode, payload):
    exe = os.path.join(ROOT, "oracle", "_ref", "cmix_O3")
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "in"), os.path.join(d, "out")
        with open(a, "wb") as f:
            f.write(payload)
        subprocess.run([exe, mode, a, b], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(b, "rb") as f:
            return f.read()


if __name__ == "__main__":
    from oracle import refharness as R
    out = {}
    for name, (p, bits) in seeded_cases().items():
        code = R.ref_encode(p, bits)
        assert (R.ref_decode(p, code) == bits).all() or name == "extremes"
        out[name + "_p"], out[name + "_bits"] = p, bits
        out[name + "_code"] = np.frombuffer(code, np.uint8)
        print(name, len(p), "bits ->", len(code), "bytes")
    g = np.load(os.path.join(HERE, "binary_64.npz"))
    out["binary_64_file"] = np.frombuffer(run_binary("-n", g["stream"][5:].tobytes()), np.uint8)
    from cmix_amd import synth
    payload = synth.enwik_like(10000, 5)
    out["header_10k_payload"] = np.frombuffer(payload, np.uint8)
    out["header_10k"] = np.frombuffer(run_binary("-n", payload)[:37], np.uint8)
    print("binary_64_file", len(out["binary_64_file"]), "bytes")
    np.savez_compressed(os.path.join(HERE, "coder_vectors.npz"), **out)
