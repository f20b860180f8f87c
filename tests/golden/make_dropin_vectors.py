#!/usr/bin/env python3
"""Whole-file fixtures for the drop-in test (tests/golden/dropin_vectors.npz): seeded payloads and the `.cmix`
files the UNMODIFIED reference binary (oracle/_ref/cmix_O3, built by oracle/Makefile) writes for them.

    case      command                         what it exercises
    raw_n     cmix -n in out                  no preprocessing (DEFAULT block), all-true vocabulary
    text_c    cmix -c in out                  preprocessor::Encode (type detection), no dictionary
    dict_c    cmix -c dict in out             WRT dictionary transform + Predictor::Pretrain over the dictionary
    text12k_c cmix -c in out                  >= 10 000 bytes: vocabulary bitmap in the header, LSTM sized by the
                                              real vocabulary (V < 256), 120 BPTT/Adam rounds
    text50k_c cmix -c in out                  50 000 bytes (only the SHA-256 and the size of the reference's file are
                                              kept): 500 BPTT rounds, mixer rows past their first weight decay

    python tests/golden/make_dropin_vectors.py
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
EXE = os.path.join(ROOT, "oracle", "_ref", "cmix_O3")


def run(args, files):
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for name, data in files:
            p = os.path.join(d, name)
            with open(p, "wb") as f:
                f.write(data)
            paths.append(p)
        out = os.path.join(d, "out")
        subprocess.run([EXE, args] + paths + [out], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        with open(out, "rb") as f:
            return f.read()


def payloads():
    from cmix_amd import synth
    rng = np.random.default_rng(4242)
    text = synth.enwik_like(6000, 31)
    raw = text[:500] + bytes(rng.integers(0, 256, 300, dtype=np.uint8)) + b"\x00" * 40 + text[500:860]
    words = sorted({w for w in text.replace(b"\n", b" ").split(b" ") if w.isalpha() and w.islower() and len(w) > 2})
    dic = b"\n".join(words[:400]) + b"\n"
    big = synth.enwik_like(12000, 77)
    return {"raw_n": raw, "text_c": text[1000:3000], "dict_c": text[3000:4500], "dict": dic, "text12k_c": big,
            "text50k_c": synth.enwik_like(50000, 91)}


if __name__ == "__main__":
    p = payloads()
    out = {k + "_payload": np.frombuffer(v, np.uint8) for k, v in p.items()}
    out["raw_n_file"] = np.frombuffer(run("-n", [("in", p["raw_n"])]), np.uint8)
    out["text_c_file"] = np.frombuffer(run("-c", [("in", p["text_c"])]), np.uint8)
    out["dict_c_file"] = np.frombuffer(run("-c", [("dict", p["dict"]), ("in", p["dict_c"])]), np.uint8)
    out["text12k_c_file"] = np.frombuffer(run("-c", [("in", p["text12k_c"])]), np.uint8)
    import hashlib
    big50 = run("-c", [("in", p["text50k_c"])])
    out["text50k_c_sha256"] = np.frombuffer(hashlib.sha256(big50).digest(), np.uint8)
    out["text50k_c_size"] = np.array([len(big50)], np.int64)
    out["text50k_c_seed"] = np.array([50000, 91], np.int64)  # synth.enwik_like(50000, 91): not stored, regenerated
    del out["text50k_c_payload"]
    print("text50k_c", len(p["text50k_c"]), "->", len(big50), "bytes")
    for k in ("raw_n", "text_c", "dict_c", "text12k_c"):
        print(k, len(p[k]), "->", len(out[k + "_file"]), "bytes; header", out[k + "_file"][:5])
    np.savez_compressed(os.path.join(HERE, "dropin_vectors.npz"), **out)
