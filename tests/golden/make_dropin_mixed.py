#!/usr/bin/env python3
"""A mixed-type file for the drop-in test (tests/golden/dropin_mixed.npz): text, x86-like code (call instructions to a handful
of targets: the reference's detector makes it an EXE block and its preprocessor rewrites the call addresses), binary records,
text again -- the `.cmix` file the UNMODIFIED reference binary (oracle/_ref/cmix_O3 -c) writes for it, and the stream its preprocessor
handed the predictor (captured from the binary's temp file while it runs). Exercises block
switching (TEXT / DEFAULT / EXE headers inside one stream) through every stage of the engine.

    python tests/golden/make_dropin_mixed.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def payload():
    from cmix_amd import synth
    r = np.random.default_rng(77)
    code = bytearray()
    targets = [int(t) for t in r.integers(0x200, 0x2800, 8)]
    while len(code) < 12000:
        # a few "instructions": mov / add / push / pop / jcc short, never E8 / E9 / 0F 8x
        for _ in range(int(r.integers(2, 7))):
            code += bytes([[0x89, 0x8B, 0x01, 0x03, 0x50, 0x58, 0x74, 0x75, 0x83, 0xC7][int(r.integers(0, 10))], int(r.integers(0, 0xE0))])
        t = targets[int(r.integers(0, 8))]
        rel = (t - (len(code) + 5)) & 0xFFFFFFFF
        code += bytes([0xE8]) + rel.to_bytes(4, "little")
    rec = b"".join(bytes([i & 255, (i >> 8) & 255, 0, 0]) + bytes(r.integers(32, 48, 12, dtype=np.uint8)) for i in range(256))
    return synth.enwik_like(5000, 5) + bytes(code[:12000]) + rec + synth.enwik_like(3000, 6)


if __name__ == "__main__":
    import subprocess
    import tempfile
    import time
    p = payload()
    d = tempfile.mkdtemp()
    open(os.path.join(d, "in"), "wb").write(p)
    pr = subprocess.Popen([os.path.join(ROOT, "oracle", "_ref", "cmix_O3"), "-c", os.path.join(d, "in"), os.path.join(d, "out")],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    time.sleep(8)   # the reference's preprocessor has written <out>.cmix.temp by now: the stream the predictor is coding
    stream = open(os.path.join(d, "out.cmix.temp"), "rb").read()
    pr.wait()
    f = open(os.path.join(d, "out"), "rb").read()
    print(len(p), "-> stream", len(stream), "-> file", len(f), "bytes")
    np.savez_compressed(os.path.join(HERE, "dropin_mixed.npz"), payload=np.frombuffer(p, np.uint8), cmix_file=np.frombuffer(f, np.uint8),
                        stream=np.frombuffer(stream, np.uint8))
