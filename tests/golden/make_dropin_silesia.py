#!/usr/bin/env python3
"""Config 4 at reduced sizes (tests/golden/dropin_silesia.npz): the twelve members of synth.silesia_like(SCALE) -- prose, wiki
text, XML, text + record tables, x86-like code + data, 16-bit samples, uniform bytes -- each compressed by the UNMODIFIED
reference binary (`cmix -c member out`, its own type detection / block framing / e8e9 transform); per member the size and
SHA-256 of the file it wrote. The payloads are regenerated from the seed on the GPU box. About 12 CPU-minutes.

    python tests/golden/make_dropin_silesia.py [scale_bytes]
"""
import hashlib
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))
SCALE, SEED = 48 * 1024, 4000

if __name__ == "__main__":
    from cmix_amd import synth
    from make_dropin_vectors import run
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else SCALE
    files = synth.silesia_like(scale, SEED)
    out = {"scale_seed": np.array([scale, SEED], np.int64), "names": np.array(sorted(files))}
    for name in sorted(files):
        t0 = time.time()
        blob = run("-c", [("in", files[name])])
        out[name + "_sha256"] = np.frombuffer(hashlib.sha256(blob).digest(), np.uint8)
        out[name + "_size"] = np.array([len(files[name]), len(blob)], np.int64)
        out[name + "_ref_seconds"] = np.array([time.time() - t0])
        print(name, len(files[name]), "->", len(blob), "bytes in %.0f s" % (time.time() - t0), flush=True)
    np.savez_compressed(os.path.join(HERE, "dropin_silesia.npz"), **out)
