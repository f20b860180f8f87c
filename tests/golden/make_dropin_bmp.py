#!/usr/bin/env python3
"""A file with a 24-bit BMP inside for the drop-in test (tests/golden/dropin_bmp.npz): text, a 96 x 48 bottom-up BMP (the reference's
detector makes it HDR + IMAGE24 blocks, its preprocessor the colour transform of preprocessor.cpp:303-324), text again -- and the `.cmix`
file the UNMODIFIED reference binary (oracle/_ref/cmix_O3 -c) writes for it. Exercises paq8's image model (im24bitModel) through the whole
engine: the paq8 stage's image kernels, every other stage on image bytes, block switching.

    python tests/golden/make_dropin_bmp.py"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def payload():
    from cmix_amd import synth
    from make_paq8_hashes import bmp_file, photo
    text = synth.enwik_like(1500, 21)
    return text[:600] + bmp_file(photo(96, 48, 3, 7)) + text[600:]


if __name__ == "__main__":
    p = payload()
    d = tempfile.mkdtemp()
    open(os.path.join(d, "in"), "wb").write(p)
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "cmix_O3"), "-c", os.path.join(d, "in"), os.path.join(d, "out")], check=True, stdout=subprocess.DEVNULL)
    f = open(os.path.join(d, "out"), "rb").read()
    print(len(p), "->", len(f), "bytes")
    np.savez_compressed(os.path.join(HERE, "dropin_bmp.npz"), payload=np.frombuffer(p, np.uint8), cmix_file=np.frombuffer(f, np.uint8))
