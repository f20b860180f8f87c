#!/usr/bin/env python3
"""A file with a baseline JPEG inside for the drop-in test (tests/golden/dropin_jpeg.npz): text, a 4:2:0 colour JPEG (Pillow's encoder; the
reference's detector makes it a JPEG block), text again -- and the `.cmix` file the UNMODIFIED reference binary (oracle/_ref/cmix_O3 -c) writes for it.

    python tests/golden/make_dropin_jpeg.py"""
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
    from make_paq8_hashes import jpeg_file, photo
    text = synth.enwik_like(1000, 29)
    return text[:400] + jpeg_file(photo(128, 96, 3, 31), quality=75) + text[400:]


if __name__ == "__main__":
    p = payload()
    d = tempfile.mkdtemp()
    open(os.path.join(d, "in"), "wb").write(p)
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "cmix_O3"), "-c", os.path.join(d, "in"), os.path.join(d, "out")], check=True, stdout=subprocess.DEVNULL)
    f = open(os.path.join(d, "out"), "rb").read()
    print(len(p), "->", len(f), "bytes")
    np.savez_compressed(os.path.join(HERE, "dropin_jpeg.npz"), payload=np.frombuffer(p, np.uint8), cmix_file=np.frombuffer(f, np.uint8))
