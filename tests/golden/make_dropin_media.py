#!/usr/bin/env python3
"""The 168 KB media file of tests/golden/make_paq8_big_media.py (a 160 x 120 24-bit BMP, a 40 KB 16-bit stereo WAV, a 320 x 240 JPEG, a 200 x 150 PGM,
4- and 8-bit BMPs, short pieces of text) and the `.cmix` file the UNMODIFIED reference binary (oracle/_ref/cmix_O3 -c) writes for it:
tests/golden/dropin_media_168k.npz. Every stage of the engine on media at sizes where a model's segment runs through many chunks. About ten minutes.

    python tests/golden/make_dropin_media.py"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, HERE]

if __name__ == "__main__":
    from make_paq8_big_media import payload
    p = payload()
    d = tempfile.mkdtemp()
    open(os.path.join(d, "in"), "wb").write(p)
    t0 = time.time()
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "cmix_O3"), "-c", os.path.join(d, "in"), os.path.join(d, "out")], check=True, stdout=subprocess.DEVNULL)
    f = open(os.path.join(d, "out"), "rb").read()
    print(len(p), "->", len(f), "bytes in", round(time.time() - t0), "s")
    np.savez_compressed(os.path.join(HERE, "dropin_media_168k.npz"), payload=np.frombuffer(p, np.uint8), cmix_file=np.frombuffer(f, np.uint8))
