#!/usr/bin/env python3
"""A file with a 16-bit stereo WAV inside for the drop-in test (tests/golden/dropin_wav.npz): text, a RIFF / WAVE file (1500 stereo samples: the
reference's preprocessor leaves it in a DEFAULT block, paq8's own detector switches wavModel + recordModel on for the samples), text again -- and
the `.cmix` file the UNMODIFIED reference binary (oracle/_ref/cmix_O3 -c) writes for it.

    python tests/golden/make_dropin_wav.py"""
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
    from make_paq8_hashes import wav_file
    text = synth.enwik_like(1200, 23)
    return text[:500] + wav_file(1500, 2, 16, 21) + text[500:]


if __name__ == "__main__":
    p = payload()
    d = tempfile.mkdtemp()
    open(os.path.join(d, "in"), "wb").write(p)
    subprocess.run([os.path.join(ROOT, "oracle", "_ref", "cmix_O3"), "-c", os.path.join(d, "in"), os.path.join(d, "out")], check=True, stdout=subprocess.DEVNULL)
    f = open(os.path.join(d, "out"), "rb").read()
    print(len(p), "->", len(f), "bytes")
    np.savez_compressed(os.path.join(HERE, "dropin_wav.npz"), payload=np.frombuffer(p, np.uint8), cmix_file=np.frombuffer(f, np.uint8))
