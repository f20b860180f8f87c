#!/usr/bin/env python3
"""Config 3's fixture (tests/golden/dropin_dict.npz): `cmix -c english.dic in out` by the UNMODIFIED reference binary with the
reference's own 44 515-word dictionary -- WRT transform of the text (2- and 3-byte codewords) and Predictor::Pretrain over the
412 KB dictionary (preprocessor.cpp:37-69) before the first coded bit -- on 64 KB of enwik-like text whose words are drawn
from that dictionary (Zipf) with the rich alphabet. The payload is stored (the dictionary is not on the GPU box as
a source of words; the file itself travels as oracle/_ref/english.dic, copied there by oracle/Makefile). About 15 CPU-minutes.

    python tests/golden/make_dropin_dict.py [nbytes]
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
DIC = "/root/reference/dictionary/english.dic"

if __name__ == "__main__":
    from cmix_amd import synth
    from make_dropin_vectors import run
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
    dic = open(DIC, "rb").read()
    words = [w.decode("latin-1") for w in dic.split(b"\n") if w]
    payload = synth.enwik_like(n, 3003, rich=True, lexicon=words)
    t0 = time.time()
    blob = run("-c", [("english.dic", dic), ("in", payload)])
    dt = time.time() - t0
    print(n, "->", len(blob), "bytes in %.0f s" % dt)
    np.savez_compressed(os.path.join(HERE, "dropin_dict.npz"), payload=np.frombuffer(payload, np.uint8),
                        sha256=np.frombuffer(hashlib.sha256(blob).digest(), np.uint8), size=np.array([len(blob)], np.int64),
                        dict_sha256=np.frombuffer(hashlib.sha256(dic).digest(), np.uint8), ref_seconds=np.array([dt]))
