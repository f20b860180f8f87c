#!/usr/bin/env python3
"""Long byte-level trace of the UNMODIFIED reference for the byte-rate learners (local fixture, not committed):
PPMd and LSTM byte distributions after every byte of a 330 KB text, i.e. past the 3000th BPTT/Adam round, where
LstmLayer's update_steps_ saturates and Adam switches to the double-precision pow() path (lstm-layer.cpp:26-30,
131-133). Per-bit state is not recorded (only Predict/Perceive are driven), so this runs at the reference's speed.

    python tests/golden/make_long_trace.py        # ~20 min, writes oracle/_ref/golden/text_330k_bytes.npz
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

if __name__ == "__main__":
    from cmix_amd import synth
    from oracle import refharness as R
    import make_golden as mg
    nbytes = int(sys.argv[1]) if len(sys.argv) > 1 else 330000
    stream = mg.text_block(synth.enwik_like(nbytes + 4096, 1003)[4096:4096 + nbytes - 6])
    r = R.Ref(R.vocab_of(stream))
    N = len(stream)
    ppmd = np.empty((N + 1, 256), np.float32)
    lstm = np.empty((N + 1, 256), np.float32)
    p16 = np.empty(8 * N, np.uint16)
    ppmd[0] = r.byte_probs(0)[0]
    lstm[0] = r.byte_probs(1)[0]
    t = 0
    for n, byte in enumerate(stream):
        for j in range(7, -1, -1):
            p = r.predict()
            p16[t] = int(1 + 65534 * float(p))  # what the coder sees (encoder.cpp:10-12); float32 product truncated
            r.perceive((byte >> j) & 1)
            t += 1
        ppmd[n + 1] = r.byte_probs(0)[0]
        lstm[n + 1] = r.byte_probs(1)[0]
        if n % 20000 == 0:
            print(n, flush=True)
    out = os.path.join(ROOT, "oracle", "_ref", "golden", "text_330k_bytes.npz")
    np.savez_compressed(out, stream=np.frombuffer(stream, np.uint8), ppmd_probs=ppmd, lstm_probs=lstm, p16=p16,
                        vocab=R.vocab_of(stream))
    print("wrote", out)
