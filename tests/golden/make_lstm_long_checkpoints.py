#!/usr/bin/env python3
"""Checkpoints of the long reference trace for the GPU box (tests/golden/lstm_330k_checkpoints.npz, ~20 KB): the
LSTM byte mixer's distribution after selected bytes of the 330 000-byte trace oracle/_ref/golden/text_330k_bytes.npz
(tests/golden/make_long_trace.py, local only: 311 MB), on both sides of byte 300 000 where LstmLayer::update_steps_
saturates and Adam's bias terms switch to the double-precision pow() path (lstm-layer.cpp:26-30). The stream is not
stored: it is regenerated from its seed; the PPMd distributions the LSTM consumes come from the host PPMd stage,
which the CPU suite pins against the same trace (tests/test_ppmd_host.py::test_text_330k_local).

    python tests/golden/make_lstm_long_checkpoints.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
AT = [0, 99, 100, 9999, 100000, 200000, 299899, 299999, 300000, 300099, 300100, 310000, 320000, 329999]


def stream_330k():
    from cmix_amd import synth
    import make_golden as mg
    nbytes = 330000
    return np.frombuffer(mg.text_block(synth.enwik_like(nbytes + 4096, 1003)[4096:4096 + nbytes - 6]), np.uint8)


if __name__ == "__main__":
    with np.load(os.path.join(ROOT, "oracle", "_ref", "golden", "text_330k_bytes.npz")) as z:
        assert (z["stream"] == stream_330k()).all()
        out = {"at": np.array(AT, np.int64), "lstm_probs": z["lstm_probs"][np.array(AT) + 1],
               "ppmd_probs": z["ppmd_probs"][np.array(AT) + 1], "vocab": z["vocab"]}
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "lstm_330k_checkpoints.npz"), **out)
    print("wrote", len(AT), "checkpoints")
