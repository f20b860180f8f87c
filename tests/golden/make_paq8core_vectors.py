#!/usr/bin/env python3
"""tests/golden/paq8core_vectors.npz: outputs of the reference's own paq8 building blocks (oracle/_ref/libcmixrefpaq8.so
= reference src/models/paq8.cpp compiled by oracle/ref_paq8core.cpp) on the seeded drives of
tests/test_oracle_paq8core.py, so that the restatement stays pinned where oracle/_ref is absent.

    python tests/golden/make_paq8core_vectors.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

if __name__ == "__main__":
    import test_oracle_paq8core as T
    from oracle import refharness as R
    L = R.paq8core_lib()
    n, m, s, w, steps, seed = 96, 500, 6, 32, 4000, 77
    ranges, xs, cx, rng = T.mixer_case(seed, n, m, s, steps)
    bits = rng.random(steps)
    h = L.refp8_mixer_new(n, m, s, w)
    p, e, _ = T.run_mixer(L.refp8_mixer_step, h, ranges, xs, cx, bits)
    out = {"mixer_shape": np.array([n, m, s, w, steps]), "mixer_seed": np.array([seed]), "mixer_p": p,
           "mixer_exported_last": e[-1].copy()}
    pr, cx1, y = T._adaptive_case(5, 6000, 1 << 10)
    a = L.refp8_apm1_new(1 << 10)
    out["apm1_p"] = np.array([L.refp8_apm1_p(a, int(y[t]), int(pr[t]), int(cx1[t]), 7) for t in range(len(pr))], np.int32)
    q = L.refp8_apm_new(1 << 10)
    out["apm_p"] = np.array([L.refp8_apm_p(q, int(y[t]), int(pr[t]), int(cx1[t]), 0xFF) for t in range(len(pr))], np.int32)
    sm = L.refp8_statemap32_new(256)
    out["sm32_p"] = np.array([L.refp8_statemap32_p(sm, int(y[t]), int(cx1[t]) & 255, 1023) for t in range(len(pr))], np.int32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "paq8core_vectors.npz"), **out)
    print({k: v.shape for k, v in out.items()})
