#!/usr/bin/env python3
"""Feasibility of the speculative segment-parallel ordered chain, without a GPU: trace N bytes of the bench
shard through the unmodified reference (all 2078 layer-0 inputs per bit), replay them through the oracle's mixing network with
the orc_mix_probe hook installed, and count per candidate scheme how often a speculative segment would hit (scripts/study/spec_chain.c).

    python scripts/spec_chain_study.py [nbytes=8192] [segments=4] [sample_every=1]   -> profiles/r03_spec_chain_study.txt
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]

if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
    nseg = [int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "4").split(",")]
    every = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    from cmix_amd import synth
    import make_golden as mg
    from oracle import oracle as O
    cache = "/tmp/spec_trace_%d.npz" % n
    if os.path.exists(cache):
        g = dict(np.load(cache))
    else:
        g = mg.trace(mg.text_block(synth.enwik_like(n - 6, 1000, rich=True)), full=True)
        np.savez(cache, **g)
    probs = mg.unpack_probs(g)
    so = "/tmp/libspecchain.so"
    subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math", "-o", so,
                           os.path.join(ROOT, "scripts", "study", "spec_chain.c"), "-lm"])
    S = C.CDLL(so)
    L = O.lib()
    hook = C.c_void_p.in_dll(L, "orc_mix_probe")
    for ns in nseg:
        S.spec_config(ns, every)
        S.spec_reset()
        hook.value = C.cast(S.spec_probe, C.c_void_p).value
        net = O.MixNet()
        out = net.run(probs, g["sel"], g["bits"])
        hook.value = None
        assert np.array_equal(out.view(np.uint32), g["p_final"].view(np.uint32)), "oracle replay != reference trace"
        sys.stdout.flush()
        S.spec_report(None)
        sys.stdout.flush()
