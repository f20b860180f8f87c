#!/usr/bin/env python3
"""Where the engine's construction time goes: wall time of each stage's create call on one GPU (first a throw-away HIP init)."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cmix_amd import engine as E  # noqa: E402

t0 = time.perf_counter(); torch.cuda.init(); torch.zeros(1, device="cuda"); torch.cuda.synchronize(); print("HIP / torch init %.2f s" % (time.perf_counter() - t0))
vocab = np.ones(256, np.uint8)
for rep in range(2):
    objs = []
    for name, mk in (("mixnet", lambda: E.MixNet(0)), ("ctxmodels", lambda: E.CtxModels(vocab, 0)), ("lstm", lambda: E.Lstm(vocab, 0)),
                     ("fxcm", lambda: E.Fxcm(None, 0)), ("p8stage", lambda: E.P8Stage(0)), ("ppmd", lambda: E.Ppmd(vocab) if hasattr(E, "Ppmd") else None)):
        t0 = time.perf_counter()
        o = mk()
        torch.cuda.synchronize()
        print("rep %d create %-10s %.2f s" % (rep, name, time.perf_counter() - t0))
        objs.append(o)
    t0 = time.perf_counter()
    for o in objs:
        if o is not None and hasattr(o, "close"):
            o.close()
    print("rep %d destroy all %.2f s" % (rep, time.perf_counter() - t0))
t0 = time.perf_counter()
p = E.Pipeline(vocab, 0, 4096); p.enable_fxcm(None); p.enable_paq8(); torch.cuda.synchronize()
print("whole pipeline (4 KB chunks) %.2f s" % (time.perf_counter() - t0))
p.close()
