#!/usr/bin/env python3
"""Time the paq8 building-block kernels at cmix's level-11 table sizes on one GPU (us per bit, torch events), with
contexts that look like a front end's (order-k hashes of enwik-like text): the ContextMap family (210 contexts), the
three ContextMap2 instances (10 / 33 / 20 contexts), the DMC forest, the match models, the mixer.
    python scripts/gpu_p8blocks_time.py [nbytes]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from cmix_amd import engine as E, synth  # noqa: E402
from test_p8cm2_host import tables  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
WARM = 512
data = np.frombuffer(synth.enwik_like(N + WARM, 77), np.uint8)
bits = torch.from_numpy(np.unpackbits(data)).cuda()
nex, stretch, ilog = tables()
MEM = 0x10000 << 11


def ctxs(total, seed):
    """order-k style contexts: slot i hashes the last 1 + i % 7 bytes with a slot-specific multiplier"""
    n = len(data)
    c = np.zeros((n, total), np.uint64)
    pad = np.concatenate([np.zeros(8, np.uint8), data]).astype(np.uint64)
    for i in range(total):
        h = np.full(n, np.uint64(seed * 1000003 + i * 7919), np.uint64)
        for k in range(1 + i % 7):
            h = (h + pad[8 - 1 - k:8 - 1 - k + n] + np.uint64(1)) * np.uint64(0x9E3779B97F4A7C15)
        c[:, i] = h
    return c


def split(c64, bits_hash):
    c32 = (c64 >> np.uint64(64 - bits_hash)).astype(np.uint32)
    k16 = ((c64 >> np.uint64(64 - bits_hash - 16)) & np.uint64(0xffff)).astype(np.uint16)
    return c32, k16


def timed(fn, label, nbits):
    fn(0, WARM)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    fn(WARM, WARM + N)
    b.record()
    torch.cuda.synchronize()
    print("%-40s %8.2f us/bit" % (label, a.elapsed_time(b) * 1e3 / nbits), flush=True)


# ---- ContextMap family at cmix's sizes (instances in contextModel2's calling order) ----
SIZES = [MEM * 2, MEM * 4, MEM, 65536, 65536, 65536, MEM, 65536, 65536, 131072, 65536, 65536, MEM * 16, MEM // 2, MEM, MEM // 4]
COUNTS = [42, 31, 3, 3, 3, 3, 16, 2, 5, 4, 3, 3, 61, 12, 15, 4]
for serial in ("0", "1"):
    os.environ["CMX_P8CM_SERIAL"] = serial
    fam = E.P8ContextMapFamily(SIZES, COUNTS, nex, stretch, ilog, 0)
    tot = sum(COUNTS)
    c64 = ctxs(tot, 1)
    c32 = np.zeros((len(data), tot), np.uint32); k16 = np.zeros((len(data), tot), np.uint16)
    s = 0
    for sz, ct in zip(SIZES, COUNTS):
        hb = int(np.log2(sz >> 6))
        c32[:, s:s + ct], k16[:, s:s + ct] = split(c64[:, s:s + ct], hb)
        s += ct
    dc, dk = torch.from_numpy(c32.view(np.int32)).cuda(), torch.from_numpy(k16.view(np.int16)).cuda()
    timed(lambda a, b: fam.run(dc[a:b].contiguous(), dk[a:b].contiguous(), bits[8 * a:8 * b].contiguous()), "ContextMap family %d ctx serial=%s" % (tot, serial), 8 * N)
    fam.close()
os.environ.pop("CMX_P8CM_SERIAL", None)

for name, size, cnt in (("ContextMap2 main", MEM * 16, 10), ("ContextMap2 text", MEM * 16, 33), ("ContextMap2 exe", MEM * 2, 20)):
    cm = E.P8ContextMap2(size, cnt, nex, stretch, ilog, 0)
    c32, k16 = split(ctxs(cnt, 5), int(np.log2(size >> 6)))
    dc, dk = torch.from_numpy(c32.view(np.int32)).cuda(), torch.from_numpy(k16.view(np.int16)).cuda()
    timed(lambda a, b: cm.run(dc[a:b].contiguous(), dk[a:b].contiguous(), bits[8 * a:8 * b].contiguous()), "%s %d ctx" % (name, cnt), 8 * N)
    cm.close()

dmc = E.P8DmcForest(11, nex, stretch, 0)
timed(lambda a, b: dmc.run(bits[8 * a:8 * b].contiguous()), "DMC forest level 11", 8 * N)
dmc.close()

from oracle import oracle as O  # noqa: E402  (tables only: ilog over 16 bits)
import ctypes as C  # noqa: E402
il = np.array([O.lib().orc_p8_ilog(i) for i in range(65536)], np.uint8)
mm = E.P8MatchModels(MEM * 2, MEM // 2, 27, nex, stretch, il, 0)
dd = torch.from_numpy(data.copy()).cuda()
timed(lambda a, b: mm.run(dd[a:b].contiguous()), "match + sparse match", 8 * N)
mm.close()

sq = np.array([O.lib().orc_p8_squash(i - 2048) for i in range(4096)], np.int16)
mx = E.P8Mixer(77472, sq, stretch, 0)
r = np.random.default_rng(3)
x = torch.from_numpy(r.integers(-2047, 2048, (8 * (N + WARM), 1552)).astype(np.int16)).cuda()
base = np.cumsum([0] + [2000] * 27)
rows = torch.from_numpy((base[None, :] + r.integers(0, 2000, (8 * (N + WARM), 28))).astype(np.int32)).cuda()
timed(lambda a, b: mx.run(x[8 * a:8 * b].contiguous(), rows[8 * a:8 * b].contiguous(), bits[8 * a:8 * b].contiguous()), "mixer 1552 x 28", 8 * N)
mx.close()
