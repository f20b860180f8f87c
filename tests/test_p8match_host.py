"""paq8's MatchModel and SparseMatchModel (with the SSCM / StationaryMap / IndirectMap read-outs) as a device building
block, without a GPU: the BODY of cmx_p8match_kernel (cmix_amd/csrc/p8match_dev.h) run on the host by
tests/host/p8match_emul.cpp against the oracle's restatement (oracle/paq8_match.c, pinned against the reference's own
classes): 17 + 11 mixer inputs per bit, the sparse model's two selectors, match length / expected byte. The same
comparison runs on the device in tests/test_zgpu_p8match.py."""
import ctypes as C
import os
import subprocess

import numpy as np

from oracle import oracle as O
from test_p8cm2_host import tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "host", "libp8matchemul.so")
SRC = os.path.join(ROOT, "tests", "host", "p8match_emul.cpp")
DEPS = [SRC] + [os.path.join(ROOT, "cmix_amd", "csrc", f) for f in ("p8match_dev.h", "p8match_build.h", "p8cm2_dev.h")]
MATCH_BYTES, SPARSE_BYTES, HIST_LOG2 = 1 << 18, 1 << 18, 16


def emul():
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", SO, SRC])
    L = C.CDLL(SO)
    L.p8m_create.restype = C.c_void_p
    L.p8m_create.argtypes = [C.c_uint64, C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.p8m_destroy.argtypes = [C.c_void_p]
    L.p8m_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    return L


def ilog_table():
    lib = O.lib()
    lib.orc_p8_ilog.argtypes = [C.c_int]
    return np.array([lib.orc_p8_ilog(x) for x in range(65536)], np.uint8)


def stream():
    """Repeated passages (long matches, extension, recovery after a miss -> delta mode), near-repeats that differ in case
    or in every other byte (what the sparse finders look for), zero runs."""
    from cmix_amd import synth
    a = synth.enwik_like(1500, 53)
    up = bytes(c ^ 0x20 if 97 <= c <= 122 else c for c in a[400:1000])
    inter = bytes(b if i % 2 == 0 else (b + 1) & 0xff for i, b in enumerate(a[100:700]))
    return np.frombuffer(a + a[200:900] + b"XY" + a[300:1200] + up + a[:700] + bytes(50) + inter + a[100:600], np.uint8)


def oracle_rows(data):
    lib = O.lib()
    lib.orc_p8_match_new.restype = C.c_void_p
    lib.orc_p8_match_new.argtypes = [C.c_uint32]
    lib.orc_p8_sparsematch_new.restype = C.c_void_p
    lib.orc_p8_sparsematch_new.argtypes = [C.c_uint64]
    lib.orc_p8_match_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_p8_sparsematch_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    m, s = lib.orc_p8_match_new(MATCH_BYTES), lib.orc_p8_sparsematch_new(SPARSE_BYTES)
    ring = np.zeros(1 << HIST_LOG2, np.uint8)
    T = 8 * len(data)
    out, stats, sets = np.zeros((T, 28), np.int16), np.zeros((T, 3), np.int32), np.zeros((T, 2), np.int32)
    o, nn, ee, ss = np.zeros(32, np.int16), C.c_int(0), C.c_int(0), np.zeros(2, np.int32)
    y, c0 = 0, 1
    for n in range(len(data)):
        for bpos in range(8):
            t = 8 * n + bpos
            stats[t, 0] = lib.orc_p8_match_step(m, y, bpos, c0, ring.ctypes.data, (1 << HIST_LOG2) - 1, n, o.ctypes.data, C.byref(nn), C.byref(ee))
            assert nn.value == 17
            out[t, :17] = o[:17]
            stats[t, 1] = ee.value
            stats[t, 2] = lib.orc_p8_sparsematch_step(s, y, bpos, c0, ring.ctypes.data, (1 << HIST_LOG2) - 1, n, o.ctypes.data, C.byref(nn), ss.ctypes.data)
            assert nn.value == 11
            out[t, 17:] = o[:11]
            sets[t] = ss
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        ring[n & ((1 << HIST_LOG2) - 1)] = data[n]
    return out, stats, sets


def test_vs_oracle():
    L = emul()
    data = stream()
    want = oracle_rows(data)
    nex, stretch, _ = tables()
    ilog = ilog_table()
    h = L.p8m_create(MATCH_BYTES, SPARSE_BYTES, HIST_LOG2, nex.ctypes.data, stretch.ctypes.data, ilog.ctypes.data)
    assert h
    T = 8 * len(data)
    out, stats, sets = np.zeros((T, 28), np.int16), np.zeros((T, 3), np.int32), np.zeros((T, 2), np.int32)
    pos = 0
    for n in [1, 5, 2000, 1 << 30]:
        n = min(n, len(data) - pos)
        if n <= 0:
            break
        b = np.ascontiguousarray(data[pos:pos + n])
        L.p8m_run(h, b.ctypes.data, n, out[8 * pos:].ctypes.data, stats[8 * pos:].ctypes.data, sets[8 * pos:].ctypes.data)
        pos += n
    L.p8m_destroy(h)
    for name, g, w in zip(("inputs", "stats", "selectors"), (out, stats, sets), want):
        bad = np.argwhere(g != w)
        assert bad.size == 0, (name, "first mismatch (bit, column):", bad[0], g[tuple(bad[0])], w[tuple(bad[0])])
    assert want[1][:, 0].max() > 400 and want[1][:, 2].max() > 20      # long matches and sparse matches occurred


def test_vs_golden_columns():
    """In paq8's input order MatchModel's 17 inputs are mixer inputs 76..92 and SparseMatchModel's 11 are 93..103
    (contextModel2, reference paq8.cpp:8155-8165), which cmix sees as squash(x) / 4095 in layer-0 columns 510..537. The
    kernel body at cmix's sizes (level 11: 256 MB / 64 MB position tables, 1 GB history ring), started the way paq8's
    Predictor starts (first call at bit position 1), must reproduce those columns of the golden trace recorded from the
    unmodified reference predictor from the second byte on (during the first byte the context maps ahead of it emit
    nothing, so the columns sit elsewhere). Fixtures only."""
    import make_golden as mg
    from conftest import load_golden
    lib = O.lib()
    lib.orc_p8_squash.argtypes = [C.c_int]
    g = load_golden("text_96")
    probs, data = mg.unpack_probs(g), np.ascontiguousarray(g["stream"], np.uint8)
    L = emul()
    L.p8m_run_from.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    nex, stretch, _ = tables()
    ilog = ilog_table()
    mem = 0x10000 << 11
    h = L.p8m_create(mem * 2, mem // 2, 30, nex.ctypes.data, stretch.ctypes.data, ilog.ctypes.data)
    assert h
    T = 8 * len(data)
    out, stats, sets = np.zeros((T, 28), np.int16), np.zeros((T, 3), np.int32), np.zeros((T, 2), np.int32)
    L.p8m_run_from(h, data.ctypes.data, len(data), 1, out.ctypes.data, stats.ctypes.data, sets.ctypes.data)
    L.p8m_destroy(h)
    sq = np.array([lib.orc_p8_squash(int(v)) for v in range(-2048, 2048)], np.int32)
    got = sq[np.clip(out[8:].astype(np.int32), -2047, 2047) + 2048].astype(np.float32) * np.float32(1.0 / 4095)
    want = np.ascontiguousarray(probs[8:T, 510:538])
    bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, ("first mismatch (bit after the first byte, input):", bad[0], got[tuple(bad[0])] * 4095, want[tuple(bad[0])] * 4095)
