"""paq8's ContextMap2 as a device building block, without a GPU: the step functions of cmx_p8s_cm2v2_kernel (cmix_amd/csrc/p8cm2_dev.h:
bucket lists, overlap check, update + mix per context lane) run on the host by tests/host/p8cm2_emul.cpp -- a loop over
lanes per barrier step in shuffled order -- against the oracle's restatement (oracle/paq8_maps.c, itself pinned against
the reference's own ContextMap2 class). All 7 inputs of every context for every bit. The same comparison runs on the
device in tests/test_zgpu_p8cm2.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "host", "libp8cm2emul.so")
SRC = os.path.join(ROOT, "tests", "host", "p8cm2_emul.cpp")
DEPS = [SRC] + [os.path.join(ROOT, "cmix_amd", "csrc", f) for f in ("p8cm2_dev.h", "p8cm2_build.h")]


def emul():
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", SO, SRC])
    L = C.CDLL(SO)
    L.p8e_create.restype = C.c_void_p
    L.p8e_create.argtypes = [C.c_uint64, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    L.p8e_destroy.argtypes = [C.c_void_p]
    L.p8e_hash.argtypes = [C.c_uint64, C.c_uint32, C.c_uint64, C.c_void_p, C.c_void_p]
    L.p8e_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.p8e_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    return L


def tables():
    """nex(), stretch(), ilog() as data, from the oracle (which holds them as dumps of the reference build)."""
    lib = O.lib()
    nex = np.zeros(1024, np.uint8)
    lib.orc_p8_state_table.argtypes = [C.c_void_p]
    lib.orc_p8_state_table(nex.ctypes.data)
    lib.orc_p8_stretch.argtypes = [C.c_int]
    lib.orc_p8_ilog.argtypes = [C.c_int]
    stretch = np.array([lib.orc_p8_stretch(p) for p in range(4096)], np.int16)
    ilog = np.array([lib.orc_p8_ilog(x) for x in range(257)], np.uint8)
    return nex, stretch, ilog


def contexts(data, count, flavour):
    """Per byte the `count` 64-bit contexts a front end would set: order-1..N chains (contextModel2 :8139-8153) or, for
    `collide`, contexts drawn from a tiny pool so that several of them land in the same bucket all the time."""
    lib = O.lib()
    lib.orc_p8_combine64.restype = C.c_uint64
    lib.orc_p8_combine64.argtypes = [C.c_uint64, C.c_uint64]
    out = np.zeros((len(data), count), np.uint64)
    r = np.random.default_rng(9)
    for n in range(len(data)):
        if flavour == "collide":
            out[n] = r.integers(0, 5, count).astype(np.uint64) + np.uint64(int(data[n - 1]) if n else 0) * np.uint64(7)
            continue
        h = 0
        for k in range(count):
            h = lib.orc_p8_combine64(h, int(data[n - 1 - k]) if n - 1 - k >= 0 else 0)
            out[n, k] = h
    return out


def oracle_rows(size_bytes, count, data, cx):
    lib = O.lib()
    lib.orc_p8_cm2_new.restype = C.c_void_p
    lib.orc_p8_cm2_new.argtypes = [C.c_uint64, C.c_uint32]
    lib.orc_p8_cm2_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.orc_p8_cm2_free.argtypes = [C.c_void_p]
    h = lib.orc_p8_cm2_new(size_bytes, count)
    rows = np.zeros((8 * len(data), 7 * count), np.int16)
    o, n_out, y = np.zeros(7 * count + 8, np.int16), C.c_int(0), 0
    for n in range(len(data)):
        c = np.ascontiguousarray(cx[n])
        for bpos in range(8):
            lib.orc_p8_cm2_step(h, y, bpos, c.ctypes.data, count, o.ctypes.data, C.byref(n_out))
            assert n_out.value == 7 * count
            rows[8 * n + bpos] = o[:7 * count]
            y = (int(data[n]) >> (7 - bpos)) & 1
    lib.orc_p8_cm2_free(h)
    return rows


def hashed(L, cx, size_bytes):
    c32, k16 = np.zeros(cx.shape, np.uint32), np.zeros(cx.shape, np.uint16)
    a, b = C.c_uint32(0), C.c_uint16(0)
    for n in range(cx.shape[0]):
        for i in range(cx.shape[1]):
            L.p8e_hash(int(cx[n, i]), i, size_bytes, C.byref(a), C.byref(b))
            c32[n, i], k16[n, i] = a.value, b.value
    return c32, k16


def run_emul(L, size_bytes, count, data, cx, chunks, seed=777, serial=0, stats=None):
    nex, stretch, ilog = tables()
    h = L.p8e_create(size_bytes, count, nex.ctypes.data, stretch.ctypes.data, ilog.ctypes.data, seed, serial)
    assert h
    c32, k16 = hashed(L, cx, size_bytes)
    bits = np.unpackbits(np.ascontiguousarray(data, np.uint8))
    out = np.zeros((8 * len(data), 7 * count), np.int16)
    pos = 0
    for n in chunks:
        n = min(n, len(data) - pos)
        if n <= 0:
            break
        a, b = np.ascontiguousarray(c32[pos:pos + n]), np.ascontiguousarray(k16[pos:pos + n])
        bb, o = np.ascontiguousarray(bits[8 * pos:8 * (pos + n)]), out[8 * pos:8 * (pos + n)]
        assert L.p8e_run(h, a.ctypes.data, b.ctypes.data, bb.ctypes.data, n, o.ctypes.data) == 0
        pos += n
    if stats is not None:
        st = np.zeros(2, np.uint64)
        L.p8e_stats(h, st.ctypes.data)
        stats.extend(int(v) for v in st)
    L.p8e_destroy(h)
    return out


CASES = [(1 << 16, 10, 3000, "orders"), (1 << 22, 33, 1500, "orders"), (1 << 16, 20, 2500, "collide")]


@pytest.mark.parametrize("size_bytes,count,nbytes,flavour", CASES)
def test_vs_oracle(size_bytes, count, nbytes, flavour):
    from cmix_amd import synth
    L = emul()
    data = np.frombuffer(synth.enwik_like(nbytes, 13), np.uint8)
    cx = contexts(data, count, flavour)
    want = oracle_rows(size_bytes, count, data, cx)
    stats = []
    got = run_emul(L, size_bytes, count, data, cx, [1, 7, 500, 1000, 4000], stats=stats)
    bad = np.argwhere(got != want)
    assert bad.size == 0, (flavour, "first mismatch (step, input):", bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
    if flavour == "collide":
        assert stats[1] > stats[0] // 10, stats      # the serial fallback is what this case is about
    else:
        assert stats[1] < stats[0] // 2, stats       # and the lane-per-context path is the common one otherwise
    got = run_emul(L, size_bytes, count, data[:600], cx[:600], [600], serial=1)
    assert np.array_equal(got, want[:8 * 600])


def test_order_n_map_vs_golden_columns():
    """contextModel2's order-N ContextMap2 (10 contexts, 2 GB at cmix's level 11; reference paq8.cpp:8102, :8139-8153) is
    the first map paq8 mixes: its 70 inputs are paq8's mixer inputs 3..72, and cmix sees every mixer input as
    squash(x) / 4095 (paq8.cpp:542-545) -- layer-0 columns 437..506 of the golden traces recorded from the unmodified
    reference predictor. The kernel body, fed with contexts computed here from the trace's bytes, must reproduce those
    columns from the second byte on (during the first byte the map holds no context and emits nothing). Fixtures only."""
    import make_golden as mg
    from conftest import load_golden
    lib = O.lib()
    lib.orc_p8_combine64.restype = C.c_uint64
    lib.orc_p8_combine64.argtypes = [C.c_uint64, C.c_uint64]
    lib.orc_p8_squash.argtypes = [C.c_int]
    g = load_golden("text_96")
    probs, stream = mg.unpack_probs(g), np.asarray(g["stream"], np.uint8)
    size_bytes, count = 1 << 31, 10
    cxt = [0] * 16
    cx = np.zeros((len(stream) - 1, count), np.uint64)
    for n in range(1, len(stream)):          # the contexts set when byte n-1 is complete, in set() order (:8141-8152)
        B = int(stream[n - 1])
        cxt[15] = (lib.orc_p8_combine64(cxt[15], ord(chr(B).lower())) & 0xffffffff) if chr(B).isalpha() and B < 128 else 0
        for i in range(14, 0, -1):
            cxt[i] = lib.orc_p8_combine64(cxt[i - 1], B) & 0xffffffff
        cx[n - 1] = [cxt[15]] + cxt[0:7] + [cxt[8], cxt[14]]
    L = emul()
    nex, stretch, ilog = tables()
    h = L.p8e_create(size_bytes, count, nex.ctypes.data, stretch.ctypes.data, ilog.ctypes.data, 99, 0)
    L.p8e_seed.argtypes = [C.c_void_p, C.c_uint32, C.c_int]
    L.p8e_seed(h, 128 | (int(stream[0]) >> 1), int(stream[0]) & 1)
    c32, k16 = hashed(L, cx, size_bytes)
    bits = np.unpackbits(stream[1:])
    out = np.zeros((len(bits), 7 * count), np.int16)
    assert L.p8e_run(h, c32.ctypes.data, k16.ctypes.data, bits.ctypes.data, len(stream) - 1, out.ctypes.data) == 0
    L.p8e_destroy(h)
    sq = np.array([lib.orc_p8_squash(int(v)) for v in range(-2048, 2048)], np.int32)
    got = sq[np.clip(out.astype(np.int32), -2047, 2047) + 2048].astype(np.float32) * np.float32(1.0 / 4095)
    want = np.ascontiguousarray(probs[8:8 + len(bits), 437:507])
    bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, ("first mismatch (bit after the first byte, input):", bad[0], got[tuple(bad[0])] * 4095, want[tuple(bad[0])] * 4095)
