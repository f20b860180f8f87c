"""The family of paq8's older ContextMap instances as a device building block, without a GPU: the step functions of the family kernels (cmx_p8s_fam2_kernel, cmx_p8s_xfam_kernel)
(cmix_amd/csrc/p8cm_dev.h: bucket lists, overlap check, ranked draws of the process-global rnd(), one lane per context)
run on the host by tests/host/p8cm_emul.cpp -- loops over lanes per barrier step in shuffled order -- against the
oracle's restatement (oracle/paq8_maps.c, pinned against the reference's own class and generator), stepped instance by
instance in the reference's order with its one global generator. All 5 inputs of every context for every bit, and the
generator ends in the same position. The same comparison runs on the device in tests/test_zgpu_p8cm.py."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
from test_p8cm2_host import tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "host", "libp8cmemul.so")
SRC = os.path.join(ROOT, "tests", "host", "p8cm_emul.cpp")
DEPS = [SRC] + [os.path.join(ROOT, "cmix_amd", "csrc", f) for f in ("p8cm_dev.h", "p8cm_build.h", "p8cm2_dev.h", "p8cm2_build.h")]


def emul():
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in DEPS):
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", SO, SRC])
    L = C.CDLL(SO)
    L.p8f_create.restype = C.c_void_p
    L.p8f_create.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_int]
    L.p8f_destroy.argtypes = [C.c_void_p]
    L.p8f_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.p8f_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    return L


def family_contexts(data, counts, seed):
    """Per byte and instance the contexts a front end would set: order-k chains with an instance-specific seed; instance
    1 draws from a tiny pool so that its contexts collide."""
    lib = O.lib()
    lib.orc_p8_combine64.restype = C.c_uint64
    lib.orc_p8_combine64.argtypes = [C.c_uint64, C.c_uint64]
    r = np.random.default_rng(seed)
    out = []
    for k, cnt in enumerate(counts):
        a = np.zeros((len(data), cnt), np.uint64)
        for n in range(len(data)):
            if k == 1:
                a[n] = r.integers(0, 4, cnt).astype(np.uint64) + np.uint64(int(data[n - 1]) if n else 0) * np.uint64(5)
                continue
            h = 1000 * k
            for i in range(cnt):
                h = lib.orc_p8_combine64(h, int(data[n - 1 - i]) if n - 1 - i >= 0 else 0)
                a[n, i] = h
        out.append(a)
    return out


def oracle_rows(sizes, counts, data, cxs):
    """The instances stepped one after the other per bit, as contextModel2 calls its sub-models, on the one generator."""
    lib = O.lib()
    lib.orc_p8_cm_new.restype = C.c_void_p
    lib.orc_p8_cm_new.argtypes = [C.c_uint64, C.c_int]
    lib.orc_p8_cm_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    lib.orc_p8_cm_free.argtypes = [C.c_void_p]
    lib.orc_p8_rnd_next.restype = C.c_uint32
    lib.orc_p8_rnd_reset()
    hs = [lib.orc_p8_cm_new(s, c) for s, c in zip(sizes, counts)]
    S = sum(counts)
    rows = np.zeros((8 * len(data), 5 * S), np.int16)
    o, n_out = np.zeros(5 * max(counts) + 8, np.int16), C.c_int(0)
    y, c0 = 0, 1
    for n in range(len(data)):
        c1 = int(data[n - 1]) if n else 0
        for bpos in range(8):
            off = 0
            for k, h in enumerate(hs):
                c = np.ascontiguousarray(cxs[k][n])
                lib.orc_p8_cm_step(h, y, bpos, c0, c1, c.ctypes.data, counts[k], o.ctypes.data, C.byref(n_out))
                assert n_out.value == 5 * counts[k]
                rows[8 * n + bpos, off:off + 5 * counts[k]] = o[:5 * counts[k]]
                off += 5 * counts[k]
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
    for h in hs:
        lib.orc_p8_cm_free(h)
    return rows, lib.orc_p8_rnd_next()


def hashed(sizes, counts, cxs):
    from test_p8cm2_host import emul as emul2
    L2 = emul2()
    n = cxs[0].shape[0]
    S = sum(counts)
    c32, k16 = np.zeros((n, S), np.uint32), np.zeros((n, S), np.uint16)
    a, b = C.c_uint32(0), C.c_uint16(0)
    off = 0
    for k, cnt in enumerate(counts):
        for j in range(n):
            for i in range(cnt):
                L2.p8e_hash(int(cxs[k][j, i]), i, sizes[k], C.byref(a), C.byref(b))
                c32[j, off + i], k16[j, off + i] = a.value, b.value
        off += cnt
    return c32, k16


def run_emul(L, sizes, counts, data, cxs, chunks, seed=4242, serial=0, stats=None):
    nex, stretch, ilog = tables()
    sz, ct = np.array(sizes, np.uint64), np.array(counts, np.int32)
    h = L.p8f_create(len(sizes), sz.ctypes.data, ct.ctypes.data, nex.ctypes.data, stretch.ctypes.data, ilog.ctypes.data, seed, serial)
    assert h
    c32, k16 = hashed(sizes, counts, cxs)
    bits = np.unpackbits(np.ascontiguousarray(data, np.uint8))
    S = sum(counts)
    out = np.zeros((8 * len(data), 5 * S), np.int16)
    pos = 0
    for n in chunks:
        n = min(n, len(data) - pos)
        if n <= 0:
            break
        a, b = np.ascontiguousarray(c32[pos:pos + n]), np.ascontiguousarray(k16[pos:pos + n])
        bb, o = np.ascontiguousarray(bits[8 * pos:8 * (pos + n)]), out[8 * pos:8 * (pos + n)]
        assert L.p8f_run(h, a.ctypes.data, b.ctypes.data, bb.ctypes.data, n, o.ctypes.data) == 0
        pos += n
    if stats is not None:
        st = np.zeros(3, np.uint64)
        L.p8f_stats(h, st.ctypes.data)
        stats.extend(int(v) for v in st)
    L.p8f_destroy(h)
    return out


SIZES, COUNTS = [1 << 22, 1 << 16, 1 << 20, 1 << 18], [12, 6, 15, 4]


def stream(nbytes):
    """Half text, half a short phrase repeated: long deterministic runs push bit histories to the states >= 204 that draw."""
    from cmix_amd import synth
    text = synth.enwik_like(nbytes // 2, 17)
    return np.frombuffer(text + (text[:37] * (nbytes // 37 + 1))[:nbytes - len(text)], np.uint8)


def test_family_vs_oracle():
    L = emul()
    data = stream(3200)
    cxs = family_contexts(data, COUNTS, 3)
    want, _ = oracle_rows(SIZES, COUNTS, data, cxs)
    stats = []
    got = run_emul(L, SIZES, COUNTS, data, cxs, [1, 9, 700, 5000], stats=stats)
    bad = np.argwhere(got != want)
    assert bad.size == 0, ("first mismatch (step, input):", bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
    steps, serial, draws = stats
    assert draws > 1000 and 0 < serial < steps, stats    # both paths ran, and the generator was actually consumed in the parallel one
    got = run_emul(L, SIZES, COUNTS, data[:500], [c[:500] for c in cxs], [500], serial=1)
    assert np.array_equal(got, want[:8 * 500])


def test_generator_position_after_the_parallel_path():
    """Without the colliding instance every step takes the lane-per-context path; the family then must have consumed
    exactly as many values of the generator as the reference order did: one more oracle step after the comparison
    would diverge otherwise, so a second chunk is compared as well."""
    L = emul()
    sizes, counts = [1 << 22, 1 << 22], [10, 7]
    data = stream(2400)
    lib = O.lib()
    lib.orc_p8_combine64.restype = C.c_uint64
    lib.orc_p8_combine64.argtypes = [C.c_uint64, C.c_uint64]
    cxs = []
    for k, cnt in enumerate(counts):
        a = np.zeros((len(data), cnt), np.uint64)
        for n in range(len(data)):
            h = 77 * (k + 1)
            for i in range(cnt):
                h = lib.orc_p8_combine64(h, int(data[n - 1 - i]) if n - 1 - i >= 0 else 0)
                a[n, i] = h
        cxs.append(a)
    want, _ = oracle_rows(sizes, counts, data, cxs)
    stats = []
    got = run_emul(L, sizes, counts, data, cxs, [1200, 1200], stats=stats)
    assert np.array_equal(got, want)
    assert stats[2] > 500, stats
