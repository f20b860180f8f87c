"""CPU: the restated numeric building blocks of the vendored paq8 model (oracle/paq8_core.c: two-layer int16 mixer with
dot_product/train, APM1, StateMap, StateMap32, APM; SURVEY.md 8a') against the reference's own classes compiled from
paq8.cpp (oracle/ref_paq8core.cpp -> oracle/_ref/libcmixrefpaq8.so) and against committed vectors produced from them
(tests/golden/paq8core_vectors.npz, tests/golden/make_paq8core_vectors.py). Integer work: bit-exact."""
import ctypes as C
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O
from oracle import refharness as R


@pytest.fixture(autouse=True)
def _oracle_scope():
    """The restatement's objects have no destructors (orc_alloc.h): what a test builds and does not free is freed when it ends."""
    with O.scope():
        yield
    import _ctypes
    while _PRIVATE_COPIES:   # the reference's state of a whole-model run lives in that copy's globals: unloading it runs their destructors
        _ctypes.dlclose(_PRIVATE_COPIES.pop()._handle)


_PRIVATE_COPIES = []

needs_ref = pytest.mark.skipif(not R.paq8core_available(), reason="oracle/_ref/libcmixrefpaq8.so not built")


def mixer_case(seed, n, m, s, steps):
    """Seeded drive of a mixer: stretch-domain inputs, selectors with locality, bits correlated with the prediction."""
    rng = np.random.default_rng(seed)
    ranges = rng.integers(1, max(2, m // s), s).astype(np.int32)
    ranges[-1] = max(1, m - int(ranges[:-1].sum()) - 1) if ranges[:-1].sum() < m - 1 else 1
    while ranges.sum() > m:
        ranges[np.argmax(ranges)] //= 2
    ranges = np.maximum(ranges, 1)
    xs = np.clip(rng.normal(0, 600, (steps, n)), -2047, 2047).astype(np.int16)
    xs[rng.random((steps, n)) < 0.02] = 32767          # saturation paths of train (2t overflows 16 bits)
    xs[rng.random((steps, n)) < 0.02] = -32768
    cx = np.zeros((steps, s), np.int32)
    cur = rng.integers(0, ranges)
    for t in range(steps):
        move = rng.random(s) < 0.3
        cur = np.where(move, rng.integers(0, ranges), cur)
        cx[t] = cur
    return ranges, xs, cx, rng


def run_mixer(step_fn, handle, ranges, xs, cx, rng_bits):
    steps, n = xs.shape
    s = len(ranges)
    out_p = np.zeros(steps, np.int32)
    exported = np.zeros((steps, n + s + 8), np.float32)
    nexp = C.c_int(0)
    y = 0
    for t in range(steps):
        p = step_fn(handle, y, xs[t].ctypes.data, n, cx[t].ctypes.data, ranges.ctypes.data, s,
                    exported[t].ctypes.data, C.byref(nexp))
        out_p[t] = p
        y = int(rng_bits[t] < p / 4096.0)
    return out_p, exported[:, :nexp.value], nexp.value


@needs_ref
def test_tables_match_the_reference():
    L = R.paq8core_lib()
    sq, st = np.zeros(4096, np.int16), np.zeros(4096, np.int16)
    dt, stt = np.zeros(1024, np.int32), np.zeros(1024, np.uint8)
    L.refp8_tables(sq.ctypes.data, st.ctypes.data, dt.ctypes.data, stt.ctypes.data)
    lib = O.lib()
    assert [lib.orc_p8_squash(d) for d in range(-2300, 2300)] == [int(sq[min(max(d, -2048), 2047) + 2048]) if -2047 <= d <= 2047
                                                                   else (4095 if d > 2047 else 0) for d in range(-2300, 2300)]
    assert [lib.orc_p8_stretch(p) for p in range(4096)] == [int(v) for v in st]


@needs_ref
@pytest.mark.parametrize("n,m,s,w,steps", [(1552, 77472, 28, 32, 400), (64, 300, 5, 32, 3000), (8, 1, 1, 0x7fff, 2000)])
def test_mixer_vs_reference(n, m, s, w, steps):
    L, lib = R.paq8core_lib(), O.lib()
    ranges, xs, cx, rng = mixer_case(11 + n, n, m, s, steps)
    bits = rng.random(steps)
    ref = L.refp8_mixer_new(n, m, s, w)
    got = lib.orc_p8_mixer_new(n, m, s, w)
    p_ref, e_ref, k_ref = run_mixer(L.refp8_mixer_step, ref, ranges, xs, cx, bits)
    p_got, e_got, k_got = run_mixer(lib.orc_p8_mixer_step, got, ranges, xs, cx, bits)
    assert k_ref == k_got == n + (s + 0 if s > 1 else 0)
    bad = np.nonzero(p_ref != p_got)[0]
    assert len(bad) == 0, f"mixer output differs first at step {bad[0]}: {p_ref[bad[0]]} vs {p_got[bad[0]]}"
    assert (e_ref.view(np.uint32) == e_got.view(np.uint32)).all()
    assert len(np.unique(p_ref)) > steps // 20  # the drive exercises the mixer
    L.refp8_mixer_free(ref)
    lib.orc_p8_mixer_free(got)


def _adaptive_case(seed, steps, ncx):
    rng = np.random.default_rng(seed)
    pr = rng.integers(0, 4096, steps).astype(np.int32)
    cx = rng.integers(0, ncx, steps).astype(np.int32)
    cx[rng.random(steps) < 0.7] = 3                     # one hot context so that counts climb to their limits
    y = (rng.random(steps) < pr / 4096.0).astype(np.int32)
    return pr, cx, y


@needs_ref
def test_apm1_statemap_statemap32_apm_vs_reference():
    L, lib = R.paq8core_lib(), O.lib()
    steps = 20000
    pr, cx, y = _adaptive_case(5, steps, 1 << 10)
    a_ref, a_got = L.refp8_apm1_new(1 << 10), lib.orc_p8_apm1_new(1 << 10)
    for rate in (7, 6):
        r = [L.refp8_apm1_p(a_ref, int(y[t]), int(pr[t]), int(cx[t]), rate) for t in range(steps)]
        g = [lib.orc_p8_apm1_p(a_got, int(y[t]), int(pr[t]), int(cx[t]), rate) for t in range(steps)]
        assert r == g
    s_ref, s_got = L.refp8_statemap_new(), lib.orc_p8_statemap_new()
    assert [L.refp8_statemap_p(s_ref, int(y[t]), int(cx[t]) & 255) for t in range(steps)] == \
           [lib.orc_p8_statemap_p(s_got, int(y[t]), int(cx[t]) & 255) for t in range(steps)]
    for n, limit in ((256, 1023), (1 << 16, 1023), (1 << 10, 127)):
        q_ref, q_got = L.refp8_statemap32_new(n), lib.orc_p8_statemap32_new(n)
        assert [L.refp8_statemap32_p(q_ref, int(y[t]), int(cx[t]) % n, limit) for t in range(steps)] == \
               [lib.orc_p8_statemap32_p(q_got, int(y[t]), int(cx[t]) % n, limit) for t in range(steps)]
    for limit in (0xFF, 0x3FF >> 2, 0x3FF):
        p_ref, p_got = L.refp8_apm_new(1 << 10), lib.orc_p8_apm_new(1 << 10)
        assert [L.refp8_apm_p(p_ref, int(y[t]), int(pr[t]), int(cx[t]), limit) for t in range(steps)] == \
               [lib.orc_p8_apm_p(p_got, int(y[t]), int(pr[t]), int(cx[t]), limit) for t in range(steps)]


def test_golden_vectors():
    """The same drives, recorded from the reference (runs wherever the fixture is, oracle/_ref not needed)."""
    path = os.path.join(GOLDEN, "paq8core_vectors.npz")
    if not os.path.exists(path):
        pytest.skip("tests/golden/paq8core_vectors.npz not generated yet")
    lib = O.lib()
    with np.load(path) as z:
        v = {k: z[k] for k in z.files}
    n, m, s, w, steps = (int(x) for x in v["mixer_shape"])
    ranges, xs, cx, rng = mixer_case(int(v["mixer_seed"][0]), n, m, s, steps)
    bits = rng.random(steps)
    got = lib.orc_p8_mixer_new(n, m, s, w)
    p_got, e_got, _ = run_mixer(lib.orc_p8_mixer_step, got, ranges, xs, cx, bits)
    assert (p_got == v["mixer_p"]).all() and (e_got[-1].view(np.uint32) == v["mixer_exported_last"].view(np.uint32)).all()
    pr, cx1, y = _adaptive_case(5, len(v["apm1_p"]), 1 << 10)
    a = lib.orc_p8_apm1_new(1 << 10)
    assert [lib.orc_p8_apm1_p(a, int(y[t]), int(pr[t]), int(cx1[t]), 7) for t in range(len(pr))] == list(v["apm1_p"])
    q = lib.orc_p8_apm_new(1 << 10)
    assert [lib.orc_p8_apm_p(q, int(y[t]), int(pr[t]), int(cx1[t]), 0xFF) for t in range(len(pr))] == list(v["apm_p"])
    sm = lib.orc_p8_statemap32_new(256)
    assert [lib.orc_p8_statemap32_p(sm, int(y[t]), int(cx1[t]) & 255, 1023) for t in range(len(pr))] == list(v["sm32_p"])


# ---- context-to-prediction structures (oracle/paq8_maps.c) -----------------------------------------------------

def _byte_contexts(data, n, count):
    """contextModel2's order-N hashes of the bytes before position n (paq8.cpp:8139-8153): combine64 chains."""
    lib = O.lib()
    cx = [0] * count
    h = 0
    for k in range(count):
        b = data[n - 1 - k] if n - 1 - k >= 0 else 0
        h = lib.orc_p8_combine64(h, int(b))
        cx[k] = h
    return np.array(cx, np.uint64)


@needs_ref
def test_hash_helpers_and_ilog_vs_reference():
    L, lib = R.paq8core_lib(), O.lib()
    rng = np.random.default_rng(3)
    for _ in range(2000):
        a, b = (int(v) for v in rng.integers(0, 1 << 63, 2, dtype=np.uint64))
        assert lib.orc_p8_hash2(a, b) == L.refp8_hash2(a, b) and lib.orc_p8_combine64(a, b) == L.refp8_combine64(a, b)
        for bits in (12, 16, 22):
            assert lib.orc_p8_finalize64(a, bits) == L.refp8_finalize64(a, bits)
            assert lib.orc_p8_checksum64(a, bits, 16) == L.refp8_checksum64(a, bits, 16)
    t = np.zeros(65536, np.uint8)
    L.refp8_ilog_table(t.ctypes.data)
    lib.orc_p8_ilog.argtypes = [C.c_int]
    assert [lib.orc_p8_ilog(i) for i in range(65536)] == [int(v) for v in t]


@needs_ref
@pytest.mark.parametrize("size_bytes,count,nbytes", [(1 << 16, 10, 6000), (1 << 22, 6, 6000)])
def test_contextmap2_vs_reference(size_bytes, count, nbytes):
    """Order-1..N contexts over text-like bytes; the small table forces bucket replacement (Find's priority rule, the
    MRU pair) all the time, the large one lets byte histories and run statistics mature."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    data = np.frombuffer(synth.enwik_like(nbytes, 13), np.uint8)
    ref, got = L.refp8_cm2_new(size_bytes, count), lib.orc_p8_cm2_new(size_bytes, count)
    o_ref, o_got = np.zeros(256, np.int16), np.zeros(256, np.int16)
    n_ref, n_got = C.c_int(0), C.c_int(0)
    y = 0
    for n in range(nbytes):
        cx = _byte_contexts(data, n, count)
        for bpos in range(8):
            r = L.refp8_cm2_step(ref, y, bpos, cx.ctypes.data, count, o_ref.ctypes.data, C.byref(n_ref))
            g = lib.orc_p8_cm2_step(got, y, bpos, cx.ctypes.data, count, o_got.ctypes.data, C.byref(n_got))
            assert r == g and n_ref.value == n_got.value == 7 * count, (n, bpos, r, g, n_ref.value, n_got.value)
            assert (o_ref[:n_ref.value] == o_got[:n_got.value]).all(), f"byte {n} bit {bpos}: {o_ref[:70]} vs {o_got[:70]}"
            y = (int(data[n]) >> (7 - bpos)) & 1
    L.refp8_cm2_free(ref)
    lib.orc_p8_cm2_free(got)


@needs_ref
def test_direct_lookup_maps_vs_reference():
    L, lib = R.paq8core_lib(), O.lib()
    rng = np.random.default_rng(8)
    nbytes = 4000
    data = rng.integers(0, 4, nbytes) * 37 % 256
    o_ref, o_got = np.zeros(8, np.int16), np.zeros(8, np.int16)
    cases = [("sscm", 0, (11, 8, 0), (7, 1, 4)), ("sscm", 0, (8, 1, 0), (6, 3, 2)),
             ("smap", 1, (16, 8, 0), (1023, 1, 4)), ("smap", 1, (10, 3, 200), (255, 2, 1)),
             ("imap", 2, (12, 8, 0), (1023, 1, 4)), ("imap", 2, (9, 2, 0), (127, 3, 2))]
    for name, kind, (boc, bpc, rate), (a, mul, div) in cases:
        if name == "sscm":
            ref = L.refp8_sscm_new(boc, bpc)
        elif name == "smap":
            ref = L.refp8_smap_new(boc, bpc, rate)
        else:
            ref = L.refp8_imap_new(boc, bpc)
        got = lib.orc_p8_dmap_new(kind, boc, bpc, rate)
        y, bit_in_ctx = 0, 0
        for n in range(nbytes):
            for bpos in range(8):
                if bit_in_ctx == 0:  # a new context every bpc bits
                    ctx = int(rng.integers(0, 1 << 20)) if n % 3 else 5
                    if name == "sscm":
                        L.refp8_sscm_set(ref, ctx)
                    elif name == "smap":
                        L.refp8_smap_set_direct(ref, ctx) if n % 2 else L.refp8_smap_set(ref, ctx * 0x9E3779B97F4A7C15 % (1 << 64))
                    else:
                        L.refp8_imap_set_direct(ref, ctx) if n % 2 else L.refp8_imap_set(ref, ctx * 0x9E3779B97F4A7C15 % (1 << 64))
                    if name == "sscm" or n % 2:
                        lib.orc_p8_dmap_set_direct(got, ctx)
                    else:
                        lib.orc_p8_dmap_set(got, ctx * 0x9E3779B97F4A7C15 % (1 << 64))
                if name == "sscm":
                    k = L.refp8_sscm_mix(ref, y, a, mul, div, o_ref.ctypes.data)
                elif name == "smap":
                    k = L.refp8_smap_mix(ref, y, mul, div, a, o_ref.ctypes.data)
                else:
                    k = L.refp8_imap_mix(ref, y, mul, div, a, o_ref.ctypes.data)
                assert lib.orc_p8_dmap_mix(got, y, a, mul, div, o_got.ctypes.data) == k == 2
                assert (o_ref[:2] == o_got[:2]).all(), (name, boc, bpc, n, bpos, o_ref[:2], o_got[:2])
                y = (int(data[n]) >> (7 - bpos)) & 1
                bit_in_ctx = (bit_in_ctx + 1) % bpc


@needs_ref
def test_global_rnd_vs_reference():
    L, lib = R.paq8core_lib(), O.lib()
    L.refp8_rnd_reset()
    lib.orc_p8_rnd_reset()
    assert [L.refp8_rnd_next() for _ in range(500)] == [lib.orc_p8_rnd_next() for _ in range(500)]


@needs_ref
@pytest.mark.parametrize("size_bytes,count,nbytes", [(1 << 16, 12, 5000), (1 << 22, 5, 12000)])
def test_contextmap_vs_reference(size_bytes, count, nbytes):
    """The older ContextMap (wordModel, sparseModel, indirectModel, ...): same bucket, u16 StateMap, and the
    process-global rnd() drawn only when a bit-history state >= 204 comes up (long deterministic runs reach them:
    the second case is a repetitive stream)."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    text = synth.enwik_like(nbytes, 17)
    data = np.frombuffer(text if count > 5 else (text[:40] * (nbytes // 40 + 1))[:nbytes], np.uint8)
    L.refp8_rnd_reset()
    lib.orc_p8_rnd_reset()
    ref, got = L.refp8_cm_new(size_bytes, count), lib.orc_p8_cm_new(size_bytes, count)
    o_ref, o_got = np.zeros(256, np.int16), np.zeros(256, np.int16)
    n_ref, n_got = C.c_int(0), C.c_int(0)
    y, c0 = 0, 1
    for n in range(nbytes):
        cx = _byte_contexts(data, n, count)
        c1 = int(data[n - 1]) if n else 0
        for bpos in range(8):
            r = L.refp8_cm_step(ref, y, bpos, c0, c1, cx.ctypes.data, count, o_ref.ctypes.data, C.byref(n_ref))
            g = lib.orc_p8_cm_step(got, y, bpos, c0, c1, cx.ctypes.data, count, o_got.ctypes.data, C.byref(n_got))
            assert r == g and n_ref.value == n_got.value == 5 * count, (n, bpos, r, g)
            assert (o_ref[:n_ref.value] == o_got[:n_got.value]).all(), f"byte {n} bit {bpos}"
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
    assert L.refp8_rnd_next() == lib.orc_p8_rnd_next()  # the same number of draws happened on both sides
    L.refp8_cm_free(ref)
    lib.orc_p8_cm_free(got)


@needs_ref
def test_runcontextmap_vs_reference():
    """RunContextMap over BH<4>: 8-way probe, move-to-front, replacement of the lower-priority of the last two."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    for m, nbytes, order in ((1 << 10, 6000, 2), (1 << 16, 6000, 4)):  # the tiny table overflows its probe windows
        data = np.frombuffer(synth.enwik_like(nbytes, 23), np.uint8)
        ref, got = L.refp8_rcm_new(m), lib.orc_p8_rcm_new(m)
        o_ref, o_got = np.zeros(4, np.int16), np.zeros(4, np.int16)
        c0 = 1
        for n in range(nbytes):
            cx = int(_byte_contexts(data, n, order)[-1])
            c1 = int(data[n - 1]) if n else 0
            L.refp8_rcm_set(ref, cx, c1)
            lib.orc_p8_rcm_set(got, cx, c1)
            for bpos in range(8):
                assert L.refp8_rcm_mix(ref, bpos, c0, o_ref.ctypes.data) == lib.orc_p8_rcm_mix(got, bpos, c0, o_got.ctypes.data)
                assert o_ref[0] == o_got[0], (n, bpos, o_ref[0], o_got[0])
                y = (int(data[n]) >> (7 - bpos)) & 1
                c0 = (c0 << 1 | y) if bpos < 7 else 1


@needs_ref
@pytest.mark.parametrize("level,nbytes", [(0, 40000), (5, 20000)])
def test_dmc_forest_vs_reference(level, nbytes):
    """Ten DMC state graphs: cloning, threshold growth, and -- at level 0, where the graphs fill within a few KB --
    the `isfull` resets of the eight fast models."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    data = np.frombuffer(synth.enwik_like(nbytes, 29), np.uint8)
    ref, got = L.refp8_dmc_new(level), lib.orc_p8_dmc_new(level)
    o_ref, o_got = np.zeros(8, np.int16), np.zeros(8, np.int16)
    y = 0
    for n in range(nbytes):
        for bpos in range(8):
            assert L.refp8_dmc_mix(ref, y, bpos, o_ref.ctypes.data) == lib.orc_p8_dmc_mix(got, y, bpos, o_got.ctypes.data) == 6
            assert (o_ref[:6] == o_got[:6]).all(), (n, bpos, o_ref[:6], o_got[:6])
            y = (int(data[n]) >> (7 - bpos)) & 1
    L.refp8_dmc_free(ref)
    lib.orc_p8_dmc_free(got)


@needs_ref
def test_linear_prediction_model_vs_reference():
    """Three OLS<double> recursive least-squares predictors + two fixed extrapolations, read through SSCMs: the only
    double-precision arithmetic on paq8's text path. Text, then a smooth ramp (where the predictors lock on), then
    noise. One reference instance per process (function-local statics)."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    rng = np.random.default_rng(31)
    text = np.frombuffer(synth.enwik_like(3000, 41), np.uint8)
    ramp = ((np.arange(3000) * 3 + (np.sin(np.arange(3000) / 9.0) * 20).astype(int)) & 255).astype(np.uint8)
    data = np.concatenate([text, ramp, rng.integers(0, 256, 2000, dtype=np.uint8)])
    got = lib.orc_p8_lpm_new()
    o_ref, o_got = np.zeros(16, np.int16), np.zeros(16, np.int16)
    hist = np.zeros(64, np.uint8)  # hist[i-1] = buf(i)
    y, c0 = 0, 1
    seen = set()
    for n in range(len(data)):
        for bpos in range(8):
            k = L.refp8_lpm_step(y, bpos, c0, hist.ctypes.data, 64, o_ref.ctypes.data)
            assert lib.orc_p8_lpm_step(got, y, bpos, c0, hist.ctypes.data, o_got.ctypes.data) == k == 10
            assert (o_ref[:10] == o_got[:10]).all(), (n, bpos, o_ref[:10], o_got[:10])
            seen.add(int(o_ref[0]))
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        hist = np.concatenate([[data[n]], hist[:-1]]).astype(np.uint8)
    assert len(seen) > 50


@needs_ref
def test_nest_distance_indirect_models_vs_reference():
    """Three of paq8's context models end to end (state machines -> hashed contexts -> ContextMap -> mixer inputs), each
    against the reference's own function. Markup-heavy text (brackets, quotes, entities) for the nesting state."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    rng = np.random.default_rng(37)
    for a, b in ((1, 2), (3, 4), (1 << 40, 7)):
        assert lib.orc_p8_hash3(a, b, 9) == L.refp8_hash3(a, b, 9) and lib.orc_p8_hash6(a, b, 3, 4, 5, 6) == L.refp8_hash6(a, b, 3, 4, 5, 6)
    text = synth.enwik_like(5000, 43) + b"&lt;&gt; (a [b {c} 'd' \"e\"]) = == 'x' / | \\ # % $ * - @ ; : \xc3\xa9\xc3\xa8 " * 20
    data = np.frombuffer(text, np.uint8)
    level = 6
    L.refp8_rnd_reset()
    lib.orc_p8_rnd_reset()
    for which, nout in ((0, 60), (1, 15), (2, 75)):
        got = lib.orc_p8_ctxmodel_new(which, level)
        o_ref, o_got = np.zeros(128, np.int16), np.zeros(128, np.int16)
        hist = np.zeros(8, np.uint8)
        y, c0, c4, f4 = 0, 1, 0, 0
        for n in range(len(data)):
            for bpos in range(8):
                k = L.refp8_ctxmodel_step(which, level, y, bpos, c0, c4, f4, n, hist.ctypes.data, 8, o_ref.ctypes.data)
                assert lib.orc_p8_ctxmodel_step(got, y, bpos, c0, c4, f4, n, hist.ctypes.data, o_got.ctypes.data) == k == nout
                assert (o_ref[:k] == o_got[:k]).all(), (which, n, bpos, o_ref[:k], o_got[:k])
                y = (int(data[n]) >> (7 - bpos)) & 1
                c0 = (c0 << 1 | y) if bpos < 7 else 1
            c4 = ((c4 << 8) | int(data[n])) & 0xffffffff
            f4 = ((f4 << 4) | (int(data[n]) >> 4)) & 0xffffffff
            hist = np.concatenate([[data[n]], hist[:-1]]).astype(np.uint8)


@needs_ref
def test_sparse_models_vs_reference():
    """sparseModel (42 skip contexts) and sparseModel1 (31 contexts + 7 SSCMs) against the reference's own functions;
    the word-level globals they read are inputs here (arbitrary but history-derived values)."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    data = np.frombuffer(synth.enwik_like(4000, 47), np.uint8)
    level = 5
    L.refp8_rnd_reset()
    lib.orc_p8_rnd_reset()
    for which, nout in ((0, 42 * 5), (1, 29 * 5 + 14)):  # sparseModel1 sets 29 of its 31 slots
        got = lib.orc_p8_sparse_new(which, level)
        o_ref, o_got = np.zeros(512, np.int16), np.zeros(512, np.int16)
        hist = np.zeros(10, np.uint8)
        g = np.zeros(9, np.uint32)
        y, c0 = 0, 1
        for n in range(len(data)):
            seen, many = int(g[2] % 7), int(g[3] % 11)
            for bpos in range(8):
                k = L.refp8_sparse_step(which, level, y, bpos, c0, g.ctypes.data, seen, many, hist.ctypes.data, 10, o_ref.ctypes.data)
                assert lib.orc_p8_sparse_step(got, y, bpos, c0, g.ctypes.data, seen, many, hist.ctypes.data, o_got.ctypes.data) == k == nout
                assert (o_ref[:k] == o_got[:k]).all(), (which, n, bpos)
                y = (int(data[n]) >> (7 - bpos)) & 1
                c0 = (c0 << 1 | y) if bpos < 7 else 1
            b = int(data[n])
            g[0] = ((int(g[0]) << 8) | b) & 0xffffffff                      # c4
            g[1] = ((int(g[1]) << 4) | (b >> 4)) & 0xffffffff               # f4
            g[2] = (int(g[2]) * 256 + b) & 0xffffffff                       # x4
            g[3] = (int(g[3]) * 4 + (b >> 6)) & 0xffffffff                  # w4
            g[4] = (int(g[4]) * 8 + (b & 7)) & 0xffffffff                   # tt
            g[5] = (int(g[5]) * 2 + (1 if chr(b).isalpha() else 0)) & 0xffffffff   # words
            g[6] = (int(g[6]) * 2 + (1 if b == 32 else 0)) & 0xffffffff     # spaces
            g[7] = b if b in (32, 61, 91) else int(g[7])                    # frstchar
            g[8] = 0 if b == 46 else min(63, int(g[8]) + 1)                 # spafdo
            hist = np.concatenate([[data[n]], hist[:-1]]).astype(np.uint8)


@needs_ref
def test_match_model_vs_reference():
    """MatchModel: repeated passages (long matches, extension, recovery after a miss -> delta mode), against the
    reference's own class over its own ring buffer."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    a = synth.enwik_like(1500, 53)
    data = np.frombuffer(a + a[200:900] + b"XY" + a[300:1200] + a[:700] + bytes(50) + a[100:600], np.uint8)
    LOG = 16
    L.refp8_buf_reset(LOG)
    ring = np.zeros(1 << LOG, np.uint8)
    ref, got = L.refp8_match_new(1 << 18), lib.orc_p8_match_new(1 << 18)
    o_ref, o_got = np.zeros(32, np.int16), np.zeros(32, np.int16)
    n_ref, n_got, e_ref, e_got = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    y, c0, longest = 0, 1, 0
    for n in range(len(data)):
        for bpos in range(8):
            r = L.refp8_match_step(ref, y, bpos, c0, o_ref.ctypes.data, C.byref(n_ref), C.byref(e_ref))
            g = lib.orc_p8_match_step(got, y, bpos, c0, ring.ctypes.data, (1 << LOG) - 1, n, o_got.ctypes.data, C.byref(n_got),
                                      C.byref(e_got))
            assert r == g and n_ref.value == n_got.value == 17, (n, bpos, r, g, n_ref.value, n_got.value)
            assert (o_ref[:17] == o_got[:17]).all(), (n, bpos, o_ref[:17], o_got[:17])
            if bpos == 0:
                assert e_ref.value == e_got.value
            longest = max(longest, r)
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        L.refp8_buf_push(int(data[n]))
        ring[n] = data[n]
    assert longest > 400


@needs_ref
def test_sparse_match_model_vs_reference():
    """SparseMatchModel: repeats with case flips (mask 0xDF), a skipped byte, every-other-byte repeats and nibble-only
    repeats, so that each of the four finders wins at some point (move-to-front order changes)."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    a = synth.enwik_like(1200, 59)
    up = bytes(c ^ 0x20 if 97 <= c <= 122 else c for c in a[100:500])
    inter = bytes(b for pair in zip(a[200:500], bytes(i % 251 for i in range(300))) for b in pair)
    nib = bytes((c & 0x0F) | 0x40 for c in a[300:700])
    data = np.frombuffer(a + up + b"#" + a[101:600] + inter + inter + nib + a[:400] + nib, np.uint8)
    LOG = 16
    L.refp8_buf_reset(LOG)
    ring = np.zeros(1 << LOG, np.uint8)
    ref, got = L.refp8_sparsematch_new(1 << 18), lib.orc_p8_sparsematch_new(1 << 18)
    o_ref, o_got = np.zeros(32, np.int16), np.zeros(32, np.int16)
    s_ref, s_got = np.zeros(8, np.int32), np.zeros(8, np.int32)
    n_ref, n_got, k_ref = C.c_int(0), C.c_int(0), C.c_int(0)
    y, c0 = 0, 1
    winners = set()
    for n in range(len(data)):
        for bpos in range(8):
            r = L.refp8_sparsematch_step(ref, y, bpos, c0, o_ref.ctypes.data, C.byref(n_ref), s_ref.ctypes.data, C.byref(k_ref))
            g = lib.orc_p8_sparsematch_step(got, y, bpos, c0, ring.ctypes.data, (1 << LOG) - 1, n, o_got.ctypes.data,
                                            C.byref(n_got), s_got.ctypes.data)
            assert r == g and n_ref.value == n_got.value == 11 and k_ref.value == 2, (n, bpos, r, g, n_ref.value, n_got.value)
            assert (o_ref[:11] == o_got[:11]).all(), (n, bpos, o_ref[:11], o_got[:11])
            assert (s_ref[:2] == s_got[:2]).all(), (n, bpos, s_ref[:2], s_got[:2])
            if r > 1:
                winners.add(int(s_ref[0]) >> 6)
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        L.refp8_buf_push(int(data[n]))
        ring[n] = data[n]
    assert len(winners) >= 3, winners


@needs_ref
def test_pic_and_record1_models_vs_reference():
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    data = np.frombuffer(synth.enwik_like(3000, 61), np.uint8)
    LOG = 16
    L.refp8_rnd_reset()
    lib.orc_p8_rnd_reset()
    for which, nout in ((0, 3), (1, (2 + 5 + 4 + 3 + 3) * 5)):
        L.refp8_buf_reset(LOG)
        ring = np.zeros(1 << LOG, np.uint8)
        got = lib.orc_p8_small_new(which)
        o_ref, o_got = np.zeros(128, np.int16), np.zeros(128, np.int16)
        y, c0, c4, f4, w5 = 0, 1, 0, 0, 0
        for n in range(len(data)):
            for bpos in range(8):
                k = L.refp8_small_step(which, y, bpos, c0, c4, f4, w5, o_ref.ctypes.data)
                assert lib.orc_p8_small_step(got, y, bpos, c0, c4, f4, w5, ring.ctypes.data, (1 << LOG) - 1, n, o_got.ctypes.data) == k == nout
                assert (o_ref[:k] == o_got[:k]).all(), (which, n, bpos, o_ref[:k], o_got[:k])
                y = (int(data[n]) >> (7 - bpos)) & 1
                c0 = (c0 << 1 | y) if bpos < 7 else 1
            b = int(data[n])
            L.refp8_buf_push(b)
            ring[n] = b
            c4 = ((c4 << 8) | b) & 0xffffffff
            f4 = ((f4 << 4) | (b >> 4)) & 0xffffffff
            w5 = (w5 * 4 + (b >> 6)) & 0xffffffff


@needs_ref
def test_record_model_vs_reference():
    """recordModel: fixed-length records with space / zero padding (length detection, the two candidates, padding
    transitions), a length change, a multiple-of-3 length (the 24-bit-image guess), then text."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    rng = np.random.default_rng(67)

    def records(n, length, pad):
        out = bytearray()
        for k in range(n):
            name = bytes(rng.integers(97, 123, int(rng.integers(3, length - 6))).astype(np.uint8))
            out += (name + bytes([pad]) * length)[:length - 4] + int(k * 37).to_bytes(4, "little")
        return bytes(out)
    data = np.frombuffer(records(120, 24, 32) + records(80, 40, 0) + records(60, 33, 32) + synth.enwik_like(1500, 71), np.uint8)
    LOG, level = 16, 4
    L.refp8_rnd_reset()
    lib.orc_p8_rnd_reset()
    L.refp8_buf_reset(LOG)
    ring = np.zeros(1 << LOG, np.uint8)
    got = lib.orc_p8_record_new(level)
    o_ref, o_got = np.zeros(256, np.int16), np.zeros(256, np.int16)
    s_ref, s_got = np.zeros(8, np.int32), np.zeros(8, np.int32)
    io_ref, io_got = np.zeros(6, np.uint32), np.zeros(6, np.uint32)
    k_ref = C.c_int(0)
    y, c0, c4 = 0, 1, 0
    lens = set()
    for n in range(len(data)):
        for bpos in range(8):
            for io in (io_ref, io_got):
                io[0], io[1], io[2] = n, (c0 * 7) & 31, 4 if n > 7000 else 0
                io[4], io[5] = (n // 50) % 3, int(data[n - 3]) if n > 3 else 0
            k = L.refp8_record_step(level, y, bpos, c0, c4, io_ref.ctypes.data, o_ref.ctypes.data, s_ref.ctypes.data, C.byref(k_ref))
            g = lib.orc_p8_record_step(got, y, bpos, c0, c4, io_got.ctypes.data, ring.ctypes.data, (1 << LOG) - 1, n, o_got.ctypes.data,
                                       s_got.ctypes.data)
            assert k == g == 149 and k_ref.value == 3, (n, bpos, k, g)
            assert (o_ref[:k] == o_got[:k]).all(), (n, bpos, np.nonzero(o_ref[:k] != o_got[:k])[0][:5])
            assert (s_ref[:3] == s_got[:3]).all() and io_ref[3] == io_got[3], (n, bpos, s_ref[:3], s_got[:3], io_ref[3], io_got[3])
            lens.add(int(io_ref[3]) >> 16)
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        b = int(data[n])
        L.refp8_buf_push(b)
        ring[n] = b
        c4 = ((c4 << 8) | b) & 0xffffffff
    assert {24, 40, 33} <= lens, lens


@needs_ref
def test_xml_model_vs_reference():
    """XMLModel: nested tags with attributes (href / http links), empty tags, comments, CDATA, indentation with spaces
    and tabs, CRLF and LF line ends, dates, times, numbers, coordinates, temperatures, ISBN -- plus enwik-like text."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    doc = (b'<?xml version="1.0"?>\r\n<root a="1" href="http://example.org/x" b=\'two\'>\r\n  <item id="7">\r\n    <date>2006-03-14</date>\n'
           b'    <time>12:34:56</time> <t2>9:05:01</t2>\n\t<empty/>\n\t\t<deep x="https://a.b/c"><!-- a comment --></deep>\n'
           b'    <![CDATA[ raw <stuff> ]]]]>\n    <coord>12\xc2\xb034\'56</coord><temp>21 \xc2\xb0C</temp><temp>7\xc2\xb0F</temp>\n'
           b'    <isbn>ISBN 0-306-40615-2</isbn><n>1234567890 text of more than eight letters</n>\n  </item>\n</root>\n'
           b'<a><b><c>31-12-1999</c></b></a><!x><1bad> < notatag >\n') * 3
    data = np.frombuffer(doc + synth.enwik_like(5000, 73), np.uint8)
    LOG, level = 16, 4
    L.refp8_rnd_reset()
    lib.orc_p8_rnd_reset()
    L.refp8_buf_reset(LOG)
    ring = np.zeros(1 << LOG, np.uint8)
    got = lib.orc_p8_xml_new(level)
    o_ref, o_got = np.zeros(64, np.int16), np.zeros(64, np.int16)
    x_ref, x_got = C.c_uint32(0), C.c_uint32(0)
    y, c0, c4 = 0, 1, 0
    states = set()
    for n in range(len(data)):
        for bpos in range(8):
            k = L.refp8_xml_step(level, y, bpos, c0, c4, o_ref.ctypes.data, C.byref(x_ref))
            g = lib.orc_p8_xml_step(got, y, bpos, c0, c4, ring.ctypes.data, (1 << LOG) - 1, n, o_got.ctypes.data, C.byref(x_got))
            assert k == g == 20 and x_ref.value == x_got.value, (n, bpos, k, g, x_ref.value, x_got.value)
            assert (o_ref[:k] == o_got[:k]).all(), (n, bpos, o_ref[:k], o_got[:k])
            states.add(x_ref.value & 7)
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        b = int(data[n])
        L.refp8_buf_push(b)
        ring[n] = b
        c4 = ((c4 << 8) | b) & 0xffffffff
    assert states == set(range(8)), states


@needs_ref
def test_exe_model_vs_reference():
    """exeModel (forced on, as contextModel2 runs it): x86-like opcode soup with prefixes, REX, 0F / 0F38 / 0F3A
    escapes, ModRM + SIB forms and immediates, random bytes (decoder errors), and text."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    rng = np.random.default_rng(79)
    ops = [b"\x55\x8b\xec", b"\x83\xec\x10", b"\x8b\x45\x08", b"\x89\x44\x24\x04", b"\xe8\x10\x00\x00\x00", b"\x0f\x84\x20\x01\x00\x00",
           b"\x66\x89\x06", b"\x48\x8b\x05\x10\x20\x00\x00", b"\x48\xb8\x01\x02\x03\x04\x05\x06\x07\x08", b"\xf3\xa4", b"\xc3", b"\x90",
           b"\x0f\x38\x00\xc1", b"\x0f\x3a\x0f\xc1\x04", b"\xff\x15\x00\x10\x40\x00", b"\x8d\x04\x8d\x00\x00\x00\x00", b"\xc8\x10\x00\x01",
           b"\x9a\x01\x02\x03\x04\x05\x06", b"\xf7\xd8", b"\xfe\xc0", b"\x64\x67\x8b\x00", b"\x0f\x0b", b"\xd9\xee", b"\xeb\xfe"]
    soup = b"".join(ops[int(k)] for k in rng.integers(0, len(ops), 600))
    data = np.frombuffer(soup + bytes(rng.integers(0, 256, 1500, dtype=np.uint8)) + synth.enwik_like(2500, 83) + soup[:800], np.uint8)
    LOG, level = 16, 3
    L.refp8_buf_reset(LOG)
    ring = np.zeros(1 << LOG, np.uint8)
    got = lib.orc_p8_exe_new(level)
    o_ref, o_got = np.zeros(256, np.int16), np.zeros(256, np.int16)
    s_ref, s_got = np.zeros(8, np.int32), np.zeros(8, np.int32)
    k_ref, x_ref, x_got = C.c_int(0), C.c_uint32(0), C.c_uint32(0)
    y, c0, c4 = 0, 1, 0
    valid_seen = False
    for n in range(len(data)):
        for bpos in range(8):
            k = L.refp8_exe_step(level, y, bpos, c0, c4, n, o_ref.ctypes.data, s_ref.ctypes.data, C.byref(k_ref), C.byref(x_ref))
            g = lib.orc_p8_exe_step(got, y, bpos, c0, c4, n, ring.ctypes.data, (1 << LOG) - 1, n, o_got.ctypes.data, s_got.ctypes.data,
                                    C.byref(x_got))
            assert k == g == 140 and k_ref.value == 6, (n, bpos, k, g, k_ref.value)
            assert x_ref.value == x_got.value, (n, bpos, hex(x_ref.value), hex(x_got.value))
            assert (s_ref[:6] == s_got[:6]).all(), (n, bpos, s_ref[:6], s_got[:6])
            assert (o_ref[:k] == o_got[:k]).all(), (n, bpos, np.nonzero(o_ref[:k] != o_got[:k])[0][:5])
            valid_seen |= bool(x_ref.value & 1)
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        b = int(data[n])
        L.refp8_buf_push(b)
        ring[n] = b
        c4 = ((c4 << 8) | b) & 0xffffffff
    assert valid_seen


@needs_ref
def test_english_stemmer_vs_reference():
    """EnglishStemmer on every word of a 6 000-word vocabulary drawn from the synthetic corpus generator's word list
    plus inflected / prefixed / possessive forms of each: stem letters, Start / End, type flags, language, both hash
    sets. (The reference's own dictionary is the natural list, but it is not available on every box; the generator's
    list is derived from it.)"""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    sig = [C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.refp8_en_stem_word.argtypes = sig
    lib.orc_p8_en_stem_word.argtypes = sig
    text = synth.enwik_like(400000, 3)
    base = sorted({w.lower() for w in text.replace(b"\n", b" ").split(b" ") if w.isalpha() and len(w) < 40})[:6000]
    extra = [b"skis", b"skies", b"dying", b"idly", b"news", b"atlas", b"inning", b"proceed", b"zinc", b"here", b"he", b"she", b"the",
             b"can't", b"won't", b"ain't", b"isn't", b"o'clock", b"'tis", b"non-linear", b"nonsense", b"overestimate", b"underground",
             b"irregular", b"unnatural", b"biggest", b"suggest", b"fullest", b"congest", b"happiest", b"smallest", b"finest", b"nearest",
             b"slowest", b"highest", b"childhood", b"neighbourhood", b"quizzing", b"squeaking", b"generously", b"communal", b"arsenic",
             b"y", b"yy", b"a", b"by", b"say", b"yellowy", b"x" * 70]
    suffixes = [b"", b"s", b"es", b"ed", b"ing", b"ly", b"ness", b"est", b"'s", b"ation", b"ational", b"fully", b"less", b"ize", b"ied", b"ies",
                b"edly", b"ingly", b"ative", b"ement", b"n't"]
    words = extra + [w + sfx for w in base for sfx in suffixes[: 1 + (len(w) % 7) * 3]]
    bufs = [(np.zeros(64, np.uint8), np.zeros(2, np.int32), np.zeros(2, np.uint64), np.zeros(4, np.uint64), np.zeros(4, np.uint64)) for _ in range(2)]
    changed = 0
    for w in words:
        out = []
        for fn, (let, se, tl, h1, h2) in ((L.refp8_en_stem_word, bufs[0]), (lib.orc_p8_en_stem_word, bufs[1])):
            r = fn(w, let.ctypes.data, se.ctypes.data, tl.ctypes.data, h1.ctypes.data, h2.ctypes.data)
            out.append((r, let.tobytes(), tuple(se), int(tl[0]), int(tl[1]) if r else -1, tuple(h1), tuple(h2)))
        assert out[0] == out[1], (w, out[0][:5], out[1][:5])
        changed += out[0][0]
    assert changed > len(words) // 4


@needs_ref
def test_french_and_german_stemmers_vs_reference():
    """FrenchStemmer and GermanStemmer (TextModel stems every completed word in all three languages): synthetic stems
    (consonant / vowel syllables with accented letters, both as Latin-1 bytes and as UTF-8 pairs, q-u / y / i vowel
    marking, sharp s) crossed with every suffix the step tables know, the exception and common words, and the
    synthetic corpus' English vocabulary. Stem letters, Start / End, type flags, language, both hash sets."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    sig = [C.c_int, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.refp8_stem_word.argtypes = sig
    lib.orc_p8_stem_word.argtypes = sig
    rng = np.random.default_rng(2024)
    fr_sfx = ("ance iqUe isme able iste eux ances iques ismes ables istes atrice ateur ation atrices ateurs ations logie logies usion ution "
              "usions utions ence ences issement issements ement ements it\xe9 it\xe9s if ive ifs ives euse euses ment ments issaient issantes "
              "iraient issante issants issions irions issais issait issant issent issiez issons irais irait irent iriez irons iront isses issez "
              "\xeemes \xeetes irai iras irez isse ies ira \xeet ie ir is it i eraient assions erions assent assiez \xe8rent erais erait eriez "
              "erons eront aient antes asses ions erai eras erez \xe2mes \xe2tes ante ants asse \xe9es era iez ais ait ant \xe9e \xe9s er ez "
              "\xe2t ai as \xe9 a i\xe8re ion ier e \xeb gu\xeb enn onn ett ell eill eaux aux amment emment s ivement ativement eusement "
              "ablement iquement i\xe8rement abilit\xe9 icit\xe9 ivit\xe9 icatif atif icateur y \xe7").split()
    de_sfx = ("em ern er e en es s nisse nissen st est end ung igung ik ig isch lich heit keit lichkeit igkeit erheit enheit ungen "
              "\xdf \xdfe \xdfen").split()
    cons, vow = list("bcdfghjklmnpqrstvwxz") + ["qu", "ch", "ss", "ll", "gu", "\xe7", "\xdf"], list("aeiouy") + ["\xe9", "\xe8", "\xea", "\xe2", "\xee", "\xf4", "\xfb", "\xe4", "\xf6", "\xfc", "ou", "ai", "ie", "ue", "uy", "ay"]

    def lat(t):
        return t.encode("latin-1").decode("unicode_escape").encode("latin-1")

    def utf(b):
        return b"".join(bytes([0xC3, c - 0x40]) if c >= 0xC0 else bytes([c]) for c in b)

    stems = [b"par", b"col", b"tap", b"", b"a", b"ou", b"monument", b"yeux", b"travaux", b"de", b"pas", b"une", b"der", b"nicht", b"sich",
             b"parl", b"fin", b"chant", b"gross", b"klein", b"freund", b"sch\xf6n".decode("unicode_escape").encode("latin-1")]
    for _ in range(900):
        k = int(rng.integers(1, 5))
        parts = []
        if rng.random() < 0.3:
            parts.append(vow[int(rng.integers(len(vow)))])
        for _ in range(k):
            parts.append(cons[int(rng.integers(len(cons)))])
            parts.append(vow[int(rng.integers(len(vow)))])
        if rng.random() < 0.6:
            parts.append(cons[int(rng.integers(len(cons)))])
        stems.append(lat("".join(parts)))
    english = sorted({w.lower() for w in synth.enwik_like(200000, 5).replace(b"\n", b" ").split(b" ") if w.isalpha() and len(w) < 40})[:3000]
    bufs = [(np.zeros(64, np.uint8), np.zeros(2, np.int32), np.zeros(2, np.uint64), np.zeros(4, np.uint64), np.zeros(4, np.uint64)) for _ in range(2)]
    for lang, sfxs in ((2, fr_sfx), (3, de_sfx)):
        words = list(english) + [b"x" * 70, b"\xc3".decode("unicode_escape").encode("latin-1") * 5]
        for st in stems:
            for j in rng.choice(len(sfxs), 12 if lang == 2 else 10, replace=False):
                w = st + lat(sfxs[int(j)])
                words.append(w)
                if any(c >= 0xC0 for c in w):
                    words.append(utf(w))
        changed = 0
        for w in words:
            if not w or 0 in w:
                continue
            out = []
            for fn, (let, se, tl, h1, h2) in ((L.refp8_stem_word, bufs[0]), (lib.orc_p8_stem_word, bufs[1])):
                r = fn(lang, w, let.ctypes.data, se.ctypes.data, tl.ctypes.data, h1.ctypes.data, h2.ctypes.data)
                out.append((r, let.tobytes(), tuple(se), int(tl[0]), int(tl[1]) if r else -1, tuple(h1), tuple(h2)))
            assert out[0] == out[1], (lang, w, out[0][:5], out[1][:5])
            changed += out[0][0]
        assert changed > len(words) // 4, (lang, changed, len(words))


@needs_ref
def test_word_model_vs_reference():
    """wordModel: enwik-like text plus hyphenated line breaks ("+\\n", "-\\r\\n"), numbers with decimal points, wiki
    markup ("==", "''", "[[..]]"), upper-case words, high bytes, long lines. 57 contexts and the nine word-level
    globals other models read."""
    from cmix_amd import synth
    L, lib = R.paq8core_lib(), O.lib()
    extra = (b"The quick-\nbrown fox+\njumps over-\r\nthe lazy+\r\ndog. 3.1415 and 12.5% of 1,000 items; == Heading ==\n''italic'' [[link]] "
             b"[[ spaced ]] : definition = value\n\nUPPER lower MiXed don't o'clock well-known \xc3\xa9t\xc3\xa9 na\xc3\xafve\n" + b"x" * 300 + b"\n") * 4
    data = np.frombuffer(synth.enwik_like(6000, 89) + extra + synth.enwik_like(2000, 97), np.uint8)
    LOG, level = 16, 2
    L.refp8_rnd_reset()
    lib.orc_p8_rnd_reset()
    L.refp8_buf_reset(LOG)
    L.refp8_word_globals_reset()  # process-wide; test_sparse_models_vs_reference leaves its inputs in them
    ring = np.zeros(1 << LOG, np.uint8)
    got = lib.orc_p8_word_new(level)
    o_ref, o_got = np.zeros(512, np.int16), np.zeros(512, np.int16)
    g_ref, g_got = np.zeros(9, np.uint32), np.zeros(9, np.uint32)
    y, c0, c4, f4, b2, b3 = 0, 1, 0, 0, 0, 0
    for n in range(len(data)):
        for bpos in range(8):
            k = L.refp8_word_step(level, y, bpos, c0, c4, f4, b3, n, o_ref.ctypes.data, g_ref.ctypes.data)
            g = lib.orc_p8_word_step(got, y, bpos, c0, c4, f4, b3, n, ring.ctypes.data, (1 << LOG) - 1, n, o_got.ctypes.data, g_got.ctypes.data)
            assert k == g == 285, (n, bpos, k, g)  # 57 of the 61 slots are set
            assert (g_ref == g_got).all(), (n, bpos, g_ref, g_got)
            assert (o_ref[:k] == o_got[:k]).all(), (n, bpos, np.nonzero(o_ref[:k] != o_got[:k])[0][:6] // 5)
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        b = int(data[n])
        L.refp8_buf_push(b)
        ring[n] = b
        c4 = ((c4 << 8) | b) & 0xffffffff
        f4 = ((f4 << 4) | (b >> 4)) & 0xffffffff
        b3, b2 = b2, b


def _text_corpus():
    from cmix_amd import synth
    fr = ("Les enfants \xc3\xa9taient heureusement arriv\xc3\xa9s; ils parlaient doucement, finissaient leurs travaux et regardaient les "
          "monuments. M. Dupont (le directeur) disait: \xab la nation fran\xc3\xa7aise est une grande nation \xbb ! Que faites-vous? "
          "Nous chanterons demain, et vous danserez apr\xc3\xa8s. ").encode("latin-1")
    de = ("Die Kinder spielten fr\xc3\xb6hlich auf der Stra\xc3\x9fe und sangen sch\xc3\xb6ne Lieder. Hr. M\xc3\xbcller sagte: "
          "die Freundlichkeit der Menschen ist nicht selbstverst\xc3\xa4ndlich! Wir haben sich mit den Nachbarn getroffen, "
          "und das Wetter war herrlich. ").encode("latin-1")
    en = (b"Mr. Smith said, \"the running dogs were happily jumping over 1,234 fences\"; Dr. Jones disagreed! Why? Because 10, 11, 12, 13 "
          b"and 100 - 99 = 1 {see [note (3)]} <tag> 'quoted words' aren't it's. St. Paul's; U.S.A. e.g. the+\nbroken+\r\nword\n\nNew "
          b"paragraph: Topic: details follow\twith\ttabs | and \\ slashes @ & ^ _ 50% 3*4/2=6\r\n\r\n")
    return np.frombuffer(synth.enwik_like(5000, 41) + en * 3 + fr * 6 + en + de * 6 + synth.enwik_like(2500, 43) + bytes(range(256)) * 2 + fr + en,
                         np.uint8)


@needs_ref
def test_text_model_vs_reference():
    """TextModel: enwik-like text, English with abbreviations / quotes / number sequences / nesting / "+\\n" hyphenation,
    French and German paragraphs long enough to flip the detected language (UTF-8 accents, guillemets, sharp s), all 256
    byte values. The 33-context map's 231 inputs, the eight mixer selectors and ModelStats::Text, bit for bit."""
    L, lib = R.paq8core_lib(), O.lib()
    data = _text_corpus()
    LOG, level = 16, 2
    L.refp8_buf_reset(LOG)
    L.refp8_text_new.restype = C.c_void_p
    lib.orc_p8_text_new.restype = C.c_void_p
    L.refp8_text_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_p8_text_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    size = (0x10000 << level) * 16
    ref, got = L.refp8_text_new(size), lib.orc_p8_text_new(size)
    ring = np.zeros(1 << LOG, np.uint8)
    o_ref, o_got = np.zeros(512, np.int16), np.zeros(512, np.int16)
    s_ref, s_got = np.zeros(64, np.int32), np.zeros(8, np.int32)
    t_ref, t_got = np.zeros(6, np.uint32), np.zeros(6, np.uint32)
    k_ref = C.c_int(0)
    y, c0 = 0, 1
    states = set()
    bases = np.concatenate([[0], np.cumsum([2048, 2048, 4096, 4096, 2048, 2048, 4096])]).astype(np.int32)  # the recording mixer stores base + selector
    for n in range(len(data)):
        for bpos in range(8):
            k = L.refp8_text_step(ref, y, bpos, c0, o_ref.ctypes.data, s_ref.ctypes.data, C.byref(k_ref), t_ref.ctypes.data)
            g = lib.orc_p8_text_step(got, y, bpos, c0, ring.ctypes.data, (1 << LOG) - 1, n, o_got.ctypes.data, s_got.ctypes.data, t_got.ctypes.data)
            assert k == g == 231 and k_ref.value == 8, (n, bpos, k, g, k_ref.value)
            if bpos == 0:
                assert (t_ref == t_got).all(), (n, bytes(data[max(0, n - 30):n]), t_ref, t_got)
                states.add(int(t_ref[0]))
            assert (s_ref[:8] - bases == s_got).all(), (n, bpos, bytes(data[max(0, n - 30):n]), s_ref[:8] - bases, s_got)
            assert (o_ref[:k] == o_got[:k]).all(), (n, bpos, bytes(data[max(0, n - 30):n]), np.nonzero(o_ref[:k] != o_got[:k])[0][:6] // 5)
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        b = int(data[n])
        L.refp8_buf_push(b)
        ring[n] = b
    assert states == set(range(8)), states


def _private_ref_copy(tmp_path):
    """contextModel2 keeps its sub-models in function-local statics: the whole-predictor runs use their own loaded
    copy of the reference library so that the single-model tests of this file do not share state with them."""
    import shutil
    dst = tmp_path / "libcmixrefpaq8_private.so"
    shutil.copy(R.PAQ8_LIB_PATH, dst)
    L = C.CDLL(str(dst))
    _PRIVATE_COPIES.append(L)
    L.refp8_predictor_new.restype = C.c_void_p
    L.refp8_predictor_new.argtypes = [C.c_int]
    L.refp8_predictor_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    return L


def _block(ftype, payload, info=None):
    """A block as cmix's preprocessor frames it for paq8 (contextModel2 :8117-8134): type, big-endian size[, info]."""
    hdr = bytes([ftype]) + len(payload).to_bytes(4, "big")
    if info is not None:
        hdr += info.to_bytes(4, "big")
    return hdr + payload


def _predictor_streams():
    from cmix_amd import synth
    rng = np.random.default_rng(311)
    ops = [b"\x55\x8b\xec", b"\x83\xec\x10", b"\x8b\x45\x08", b"\xe8\x10\x00\x00\x00", b"\x0f\x84\x20\x01\x00\x00", b"\x48\x8b\x05\x10\x20\x00\x00",
           b"\xc3", b"\x90", b"\xff\x15\x00\x10\x40\x00", b"\xeb\xfe", b"\x89\x44\x24\x04"]
    exe = b"".join(ops[int(k)] for k in rng.integers(0, len(ops), 500))
    recs = b"".join(int(i).to_bytes(4, "little") + bytes([i % 7, 0, 0, 0]) + b"name%04d" % (i % 50) + b"\x00\x00\x28\x00" for i in range(150))
    text = bytes(_text_corpus()[:9000])
    framed = (_block(4, text[:5000], 0) + _block(0, recs + bytes(rng.integers(0, 256, 600, dtype=np.uint8))) + _block(3, exe[:1500]) +
              _block(4, text[5000:8000], 0) + _block(1, b"hdr-ish \x00\x01\x02" * 20))
    raw = synth.enwik_like(3000, 71) + recs[:1200] + bytes(range(256))
    return {"framed": framed, "raw": raw}


@needs_ref
@pytest.mark.parametrize("name", ["framed", "raw"])
def test_whole_paq8_predictor_vs_reference(name, tmp_path):
    """The assembled oracle (oracle/paq8_predictor.c) against the reference's own paq8::Predictor, all 1591 values
    PAQ8::Predict() hands to cmix after every bit plus the final probability: a stream framed the way cmix's
    preprocessor frames blocks (TEXT with English / French / German, DEFAULT with records and noise, EXE, HDR) and a
    raw stream whose first bytes get read as a block header. Floats are int * (1/4095.f): compared bit for bit."""
    L, lib = _private_ref_copy(tmp_path), O.lib()
    lib.orc_p8_predictor_new.restype = C.c_void_p
    lib.orc_p8_predictor_new.argtypes = [C.c_int]
    lib.orc_p8_predictor_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    level = 2
    data = _predictor_streams()[name]
    lib.orc_p8_rnd_reset()
    ref, got = L.refp8_predictor_new(level), lib.orc_p8_predictor_new(level)
    o_ref, o_got = np.zeros(1591, np.float32), np.zeros(1591, np.float32)
    for n, byte in enumerate(data):
        for bpos in range(8):
            y = (byte >> (7 - bpos)) & 1
            pr = L.refp8_predictor_update(ref, y, o_ref.ctypes.data)
            pg = lib.orc_p8_predictor_update(got, y, o_got.ctypes.data)
            assert pg >= 0, (n, bpos, pg)
            bad = np.nonzero(o_ref.view(np.uint32) != o_got.view(np.uint32))[0]
            assert bad.size == 0, (name, n, bpos, bad[:8], o_ref[bad[:4]] * 4095, o_got[bad[:4]] * 4095)
            assert pr == pg, (n, bpos, pr, pg)


@needs_ref
@pytest.mark.parametrize("level, log2size", [(2, 21), (4, 23)])
def test_whole_paq8_predictor_vs_reference_across_the_history_rings_end(level, log2size, tmp_path):
    """paq8's byte position `pos` (paq8.cpp:167) indexes a ring of MEM() * 8 bytes (:169-186, :8368: 2^30 at cmix's level 11, 2^21 / 2^23 at the levels
    run here) and is stored / compared unmasked by the match models and the detectors. State injection (round 6's wrap / threshold audit): the reference's
    own paq8::Predictor and the oracle both start 2500 bytes below the ring's size and run across it -- all 1591 values after every bit. (At level 11 the
    same is pinned by the per-step hashes of tests/golden/paq8_cols_pos_1g_6k.npz on the stage's front end and on the device.)"""
    from cmix_amd import synth
    L, lib = _private_ref_copy(tmp_path), O.lib()
    lib.orc_p8_predictor_new.restype = C.c_void_p
    lib.orc_p8_predictor_new.argtypes = [C.c_int]
    lib.orc_p8_predictor_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_p8_predictor_set_pos.argtypes = [C.c_void_p, C.c_int]
    L.refp8_set_pos.argtypes = [C.c_int]
    data = (synth.enwik_like(2200, 88, rich=True) * 2)[:4000]
    lib.orc_p8_rnd_reset()
    ref, got = L.refp8_predictor_new(level), lib.orc_p8_predictor_new(level)
    pos0 = (1 << log2size) - 2500
    L.refp8_set_pos(pos0)
    lib.orc_p8_predictor_set_pos(got, pos0)
    o_ref, o_got = np.zeros(1591, np.float32), np.zeros(1591, np.float32)
    for n, byte in enumerate(data):
        for bpos in range(8):
            y = (byte >> (7 - bpos)) & 1
            pr = L.refp8_predictor_update(ref, y, o_ref.ctypes.data)
            pg = lib.orc_p8_predictor_update(got, y, o_got.ctypes.data)
            assert pg >= 0, (n, bpos, pg)
            bad = np.nonzero(o_ref.view(np.uint32) != o_got.view(np.uint32))[0]
            assert bad.size == 0 and pr == pg, (level, n, bpos, bad[:8], pr, pg)


def test_paq8_predictor_refuses_streams_it_does_not_model():
    """JPEG / BMP / WAV payloads and image-typed blocks switch on sub-models the oracle does not restate: it must
    return an error code, never a number."""
    lib = O.lib()
    lib.orc_p8_predictor_new.restype = C.c_void_p
    lib.orc_p8_predictor_new.argtypes = [C.c_int]
    lib.orc_p8_predictor_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    bmp = b"\x28\x00\x00\x00" + (16).to_bytes(4, "little") + (16).to_bytes(4, "little") + b"\x01\x00\x18\x00" + bytes(24)
    wav = (b"RIFF" + (1000).to_bytes(4, "little") + b"WAVEfmt " + (16).to_bytes(4, "little") + b"\x01\x00\x02\x00" + (44100).to_bytes(4, "little") +
           (176400).to_bytes(4, "little") + b"\x04\x00\x10\x00" + b"data" + (800).to_bytes(4, "little") + bytes(16))
    cases = {-2: _block(0, b"abc\xff\xd8\xff\xe0 more"), -3: _block(0, b"x" * 10 + bmp + b"y" * 20), -5: _block(0, b"zz" + wav),
             -1: _block(7, bytes(64), 8)}
    for code, stream in cases.items():
        lib.orc_p8_rnd_reset()
        h = lib.orc_p8_predictor_new(0)
        seen = None
        for byte in stream:
            for bpos in range(8):
                r = lib.orc_p8_predictor_update(h, (byte >> (7 - bpos)) & 1, None)
                if r < 0:
                    seen = r
                    break
            if seen is not None:
                break
        assert seen == code, (code, seen)
