"""CPU: the restated numeric building blocks of the vendored fxcm model (oracle/fxcm_core.c; SURVEY.md 8a') against the
reference's own structs compiled from fxcmv1.cpp (oracle/ref_fxcmcore.cpp -> oracle/_ref/libcmixreffxcm.so). Integer
work: bit-exact."""
import ctypes as C

import numpy as np
import pytest

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

needs_ref = pytest.mark.skipif(not R.fxcmcore_available(), reason="oracle/_ref/libcmixreffxcm.so not built")
P = C.c_void_p


def _libs():
    return R.fxcmcore_lib(), O.lib()


def _bits(rng, n, p1=0.5):
    return (rng.random(n) < p1).astype(np.int32)


@needs_ref
def test_tables_match_the_reference():
    """squash / stretch (float exp / log + round in the reference's constructor), ilog, dt, the six generated state
    tables and the tables derived from them: the committed numbers (oracle/fxcm_tables.h) and the oracle's formulas
    against what the reference computes at start-up."""
    L, lib = _libs()
    outs = []
    for fn in (L.reffx_tables, lib.orc_fx_tables):
        a = [np.zeros(4095, np.int16), np.zeros(4096, np.int16), np.zeros(256, np.uint8), np.zeros(1024, np.int32), np.zeros(6144, np.uint8),
             np.zeros(256, np.int16), np.zeros(4096, np.int16), np.zeros(4096, np.int16)]
        fn.argtypes = [P] * 8
        fn(*[x.ctypes.data for x in a])
        outs.append(a)
    for x, y in zip(*outs):
        assert (x == y).all()
    assert outs[0][4].reshape(6, 256, 4)[:, :, :2].max() > 200          # real state tables, not zeros
    assert [lib.orc_fx_squash(d) for d in (-3000, -2047, 0, 2047, 3000)] == [1, int(outs[0][0][0]), int(outs[0][0][2047]), int(outs[0][0][4094]), 4095]


@needs_ref
@pytest.mark.parametrize("n,m,shift,elim0,uperr,use_p1", [(512, 64, 14, 0, 24, 1), (512, 2048, 40, 27, 28, 1), (16, 8, 16, 0, 14, 0), (16, 256, 80, 3, 20, 0)])
def test_mixer1_vs_reference(n, m, shift, elim0, uperr, use_p1):
    """Mixer1: stretch-domain inputs with saturating outliers, contexts with locality, bits correlated with the
    prediction, the dead zone `elim` changing over time (the model adapts it per byte)."""
    L, lib = _libs()
    L.reffx_mixer_new.restype = P
    lib.orc_fx_mixer_new.restype = P
    sig = [P, C.c_int, P, C.c_int, C.c_int, C.c_int, P]
    L.reffx_mixer_step.argtypes = sig
    lib.orc_fx_mixer_step.argtypes = sig
    rng = np.random.default_rng(n * 31 + m)
    ref, got = L.reffx_mixer_new(n, m, shift, elim0, uperr), lib.orc_fx_mixer_new(n, m, shift, elim0, uperr)
    steps = 3000
    xs = np.clip(rng.normal(0, 500, (steps, n)), -2047, 2047).astype(np.int16)
    xs[rng.random((steps, n)) < 0.01] = 32767
    xs[rng.random((steps, n)) < 0.01] = -32768
    xs[:, n // 2:] *= (rng.random((steps, n - n // 2)) < 0.3)
    cur, y = 0, 0
    pr_r, pr_g = C.c_int(0), C.c_int(0)
    for t in range(steps):
        if rng.random() < 0.3:
            cur = int(rng.integers(0, m))
        elim = elim0 + (t // 500) % 3 * 9
        a = L.reffx_mixer_step(ref, y, xs[t].ctypes.data, cur, elim, use_p1, C.byref(pr_r))
        b = lib.orc_fx_mixer_step(got, y, xs[t].ctypes.data, cur, elim, use_p1, C.byref(pr_g))
        assert (a, pr_r.value) == (b, pr_g.value), (t, a, b, pr_r.value, pr_g.value)
        y = int(rng.random() < pr_r.value / 4096.0)


@needs_ref
def test_statemaps_and_apm_vs_reference():
    L, lib = _libs()
    rng = np.random.default_rng(5)
    for name in ("statemap_new", "statemap1_new", "apm_new"):
        getattr(L, "reffx_" + name).restype = P
        getattr(lib, "orc_fx_" + name).restype = P
    for fn in (L.reffx_statemap_set, lib.orc_fx_statemap_set, L.reffx_statemap1_set, lib.orc_fx_statemap1_set):
        fn.argtypes = [P, C.c_int, C.c_int]
    for fn in (L.reffx_apm_p, lib.orc_fx_apm_p):
        fn.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_int]
    for which in range(6):                      # StateMap over each state table
        ref, got = L.reffx_statemap_new(256, which), lib.orc_fx_statemap_new(256, which)
        cx = rng.integers(0, 256, 20000) * (rng.random(20000) < 0.7)
        for t, y in enumerate(_bits(rng, 20000, 0.3 + 0.1 * which)):
            assert L.reffx_statemap_set(ref, int(y), int(cx[t])) == lib.orc_fx_statemap_set(got, int(y), int(cx[t])), (which, t)
    for n, limit in ((1 << 16, 1023), (1 << 10, 127), (1 << 8, 1)):   # StateMap1: few contexts so that counts reach the limit
        ref, got = L.reffx_statemap1_new(n, limit), lib.orc_fx_statemap1_new(n, limit)
        cx = rng.integers(0, 1 << 20, 40000) % rng.choice([7, 64, n * 4])
        for t, y in enumerate(_bits(rng, 40000, 0.9)):
            assert L.reffx_statemap1_set(ref, int(y), int(cx[t])) == lib.orc_fx_statemap1_set(got, int(y), int(cx[t])), (n, limit, t)
    ref, got = L.reffx_apm_new(), lib.orc_fx_apm_new(1024)
    for t in range(30000):
        pr, cx, rate, y = int(rng.integers(0, 4096)), int(rng.integers(0, 1024) % (1 + t % 37)), int(rng.integers(5, 9)), int(rng.random() < 0.6)
        assert L.reffx_apm_p(ref, pr, cx, rate, y) == lib.orc_fx_apm_p(got, pr, cx, rate, y), t


@needs_ref
def test_run_map_sscm_and_direct_state_map_vs_reference():
    L, lib = _libs()
    rng = np.random.default_rng(17)
    for name in ("rcm_new", "sscm_new", "dsm_new"):
        getattr(L, "reffx_" + name).restype = P
        getattr(lib, "orc_fx_" + name).restype = P
    # RunContextMap: a small table (constant replacement in the 4-way sets), bytes with runs
    for fn in (L.reffx_rcm_set, lib.orc_fx_rcm_set):
        fn.argtypes = [P, C.c_uint32, C.c_int]
    for fn in (L.reffx_rcm_mix, lib.orc_fx_rcm_mix):
        fn.argtypes = [P, C.c_int, C.c_int, C.c_int, P]
    for m, ml in ((1 << 10, 8), (1 << 16, 5)):
        ref, got = L.reffx_rcm_new(m, ml), lib.orc_fx_rcm_new(m, ml)
        data = (rng.integers(0, 4, 6000) * 40 + 30).astype(np.uint8)
        data[1000:1400] = 65
        o_r, o_g = np.zeros(4, np.int16), np.zeros(4, np.int16)
        h, y, c0 = 0, 0, 1
        for n, b in enumerate(data):
            for bpos in range(8):
                a = L.reffx_rcm_mix(ref, y, bpos, c0, o_r.ctypes.data)
                g = lib.orc_fx_rcm_mix(got, y, bpos, c0, o_g.ctypes.data)
                assert (a, int(o_r[0])) == (g, int(o_g[0])), (m, n, bpos)
                y = (int(b) >> (7 - bpos)) & 1
                c0 = (c0 << 1 | y) if bpos < 7 else 1
            h = (h * 773 + int(b) + 1) & 0xffffffff if n % 5 else int(b)
            L.reffx_rcm_set(ref, h, int(b))
            lib.orc_fx_rcm_set(got, h, int(b))
    # SmallStationaryContextMap: 8 and 1 input bits, the exported slot of the second input is given back
    for fn in (L.reffx_sscm_set, lib.orc_fx_sscm_set):
        fn.argtypes = [P, C.c_uint32]
    for fn in (L.reffx_sscm_mix, lib.orc_fx_sscm_mix):
        fn.argtypes = [P, C.c_int, C.c_int, P, P, P]
    for boc, ib in ((11, 8), (8, 1), (16, 8)):
        ref, got = L.reffx_sscm_new(boc, ib), lib.orc_fx_sscm_new(boc, ib)
        o_r, o_g, e_r, e_g = np.zeros(4, np.int16), np.zeros(4, np.int16), np.zeros(4, np.float32), np.zeros(4, np.float32)
        k_r, k_g = C.c_int(0), C.c_int(0)
        for t, y in enumerate(_bits(rng, 24000, 0.4)):
            if t % ib == 0:
                cx = int(rng.integers(0, 1 << 20) % 300)
                L.reffx_sscm_set(ref, cx)
                lib.orc_fx_sscm_set(got, cx)
            rate = t // 8000
            a = L.reffx_sscm_mix(ref, int(y), rate, o_r.ctypes.data, e_r.ctypes.data, C.byref(k_r))
            g = lib.orc_fx_sscm_mix(got, int(y), rate, o_g.ctypes.data, e_g.ctypes.data, C.byref(k_g))
            assert a == g == 2 and k_r.value == k_g.value == 1 and (o_r[:2] == o_g[:2]).all() and e_r[0] == e_g[0], (boc, ib, t)
    # DirectStateMap: 3 contexts per bit over a 2^m table of states
    for fn in (L.reffx_dsm_step, lib.orc_fx_dsm_step):
        fn.argtypes = [P, C.c_int, P, C.c_int, P, P, P]
    for which in (5, 0):
        ref, got = L.reffx_dsm_new(12, 3, which), lib.orc_fx_dsm_new(12, 3, which)
        o_r, o_g, e_r, e_g = np.zeros(8, np.int16), np.zeros(8, np.int16), np.zeros(8, np.float32), np.zeros(8, np.float32)
        k_r, k_g = C.c_int(0), C.c_int(0)
        for t, y in enumerate(_bits(rng, 20000, 0.35)):
            cx = (rng.integers(0, 1 << 16, 3) % np.array([50, 700, 5000])).astype(np.uint32)
            a = L.reffx_dsm_step(ref, int(y), cx.ctypes.data, 3, o_r.ctypes.data, e_r.ctypes.data, C.byref(k_r))
            g = lib.orc_fx_dsm_step(got, int(y), cx.ctypes.data, 3, o_g.ctypes.data, e_g.ctypes.data, C.byref(k_g))
            assert a == g == 6 and k_r.value == k_g.value == 0 and (o_r[:6] == o_g[:6]).all(), (which, t, o_r[:6], o_g[:6])


@needs_ref
@pytest.mark.parametrize("kind,m,ncx,cr,cs,s3,sta,s4,keep,u,st2", [
    (0, 16 * 4096, 7, 4, 34, 35, 1, 8, 0, 1, 1),          # cmC[0]: 7 contexts, STA2, st2_p1
    (0, 2 * 4096, 2, 6, 36, 30, 1, 13, 0xf0, 0, 0),       # cmC[2]: tiny table (constant replacement), keep flag, no st2 input
    (0, 32 * 4096, 2, 4, 33, 32, 1, 12, 0, 1, 2),         # cmC[3]: st2_p2
    (1, 32 * 4096, 2, 3, 35, 35, 4, 12, 0, 0, 0),         # cmC1[0]: 32-byte buckets of 3
    (1, 16 * 4096, 5, 3, 33, 31, 5, 7, 0, 1, 1),          # cmC1[4]: 5 contexts, STA7
    (2, 64 * 4096, 3, 3, 28, 43, 4, 9, 0xf0, 1, 1),       # cmC2[0] shape at a small size: 128-byte buckets of 14
    (2, 8 * 4096, 6, 6, 31, 29, 4, 12, 0xf0, 1, 1),       # cmC2[5] shape, heavy replacement
    (2, 16 * 4096, 1, 2, 33, 32, 0, 15, 0, 1, 1),         # cmC2[6]: a single context (the `1<C` initial mask quirk), STA1
])
def test_context_maps_vs_reference(kind, m, ncx, cr, cs, s3, sta, s4, keep, u, st2):
    """ContextMap / ContextMap1 / ContextMap2 with the parameter sets the model uses, at table sizes small enough that
    buckets fill and get replaced: order-1..n byte-history contexts over text with long repeats (run model, deferred
    creation of the bits 2-7 histories on the second visit), some slots skipped the way the model skips them. Every
    input, every exported probability and the confidence count, per bit."""
    from cmix_amd import synth
    L, lib = _libs()
    L.reffx_cm_new.restype = P
    lib.orc_fx_cm_new.restype = P
    new_sig = [C.c_int, C.c_uint32] + [C.c_int] * 7
    L.reffx_cm_new.argtypes = new_sig
    lib.orc_fx_cm_new.argtypes = new_sig
    L.reffx_cm_step.argtypes = [C.c_int, P, C.c_int, C.c_int, C.c_int, C.c_uint32, P, P, C.c_int, P, P, P, P]
    lib.orc_fx_cm_step.argtypes = [P, C.c_int, C.c_int, C.c_int, C.c_uint32, P, P, C.c_int, P, P, P, P]
    c = ncx | (cr << 8) | (cs << 16)
    ref, got = L.reffx_cm_new(kind, m, c, s3, sta, s4, keep, u, st2), lib.orc_fx_cm_new(kind, m, c, s3, sta, s4, keep, u, st2)
    rng = np.random.default_rng(kind * 100 + ncx)
    text = synth.enwik_like(3000, 7 + kind)
    data = np.frombuffer(text[:1500] + text[200:900] + bytes(rng.integers(0, 256, 500, dtype=np.uint8)) + text[:1200] + b"a" * 300 + text[1500:], np.uint8)
    o_r, o_g, e_r, e_g = np.zeros(64, np.int16), np.zeros(64, np.int16), np.zeros(64, np.float32), np.zeros(64, np.float32)
    k_r, k_g, n_r, n_g = C.c_int(0), C.c_int(0), C.c_int(0), C.c_int(0)
    cx, skip = np.zeros(8, np.uint32), np.zeros(8, np.uint8)
    y, c0, c4, hist = 0, 1, 0, [0] * 8
    per = 6 if u == 1 else 5
    for n in range(len(data)):
        for bpos in range(8):
            if bpos == 0:
                for i in range(ncx):
                    h = 0
                    for j in range(i + 1):
                        h = (h * 0x2F0B4A47 + hist[j] + 1) & 0xffffffff
                    cx[i] = h
                    skip[i] = 1 if (i >= 2 and (n // 97 + i) % 5 == 0) else 0
            a = L.reffx_cm_step(kind, ref, y, bpos, c0, c4, cx.ctypes.data, skip.ctypes.data, ncx, o_r.ctypes.data, e_r.ctypes.data, C.byref(k_r), C.byref(n_r))
            g = lib.orc_fx_cm_step(got, y, bpos, c0, c4, cx.ctypes.data, skip.ctypes.data, ncx, o_g.ctypes.data, e_g.ctypes.data, C.byref(k_g), C.byref(n_g))
            cn = ncx if n > 0 or bpos == 0 else ncx
            assert a == g and n_r.value == n_g.value and k_r.value == k_g.value, (n, bpos, a, g, n_r.value, n_g.value, k_r.value, k_g.value)
            if n > 0:
                assert n_r.value == per * cn and k_r.value == (per - 1) * cn, (n, bpos, n_r.value, k_r.value)
            assert (o_r[:n_r.value] == o_g[:n_r.value]).all(), (n, bpos, o_r[:n_r.value], o_g[:n_r.value])
            assert (e_r[:k_r.value].view(np.uint32) == e_g[:k_r.value].view(np.uint32)).all(), (n, bpos)
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        b = int(data[n])
        c4 = ((c4 << 8) | b) & 0xffffffff
        hist = [b] + hist[:7]


@needs_ref
def test_sparse_match_model_vs_reference():
    """fxcm's SparseMatchModel: text with long repeats, repeats with every other byte changed (the stride-2 finder),
    random bytes. Match length, the finder that won, the match position, the expected byte and both inputs."""
    from cmix_amd import synth
    L, lib = _libs()
    L.reffx_sparsematch_new.restype = P
    lib.orc_fx_sparsematch_new.restype = P
    L.reffx_sparsematch_p.argtypes = [P, C.c_int, C.c_int, C.c_int, P, P]
    lib.orc_fx_sparsematch_p.argtypes = [P, C.c_int, C.c_int, P, C.c_uint32, C.c_int, P, P]
    rng = np.random.default_rng(23)
    text = synth.enwik_like(4000, 29)
    alt = bytearray(text[300:1300])
    alt[1::2] = bytes(rng.integers(97, 123, len(alt[1::2]), dtype=np.uint8))
    data = np.frombuffer(text[:2000] + text[300:1300] + bytes(alt) + bytes(rng.integers(0, 256, 400, dtype=np.uint8)) + text[100:1500] + bytes(alt[:600]), np.uint8)
    LOG = 16
    ring = np.zeros(1 << LOG, np.uint8)
    L.reffx_buf_reset()
    ref, got = L.reffx_sparsematch_new(), lib.orc_fx_sparsematch_new()
    o_r, o_g, s_r, s_g = np.zeros(4, np.int16), np.zeros(4, np.int16), np.zeros(4, np.int32), np.zeros(4, np.int32)
    y, c0 = 0, 1
    winners, longest = set(), 0
    for n in range(len(data)):
        for bpos in range(8):
            a = L.reffx_sparsematch_p(ref, y, bpos, c0, o_r.ctypes.data, s_r.ctypes.data)
            g = lib.orc_fx_sparsematch_p(got, bpos, c0, ring.ctypes.data, (1 << LOG) - 1, n, o_g.ctypes.data, s_g.ctypes.data)
            assert a == g and (s_r == s_g).all() and (o_r[:2] == o_g[:2]).all(), (n, bpos, a, g, s_r, s_g, o_r[:2], o_g[:2])
            if a:
                winners.add(int(s_r[0]))
                longest = max(longest, a)
            y = (int(data[n]) >> (7 - bpos)) & 1
            c0 = (c0 << 1 | y) if bpos < 7 else 1
        L.reffx_buf_push(int(data[n]))
        ring[n] = data[n]
    assert longest == 64 and winners >= {0, 2}, (longest, winners)   # the stride-1 and the stride-2 finder both win


@needs_ref
def test_fxcm_english_stemmer_vs_reference():
    """fxcm's EnglishStemmer: the synthetic corpus' vocabulary crossed with inflection suffixes, prefixes (incl. the
    "anti-" / "dis-" forms this variant adds), apostrophes on both ends, trailing hyphens, the closed word classes,
    both sides of the block-position switch of the auxiliary-verb list. Stem letters, Start / End, the stem hash and
    the three flag words."""
    from cmix_amd import synth
    L, lib = _libs()
    sig = [C.c_char_p, C.c_int, P, P, P]
    L.reffx_stem_word.argtypes = sig
    lib.orc_fx_stem_word.argtypes = sig
    text = synth.enwik_like(400000, 3)
    base = sorted({w.lower() for w in text.replace(b"\n", b" ").split(b" ") if w.isalpha() and len(w) < 40})[:5000]
    extra = [b"skis", b"skies", b"dying", b"idly", b"news", b"atlas", b"texas", b"inning", b"proceed", b"zinc", b"here", b"he", b"she", b"the", b"an",
             b"can't", b"won't", b"ain't", b"isn't", b"o'clock", b"'tis", b"''quoted''", b"'word'", b"non-linear", b"nonsense", b"overestimate",
             b"underground", b"irregular", b"unnatural", b"anti-hero", b"antimatter", b"dis-agree", b"disagree", b"biggest", b"suggest", b"fullest",
             b"happiest", b"smallest", b"childhood", b"neighbourhood", b"quizzing", b"generously", b"communal", b"y", b"yy", b"a", b"by", b"say",
             b"yellowy", b"well-", b"co-", b"because", b"between", b"also", b"thus", b"would", b"been", b"seven", b"million", b"x" * 70]
    suffixes = [b"", b"s", b"es", b"ed", b"ing", b"ly", b"ness", b"est", b"'s", b"ation", b"ational", b"fully", b"less", b"ize", b"ied", b"ies",
                b"edly", b"ingly", b"ative", b"ement", b"n't", b"-"]
    words = extra + [w + sfx for w in base for sfx in suffixes[: 1 + (len(w) % 7) * 3]] + [b"anti-" + w for w in base[:300]] + [b"'" + w + b"'" for w in base[:300]]
    bufs = [(np.zeros(64, np.uint8), np.zeros(2, np.int32), np.zeros(4, np.uint32)) for _ in range(2)]
    changed = 0
    for k, w in enumerate(words):
        blpos = 451531986 + 5 if k % 11 == 0 else 1000
        out = []
        for fn, (let, se, h) in ((L.reffx_stem_word, bufs[0]), (lib.orc_fx_stem_word, bufs[1])):
            r = fn(w, blpos, let.ctypes.data, se.ctypes.data, h.ctypes.data)
            out.append((r, let.tobytes(), tuple(se), tuple(h)))
        assert out[0] == out[1], (w, out[0][0], out[0][2:], out[1][0], out[1][2:])
        changed += out[0][0]
    assert changed > len(words) // 4


def _private_fx_copy(tmp_path):
    """All of fxcm's state is in namespace-level globals: the whole-model runs use their own loaded copy of the
    reference library so that the single-block tests of this file do not share state with them."""
    import shutil
    dst = tmp_path / "libcmixreffxcm_private.so"
    shutil.copy(R.FXCM_LIB_PATH, dst)
    L = C.CDLL(str(dst))
    _PRIVATE_COPIES.append(L)
    L.reffx_model_new.restype = P
    L.reffx_model_update.argtypes = [P, C.c_int, C.c_int, C.c_int, P]
    L.reffx_model_debug.argtypes = [P]
    L.reffx_model_contexts.argtypes = [P]
    return L


@needs_ref
@pytest.mark.parametrize("name", ["wiki", "mixed", "binary"])
def test_whole_fxcm_model_vs_reference(name, tmp_path):
    """The assembled oracle (oracle/fxcm_model.c) against the reference's own fxcmv1::Predictor after every bit: all 431
    values FXCM::Predict() hands to cmix, the model's final probability, the ten mixer selectors and a set of parser
    registers, and the 256 context-slot hashes the 32 maps hold. Streams: wiki-like text (markup, links, tables,
    numbers), prose in three languages with abbreviations / quotes / nesting / all 256 byte values, and binary records.
    The LSTM hints the model reads before each update are seeded random numbers (same on both sides)."""
    from cmix_amd import synth
    import test_oracle_paq8core as T
    L, lib = _private_fx_copy(tmp_path), O.lib()
    lib.orc_fx_model_new.restype = P
    lib.orc_fx_model_update.argtypes = [P, C.c_int, C.c_int, C.c_int, P]
    lib.orc_fx_model_debug.argtypes = [P, P]
    lib.orc_fx_model_contexts.argtypes = [P, P]
    rng = np.random.default_rng(77)
    wiki = (b"{{Infobox|name=Test|value=12}}\n{|\n|-\n| cell one || cell two\n|-\n| 3.14 || [[link|text]]\n|}\n* item one\n* item [[two]], three\n"
            b"== Heading ==\n'''Bold''' and ''italic'' text. See [http://example.org/page link] &amp; more; x &lt; y.\n\n")
    data = {"wiki": synth.enwik_like(2500, 51) + wiki * 3 + synth.enwik_like(800, 52),
            "mixed": bytes(T._text_corpus()[:3500]),
            "binary": b"".join(int(i).to_bytes(4, "little") + bytes([i % 7, 0, 255, 12]) + b"rec%03d" % (i % 40) for i in range(150)) +
                      bytes(rng.integers(0, 256, 800, dtype=np.uint8))}[name]
    ref, got = L.reffx_model_new(), lib.orc_fx_model_new()
    a, b = np.zeros(431, np.float32), np.zeros(431, np.float32)
    da, db, ca, cb = np.zeros(48, np.uint32), np.zeros(48, np.uint32), np.zeros(256, np.uint32), np.zeros(256, np.uint32)
    for n, byte in enumerate(data):
        for bpos in range(8):
            y = (byte >> (7 - bpos)) & 1
            hp, hx = int(rng.integers(1, 4095)), int(rng.integers(0, 256))
            pr, pg = L.reffx_model_update(ref, y, hp, hx, a.ctypes.data), lib.orc_fx_model_update(got, y, hp, hx, b.ctypes.data)
            bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
            assert bad.size == 0 and pr == pg, (name, n, bpos, bytes(data[max(0, n - 20):n]), bad[:8], pr, pg)
            if bpos == 7:
                k = L.reffx_model_debug(da.ctypes.data)
                assert k == lib.orc_fx_model_debug(got, db.ctypes.data) and (da == db).all(), (name, n, np.nonzero(da != db)[0])
                L.reffx_model_contexts(ca.ctypes.data)
                lib.orc_fx_model_contexts(got, cb.ctypes.data)
                assert (ca == cb).all(), (name, n, np.nonzero(ca != cb)[0])


@needs_ref
@pytest.mark.parametrize("blpos", [14 * 256 * 1024 - 700, 28 * 512 * 1024 - 700, 448131719 - 500, 451531986 - 500, 463139793 - 500])
def test_whole_fxcm_model_vs_reference_across_block_position_thresholds(blpos, tmp_path):
    """fxcm reads its position in the block at five places: the SSCM / APM rates step at 3.67 MB and 14.7 MB (update1, fxcmv1.cpp:4772-4774) and
    modelPrediction changes three decisions between 448 MB and 463 MB (:3200, :3918 / :3929, :3965). The 4 and 8 MiB files cross the first for real;
    the others lie beyond every stream that was ever run here. State injection (round 6's wrap / threshold audit): the reference's own
    fxcmv1::Predictor and the oracle are both placed shortly before each threshold and run across it -- all 431 values after every bit. The stage's
    host bodies and device kernels are pinned on the oracle at the same places (tests/test_fxcm_stage_host.py, test_zgpu_stage_fxcm.py)."""
    from cmix_amd import synth
    L, lib = _private_fx_copy(tmp_path), O.lib()
    lib.orc_fx_model_new.restype = P
    lib.orc_fx_model_update.argtypes = [P, C.c_int, C.c_int, C.c_int, P]
    lib.orc_fx_model_set_blpos.argtypes = [P, C.c_int]
    L.reffx_model_set_blpos.argtypes = [P, C.c_int]
    rng = np.random.default_rng(blpos & 0xffff)
    data = synth.enwik_like(1400, 61, rich=True)
    ref, got = L.reffx_model_new(), lib.orc_fx_model_new()
    L.reffx_model_set_blpos(ref, blpos)
    lib.orc_fx_model_set_blpos(got, blpos)
    a, b = np.zeros(431, np.float32), np.zeros(431, np.float32)
    for n, byte in enumerate(data):
        for bpos in range(8):
            y = (byte >> (7 - bpos)) & 1
            hp, hx = int(rng.integers(1, 4095)), int(rng.integers(0, 256))
            pr, pg = L.reffx_model_update(ref, y, hp, hx, a.ctypes.data), lib.orc_fx_model_update(got, y, hp, hx, b.ctypes.data)
            bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
            assert bad.size == 0 and pr == pg, (blpos, n, bpos, bad[:8], pr, pg)


DICTIONARY = "/root/reference/dictionary/english.dic"


def _wrt_codewords():
    """Inverse of fxcm's decodeCodeWord (reference src/models/fxcmv1.cpp:389-411): dictionary index -> the 1..3 codeword
    bytes (128..255) cmix's WRT preprocessor writes for it."""
    inv = {}
    for s0 in range(128):
        if s0 < 80:
            inv[s0] = bytes([128 + s0])
            continue
        for s1 in range(128):
            i = 80 * (s0 - 80)
            if s1 < 80:
                inv.setdefault(i + s1 + 80, bytes([128 + s0, 128 + s1]))
                continue
            j = (i - 80 * 32) * 32 + 80 * (s1 - 80)
            for s2 in range(128):
                if j + s2 + 80 * 49 > 0:
                    inv.setdefault(j + s2 + 80 * 49, bytes([128 + s0, 128 + s1, 128 + s2]))
    return inv


@needs_ref
@pytest.mark.skipif(not __import__("os").path.exists(DICTIONARY), reason="the reference's dictionary is not on this box")
def test_whole_fxcm_model_with_wrt_dictionary_vs_reference(tmp_path):
    """cmix -c <dictionary>: the stream carries WRT codewords (bytes 128..255) that fxcm decodes through the
    dictionary; the decoded words feed the stemmer and the <text> / <math> / <pre> / <nowiki> / </page> detectors, the
    dictionary index selects a mixer weight set. A wiki page written the way the preprocessor writes it (swapped
    punctuation, codewords for dictionary words, literal letters otherwise), plus undecodable and over-long codewords."""
    L, lib = _private_fx_copy(tmp_path), O.lib()
    L.reffx_model_new_dict.restype = P
    L.reffx_model_new_dict.argtypes = [C.c_char_p]
    lib.orc_fx_model_new_dict.restype = P
    lib.orc_fx_model_new_dict.argtypes = [C.c_char_p]
    lib.orc_fx_model_update.argtypes = [P, C.c_int, C.c_int, C.c_int, P]
    lib.orc_fx_model_debug.argtypes = [P, P]
    lib.orc_fx_model_contexts.argtypes = [P, P]
    words = [w.rstrip(b"\n") for w in open(DICTIONARY, "rb").read().split(b"\n")]
    index = {w: i for i, w in enumerate(words) if w}
    inv = _wrt_codewords()
    rng = np.random.default_rng(5)

    def enc(text):
        """lower-case letters runs -> codeword when in the dictionary; punctuation swapped the way cmix swaps it"""
        import re
        out = bytearray()
        for tok in re.findall(rb"[a-z]+|[^a-z]", text):
            if tok.isalpha() and tok in index and index[tok] in inv and rng.random() < 0.85:
                out += inv[index[tok]]
            else:
                for c in tok:
                    if ord("{") <= c < 127:
                        c += ord("P") - ord("{")
                    elif ord(":") <= c <= ord("?"):
                        c ^= 0x70
                    out.append(c)
        return bytes(out)

    page = (b"<page> <title>the first page</title> <text xml:space=\"preserve\">'''the page''' is about the history of science and the people.\n"
            b"it has <math>a + b</math> and <nowiki>some [[raw]] text</nowiki> and <pre>fixed\n  text</pre> here. see [[category:history]] "
            b"[[image:people.png|the caption]] [[wikipedia:about]] and: more words, seven hundred million times.\n\n* one item\n* another item\n"
            b"</text> </page>\n")
    data = enc(page) * 4 + bytes([0xff, 0xfe, 0xfd, 0xfc, 0xfb]) + enc(b" after a long codeword ") + bytes([200, 255]) + enc(b" end of the text.\n" * 20)
    assert max(data) > 127
    ref, got = L.reffx_model_new_dict(DICTIONARY.encode()), lib.orc_fx_model_new_dict(DICTIONARY.encode())
    a, b = np.zeros(431, np.float32), np.zeros(431, np.float32)
    da, db, ca, cb = np.zeros(48, np.uint32), np.zeros(48, np.uint32), np.zeros(256, np.uint32), np.zeros(256, np.uint32)
    flags, decoded = 0, set()
    for n, byte in enumerate(data):
        for bpos in range(8):
            y = (byte >> (7 - bpos)) & 1
            hp, hx = int(rng.integers(1, 4095)), int(rng.integers(0, 256))
            pr, pg = L.reffx_model_update(ref, y, hp, hx, a.ctypes.data), lib.orc_fx_model_update(got, y, hp, hx, b.ctypes.data)
            bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
            assert bad.size == 0 and pr == pg, (n, bpos, bytes(data[max(0, n - 20):n]), bad[:8], pr, pg)
        k = L.reffx_model_debug(da.ctypes.data)
        flags |= int(da[k - 3])
        decoded.add(int(da[k - 1]))
        lib.orc_fx_model_debug(got, db.ctypes.data)
        assert (da == db).all(), (n, bytes(data[max(0, n - 20):n + 1]), np.nonzero(da != db)[0], da[np.nonzero(da != db)[0]], db[np.nonzero(da != db)[0]])
        L.reffx_model_contexts(ca.ctypes.data)
        lib.orc_fx_model_contexts(got, cb.ctypes.data)
        assert (ca == cb).all(), (n, bytes(data[max(0, n - 20):n + 1]), np.nonzero(ca != cb)[0])
    assert flags & 7 == 7 and len(decoded) > 20, (flags, len(decoded))   # <text>, <math>, <pre> states entered ("nowiki" is not a dictionary word); many words decoded
