"""The paq8 stage without a GPU: the product's host front end (cmix_amd/csrc/p8front/) + the product's device bodies
(p8cm_dev.h, p8cm2_dev.h, p8dmc_dev.h, p8stage_dev.h) run on the host by tests/host/p8stage_emul.cpp, against
  * columns 434..2024 of the committed traces of the UNMODIFIED reference predictor (tests/golden/*.npz,
    tests/golden/make_golden.py): all 1591 values PAQ8::Predict() hands to cmix before every bit, bit for bit;
  * per-step hashes of the same columns over longer reference traces (tests/golden/paq8_cols_*.npz,
    tests/golden/make_paq8_hashes.py);
  * the oracle's restatement (oracle/paq8_predictor.c, itself pinned to the reference's paq8::Predictor) on streams
    the fixtures do not hold.
The same comparisons run on the device in tests/test_zgpu_p8stage.py."""
import ctypes as C
import glob
import os
import subprocess

import numpy as np
import pytest

from conftest import load_golden
import make_golden as mg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tests", "host", "libp8stageemul.so")
SRC = os.path.join(ROOT, "tests", "host", "p8stage_emul.cpp")
CSRC = os.path.join(ROOT, "cmix_amd", "csrc")
FRONT = sorted(glob.glob(os.path.join(CSRC, "p8front", "*.c")))
DEPS = [SRC] + FRONT + glob.glob(os.path.join(CSRC, "p8front", "*.h")) + [os.path.join(CSRC, f) for f in (
    "p8_rec.h", "p8stage_dev.h", "p8stage_build.h", "p8fam_dev.h", "p8cm2v2_dev.h", "p8cm_dev.h", "p8cm_build.h", "p8cm2_dev.h", "p8cm2_build.h", "p8dmc_dev.h", "p8dmc_build.h")]


def emul():
    if not os.path.exists(SO) or any(os.path.getmtime(d) > os.path.getmtime(SO) for d in DEPS):
        obj = os.path.join(ROOT, "tests", "host", "_p8obj")
        os.makedirs(obj, exist_ok=True)
        objs = []
        for f in FRONT:
            o = os.path.join(obj, os.path.basename(f)[:-2] + ".o")
            subprocess.check_call(["gcc", "-std=gnu11", "-O2", "-fPIC", "-ffp-contract=off", "-w", "-include", os.path.join(CSRC, "p8front", "p8f_alloc.h"), "-c", f, "-o", o])
            objs.append(o)
        subprocess.check_call(["g++", "-std=c++17", "-O2", "-fPIC", "-shared"] + os.environ.get("CMX_EMUL_FLAGS", "").split() + ["-o", SO, SRC] + objs + ["-lm"])
    L = C.CDLL(SO)
    L.p8s_create.restype = C.c_void_p
    L.p8s_create.argtypes = [C.c_int]
    L.p8s_destroy.argtypes = [C.c_void_p]
    L.p8s_run.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    L.p8s_stats.argtypes = [C.c_void_p, C.c_void_p]
    return L


def run_stage(data, chunks=None, miniwalk=None, mini_stats=None, late=False):
    """PAQ8::Predict()'s 1591 values before every bit of data, through the emulated stage in the given chunk sizes. miniwalk: how the
    ContextMap family resolves an overlap at a lookup bit (None / 1: as the kernel, 0: whole-instance walks, 2: the fall-back path forced)."""
    L = emul()
    data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
    h = L.p8s_create(11)
    assert h
    if miniwalk is not None:
        L.p8s_set_miniwalk.argtypes = [C.c_void_p, C.c_int]
        L.p8s_set_miniwalk(h, miniwalk)
    if late:
        L.p8s_set_late.argtypes = [C.c_void_p, C.c_int]
        L.p8s_set_late(h, int(late))   # 1: the decoder's order; 3: for the steps of the models with their own tables too
    out = np.zeros((8 * len(data), 1591), np.float32)
    pos, k = 0, 0
    chunks = chunks or [len(data)]
    while pos < len(data):
        n = min(chunks[k % len(chunks)], len(data) - pos)
        k += 1
        part = out[8 * pos:8 * (pos + n)]
        assert L.p8s_run(h, data[pos:].ctypes.data, n, part.ctypes.data) == 0
        pos += n
    st = np.zeros(3, np.uint64)
    L.p8s_stats(h, st.ctypes.data)
    if mini_stats is not None:
        L.p8s_miniwalk_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.p8s_miniwalk_stats(h, mini_stats.ctypes.data)
    L.p8s_destroy(h)
    return out, st


@pytest.mark.parametrize("name", ["text_96", "binary_64"])
def test_stage_reproduces_golden_columns(name):
    g = load_golden(name)
    probs = mg.unpack_probs(g)
    got, _ = run_stage(g["stream"], chunks=[1, 1, 7, 30])
    want = np.ascontiguousarray(probs[:, 434:2025])
    bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, (name, "first mismatch (step, column):", bad[0], got[tuple(bad[0])] * 4095, want[tuple(bad[0])] * 4095)


def oracle_columns(data):
    from oracle import oracle as O
    lib = O.lib()
    lib.orc_p8_predictor_new.restype = C.c_void_p
    lib.orc_p8_predictor_new.argtypes = [C.c_int]
    lib.orc_p8_predictor_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    lib.orc_p8_rnd_reset()
    h, tag = O.new_owned(lib.orc_p8_predictor_new, 11)
    bits = np.unpackbits(np.frombuffer(bytes(data), np.uint8))
    out = np.full((len(bits), 1591), 0.5, np.float32)
    for t in range(len(bits) - 1):
        assert lib.orc_p8_predictor_update(h, int(bits[t]), out[t + 1].ctypes.data) >= 0
    O.release(tag)
    return out


@pytest.mark.parametrize("kind", ["text", "wiki", "binary", "runs"])
def test_stage_vs_oracle(kind):
    """Streams the fixtures do not hold, against the oracle's monolithic restatement (pinned to paq8::Predictor)."""
    from cmix_amd import synth
    r = np.random.default_rng(7)
    if kind == "text":
        data = synth.enwik_like(6000, 11)
    elif kind == "wiki":
        data = (b"== Heading ==\n[[Link|text]] and ''italic'' {{template|a=1}}\n* item one\n* item two\n<ref name=\"x\">cite</ref>\n" * 40)[:4000]
    elif kind == "binary":
        data = bytes(r.integers(0, 256, 3000, dtype=np.uint8))
    else:
        data = b"\x00" * 700 + b"abcabcabc" * 100 + b"\xff" * 300 + bytes(r.integers(0, 4, 1000, dtype=np.uint8))
    got, st = run_stage(data, chunks=[1, 100, 1000])
    want = oracle_columns(data)
    bad = np.argwhere(got.view(np.uint32) != want.view(np.uint32))
    assert bad.size == 0, (kind, "first mismatch (step, column):", bad[0], got[tuple(bad[0])] * 4095, want[tuple(bad[0])] * 4095, st)


def load_hashes(name):
    path = os.path.join(ROOT, "tests", "golden", "paq8_cols_%s.npz" % name)
    with np.load(path) as z:
        return z["stream"].copy(), z["hash"].copy()


# a 32-bit PAM (IMAGE32 block: the alpha path) and a JPEG with a thumbnail JPEG inside its APP1 segment;
# TGA payloads (true colour as an IMAGE24 block; grayscale, colour-mapped and 32-bit ones found by paq8's own detector); JPEG with 4:4:4 chroma, a progressive
# JPEG (the model stays off) and a JPEG cut off in the middle of its scan;
# one file with a BMP, a WAV, a JPEG and a PGM between pieces of text (every switch between the generic models and a model with its own tables);
# baseline JPEG (jpegModel: the marker parser, the Huffman decoder and the coefficient predictors on the host; 32 contexts on a BH<9> table, the model's
# own mixer and two APM stages on one lane; exports interleaved in call order): a 4:2:0 colour picture in a JPEG block, a grayscale one with restart
# markers in a DEFAULT block;
# a 4-bit BMP (im4bitModel: 14 contexts on a HashTable<16> walked by one lane);
# 1-bit images (im1bitModel: a PBM the preprocessor makes an IMAGE1 block of, a 1-bit BMP inside a DEFAULT block);
# PCM audio in RIFF / WAVE files inside DEFAULT blocks (paq8's detector switches audio8bModel / wavModel + recordModel on: the model's family
# holds recordModel's generic ContextMaps, whose state changes hands at every switch; wavModel's long-double Cholesky on the host);
# 8-bit images (im8bitModel: the grayscale face on a PGM the preprocessor makes an IMAGE8GRAY block of and on a BMP whose gray palette paq8's
# detector walks, the palette face on a BMP with a colour palette), and 24 / 32-bit images (im24bitModel through the model's own ContextMap and lane table, 13 weight sets, Image.Color's APM
# chain): an IMAGE24 block between other blocks; 32-bit and 24-bit BMP files inside DEFAULT blocks, where paq8's own detector switches the
# model on -- and, in the short 32-bit one, off again before the stream ends
@pytest.mark.parametrize("name,nbytes", [("text_32k", 6144), ("wiki_12k", 4096), ("records_8k", 4096), ("mixed_24k", 6400), ("rich_16k", 16384), ("hdrs_4k", 3560),
                                         ("bmp24_14k", 14602), ("bmp32_8k", 8195), ("bmp24_raw_9k", 8907),
                                         ("pgm8_4k", 4116), ("bmp8_gray_raw_5k", 4711), ("bmp8_pal_raw_5k", 4711),
                                         ("wav16s_6k", 6099), ("wav8s_4k", 3949), ("wav16m_3k", 2849), ("wav8m_2k", 1949),
                                         ("pbm1_2k", 1965), ("bmp1_raw_2k", 1917), ("bmp4_raw_3k", 3061),
                                         ("jpeg_5k", 2124), ("jpeg_rst_raw_3k", 1287), ("mixed_media_12k", 10168),
                                         ("tga24_5k", 4950), ("tga_gray_map_32_raw_9k", 8679), ("jpeg_444_prog_cut_6k", 4051), ("pam32_thumb_8k", 7085),
                                         ("media_in_text_9k", 6138)])
def test_stage_vs_reference_hashes(name, nbytes):
    """Prefixes of the reference-derived fixtures of tests/golden/make_paq8_hashes.py (the device test runs them whole)."""
    from make_paq8_hashes import row_hash
    stream, want = load_hashes(name)
    got, _ = run_stage(stream[:nbytes], chunks=[1000, 333])
    h = row_hash(got)
    bad = np.nonzero(h != want[:8 * nbytes])[0]
    assert bad.size == 0, (name, "first differing step:", bad[0], "of", 8 * nbytes)


@pytest.mark.parametrize("mode", [0, 2])
def test_family_overlap_paths(mode):
    """An overlap of the ContextMap family at a lookup bit walks only the contexts that share a key (p8f_miniwalk); the whole-instance walk
    (mode 0) and the narrowed walk's fall-back for a second visit phase 1 did not list (mode 2: forced at every second visit) must give the
    same columns -- the reference's."""
    from make_paq8_hashes import row_hash
    stream, want = load_hashes("rich_16k")
    nbytes = 5000
    ms = np.zeros(2, np.uint64)
    got, _ = run_stage(stream[:nbytes], chunks=[1000, 333], miniwalk=mode, mini_stats=ms)
    bad = np.nonzero(row_hash(got) != want[:8 * nbytes])[0]
    assert bad.size == 0, (mode, "first differing step:", bad[0])
    if mode == 2:
        assert ms[0] > 500 and ms[1] > 50, ms   # narrowed walks happened, and a good part of them took the fall-back


def test_cm2_walk_and_reload_path(monkeypatch):
    """ContextMap2's second design falls back to a serial walk of the instance on overlaps, which the 2 GB tables almost
    never produce: force the walk on two bits out of five and check that walk -> reload -> lane-parallel stays exact."""
    monkeypatch.setenv("CMX_P8C2_FORCE_WALK", "1")
    g = load_golden("text_96")
    probs = mg.unpack_probs(g)
    got, st = run_stage(g["stream"], chunks=[5, 40])
    assert st[2] > 100
    want = np.ascontiguousarray(probs[:, 434:2025])
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_decoders_order_of_operations_gives_the_same_columns():
    """The late-bit protocol (cmix_amd/csrc/cmx_late.h): a decoder's front end emits the records of a step only after the bit before it
    has been decoded (p8f_front_emit_step / p8f_front_set_bit instead of p8f_front_run over known bytes), and the maps' uniform
    registers take that bit at the top of the step (p8d_bit_y; p8f_uni_tail of step t - 1, then p8f_uni_head of step t) instead of
    reading the chunk's bytes ahead. Same values, bit for bit, in ragged chunks -- against the reference's own columns."""
    g = load_golden("text_96")
    probs = mg.unpack_probs(g)
    out, _ = run_stage(g["stream"], chunks=[1, 7, 40, 13], late=True)
    a, b = out.view(np.uint32), np.ascontiguousarray(probs[:, 434:2025]).view(np.uint32)
    bad = np.argwhere(a != b)
    assert len(bad) == 0, f"late order: column {434 + bad[0][1]} differs first at bit {bad[0][0]}"
    rng = np.random.default_rng(3)
    data = bytes(rng.integers(0, 256, 1500, dtype=np.uint8)) + b"the quick brown fox jumps over the lazy dog. " * 40
    x, _ = run_stage(data, chunks=[512])
    y, _ = run_stage(data, chunks=[512, 1, 300], late=True)
    assert (x.view(np.uint32) == y.view(np.uint32)).all()


@pytest.mark.parametrize("name,nbytes", [("mixed_media_12k", 10168), ("bmp4_raw_3k", 3061), ("jpeg_5k", 2124), ("wav8m_2k", 1949), ("pgm8_4k", 4116),
                                         ("pam32_thumb_8k", 7085)])
def test_decoders_order_of_operations_on_media_streams(name, nbytes):
    """What the decoder's form of the image / audio / JPEG steps has to compute (the device form is not built yet: `cmix_dropin -d` refuses
    such a file): the front end emits a model step's records only after the bit before it is known (its detectors, the JPEG parser and
    the OLS predictors advance bit by bit), the generic maps follow the bits of a model's bytes through the tail of the step before
    (never reading a bit ahead), and the generator / the shared instances change hands at the same byte boundaries. Same values as the
    reference's, across every switch between the generic models and a model with tables of its own."""
    from make_paq8_hashes import row_hash
    stream, want = load_hashes(name)
    got, _ = run_stage(stream[:nbytes], chunks=[512, 77, 300], late=3)
    bad = np.nonzero(row_hash(got) != want[:8 * nbytes])[0]
    assert bad.size == 0, (name, "first differing step:", bad[0], "of", 8 * nbytes)


def test_model_step_inside_a_text_block_is_coded():
    """Audio (or a 1- / 4-bit image, or JPEG) data that paq8's own detectors find INSIDE A TEXT BLOCK ends in the text chain of final APM stages
    (paq8.cpp:8281-8296). Rounds 3-4 refused such a stream (P8F_ERR_MODEL_IN_TEXT: the step's own fields shared the record's cells with the chain's
    contexts); since round 5 the record has fields of its own for them (P8ApmRec::m) and the step runs the chain of the block's type -- the values are the
    reference's (test_stage_vs_reference_hashes[media_in_text_9k], and on the device tests/test_zgpu_p8stage.py); here: both framings of the same data run."""
    from make_paq8_hashes import wav_file
    from cmix_amd import synth
    text = synth.enwik_like(900, 21)
    L = emul()
    for framing in (mg.text_block, mg.default_block):
        data = np.ascontiguousarray(np.frombuffer(bytes(framing(text[:400] + wav_file(300, 1, 8, 3) + text[400:700])), np.uint8))
        h = L.p8s_create(11)
        out = np.zeros((8 * len(data), 1591), np.float32)
        rc = L.p8s_run(h, data.ctypes.data, len(data), out.ctypes.data)
        L.p8s_destroy(h)
        assert rc == 0, (framing.__name__, rc)


@pytest.mark.skipif(os.environ.get("CMX_LONG") != "1", reason="168 KB through the host emulation: about two minutes (CMX_LONG=1); the device test runs it always")
def test_big_media_stream_digests():
    """tests/golden/make_paq8_big_media.py: 168 KB of images, audio and a JPEG at sizes where a model's segment runs through many chunks -- every
    256 steps' digest of the 1591 values against the unmodified reference's."""
    from make_paq8_hashes import row_hash
    from make_paq8_big_media import digest
    with np.load(os.path.join(ROOT, "tests", "golden", "paq8_big_media_168k.npz")) as z:
        stream, want = z["stream"].copy(), z["digest"].copy()
    L = emul()
    h = L.p8s_create(11)
    hashes = np.zeros(8 * len(stream), np.uint32)
    pos = 0
    while pos < len(stream):
        n = min(4096, len(stream) - pos)
        out = np.zeros((8 * n, 1591), np.float32)
        assert L.p8s_run(h, stream[pos:].ctypes.data, n, out.ctypes.data) == 0
        hashes[8 * pos:8 * (pos + n)] = row_hash(out)
        pos += n
    L.p8s_destroy(h)
    bad = np.nonzero(digest(hashes) != want)[0]
    assert bad.size == 0, ("first differing block of 256 steps:", bad[0], "of", len(want))


@pytest.mark.parametrize("start", [(1 << 31) - 64 * 15000, (1 << 32) - 64 * 15000])
def test_shared_generator_counter_wraps_like_the_references(start):
    """The ContextMap family's shared rnd() (paq8.cpp:152-165) keeps an `int` counter that only matters modulo 64; it passes 2^31 after ~4 MB and 2^32
    after 8.0 MB of enwik-like text (34..67 draws per bit). Round 5's 8 MiB run left the reference exactly there: the look-ahead ring of generator values
    was refilled up to an index compared with `<=`, which fails in the step the counter wraps. Here the counter is placed shortly before 2^31 / 2^32
    (a multiple of 64: the 64 table words keep their places, so the VALUES are those of a fresh generator) and the outputs must stay the reference's."""
    from make_paq8_hashes import row_hash
    L = emul()
    L.p8s_rnd_i.restype = C.c_uint32
    L.p8s_rnd_i.argtypes = [C.c_void_p]
    L.p8s_set_rnd_i.argtypes = [C.c_void_p, C.c_uint32]
    stream, want = load_hashes("rich_16k")
    n = 7000
    h = L.p8s_create(11)
    L.p8s_set_rnd_i(h, start & 0xFFFFFFFF)
    d = np.ascontiguousarray(stream[:n])
    out = np.zeros((8 * n, 1591), np.float32)
    pos = 0
    while pos < n:
        m = min(1000, n - pos)
        assert L.p8s_run(h, d[pos:].ctypes.data, m, out[8 * pos:].ctypes.data) == 0
        pos += m
    end = L.p8s_rnd_i(h)
    L.p8s_destroy(h)
    assert end < (start & 0xFFFFFFFF) or (start < (1 << 31) <= end), "the counter did not pass the boundary: lengthen the stream"
    bad = np.nonzero(row_hash(out) != want[:8 * n])[0]
    assert bad.size == 0, ("first differing step:", int(bad[0]), "byte", int(bad[0]) // 8)


def test_byte_position_passes_the_end_of_the_history_ring_like_the_references():
    """paq8's `pos` (paq8.cpp:167) indexes a 2^30-byte ring at cmix's level 11 (Buf :169-186: every read is (pos - i) & (size - 1)); the match models and the
    detectors store and compare the unmasked value. A stream gets there after 1 GB. State injection (round 6's wrap / threshold audit): the unmodified
    paq8::Predictor started 3000 bytes below 2^30 (oracle/ref_paq8core.cpp refp8_set_pos) gave the fixture; the front end is placed the same way."""
    from make_paq8_hashes import row_hash
    L = emul()
    L.p8s_set_pos.argtypes = [C.c_void_p, C.c_int]
    with np.load(os.path.join(ROOT, "tests", "golden", "paq8_cols_pos_1g_6k.npz")) as z:
        stream, want, pos0 = z["stream"].copy(), z["hash"].copy(), int(z["inject_pos"][0])
    n = len(stream)
    assert pos0 < (1 << 30) < pos0 + n
    h = L.p8s_create(11)
    L.p8s_set_pos(h, pos0)
    d = np.ascontiguousarray(stream)
    out = np.zeros((8 * n, 1591), np.float32)
    pos = 0
    for m in (2990, 10, 1, 1000, n):   # a chunk that ends 2990 + 10 = 3000 bytes in: exactly at the ring's end; the byte behind it alone
        m = min(m, n - pos)
        if m:
            assert L.p8s_run(h, d[pos:].ctypes.data, m, out[8 * pos:].ctypes.data) == 0
        pos += m
    L.p8s_destroy(h)
    bad = np.nonzero(row_hash(out) != want)[0]
    assert bad.size == 0, ("first differing step:", int(bad[0]), "byte", int(bad[0]) // 8, "(the ring ends in front of byte 3000)")
    # and the fixture does depend on the injection: from position 0 the same stream gives other values
    h = L.p8s_create(11)
    out2 = np.zeros((8 * 400, 1591), np.float32)
    assert L.p8s_run(h, d.ctypes.data, 400, out2.ctypes.data) == 0
    L.p8s_destroy(h)
    assert (row_hash(out2) != want[:8 * 400]).any()
