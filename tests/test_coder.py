"""CPU: the callers of the path -- arithmetic coder (Encoder/Decoder) and container header -- through the C ABI,
against (a) golden vectors produced by the unmodified reference coder and by the reference binary
(tests/golden/coder_vectors.npz, tests/golden/make_coder_vectors.py), (b) the reference coder itself where
oracle/_ref is built, (c) the oracle restatement, and (d) encode -> decode round trips at sizes the trace-based
checks cannot reach. Integer/byte work: everything bit-exact."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as O
from oracle import refharness as R

CASES = ["confident", "uniform", "extremes", "grid_edges", "empty", "one_bit"]


def _vec():
    with np.load(os.path.join(GOLDEN, "coder_vectors.npz")) as z:
        return {k: z[k] for k in z.files}


def _encode(p, bits, chunks=None):
    from cmix_amd import engine as E
    e = E.Encoder()
    edges = [0, len(p)] if not chunks else sorted(set([0, len(p)] + [c for c in chunks if c < len(p)]))
    for a, b in zip(edges[:-1], edges[1:]):
        e.encode_bits(p[a:b], bits[a:b])
    e.flush()
    out = e.data()
    e.close()
    return out


@pytest.mark.parametrize("name", CASES)
def test_encoder_golden(name):
    v = _vec()
    p, bits, code = v[name + "_p"], v[name + "_bits"], v[name + "_code"].tobytes()
    assert _encode(p, bits) == code
    assert _encode(p, bits, chunks=[1, 2, 777, 4096]) == code  # streaming = one shot
    assert O.coder_encode(p, bits) == code                     # the oracle restatement, same vectors


@pytest.mark.parametrize("name", CASES)
def test_decoder_golden(name):
    from cmix_amd import engine as E
    v = _vec()
    p, bits, code = v[name + "_p"], v[name + "_bits"], v[name + "_code"].tobytes()
    want = O.coder_decode(p, code)
    if name != "extremes":  # p = 0 / 1 with a contradicting bit is not decodable, by construction of the coder
        assert (want == bits).all()
    d = E.Decoder(code)
    assert (d.decode_bits(p) == want).all()
    d.close()
    d = E.Decoder(code)
    assert [d.decode(float(q)) for q in p[:500]] == list(want[:500])  # bit-at-a-time entry point
    d.close()


def test_whole_file_of_the_reference_binary():
    """`cmix -n` on the payload of the binary_64 trace: 5-byte header + the code of that trace's final
    probabilities, MSB-first bytes (runner.cpp:34-52,101-121)."""
    from cmix_amd import engine as E
    v = _vec()
    with np.load(os.path.join(GOLDEN, "binary_64.npz")) as g:
        stream, p = g["stream"], g["p_final"]
    e = E.Encoder()
    e.encode_bytes(p[:8 * 10], stream[:10].tobytes())
    e.encode_bytes(p[8 * 10:], stream[10:].tobytes())
    e.flush()
    got = E.header_write(len(stream), np.ones(256, np.uint8)) + e.data()
    assert got == v["binary_64_file"].tobytes()
    length, dic, vocab, used = E.header_read(got)
    assert (length, dic, used) == (len(stream), False, 5) and vocab.all()
    d = E.Decoder(got[used:])
    bits = d.decode_bits(p)
    assert (np.packbits(bits) == stream).all()


def test_header_with_vocabulary_bitmap():
    from cmix_amd import engine as E
    v = _vec()
    payload = v["header_10k_payload"]
    stream = np.concatenate([np.array([0, 0, 0, 0x27, 0x10], np.uint8), payload])  # NoPreprocess block header
    vocab = np.zeros(256, np.uint8)
    vocab[np.unique(stream)] = 1
    want = v["header_10k"].tobytes()
    assert E.header_write(len(stream), vocab) == want
    assert O.header_write(len(stream), vocab) == want
    length, dic, got_vocab, used = E.header_read(want)
    assert (length, dic, used) == (len(stream), False, 37) and (got_vocab == vocab).all()
    h = E.header_write(12345678901, vocab, dictionary_used=True)
    length, dic, got_vocab, used = E.header_read(h)
    assert (length, dic, used) == (12345678901, True, 37) and (got_vocab == vocab).all()
    assert E.header_read(E.header_write(0, vocab))[0] == 0
    with pytest.raises(E.CmxError):
        E.header_write(1 << 39, vocab)
    with pytest.raises(E.CmxError):
        E.header_read(want[:20])


@pytest.mark.skipif(not R.coder_available(), reason="oracle/_ref/libcmixrefcoder.so not built")
def test_vs_reference_coder_fresh_seeds():
    rng = np.random.default_rng(99)
    for n, skew in [(3, 1), (1000, 1), (30000, 6), (30000, 30)]:
        bits = rng.integers(0, 2, n, dtype=np.uint8)
        conf = rng.random(n).astype(np.float32) ** skew
        p = np.where(bits == 1, 1 - 0.5 * conf, 0.5 * conf).astype(np.float32)
        code = R.ref_encode(p, bits)
        assert _encode(p, bits, chunks=[n // 3]) == code
        assert O.coder_encode(p, bits) == code
        assert (R.ref_decode(p, code) == bits).all()


def test_round_trip_4m_bits():
    """Size-independent property: decode(encode(bits)) == bits, and the code length is within a few bytes of the
    ideal -sum(log2 P(bit)) at 16-bit precision."""
    from cmix_amd import engine as E
    rng = np.random.default_rng(7)
    n = 1 << 22
    bits = rng.integers(0, 2, n, dtype=np.uint8)
    conf = rng.random(n).astype(np.float32) ** 3
    p = np.clip(np.where(bits == 1, 1 - 0.5 * conf, 0.5 * conf), 1e-4, 1 - 1e-4).astype(np.float32)
    flip = rng.random(n) < 0.02
    bits[flip] ^= 1
    code = _encode(p, bits, chunks=list(range(0, n, 100003)))
    d = E.Decoder(code)
    assert (d.decode_bits(p) == bits).all()
    p16 = (1 + 65534 * p.astype(np.float64)).astype(np.int64) / 65536.0
    ideal = -(np.log2(np.where(bits == 1, p16, 1 - p16))).sum() / 8
    assert ideal - 8 <= len(code) <= ideal * 1.001 + 8  # Flush() emits one byte of the final interval, not four


def test_bad_arguments_fail_loudly():
    from cmix_amd import engine as E
    e = E.Encoder()
    with pytest.raises(E.CmxError, match="outside"):
        e.encode_bits(np.array([1.5], np.float32), np.array([1], np.uint8))
    with pytest.raises(E.CmxError, match="outside"):
        e.encode_bits(np.array([np.nan], np.float32), np.array([1], np.uint8))
    e.flush()
    with pytest.raises(E.CmxError, match="flushed"):
        e.encode_bits(np.array([0.5], np.float32), np.array([1], np.uint8))
    d = E.Decoder(b"")
    assert d.decode(0.5) in (0, 1)  # reads past the end return zeros, like the reference's ReadByte
    with pytest.raises(E.CmxError):
        d.decode(2.0)
