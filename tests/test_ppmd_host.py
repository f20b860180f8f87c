"""CPU: the host PPMd stage (through the C ABI) vs the byte distributions of the unmodified reference's
PPMD byte model recorded in the golden traces. Bit-exact on all 256 floats after every byte."""
import numpy as np
import pytest

from conftest import bits_equal, load_golden


def _check(name, big=False, chunks=None):
    from cmix_amd import engine as E
    g = load_golden(name, big)
    stream = g["stream"]
    N = len(stream)
    p = E.Ppmd(g["vocab"])
    edges = [0, N] if not chunks else sorted(set([0, N] + list(chunks)))
    outs = [p.run(stream[a:b].tobytes()) for a, b in zip(edges[:-1], edges[1:])]
    got = np.concatenate(outs)
    p.close()
    bad = np.nonzero(~bits_equal(got, g["ppmd_probs"][1:N + 1]).all(axis=1))[0]
    assert len(bad) == 0, f"{name}: PPMd distribution differs first after byte {bad[0]} of {N}"


def test_text_96():
    _check("text_96")


def test_binary_64():
    _check("binary_64", chunks=[1, 33])


def test_brackets_1k():
    _check("brackets_1k")  # long repeats: deep contexts, frequency rescaling


def test_text_2k():
    _check("text_2k_nofull", chunks=[1000])


def test_text_32k_local():
    _check("text_32k", big=True)


def test_random_160k_local():
    _check("random_160k", big=True)  # incompressible: widest contexts, constant escapes


def test_text_330k_local():
    _check("text_330k_bytes", big=True, chunks=[100000, 200001])  # tests/golden/make_long_trace.py


def test_vocab_mask_and_errors():
    from cmix_amd import engine as E
    v = np.zeros(256, np.uint8)
    v[[97, 98, 99]] = 1
    p = E.Ppmd(v)
    out = p.run(b"abcabcabcabc")
    assert out.shape == (12, 256) and (out[:, v == 0] == 0).all()
    assert np.allclose(out.sum(1), 1, atol=1e-5) and out[-1, 97] > 0.5  # 'a' follows "abc" repeats
    assert E.lib().cmx_ppmd_run(p.h, None, 3, None) != 0 and "bad argument" in E.last_error()
    p.close()


def test_small_arena_fills_up_then_fails_loudly():
    """1 MB arena (reference PPMD(25, 1) stand-alone): bit-exact while the reference works (50 000 bytes, arena
    full to the brim), then -- where the reference build segfaults in its restore path -- a clean error."""
    from cmix_amd import engine as E
    g = load_golden("ppmd_1mb_50k")
    data = g["stream"]
    N = len(g["p_next"])
    p = E.Ppmd(np.ones(256, np.uint8), 25, 1)
    out = p.run(data[:N].tobytes())
    nxt = out[np.arange(N), data[1:N + 1]]
    bad = np.nonzero(~bits_equal(nxt, g["p_next"]))[0]
    assert len(bad) == 0, f"p(next byte) differs first after byte {bad[0]}"
    chk = (out.astype(np.float64) * np.arange(1, 257, dtype=np.float64)).sum(1)
    assert np.array_equal(chk, g["chk"]), "distribution checksum differs"
    with pytest.raises(E.CmxError, match="exhausted"):
        p.run(data[N:N + 10000].tobytes())
    with pytest.raises(E.CmxError, match="exhausted"):  # stays failed
        p.run(b"abc")
    p.close()
