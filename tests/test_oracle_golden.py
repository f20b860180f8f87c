"""CPU: the C restatement (oracle/) against the committed golden traces of the
unmodified reference (tests/golden/*.npz, produced by tests/golden/make_golden.py)."""
import numpy as np

from conftest import bits_equal, load_golden
import make_golden as mg
from oracle import oracle as O


def _check(name, big=False):
    g = load_golden(name, big)
    probs = mg.unpack_probs(g)
    net = O.MixNet()
    T = len(g["bits"])
    for t in range(T):
        p, mix = net.step(probs[t], g["sel"][t], g["bits"][t], want_mix=True)
        assert bits_equal(p, g["p_final"][t]).all(), f"{name}: final p differs at bit {t}"
        assert bits_equal(mix, g["mix_out"][t]).all(), f"{name}: mixer outputs differ at bit {t}"
        assert net.aux_context() == int(g["sel"][t][12]), f"{name}: aux context differs at bit {t}"


def test_mixnet_text_golden():
    _check("text_96")


def test_mixnet_binary_golden():
    _check("binary_64")


def test_mixnet_text_4k_local():
    _check("text_4k", big=True)


def test_stretch_matches_reference_layer0():
    """MixerInput::SetInput: raw probs -> stretch; aux selector derived from them."""
    g = load_golden("text_96")
    probs = mg.unpack_probs(g)
    lut = O.logit_table()
    assert np.isfinite(lut).all() and lut[0] < -9 and lut[-1] > 9
    # paq8/fxcm outputs are k/4095: almost everything must be on the grid
    assert (g["probs_q"] != 0xFFFF).mean() > 0.9
    s = np.array([O.stretch(p) for p in probs[5]], np.float32)
    assert np.all(np.abs(s) <= 9.3)


def test_sse_tables_shape():
    st, sq = O.sse_tables()
    assert st[0] == 0 and st[16384] in (16383, 16384) and sq[1] > 32000 and sq[32767] < 100
    assert np.all(np.diff(st[1:].astype(np.int32)) <= 0)  # stretch((1-p)/p) decreases with p
