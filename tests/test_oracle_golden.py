"""CPU: the C restatement (oracle/) against the committed golden traces of the
unmodified reference (tests/golden/*.npz, produced by tests/golden/make_golden.py)."""
import os

import numpy as np
import pytest

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


@pytest.mark.parametrize("name", ["text_96", "binary_64"])
def test_paq8_oracle_reproduces_golden_columns(name):
    """The assembled paq8 restatement (oracle/paq8_predictor.c) at cmix's own setting (level 11, reference
    src/predictor.cpp:85) against layer-0 columns 434..2024 of the traces recorded from the unmodified reference
    predictor: all 1591 values after every coded bit, bit for bit. Needs nothing but the committed fixtures."""
    import ctypes as C
    lib = O.lib()
    lib.orc_p8_predictor_new.restype = C.c_void_p
    lib.orc_p8_predictor_new.argtypes = [C.c_int]
    lib.orc_p8_predictor_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    g = load_golden(name)
    probs, bits = mg.unpack_probs(g), g["bits"]
    lib.orc_p8_rnd_reset()
    h, tag = O.new_owned(lib.orc_p8_predictor_new, 11)
    out = np.zeros(1591, np.float32)
    assert (probs[0, 434:2025] == 0.5).all()          # PAQ8::Predict() before the first Perceive
    for t in range(len(bits) - 1):
        assert lib.orc_p8_predictor_update(h, int(bits[t]), out.ctypes.data) >= 0
        want = np.ascontiguousarray(probs[t + 1, 434:2025])
        bad = np.nonzero(want.view(np.uint32) != out.view(np.uint32))[0]
        assert bad.size == 0, (name, t, bad[:8], want[bad[:4]] * 4095, out[bad[:4]] * 4095)
    O.release(tag)


@pytest.mark.parametrize("name", ["text_96", "binary_64"])
def test_fxcm_oracle_reproduces_golden_columns(name):
    """The assembled fxcm restatement (oracle/fxcm_model.c) against layer-0 columns 3..433 of the traces recorded from
    the unmodified reference predictor: all 431 values after every coded bit, bit for bit. fxcm reads two hints from
    the LSTM before every update (reference predictor.cpp:462-465: lstmpr = 1 + 4094 * p of the NEXT bit, lstmex = the
    likeliest byte of the current interval); they come from the LSTM restatement stepped over the same trace. Needs
    nothing but the committed fixtures."""
    import ctypes as C
    lib = O.lib()
    lib.orc_fx_model_new.restype = C.c_void_p
    lib.orc_fx_model_update.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p]
    g = load_golden(name)
    probs, bits, stream = mg.unpack_probs(g), g["bits"], g["stream"]
    assert (probs[0, 3:434] == 0.5).all()              # FXCM::Predict() before the first Perceive
    l = O.Lstm(g["vocab"])
    h, tag = O.new_owned(lib.orc_fx_model_new)
    out = np.zeros(431, np.float32)
    t = 0
    for n in range(len(stream)):
        for j in range(7, -1, -1):
            bit = (int(stream[n]) >> j) & 1
            assert bit == int(bits[t])
            l.bit_perceive(bit)
            if j == 0:
                l.byte_update(g["ppmd_probs"][n + 1], stream[n])
            if t + 1 == len(bits):
                break
            p_next = np.float32(l.bit_predict())
            assert bits_equal(p_next, probs[t + 1, 2077]).all()
            lstmpr = int(np.float32(1) + np.float32(4094) * p_next)
            assert lib.orc_fx_model_update(h, bit, lstmpr, int(l.ex()), out.ctypes.data) >= 0
            want = np.ascontiguousarray(probs[t + 1, 3:434])
            bad = np.nonzero(want.view(np.uint32) != out.view(np.uint32))[0]
            assert bad.size == 0, (name, t, bad[:8], want[bad[:4]] * 4095, out[bad[:4]] * 4095)
            t += 1
    O.release(tag)


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


def _check_lstm(name):
    """LSTM byte mixer restatement vs the reference's byte_mixers_[0] (per byte: 256-way
    distribution; per bit: ByteModel::Predict value = layer-0 input 2077)."""
    g = load_golden(name)
    l = O.Lstm(g["vocab"])
    stream = g["stream"]
    probs = mg.unpack_probs(g) if "probs_q" in g else None
    t = 0
    for n in range(len(stream)):
        for j in range(7, -1, -1):
            p = l.bit_predict()
            if probs is not None:
                assert bits_equal(p, probs[t, 2077]).all(), f"{name}: LSTM bit prediction differs at bit {t}"
            l.bit_perceive((int(stream[n]) >> j) & 1)
            t += 1
        out = l.byte_update(g["ppmd_probs"][n + 1], stream[n])
        assert bits_equal(out, g["lstm_probs"][n + 1]).all(), f"{name}: LSTM distribution differs after byte {n}"


@pytest.mark.skipif(os.environ.get("CMX_LONG") != "1", reason="~5 min; set CMX_LONG=1")
def test_lstm_330k_golden_past_3000_adam_rounds():
    """330 000 bytes = 3300 BPTT rounds: LstmLayer::update_steps_ saturates at 3000 and Adam's bias terms switch
    to the double-precision pow() path (lstm-layer.cpp:26-30). Fixture: tests/golden/make_long_trace.py (local)."""
    g = load_golden("text_330k_bytes", big=True)
    l = O.Lstm(g["vocab"])
    stream = g["stream"]
    for n in range(len(stream)):
        out = l.byte_update(g["ppmd_probs"][n + 1], stream[n])
        if n % 997 == 0 or n >= len(stream) - 300:
            assert bits_equal(out, g["lstm_probs"][n + 1]).all(), f"LSTM distribution differs after byte {n}"


def test_lstm_text_golden():
    _check_lstm("text_96")


def test_lstm_binary_golden():
    _check_lstm("binary_64")


def test_lstm_2k_golden_20_bptt_rounds():
    _check_lstm("text_2k_nofull")  # 2048 bytes: 21 BPTT + Adam rounds (every 100 bytes)


def _check_ctxmodels(name, big=False):
    """ContextManager + contexts + 54 small models restatement vs the reference: per bit the 54 model
    outputs (layer-0 columns 0,1,2,2025..2075) and all 47 mixer selectors; per byte the manager
    registers, the 54 byte contexts; per bit the 8 bit contexts."""
    g = load_golden(name, big)
    c = O.CtxModels(g["vocab"])
    if "pretrain" in g:  # Predictor::Pretrain = the same transitions for these models, outputs unused
        c.run(g["pretrain"].tobytes())
    stream = g["stream"]
    probs = mg.unpack_probs(g)[:, O.SMALL_COLS] if "probs_q" in g else g["small_probs"][:, :54]
    t = 0
    for n in range(len(stream)):
        for j in range(7, -1, -1):
            p, sel = c.predict()
            if probs is not None:
                bad = np.nonzero(~bits_equal(p, probs[t]))[0]
                assert len(bad) == 0, f"{name}: small model {bad[0]} (col {O.SMALL_COLS[bad[0]]}) differs at bit {t}"
            want = g["sel"][t].copy()
            want[12] = 0
            bad = np.nonzero(sel != want)[0]
            assert len(bad) == 0, f"{name}: selector {bad[0]} differs at bit {t}: {sel[bad[0]]} vs {want[bad[0]]}"
            assert (c.manager()[2] == g["bitctx"][t]).all(), f"{name}: bit contexts differ at bit {t}"
            c.perceive((int(stream[n]) >> j) & 1)
            t += 1
        regs, ctx, _ = c.manager()
        want = g["regs"][n + 1].copy()
        want[6] = 0  # auxiliary_context_ belongs to the mixing network
        assert (regs == want).all(), f"{name}: manager registers differ after byte {n}: {regs} vs {want}"
        bad = np.nonzero(ctx != g["ctx"][n + 1])[0]
        assert len(bad) == 0, f"{name}: context {bad[0]} differs after byte {n}"
        assert bits_equal(c.bracket_probs(), g["bracket_probs"][n + 1]).all(), f"{name}: Bracket dist after byte {n}"
    c.close()


def test_ctxmodels_text_golden():
    _check_ctxmodels("text_96")


def test_ctxmodels_binary_golden():
    _check_ctxmodels("binary_64")


def test_ctxmodels_2k_golden():
    _check_ctxmodels("text_2k_nofull")


def test_ctxmodels_brackets_golden():
    g = load_golden("brackets_1k")
    assert g["regs"][:, 5].max() >= 1  # longest_match_ reaches >= 1 (a Match model saw >= 32 matching bits)
    _check_ctxmodels("brackets_1k")


def test_ctxmodels_random_160k_local():
    _check_ctxmodels("random_160k", big=True)  # DirectHash evictions (20-probe reset), full hashed tables


def test_ctxmodels_pretrained_golden():
    _check_ctxmodels("pretrained_128")  # 300 dictionary bytes through Predictor::Pretrain first


def test_oracle_models_are_freed_by_their_scope():
    """orc_alloc.h: what a constructor allocates between new_owned() and release() is gone afterwards, blocks the oracle frees itself
    are not freed twice, and nothing outside a scope is touched."""
    import ctypes as C
    lib = O.lib()
    lib.orc_live_bytes.restype = C.c_size_t
    lib.orc_p8_mixer_new.restype = C.c_void_p
    keep = lib.orc_p8_mixer_new(8, 4, 1, 0)                   # outside any scope
    base = lib.orc_live_bytes()
    lib.orc_p8_predictor_new.restype = C.c_void_p
    lib.orc_p8_predictor_new.argtypes = [C.c_int]
    h, tag = O.new_owned(lib.orc_p8_predictor_new, 11)
    assert h and lib.orc_live_bytes() - base > 1 << 30        # the level-11 tables
    with O.scope():
        m = lib.orc_p8_mixer_new(8, 4, 1, 0)
        lib.orc_p8_mixer_free(m)                              # freed by the oracle itself inside the scope
        lib.orc_p8_mixer_new(8, 4, 1, 0)                      # left to the scope
    O.release(tag)
    assert lib.orc_live_bytes() == base
    lib.orc_p8_mixer_free(keep)
    assert lib.orc_live_bytes() < base
