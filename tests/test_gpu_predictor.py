"""GPU parity: the whole-predictor surface (cmx_create / cmx_predict / cmx_perceive / cmx_pretrain) in the
reference's strict per-bit protocol -- bit t+1's prediction is requested only after bit t has been perceived, as
a Decoder does -- against golden traces of the unmodified reference. Only the fxcm/paq8 columns are replayed from
the trace (handed in per bit); every other number is produced by the engine. Bit-exact."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits_equal, load_golden
import make_golden as mg

pytestmark = pytest.mark.gpu


def _walk(name, nbytes=None):
    from cmix_amd import engine as E
    g = load_golden(name)
    stream = g["stream"] if nbytes is None else g["stream"][:nbytes]
    ref = mg.unpack_probs(g)
    pr = E.Predictor(g["vocab"], 0)
    t = 0
    for byte in stream:
        for j in range(7, -1, -1):
            pr.set_model_outputs(ref[t, 3:2025])
            p = pr.Predict()
            assert bits_equal(p, g["p_final"][t]).all(), f"{name}: Predict() differs at bit {t}: {p} vs {g['p_final'][t]}"
            pr.Perceive((int(byte) >> j) & 1)
            if t + 1 < len(ref):  # lstmpr = Discretize(next LSTM bit prediction), predictor.cpp:180-182,463-464
                want = int(np.float32(1) + np.float32(4094) * ref[t + 1, 2077])
                assert pr.lstm_hint()[0] == want, f"{name}: lstmpr after bit {t}"
            t += 1
    pr.close()


def test_predictor_text_96():
    _walk("text_96")


def test_predictor_binary_64():
    _walk("binary_64")


def _walk_owned_columns(name, nbytes, pretrain=False):
    """Traces without the fxcm/paq8 columns: neutral stand-ins go in, and only the columns the engine owns
    (0-2, 2025-2077) are compared, read back from the row the last Predict() assembled."""
    from cmix_amd import engine as E
    g = load_golden(name)
    pr = E.Predictor(g["vocab"], 0)
    if pretrain:  # Predictor::Pretrain bit by bit, as preprocessor::Pretrain calls it (preprocessor.cpp:37-69)
        for byte in g["pretrain"]:
            for j in range(7, -1, -1):
                pr.Pretrain((int(byte) >> j) & 1)
    stand_in = np.full(2022, 0.5, np.float32)
    cols = E.SMALL_COLS + [2076, 2077]
    t = 0
    for byte in g["stream"][:nbytes]:
        for j in range(7, -1, -1):
            pr.set_model_outputs(stand_in)
            pr.Predict()
            row = np.ctypeslib.as_array(E.C.cast(E.lib().cmx_debug_last_row(pr.h), E.C.POINTER(E.C.c_float)), (2078,))
            bad = np.nonzero(~bits_equal(row[cols], g["small_probs"][t]))[0]
            assert len(bad) == 0, f"{name}: column {cols[bad[0]]} differs at bit {t}"
            pr.Perceive((int(byte) >> j) & 1)
            t += 1
    return pr


def test_predictor_brackets_prefix():
    # nested brackets: Bracket byte model + BracketContext in bit-synchronous form
    _walk_owned_columns("brackets_1k", 200).close()


def test_decode_the_reference_binarys_file():
    """Decompression: the file `cmix -n` wrote for binary_64's payload is decoded with cmx_decoder_* driving the
    per-bit surface -- no bit is known before its Predict() returned -- and yields the original stream."""
    from cmix_amd import engine as E
    g = load_golden("binary_64")
    with np.load(os.path.join(GOLDEN, "coder_vectors.npz")) as v:
        blob = v["binary_64_file"].tobytes()
    ref = mg.unpack_probs(g)
    length, _, vocab, used = E.header_read(blob)
    assert length == len(g["stream"])
    dec = E.Decoder(blob[used:])
    pr = E.Predictor(vocab, 0)
    out = bytearray()
    t = 0
    for _ in range(length):
        byte = 1
        while byte < 256:  # runner.cpp:127-131
            pr.set_model_outputs(ref[t, 3:2025])
            bit = dec.decode(pr.Predict())
            pr.Perceive(bit)
            byte += byte + bit
            t += 1
        out.append(byte & 255)
    assert bytes(out) == g["stream"].tobytes()
    pr.close()


def test_pretrain_then_code():
    from cmix_amd import engine as E
    pr = _walk_owned_columns("pretrained_128", 40, pretrain=True)
    with pytest.raises(E.CmxError, match="before the first predict"):
        pr.Pretrain(1)
    pr.close()


def test_protocol_errors():
    from cmix_amd import engine as E
    pr = E.Predictor(np.ones(256, np.uint8), 0)
    assert pr.mode() == (0, 0)   # nothing is built before the first call that needs device state
    # (a Predict() with neither staged input nor caller-supplied columns is no error since round 4: it is a decoder's first call and
    #  builds the late-bit form of the whole engine, mode 3 -- tests/test_gpu_late.py, tests/test_gpu_dropin.py::test_dropin_decodes_*)
    pr.set_model_outputs(np.full(2022, 0.5, np.float32))
    pr.Predict()
    with pytest.raises(E.CmxError, match="twice"):
        pr.Predict()
    pr.Perceive(1)
    with pytest.raises(E.CmxError, match="no pending"):
        pr.Perceive(0)
    assert pr.mode()[0] == 1
    with pytest.raises(E.CmxError, match="per-bit stages of this handle already exist"):
        pr.stage_input(b"abc")
    pr.close()


# ---- look-ahead mode of the same surface (cmx_stage_input): the chunk pipeline behind Predict() / Perceive() ----------

def _lookahead_walk(name, pretrain=False, split=None):
    """Stage the trace's bytes, then drive Predict / Perceive bit by bit as Encoder::Encode does (encoder.cpp:14-30): every
    returned float must equal the reference trace's final probability, with NO column handed in by the caller."""
    from cmix_amd import engine as E
    g = load_golden(name)
    pr = E.Predictor(g["vocab"], 0)
    if pretrain:
        for byte in g["pretrain"]:
            for j in range(7, -1, -1):
                pr.Pretrain((int(byte) >> j) & 1)
    data = g["stream"].tobytes()
    if split:   # staged in pieces, as a streaming caller would
        pr.stage_input(data[:split], end=False)
        pr.stage_input(data[split:], end=True)
    else:
        pr.stage_input(data)
    assert pr.mode()[0] == 2
    t = 0
    for byte in data:
        for j in range(7, -1, -1):
            p = pr.Predict()
            assert np.float32(p).view(np.uint32) == g["p_final"][t].view(np.uint32), f"{name}: p differs at bit {t}"
            pr.Perceive((byte >> j) & 1)
            t += 1
    return pr, g


def test_lookahead_mode_reproduces_the_reference_trace():
    from cmix_amd import engine as E
    pr, _ = _lookahead_walk("text_96", split=50)
    with pytest.raises(E.CmxError, match="no staged input is left"):
        pr.Predict()
    pr.close()


def test_lookahead_mode_after_pretrain():
    pr, _ = _lookahead_walk("pretrained_128", pretrain=True)
    pr.close()


def test_lookahead_mode_protocol_errors():
    from cmix_amd import engine as E
    g = load_golden("text_96")
    pr = E.Predictor(g["vocab"], 0)
    pr.stage_input(g["stream"].tobytes()[:8])
    with pytest.raises(E.CmxError, match="takes no columns"):
        pr.set_model_outputs(np.full(2022, 0.5, np.float32))
    with pytest.raises(E.CmxError, match="after the end-of-input mark"):
        pr.stage_input(b"x")
    pr.Predict()
    with pytest.raises(E.CmxError, match="twice"):
        pr.Predict()
    first = int(g["stream"][0]) >> 7
    with pytest.raises(E.CmxError, match="differs from the staged input"):
        pr.Perceive(1 - first)
    with pytest.raises(E.CmxError, match="failed part-way"):   # the device has learnt the staged bit: the handle is void
        pr.Predict()
    pr.close()
