"""GPU parity: HIP LSTM byte-mixer stage (through the C ABI) vs the oracle and vs golden traces of
the unmodified reference. Bit-exact on every float (256-way distributions and per-bit predictions)."""
import numpy as np
import pytest

from conftest import bits_equal, load_golden
import make_golden as mg

pytestmark = pytest.mark.gpu


def _run_gpu(vocab, in_probs, data, chunks=None):
    import torch
    from cmix_amd import engine as E
    l = E.Lstm(vocab, 0)
    N = len(data)
    d_in = torch.from_numpy(np.ascontiguousarray(in_probs, np.float32)).cuda()
    d_b = torch.from_numpy(np.ascontiguousarray(data, np.uint8)).cuda()
    edges = [0, N] if not chunks else sorted(set([0, N] + list(chunks)))
    outs, bps, bxs = [], [], []
    for a, b in zip(edges[:-1], edges[1:]):
        o, bp, bx = l.run(d_in[a:b], d_b[a:b])
        outs.append(o); bps.append(bp); bxs.append(bx)
    torch.cuda.synchronize()
    res = (torch.cat(outs).cpu().numpy(), torch.cat(bps).cpu().numpy(), torch.cat(bxs).cpu().numpy())
    l.close()
    return res


def _check_golden(name, nbytes=None, chunks=None):
    g = load_golden(name)
    N = len(g["stream"]) if nbytes is None else nbytes
    out, bp, bx = _run_gpu(g["vocab"], g["ppmd_probs"][1:N + 1], g["stream"][:N], chunks)
    bad = np.nonzero(~bits_equal(out, g["lstm_probs"][1:N + 1]).all(axis=1))[0]
    assert len(bad) == 0, f"{name}: LSTM distribution differs first after byte {bad[0]}"
    if "probs_q" in g:
        want = mg.unpack_probs(g)[:8 * N, 2077].reshape(N, 8)
        badb = np.argwhere(~bits_equal(bp, want))
        assert len(badb) == 0, f"{name}: bit prediction differs first at (byte, bit) {badb[0]}"


def test_initial_weights_match_oracle():
    from cmix_amd import engine as E
    from oracle import oracle as O
    g = load_golden("text_96")
    dev = E.Lstm(g["vocab"], 0)
    orc = O.Lstm(g["vocab"])
    for layer in range(2):
        for gate in range(3):
            assert bits_equal(dev.gate_weights(layer, gate), orc.gate_weights(layer, gate)).all()
    dev.close()


def test_golden_text_96():
    _check_golden("text_96")


def test_golden_binary_64():
    _check_golden("binary_64")


def test_golden_2k_with_bptt_rounds():
    _check_golden("text_2k_nofull", nbytes=450)   # BPTT + Adam at bytes 0, 100, 200, 300, 400


def test_golden_ragged_chunks_across_bptt_boundary():
    _check_golden("text_2k_nofull", nbytes=230, chunks=[1, 2, 99, 100, 101, 199, 205])


def test_tolerance_mode_mfma_weight_update_deviation():
    """TOLERANCE mode (cmx_lstm_set_tolerance; NOT bit-exact): the BPTT round's weight-update contraction -- per gate a 200 x rowlen x 100 product of
    the round's error signals and layer inputs (lstm-layer.cpp:182-186) -- as v_mfma_f32_16x16x4_f32 tiles (fused products, the epochs summed
    upwards) instead of the reference's ordered chain of separately rounded operations. Measured on the MI355X after five rounds (profiles/r04_lstm_mfma_tolerance.txt):
    max |dp| = 2.1e-6 over the 256-way output distributions, 5.8e-6 relative on the gate weights (Adam divides the deviating sums by the square root of
    their own second moment: a last-place difference of an f32 sum becomes a few 1e-6 of a weight) -- north_star's 1e-6 is NOT met on the weights, and the
    test says so with the bounds that do hold: 1e-5 on the distributions, 2e-5 on the weights. Strict mode is untouched."""
    import torch
    from cmix_amd import engine as E
    g = load_golden("text_2k_nofull")
    N = 450   # BPTT + Adam at bytes 0, 100, 200, 300, 400
    res, wts = [], []
    for tol in (False, True):
        l = E.Lstm(g["vocab"], 0)
        if tol:
            l.set_tolerance(True)
        d_in = torch.from_numpy(np.ascontiguousarray(g["ppmd_probs"][1:N + 1], np.float32)).cuda()
        d_b = torch.from_numpy(np.ascontiguousarray(g["stream"][:N], np.uint8)).cuda()
        o, _, _ = l.run(d_in, d_b)
        torch.cuda.synchronize()
        res.append(o.cpu().numpy())
        wts.append([l.gate_weights(layer, gate).copy() for layer in range(2) for gate in range(3)])
        l.close()
    dp = float(np.abs(res[0].astype(np.float64) - res[1]).max())
    dw = max(float(np.abs(a.astype(np.float64) - b).max() / np.abs(a).max()) for a, b in zip(*wts))
    same = float((res[0].view(np.uint32) == res[1].view(np.uint32)).mean())
    print("LSTM tolerance mode after 5 BPTT rounds: max |dp| = %.3g over the 256-way distributions (%.1f %% of the values bit-identical), max relative gate-weight deviation %.3g" % (dp, 100 * same, dw))
    assert bits_equal(res[0][:100], res[1][:100]).all()   # the first update lands after byte 100's round: before it nothing differs
    assert 0 < dw < 2e-5 and dp < 1e-5


def test_small_vocabulary_vs_oracle():
    """V = 3 (ragged rows, tiny softmax) and a vocabulary-complete random distribution stream."""
    from oracle import oracle as O
    rng = np.random.default_rng(2)
    vocab = np.zeros(256, np.uint8)
    vocab[[10, 65, 200]] = 1
    N = 130
    data = rng.choice([10, 65, 200], N).astype(np.uint8)
    probs = np.zeros((N, 256), np.float32)
    probs[:, [10, 65, 200]] = rng.dirichlet([1, 1, 1], N).astype(np.float32)
    orc = O.Lstm(vocab)
    want, wantb = [], []
    for n in range(N):
        for j in range(7, -1, -1):
            wantb.append(orc.bit_predict())
            orc.bit_perceive((int(data[n]) >> j) & 1)
        want.append(orc.byte_update(probs[n], data[n]))
    out, bp, _ = _run_gpu(vocab, probs, data)
    assert bits_equal(out, np.array(want)).all()
    assert bits_equal(bp.reshape(-1), np.array(wantb, np.float32)).all()


def test_bytemodel_bits_matches_oracle_ex():
    """`ex` (arg-max symbol inside the current interval, byte-model.cpp:13-20) feeds fxcm as lstmex."""
    from oracle import oracle as O
    g = load_golden("text_96")
    N = 40
    out, bp, bx = _run_gpu(g["vocab"], g["ppmd_probs"][1:N + 1], g["stream"][:N])
    orc = O.Lstm(g["vocab"])
    for n in range(N):
        for j in range(8):
            orc.bit_predict()
            assert orc.ex() == bx[n, j], (n, j)
            orc.bit_perceive((int(g["stream"][n]) >> (7 - j)) & 1)
        orc.byte_update(g["ppmd_probs"][n + 1], g["stream"][n])


def test_330k_bytes_past_the_adam_step_limit():
    """330 000 bytes = 3300 BPTT/Adam rounds on the device: LstmLayer::update_steps_ saturates at 3000 (byte 300 000)
    and the bias terms take the double-precision pow() path (lstm-layer.cpp:26-30). The stream is regenerated from
    its seed, the host PPMd stage supplies the input distributions (and is itself checked at the same bytes), and
    the LSTM's output distribution is compared, bit for bit, with checkpoints of the reference trace on both sides
    of the switch (tests/golden/make_lstm_long_checkpoints.py)."""
    import torch
    from cmix_amd import engine as E
    import make_lstm_long_checkpoints as mk
    g = load_golden("lstm_330k_checkpoints")
    stream = mk.stream_330k()
    N = len(stream)
    ppmd = E.Ppmd(g["vocab"])
    lstm = E.Lstm(g["vocab"], 0)
    want = {int(n): i for i, n in enumerate(g["at"])}
    BLK = 30000
    for a in range(0, N, BLK):
        b = min(N, a + BLK)
        pp = ppmd.run(stream[a:b].tobytes())
        for n in range(a, b):
            if n in want:
                assert bits_equal(pp[n - a], g["ppmd_probs"][want[n]]).all(), f"PPMd distribution after byte {n}"
        out, _, _ = lstm.run(torch.from_numpy(pp).cuda(), torch.from_numpy(stream[a:b].copy()).cuda(), want_bits=False)
        torch.cuda.synchronize()
        rows = [n for n in want if a <= n < b]
        if rows:
            got = out[torch.tensor([n - a for n in rows], device="cuda")].cpu().numpy()
            for k, n in enumerate(rows):
                assert bits_equal(got[k], g["lstm_probs"][want[n]]).all(), f"LSTM distribution after byte {n}"
    ppmd.close()
    lstm.close()
