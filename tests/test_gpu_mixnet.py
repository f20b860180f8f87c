"""GPU parity: the HIP mixing-network stage (through the C ABI) against the oracle
and against golden traces of the unmodified reference. Bit-exact (f32 bit patterns)."""
import numpy as np
import pytest

from conftest import bits_equal, load_golden, synth_mixnet_inputs
import make_golden as mg

pytestmark = pytest.mark.gpu


def _gpu_run(probs, sel, bits, chunks=None, tolerance=False):
    import torch
    from cmix_amd import engine as E
    net = E.MixNet(0)
    if tolerance:
        net.set_tolerance(True)
    T = len(bits)
    d_probs = torch.from_numpy(probs).cuda()
    d_sel = torch.from_numpy((sel & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)).cuda()
    d_bits = torch.from_numpy(bits).cuda()
    p = torch.empty(T, dtype=torch.float32, device="cuda")
    mix = torch.empty((T, 47), dtype=torch.float32, device="cuda")
    edges = [0, T] if not chunks else sorted(set([0, T] + list(chunks)))
    for a, b in zip(edges[:-1], edges[1:]):
        net.run(d_probs[a:b], d_sel[a:b], d_bits[a:b], p[a:b], mix[a:b])
    torch.cuda.synchronize()
    assert net.bits_done() == T
    out = p.cpu().numpy(), mix.cpu().numpy()
    net.close()
    return out


def _check_golden(name, big=False, chunks=None):
    g = load_golden(name, big)
    probs = mg.unpack_probs(g)
    p, mix = _gpu_run(probs, g["sel"], g["bits"], chunks)
    bad_mix = np.argwhere(~bits_equal(mix, g["mix_out"]))
    assert len(bad_mix) == 0, f"{name}: first mixer mismatch (bit, mixer) = {bad_mix[0]}"
    bad = np.nonzero(~bits_equal(p, g["p_final"]))[0]
    assert len(bad) == 0, f"{name}: final p differs first at bit {bad[0]}: {p[bad[0]]} vs {g['p_final'][bad[0]]}"


def test_golden_text_96():
    _check_golden("text_96")


def test_golden_binary_64():
    _check_golden("binary_64")


def test_golden_text_96_ragged_chunks():
    # the stream may be cut anywhere, including mid-byte and 1-bit chunks
    _check_golden("text_96", chunks=[1, 2, 9, 64, 65, 300, 511])


def test_vs_oracle_synthetic_long():
    """3000 bits with few distinct contexts so rows pass 1024 steps (periodic weight decay)."""
    from oracle import oracle as O
    T = 3000
    probs, sel, bits = synth_mixnet_inputs(T, seed=11, n_ctx_bits=1)
    net = O.MixNet()
    ref = net.run(probs, sel, bits)
    p, _ = _gpu_run(probs, sel, bits)
    bad = np.nonzero(~bits_equal(p, ref))[0]
    assert len(bad) == 0, f"first mismatch at bit {bad[0]}"


def test_row_cap_overflow_row():
    """More than 10000 distinct selector keys on some mixers: shared 0xDEADBEEF row (mixer.cpp:16-36)."""
    from oracle import oracle as O
    T = 12000
    probs, sel, bits = synth_mixnet_inputs(T, seed=5)
    sel[:, 11] = np.arange(T, dtype=np.uint64) * np.uint64(2654435761)  # all distinct, 64-bit wide
    sel[:, 35] = np.arange(T, dtype=np.uint64) + np.uint64(1 << 33)     # truncation to 32 bits matters
    net = O.MixNet()
    ref = net.run(probs, sel, bits)
    p, _ = _gpu_run(probs, sel, bits)
    bad = np.nonzero(~bits_equal(p, ref))[0]
    assert len(bad) == 0, f"first mismatch at bit {bad[0]}"


def test_lstm_override_and_extreme_inputs():
    """p == 0 / 1 from the byte mixer overrides the output (predictor.cpp:383,415-417);
    inputs outside [1e-4, 1-1e-4] are clamped (mixer-input.cpp:11-15)."""
    from oracle import oracle as O
    T = 256
    probs, sel, bits = synth_mixnet_inputs(T, seed=3)
    probs[10, 2077] = 0.0
    probs[20, 2077] = 1.0
    probs[30, :100] = 0.0
    probs[31, :100] = 1.0
    net = O.MixNet()
    ref = net.run(probs, sel, bits)
    assert ref[10] == 0.0 and ref[20] == 1.0
    p, _ = _gpu_run(probs, sel, bits)
    assert bits_equal(p, ref).all()


def test_bit_synchronous_equals_chunk_mode():
    """Predict()/Perceive() one bit at a time (decoder protocol) == look-ahead chunk mode."""
    from cmix_amd import engine as E
    g = load_golden("text_96")
    probs = mg.unpack_probs(g)
    T = 200
    net = E.MixNet(0)
    sel32 = (g["sel"] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    out = np.empty(T, np.float32)
    for t in range(T):
        out[t] = net.predict(probs[t], sel32[t])
        net.perceive(int(g["bits"][t]))
    net.close()
    assert bits_equal(out, g["p_final"][:T]).all()


def test_protocol_errors():
    from cmix_amd import engine as E
    net = E.MixNet(0)
    with pytest.raises(E.CmxError):
        net.perceive(0)  # no pending predict
    net.predict(np.full(2078, 0.5, np.float32), np.zeros(47, np.uint32))
    with pytest.raises(E.CmxError):
        net.predict(np.full(2078, 0.5, np.float32), np.zeros(47, np.uint32))
    net.perceive(1)
    net.close()
    assert E.lib().cmx_create(None, None, 0) is None  # the whole-predictor surface: tests/test_gpu_predictor.py
    assert "null vocab" in E.last_error()


def test_tolerance_mode_stays_within_its_tolerance(monkeypatch):
    """Tolerance mode (opt-in through cmx_mixnet_set_tolerance -- an API switch, never an environment variable; not bit-exact): the
    layer-0 dot products as f64 tree sums rounded once instead of the reference's 2078 sequentially rounded f32 adds.
    What the mode changes is the 26 layer-0 sums: over the first 64 bits of a stream (before the two weight histories have had time
    to drift apart through the learning feedback) every sum agrees with strict mode's to 1e-5 of its scale (measured on the MI355X:
    4.0e-6 -- the accumulated rounding of 2078 sequential f32 adds, which the tree sum does not have), i.e. the layer-0 mixers'
    probabilities squash(sum) to 2e-6: north_star's "within 1e-6" holds for the probabilities to that scale, not to the letter. The final probability comes out of integer SSE tables (steps of 1/32766 = 3.05e-5), so a last-bit
    difference of a sum either vanishes or shows as whole steps: its deviation is REPORTED, with a loose guard that it neither builds
    up nor runs away (measured on the MI355X, profiles/r03_mixnet_tolerance_mode.txt: identical on the reference's own trace, 99.6 %
    identical and at most ten steps on a 3000-bit synthetic trace with rows past 1024 steps). The environment variable of round 3
    must no longer switch anything."""
    g = load_golden("text_96")
    probs = mg.unpack_probs(g)
    p_tol, m_tol = _gpu_run(probs, g["sel"], g["bits"], tolerance=True)
    T = 3000
    sp, ss, sb = synth_mixnet_inputs(T, seed=11, n_ctx_bits=1)
    q_tol, n_tol = _gpu_run(sp, ss, sb, tolerance=True)
    monkeypatch.setenv("CMX_MIXNET_TOLERANCE", "1")   # ignored since round 4
    p_strict, m_strict = _gpu_run(probs, g["sel"], g["bits"])
    monkeypatch.delenv("CMX_MIXNET_TOLERANCE")
    q_strict, n_strict = _gpu_run(sp, ss, sb)
    assert bits_equal(p_strict, g["p_final"]).all()
    for name, a, b, ma, mb in (("text_96", p_tol, p_strict, m_tol, m_strict), ("synthetic 3000 bits, rows past 1024 steps", q_tol, q_strict, n_tol, n_strict)):
        d = np.abs(a.astype(np.float64) - b.astype(np.float64))
        s0 = np.abs(ma[:64, :26].astype(np.float64) - mb[:64, :26].astype(np.float64)) / np.maximum(1.0, np.abs(mb[:64, :26].astype(np.float64)))
        sall = np.abs(ma[:, :26].astype(np.float64) - mb[:, :26].astype(np.float64)) / np.maximum(1.0, np.abs(mb[:, :26].astype(np.float64)))
        print("tolerance mode, %s: layer-0 sums, first 64 bits: max rel dev %.3g; all %d bits: max %.3g; final p: max |dp| = %.3g, mean %.3g; %d of %d bit-identical"
              % (name, s0.max(), len(d), sall.max(), d.max(), d.mean(), int(bits_equal(a, b).sum()), len(d)))
        sg = lambda v: 1.0 / (1.0 + np.exp(-v.astype(np.float64)))
        pg = np.abs(sg(ma[:64, :26]) - sg(mb[:64, :26]))
        print("    layer-0 mixer probabilities squash(sum), first 64 bits: max |dp| = %.3g" % pg.max())
        assert s0.max() <= 1e-5 and pg.max() <= 2e-6, (name, s0.max(), pg.max())
        assert d.max() < 1e-3 and bits_equal(a, b).mean() > 0.99, (name, d.max(), bits_equal(a, b).mean())


@pytest.mark.parametrize("switch", ["CMX_MIXNET_XCD=7", "CMX_MIXNET_XCD=2", "CMX_MIXNET_JITTER=3", "CMX_MIXNET_JITTER=9", "CMX_MIXNET_ROTATE=1"])
def test_kernel_switches_are_bit_exact(monkeypatch, switch):
    """Run-time forms of the speculative kernel: the one-XCD placement (all workgroups on one XCD, the hand-off words in its L2; an XCD number below and
    above 4 -- the flag of a diagnostic once shared a bit with that number's field, which sent XCD 4..7 down the wrong branch: profiles/r05_xcd_fault.txt)
    and the jitter hook (pseudo-random stalls in every role but the gather wave: every lead / lag the roles can have towards each other). Same ordered
    f32 sums by construction: the same bits as the default kernel and as the oracle, over two launches. (The slower forms of the helper chain that round 5
    measured -- eight / sixteen segments, 128 candidates, re-runs in pieces, padded or sleepy hand-off words -- left the product: scripts/study/.)"""
    import torch
    from cmix_amd import engine as E
    from oracle import oracle as O
    T = 3072
    probs, sel, bits = synth_mixnet_inputs(T, seed=2)
    ref = O.MixNet().run(probs[:1024], sel[:1024], bits[:1024])
    d_probs = torch.from_numpy(probs).cuda()
    d_sel = torch.from_numpy((sel & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)).cuda()
    d_bits = torch.from_numpy(bits).cuda()
    out = {}
    for on in (False, True):
        name, val = switch.split("=")
        if on:
            monkeypatch.setenv(name, val)
        else:
            monkeypatch.delenv(name, raising=False)
        net = E.MixNet(0)
        p = torch.empty(T, dtype=torch.float32, device="cuda")
        mix = torch.empty((T, 47), dtype=torch.float32, device="cuda")
        for a, b in ((0, 700), (700, T)):
            net.run(d_probs[a:b], d_sel[a:b], d_bits[a:b], p[a:b], mix[a:b])
            torch.cuda.synchronize()
        out[on] = (p.cpu().numpy(), mix.cpu().numpy())
        net.close()
    assert np.array_equal(out[False][0][:1024].view(np.uint32), ref.view(np.uint32)), "default kernel != oracle"
    assert np.array_equal(out[True][0].view(np.uint32), out[False][0].view(np.uint32)), switch + ": final p differs"
    assert np.array_equal(out[True][1].view(np.uint32), out[False][1].view(np.uint32)), switch + ": a mixer output differs"


def test_bit_exact_under_foreign_load_and_random_lead(monkeypatch):
    """Round 5 saw, ONCE, the final probabilities of a stream leave the reference's while torch kernels shared the device (DESIGN.md 5 "Long streams");
    round 6 repeated that run three times without reproducing it (profiles/r06_foreign_load.txt) and then saw it once more, 33.9 MB into a stream
    (profiles/r06_two_runs_50m.txt; the network given identical inputs is excluded by scripts/gpu_mixnet_vote.py). What a disturbance could change in the
    network's kernel is pinned here for good: the mixing network's 27 workgroups hand values to each other through counters and value|tag words only, so neither foreign kernels on the device
    (memory-bound copies, LDS-heavy sorts and scans, hundreds of tiny launches, allocation churn: scripts/gpu_foreign_load.py) nor any lead / lag between
    its roles (CMX_MIXNET_JITTER: pseudo-random stalls of up to 100 us in every role) may change a single bit. 16 launches of 1024 bits under each
    disturbance against a clean run of the same inputs, and the first 2048 bits against the oracle."""
    import os
    import sys
    import torch
    from cmix_amd import engine as E
    from oracle import oracle as O
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "scripts"))
    from gpu_foreign_load import Foreign
    T, C = 16384, 1024
    probs, sel, bits = synth_mixnet_inputs(T, seed=11)
    ref = O.MixNet().run(probs[:2048], sel[:2048], bits[:2048])
    dev = torch.device("cuda", 0)
    d_probs = torch.from_numpy(probs).to(dev)
    d_sel = torch.from_numpy((sel & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)).to(dev)
    d_bits = torch.from_numpy(bits).to(dev)

    def run(load, jitter):
        if jitter:
            monkeypatch.setenv("CMX_MIXNET_JITTER", str(jitter))
        else:
            monkeypatch.delenv("CMX_MIXNET_JITTER", raising=False)
        F = Foreign(load, dev) if load else None
        net = E.MixNet(0)
        p = torch.zeros(T, dtype=torch.float32, device=dev)
        mix = torch.zeros((T, 47), dtype=torch.float32, device=dev)
        for lo in range(0, T, C):
            net.run(d_probs[lo:lo + C], d_sel[lo:lo + C], d_bits[lo:lo + C], p[lo:lo + C], mix[lo:lo + C])
            if F:
                F.step()
                F.step()
        torch.cuda.synchronize()
        net.close()
        return p.cpu().numpy(), mix.cpu().numpy()

    p0, m0 = run(None, 0)
    assert np.array_equal(p0[:2048].view(np.uint32), ref.view(np.uint32)), "clean run != oracle"
    for load, jitter in (("all", 0), (None, 5), ("all", 12), ("digest", 7)):
        p1, m1 = run(load, jitter)
        bad = np.argwhere(m1.view(np.uint32) != m0.view(np.uint32))
        assert bad.size == 0, "load %s, jitter %d: first differing (bit, mixer) %s" % (load, jitter, bad[0])
        assert np.array_equal(p1.view(np.uint32), p0.view(np.uint32)), "load %s, jitter %d: final p differs" % (load, jitter)
