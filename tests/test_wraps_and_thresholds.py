"""Counters, positions and thresholds that a stream only reaches after megabytes (or hundreds of them), pinned by STATE INJECTION: the unmodified
reference's counters are placed just below the threshold (oracle/ref_harness.cpp ref_debug_set_mixer_steps / ref_debug_set_history), its own code runs
across it, and the trace is the fixture (tests/golden/make_wrap_traces.py). The oracle (CPU) and the engine (GPU) are placed the same way through their
own hooks (orc_*_set_*, cmx_*_debug_set_*) and must reproduce every value. Round 5's 8 MiB defect was of this kind (a comparison of two 32-bit indices
that fails once, in the step in which a counter passes 2^32; tests/test_p8stage_host.py pins that one) -- DESIGN.md 5 has the audit table.

  wrap_mixsteps_2p32 / _12m / _2p24   Mixer::steps_ (mixer.cpp:58,61) across 2^32 (512 MB into a stream), 12 000 000 (the decay schedule's pow() argument
                                      crosses 2.0: the place where round 5's unexplained digest difference began) and 2^24 (float's last exact integer)
  wrap_history_100m                   ContextManager::history_pos_ across the 100 000 000-byte ring's end (context-manager.cpp:24-27) with Match models
                                      following matches that straddle it (match.cpp:43-56)
"""
import numpy as np
import pytest

from conftest import bits_equal, load_golden
import make_golden as mg

MIX_CASES = ["wrap_mixsteps_2p32", "wrap_mixsteps_12m", "wrap_mixsteps_2p24"]
COLS = np.array([0, 1, 2] + list(range(2025, 2076)))


@pytest.mark.parametrize("name", MIX_CASES)
def test_oracle_mixing_network_across_step_counter_thresholds(name):
    from oracle import oracle as O
    g = load_golden(name)
    probs = mg.unpack_probs(g)
    steps0 = int(g["inject_mixer_steps"][0])
    net = O.MixNet()
    net.set_steps(steps0)
    for t in range(len(g["bits"])):
        p, mix = net.step(probs[t], g["sel"][t], g["bits"][t], want_mix=True)
        assert bits_equal(mix, g["mix_out"][t]).all(), f"{name}: mixer outputs differ at bit {t} (steps_ = {steps0 + t})"
        assert bits_equal(p, g["p_final"][t]).all(), f"{name}: final p differs at bit {t}"
    net.close()


def test_the_injection_matters():
    """The fixtures would be worthless if the placed counter changed nothing: without the injection the oracle must leave the trace within a few bits."""
    from oracle import oracle as O
    g = load_golden("wrap_mixsteps_2p32")
    probs = mg.unpack_probs(g)
    net = O.MixNet()
    differs = False
    for t in range(64):
        p, mix = net.step(probs[t], g["sel"][t], g["bits"][t], want_mix=True)
        differs = differs or not bits_equal(mix, g["mix_out"][t]).all()
    net.close()
    assert differs


def _history_fixture():
    import make_wrap_traces as mw
    g = load_golden("wrap_history_100m")
    lo, hi = (int(x) for x in g["window"])
    return g, mw, 8 * lo, 8 * hi


def test_oracle_context_stage_across_the_history_ring_wrap():
    from oracle import oracle as O
    g, mw, lo, hi = _history_fixture()
    c = O.CtxModels(g["vocab"])
    c.set_history(int(g["inject_history_pos"][0]), g["inject_history_tail"].tobytes())
    probs, sel = c.run(g["stream"].tobytes())
    dig = mw.row_digest(probs, sel)
    bad = np.nonzero(dig != g["row_digest"])[0]
    if len(bad) and lo <= bad[0] < hi:
        t = bad[0]
        pb = np.nonzero(~bits_equal(probs[t], g["small_probs"][t - lo]))[0]
        raise AssertionError(f"bit {t} (byte {t // 8}): small models {pb} differ")
    assert len(bad) == 0, f"rows differ first at bit {bad[0]} (byte {bad[0] // 8}; the ring wraps in front of stream byte 1500)"
    assert bits_equal(probs[lo:hi], g["small_probs"]).all()
    regs = c.manager()[0]
    want = g["regs"][-1].copy()
    want[6] = 0
    assert (regs == want).all() and int(regs[3]) == len(g["stream"]) - 1500
    # the Match models did follow matches across the wrap: the longest-match register was set while the ring position was small again
    assert (g["regs"][1501:, 5] > 0).any()
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", MIX_CASES)
def test_gpu_mixing_network_across_step_counter_thresholds(name):
    import torch
    from cmix_amd import engine as E
    g = load_golden(name)
    probs = mg.unpack_probs(g)
    T = len(g["bits"])
    net = E.MixNet(0)
    net.debug_set_steps(int(g["inject_mixer_steps"][0]))
    d_probs = torch.from_numpy(probs).cuda()
    d_sel = torch.from_numpy((g["sel"] & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)).cuda()
    d_bits = torch.from_numpy(g["bits"]).cuda()
    p = torch.empty(T, dtype=torch.float32, device="cuda")
    mix = torch.empty((T, 47), dtype=torch.float32, device="cuda")
    for a, b in ((0, 200), (200, 301), (301, T)):   # the threshold falls at bit 300: inside a launch, and as a launch's last bit
        net.run(d_probs[a:b], d_sel[a:b], d_bits[a:b], p[a:b], mix[a:b])
    torch.cuda.synchronize()
    assert net.bits_done() == int(g["inject_mixer_steps"][0]) + T
    bad = np.argwhere(~bits_equal(mix.cpu().numpy(), g["mix_out"]))
    assert len(bad) == 0, f"{name}: first mixer mismatch (bit, mixer) = {bad[0]}"
    assert bits_equal(p.cpu().numpy(), g["p_final"]).all()
    net.close()


@pytest.mark.gpu
def test_gpu_context_stage_across_the_history_ring_wrap():
    import torch
    from cmix_amd import engine as E
    g, mw, lo, hi = _history_fixture()
    c = E.CtxModels(g["vocab"], 0)
    c.debug_set_history(int(g["inject_history_pos"][0]), g["inject_history_tail"].tobytes())
    data = np.ascontiguousarray(g["stream"])
    N = len(data)
    d = torch.from_numpy(data.copy()).cuda()
    probs = torch.full((8 * N, 2078), -1.0, dtype=torch.float32, device="cuda")
    sel = torch.full((8 * N, 47), -1, dtype=torch.int32, device="cuda")
    for a, b in ((0, 1000), (1000, 1500), (1500, 1501), (1501, N)):   # a chunk that ends exactly at the wrap (stream byte 1500 lands on ring position 0), one that is that byte alone
        c.run(d[a:b], probs[8 * a:8 * b], sel[8 * a:8 * b])
    c.sync()
    p = probs.cpu().numpy()[:, COLS]
    s = sel.cpu().numpy().view(np.uint32)
    dig = mw.row_digest(p, s)
    bad = np.nonzero(dig != g["row_digest"])[0]
    assert len(bad) == 0, f"rows differ first at bit {bad[0]} (byte {bad[0] // 8}; the ring wraps in front of stream byte 1500)"
    assert bits_equal(p[lo:hi], g["small_probs"]).all()
    regs = c.manager()[0]
    want = g["regs"][-1].copy()
    want[6] = 0
    assert (regs == want).all()
    c.close()
