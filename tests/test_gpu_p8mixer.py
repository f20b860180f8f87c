"""GPU parity: the paq8 two-layer int16 mixer kernel (building block of the paq8 stage; C ABI cmx_p8mixer_*) against
the oracle restatement (oracle/paq8_core.c, itself pinned against the reference's own Mixer class in
tests/test_oracle_paq8core.py) on the reference's real shape: 1552 inputs, 28 weight sets out of 77 472 rows,
second layer of 28. Integer work: bit-exact, every first-layer output and every final probability."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N, M, S, W = 1552, 77472, 28, 32


def _tables():
    from oracle import oracle as O
    lib = O.lib()
    sq = np.array([lib.orc_p8_squash(d) for d in range(-2048, 2048)], np.int16)
    st = np.array([lib.orc_p8_stretch(p) for p in range(4096)], np.int16)
    return sq, st


def test_chunks_match_the_oracle_and_time():
    import torch
    from cmix_amd import engine as E
    from oracle import oracle as O
    import test_oracle_paq8core as T8
    T = 6000
    ranges, xs, cx, rng = T8.mixer_case(2026, N, M, S, T)
    bits = (rng.random(T) < 0.5).astype(np.uint8)
    base = np.concatenate([[0], np.cumsum(ranges)[:-1]]).astype(np.int32)
    rows = (cx + base[None, :]).astype(np.int32)
    assert rows.max() < M
    # oracle: one step per bit, update with the previous bit
    lib = O.lib()
    h = lib.orc_p8_mixer_new(N, M, S, W)
    want = np.zeros(T, np.int32)
    exported = np.zeros(N + S + 8, np.float32)
    nexp = C.c_int(0)
    for t in range(T):
        want[t] = lib.orc_p8_mixer_step(h, int(bits[t - 1]) if t else 0, xs[t].ctypes.data, N, cx[t].ctypes.data,
                                        ranges.ctypes.data, S, exported.ctypes.data, C.byref(nexp))
    lib.orc_p8_mixer_free(h)
    sq, st = _tables()
    mix = E.P8Mixer(M, sq, st, 0)
    dx, dr, db = torch.from_numpy(xs).cuda(), torch.from_numpy(rows).cuda(), torch.from_numpy(bits).cuda()
    got = []
    for a, b in ((0, 1), (1, 1000), (1000, 1003), (1003, T)):  # ragged chunks: state carries across launches
        p, pr = mix.run(dx[a:b].contiguous(), dr[a:b].contiguous(), db[a:b].contiguous(), want_pr=True)
        got.append(p.cpu().numpy())
    got = np.concatenate(got)
    bad = np.nonzero(got != want)[0]
    assert len(bad) == 0, f"mixer output differs first at bit {bad[0]}: {got[bad[0]]} vs {want[bad[0]]}"
    assert len(np.unique(want)) > 200
    # timing of one 4096-bit chunk (state irrelevant): us per bit and algorithmic bytes (28 rows read + written per bit)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    mix.run(dx[:4096].contiguous(), dr[:4096].contiguous(), db[:4096].contiguous())
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 4096
    print(f"\np8mixer: {us:.2f} us/bit, {2 * S * N * 2 / us / 1e3:.1f} GB/s algorithmic (28 rows x 1552 i16 read + written per bit)")
    mix.close()
