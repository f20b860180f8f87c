"""GPU parity: HIP context / small-model stage (through the C ABI) vs golden traces of the unmodified
reference and vs the oracle on seeded inputs. Bit-exact: 54 model outputs per bit (layer-0 columns
0,1,2,2025..2075), all 47 selectors per bit (as the 32-bit keys the mixers use), manager registers and
the 54 byte contexts at chunk ends."""
import numpy as np
import pytest

from conftest import bits_equal, load_golden
import make_golden as mg

pytestmark = pytest.mark.gpu

COLS = np.array([0, 1, 2] + list(range(2025, 2076)))


def _run_gpu(vocab, data, chunks=None, want_mgr=False, pretrain=None):
    import torch
    from cmix_amd import engine as E
    c = E.CtxModels(vocab, 0)
    if pretrain is not None:
        c.pretrain(torch.from_numpy(np.ascontiguousarray(pretrain)).cuda())
    data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
    N = len(data)
    d = torch.from_numpy(data.copy()).cuda()
    probs = torch.full((8 * N, 2078), -1.0, dtype=torch.float32, device="cuda")
    sel = torch.full((8 * N, 47), -1, dtype=torch.int32, device="cuda")
    edges = [0, N] if not chunks else sorted(set([0, N] + list(chunks)))
    mgr = []
    for a, b in zip(edges[:-1], edges[1:]):
        c.run(d[a:b], probs[8 * a:8 * b], sel[8 * a:8 * b])
        if want_mgr:
            mgr.append((b, c.manager()))
    c.sync()
    p = probs.cpu().numpy()
    s = sel.cpu().numpy().view(np.uint32)
    assert (np.delete(p, COLS, axis=1) == -1.0).all(), "stage wrote outside its columns"
    c.close()
    return p[:, COLS], s, mgr


def _check_golden(name, chunks=None, big=False):
    g = load_golden(name, big)
    stream = g["stream"]
    want_p = mg.unpack_probs(g)[:, COLS] if "probs_q" in g else g["small_probs"][:, :54]
    p, s, mgr = _run_gpu(g["vocab"], stream, chunks, want_mgr=True, pretrain=g.get("pretrain"))
    bad = np.argwhere(~bits_equal(p, want_p))
    assert len(bad) == 0, f"{name}: model {bad[0][1]} (col {COLS[bad[0][1]]}) differs first at bit {bad[0][0]}"
    want_s = (g["sel"] & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    want_s[:, 12] = 0
    bad = np.argwhere(s != want_s)
    assert len(bad) == 0, f"{name}: selector {bad[0][1]} differs first at bit {bad[0][0]}"
    for n, (regs, ctx, bctx) in mgr:
        want = g["regs"][n].copy()
        want[6] = 0
        assert (regs == want).all(), f"{name}: manager registers after byte {n}: {regs} vs {want}"
        assert (ctx == g["ctx"][n]).all(), f"{name}: contexts after byte {n}"
        if n < len(stream):
            assert (bctx == g["bitctx"][8 * n]).all(), f"{name}: bit contexts after byte {n}"


def test_golden_text_96():
    _check_golden("text_96")


def test_golden_binary_64():
    _check_golden("binary_64")


def test_golden_brackets_1k_ragged_chunks():
    _check_golden("brackets_1k", chunks=[1, 2, 9, 100, 101, 640])


def test_golden_pretrained_128():
    _check_golden("pretrained_128", chunks=[50])  # Predictor::Pretrain over 300 dictionary bytes first


def test_golden_text_2k():
    _check_golden("text_2k_nofull", chunks=[1000])


def test_vs_oracle_random_160k():
    """Incompressible bytes fill the hashed tables: the 100000-row DirectHash runs out of free rows within
    its 20 probes and re-initialises rows (direct-hash.cpp:38-42). The oracle is pinned on this case against
    the reference by tests/test_oracle_golden.py::test_ctxmodels_random_160k_local."""
    from oracle import oracle as O
    data = np.random.default_rng(11).integers(0, 256, 160000, dtype=np.uint8).tobytes()
    vocab = np.ones(256, np.uint8)
    orc = O.CtxModels(vocab)
    want_p, want_s = orc.run(data)
    oregs, octx, _ = orc.manager()
    p, s, mgr = _run_gpu(vocab, data, chunks=[70001], want_mgr=True)
    bad = np.argwhere(~bits_equal(p, want_p))
    assert len(bad) == 0, f"model {bad[0][1]} (col {COLS[bad[0][1]]}) differs first at bit {bad[0][0]}"
    bad = np.argwhere(s != (want_s & np.uint64(0xFFFFFFFF)).astype(np.uint32))
    assert len(bad) == 0, f"selector {bad[0][1]} differs first at bit {bad[0][0]}"
    regs, ctx, _ = mgr[-1][1]
    assert (regs == oregs).all() and (ctx == octx).all()


def test_vs_oracle_seeded_32k_mixed():
    """Seeded mixed input (text, runs, random bytes, nested brackets): long matches, DirectHash probing,
    WRT contexts (bytes >= 0x80), line breaks; chunked raggedly; compared with the C oracle."""
    from oracle import oracle as O
    from cmix_amd import synth
    rng = np.random.default_rng(123)
    text = synth.enwik_like(16000, 77)
    parts = [text[:6000], bytes(rng.integers(0, 256, 3000, dtype=np.uint8)), text[2000:9000], b"\n" * 50,
             b"((([[[{{{<<<" * 40, text[:6000], bytes(rng.integers(128, 256, 2000, dtype=np.uint8)), b"a" * 1500,
             text[9000:14000]]
    data = b"".join(parts)[:32768]
    vocab = np.zeros(256, np.uint8)
    vocab[np.unique(np.frombuffer(data, np.uint8))] = 1
    orc = O.CtxModels(vocab)
    want_p, want_s = orc.run(data)
    oregs, octx, _ = orc.manager()
    p, s, mgr = _run_gpu(vocab, data, chunks=[4096, 4097, 20000], want_mgr=True)
    bad = np.argwhere(~bits_equal(p, want_p))
    assert len(bad) == 0, f"model {bad[0][1]} (col {COLS[bad[0][1]]}) differs first at bit {bad[0][0]}"
    ws = (want_s & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    bad = np.argwhere(s != ws)
    assert len(bad) == 0, f"selector {bad[0][1]} differs first at bit {bad[0][0]}"
    regs, ctx, _ = mgr[-1][1]
    assert (regs == oregs).all() and (ctx == octx).all()
    assert want_s[:, 10].max() >= 2  # longest_match_ exercised


def test_empty_chunk_and_bad_args():
    import torch
    from cmix_amd import engine as E
    c = E.CtxModels(np.ones(256, np.uint8), 0)
    d = torch.zeros(0, dtype=torch.uint8, device="cuda")
    c.run(d)  # no-op
    c.sync()
    d8 = torch.zeros(8, dtype=torch.uint8, device="cuda")
    rc = E.lib().cmx_ctxmodels_run(c.h, d8.data_ptr(), 8, None, 2078, None, None)  # null output pointers
    assert rc != 0 and "bad argument" in E.last_error()
    c.close()


def test_peek_is_exact_and_leaves_no_trace():
    """Bit-synchronous mode: a dry pass over byte n (cmx_ctxmodels_peek) followed by the committing run of byte n,
    for 96 000 random bytes, against a second instance that only ever ran in chunk mode. Random bytes make two
    Indirect models' windows of the shared map overlap now and then (about once per 12 000 bytes), which is the one place where a dry
    pass has to write to HBM and roll back -- the test requires that path to have been taken."""
    import torch
    from cmix_amd import engine as E
    rng = np.random.default_rng(2024)
    N, BLK = 96000, 4000
    data = rng.integers(0, 256, N, dtype=np.uint8)
    vocab = np.ones(256, np.uint8)
    a, b = E.CtxModels(vocab, 0), E.CtxModels(vocab, 0)
    d = torch.from_numpy(data).cuda()
    cols = torch.tensor(list(COLS), device="cuda")
    peek_p = torch.empty((8 * BLK, 2078), dtype=torch.float32, device="cuda")
    peek_s = torch.empty((8 * BLK, 47), dtype=torch.int32, device="cuda")
    run_p, run_s = torch.empty_like(peek_p), torch.empty_like(peek_s)
    for blk in range(0, N, BLK):
        want_p, want_s = b.run(d[blk:blk + BLK])
        for i in range(BLK):
            n = blk + i
            a.peek(d[n:n + 1], peek_p[8 * i:8 * i + 8], peek_s[8 * i:8 * i + 8])
            a.run(d[n:n + 1], run_p[8 * i:8 * i + 8], run_s[8 * i:8 * i + 8])
        torch.cuda.synchronize()
        for name, got_p, got_s in (("peek", peek_p, peek_s), ("run after peek", run_p, run_s)):
            assert torch.equal(got_p[:, cols].view(torch.int32), want_p[:, cols].view(torch.int32)), f"{name}: outputs, block {blk}"
            assert torch.equal(got_s, want_s), f"{name}: selectors, block {blk}"
    ra, rb = a.manager(), b.manager()
    assert all((x == y).all() for x, y in zip(ra, rb))
    committed, dry = a.slow_bytes()
    assert committed == b.slow_bytes()[0] and dry == committed and committed >= 3, (committed, dry)
    a.close()
    b.close()


def test_vs_oracle_fuzz_flavours():
    """The 8 adversarial stream flavours of tests/golden/fuzz_vs_reference.py (high bytes, 99+ character lines, bracket
    stacks past every limit, long exact repeats, byte runs, random, tiny alphabet, text) x 2 seeds, ragged chunks,
    against the C oracle -- which that script pins against the reference on the very same streams."""
    from oracle import oracle as O
    import fuzz_vs_reference as fz
    for seed in range(16):
        data, name = fz.make_stream(seed)
        vocab = np.ones(256, np.uint8)
        orc = O.CtxModels(vocab)
        want_p, want_s = orc.run(data)
        p, s, _ = _run_gpu(vocab, data, chunks=[1, 7, 300, 301])
        bad = np.argwhere(~bits_equal(p, want_p))
        assert len(bad) == 0, f"seed {seed} ({name}): model {bad[0][1]} (col {COLS[bad[0][1]]}) differs first at bit {bad[0][0]}"
        ws = (want_s & np.uint64(0xFFFFFFFF)).astype(np.uint32)
        bad = np.argwhere(s != ws)
        assert len(bad) == 0, f"seed {seed} ({name}): selector {bad[0][1]} differs first at bit {bad[0][0]}"
