"""GPU integration parity: the three device stages chained through HBM exactly as the engine runs them.

    bytes ──► ctx/small-model stage ──► layer-0 columns 0,1,2,2025..2075 + 47 selectors ─┐
    bytes ──► PPMd (host stage) ──► byte distributions ──► ByteModel bits ──► column 2076   ├─► mixing network ─► p
                                                      └──► LSTM byte mixer ──► column 2077  │
    fxcm / paq8 columns 3..2024 (trace of the unmodified reference) ───────────────────────┘

Only what has no stage yet (fxcm, paq8) is replayed from the golden trace of the reference; every other number
is produced by the engine (PPMd on a host core, the rest on the MI355X). The final probability must equal Predictor::Predict()'s float bit
for bit, for every coded bit (predictor.cpp:361-419)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, bits_equal, load_golden
import make_golden as mg

pytestmark = pytest.mark.gpu


def _pipeline(name, chunks=None, big=False):
    import torch
    from cmix_amd import engine as E
    g = load_golden(name, big)
    stream = np.ascontiguousarray(g["stream"])
    N = len(stream)
    ref = mg.unpack_probs(g)
    layer0 = torch.from_numpy(ref.copy()).cuda()
    own = E.SMALL_COLS + [2076, 2077]
    layer0[:, own] = float("nan")  # the device must produce these
    sel = torch.full((8 * N, 47), -1, dtype=torch.int32, device="cuda")
    d = torch.from_numpy(stream).cuda()
    host_ppmd = E.Ppmd(g["vocab"])
    pp = np.empty((N + 1, 256), np.float32)
    pp[0] = np.float32(1.0 / 256)  # ByteModel constructor (byte-model.cpp:5-6)
    pp[1:] = host_ppmd.run(stream.tobytes())
    host_ppmd.close()
    assert bits_equal(pp, g["ppmd_probs"]).all(), "host PPMd stage != reference"
    ppmd = torch.from_numpy(pp).cuda()  # [N+1,256]
    bits = torch.from_numpy(np.ascontiguousarray(g["bits"])).cuda()
    ctx, lstm, net = E.CtxModels(g["vocab"], 0), E.Lstm(g["vocab"], 0), E.MixNet(0)
    p = torch.empty(8 * N, dtype=torch.float32, device="cuda")
    edges = [0, N] if not chunks else sorted(set([0, N] + list(chunks)))
    for a, b in zip(edges[:-1], edges[1:]):
        rows = slice(8 * a, 8 * b)
        ctx.run(d[a:b], layer0[rows], sel[rows])
        E.bytemodel_bits(ppmd[a], ppmd[a + 1:b + 1], d[a:b], layer0[rows], 2076)
        lstm.run(ppmd[a + 1:b + 1], d[a:b], layer0=layer0[rows])
        net.run(layer0[rows], sel[rows], bits[rows], p[rows])
    torch.cuda.synchronize()
    ctx.sync()
    net.sync()
    got_l0 = layer0.cpu().numpy()
    bad = np.argwhere(~bits_equal(got_l0, ref))
    assert len(bad) == 0, f"{name}: layer-0 input {bad[0][1]} differs first at bit {bad[0][0]}"
    got = p.cpu().numpy()
    bad = np.nonzero(~bits_equal(got, g["p_final"]))[0]
    assert len(bad) == 0, f"{name}: final probability differs first at bit {bad[0]} of {8 * N}"
    for o in (ctx, lstm, net):
        o.close()


def test_pipeline_text_96():
    _pipeline("text_96")


def test_pipeline_binary_64_ragged():
    _pipeline("binary_64", chunks=[1, 7, 40])


def _native(name, chunks=None, big=False):
    """Same check through cmx_pipeline_* (the native C++ orchestration: PPMd host stage, three HIP streams, up to four
    chunks in flight): only the fxcm/paq8 columns come from the trace."""
    import torch
    from cmix_amd import engine as E
    g = load_golden(name, big)
    stream = np.ascontiguousarray(g["stream"])
    N = len(stream)
    ref = mg.unpack_probs(g)
    layer0 = torch.from_numpy(ref.copy()).cuda()
    layer0[:, E.SMALL_COLS + [2076, 2077]] = float("nan")
    p = torch.full((8 * N,), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    pipe = E.Pipeline(g["vocab"], 0, max_chunk_bytes=max(N, 1))
    edges = [0, N] if not chunks else sorted(set([0, N] + list(chunks)))
    for a, b in zip(edges[:-1], edges[1:]):
        pipe.submit(stream[a:b].tobytes(), layer0[8 * a:8 * b], p[8 * a:8 * b])
    pipe.sync()
    assert all(v > 0 for v in pipe.last_stage_ms().values())
    bad = np.argwhere(~bits_equal(layer0.cpu().numpy(), ref))
    assert len(bad) == 0, f"{name}: layer-0 input {bad[0][1]} differs first at bit {bad[0][0]}"
    bad = np.nonzero(~bits_equal(p.cpu().numpy(), g["p_final"]))[0]
    assert len(bad) == 0, f"{name}: final probability differs first at bit {bad[0]} of {8 * N}"
    pipe.close()


def test_native_pipeline_text_96_ragged():
    _native("text_96", chunks=[1, 2, 3, 10, 11, 50, 51])  # more chunks (8) than slots (4): buffer recycling


def test_native_pipeline_binary_64():
    _native("binary_64")


def test_native_pipeline_pretrained_128():
    """Predictor::Pretrain over 300 dictionary bytes, then 128 coded bytes: every column a stage owns (0-2,
    2025-2077) against the reference trace; pretraining after the first submit is refused."""
    import torch
    from cmix_amd import engine as E
    g = load_golden("pretrained_128")
    stream = np.ascontiguousarray(g["stream"])
    N = len(stream)
    cols = E.SMALL_COLS + [2076, 2077]
    layer0 = torch.full((8 * N, 2078), 0.5, dtype=torch.float32, device="cuda")
    p = torch.full((8 * N,), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    pipe = E.Pipeline(g["vocab"], 0, max_chunk_bytes=N)
    pipe.pretrain(g["pretrain"].tobytes())
    pipe.submit(stream[:50].tobytes(), layer0[:400], p[:400])
    with pytest.raises(E.CmxError, match="before the first submit"):
        pipe.pretrain(b"abc")
    pipe.submit(stream[50:].tobytes(), layer0[400:], p[400:])
    pipe.sync()
    got = layer0.cpu().numpy()
    bad = np.argwhere(~bits_equal(got[:, cols], g["small_probs"]))
    assert len(bad) == 0, f"column {cols[bad[0][1]]} differs first at bit {bad[0][0]}"
    assert (np.delete(got, cols, axis=1) == 0.5).all()
    pv = p.cpu().numpy()
    assert ((pv > 0) & (pv < 1)).all()
    pipe.close()


def test_compress_stream_reproduces_the_reference_file():
    """Device pipeline + host coder + header = the file `cmix -n` writes for the same payload, byte for byte (the
    fxcm/paq8 columns come from the trace of that very run); then decoded back with the same probabilities."""
    import torch
    from cmix_amd import engine as E
    from cmix_amd.pipeline import compress_stream
    g = load_golden("binary_64")
    with np.load(os.path.join(GOLDEN, "coder_vectors.npz")) as v:
        want = v["binary_64_file"].tobytes()
    stream = np.ascontiguousarray(g["stream"])
    layer0 = torch.from_numpy(mg.unpack_probs(g).copy()).cuda()
    layer0[:, E.SMALL_COLS + [2076, 2077]] = float("nan")
    got = compress_stream(stream, g["vocab"], layer0, chunk_bytes=24)
    assert got == want
    length, _, _, used = E.header_read(got)
    bits = E.Decoder(got[used:]).decode_bits(g["p_final"])
    assert length == len(stream) and (np.packbits(bits) == stream).all()


def test_native_pipeline_in_steps_with_host_columns():
    """cmx_pipeline_begin / _hints / _finish (look-ahead coding with fxcm/paq8 on the host): two chunks begun before
    the first is finished; the hints are the LSTM column of the trace (plus the first bit after the chunk), the
    fxcm/paq8 columns go in from HOST memory, and p equals Predictor::Predict() bit for bit."""
    import torch
    from cmix_amd import engine as E
    g = load_golden("text_96")
    stream = np.ascontiguousarray(g["stream"])
    N, A = len(stream), 40
    ref = mg.unpack_probs(g)
    layer0 = torch.full((8 * N, 2078), float("nan"), dtype=torch.float32, device="cuda")
    p = torch.full((8 * N,), -1.0, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    pipe = E.Pipeline(g["vocab"], 0, max_chunk_bytes=64)
    pipe.begin(stream[:A].tobytes(), layer0[:8 * A])
    pipe.begin(stream[A:].tobytes(), layer0[8 * A:])
    hp, hx = pipe.hints(A)
    assert bits_equal(hp, ref[:8 * A + 1, 2077]).all()  # entry 8A = first bit of the next chunk
    pipe.finish(ref[:8 * A, 3:2025], p[:8 * A])
    hp2, hx2 = pipe.hints(N - A)
    assert bits_equal(hp2[:-1], ref[8 * A:, 2077]).all()
    assert ((hx >= 0) & (hx < 256)).all() and ((hx2 >= 0) & (hx2 < 256)).all()
    pipe.finish(ref[8 * A:, 3:2025], p[8 * A:])
    pipe.sync()
    got = layer0.cpu().numpy()
    assert bits_equal(got, ref).all()
    assert bits_equal(p.cpu().numpy(), g["p_final"]).all()
    with pytest.raises(E.CmxError, match="no begun chunk"):
        pipe.finish(None, p[:8])
    with pytest.raises(E.CmxError, match="no begun chunk is waiting"):
        pipe.hints(1)
    pipe.close()


def test_native_pipeline_bad_args():
    from cmix_amd import engine as E
    pipe = E.Pipeline(np.ones(256, np.uint8), 0, max_chunk_bytes=16)
    assert E.lib().cmx_pipeline_submit(pipe.h, None, 4, None, None) != 0 and "bad argument" in E.last_error()
    pipe.close()


@pytest.mark.gpu
def test_engines_beyond_the_compute_units_are_refused():
    """A stream's stage kernels run for a whole chunk and wait for each other inside the launch, so their workgroups must be co-resident.
    The library keeps count per device (80 workgroups for the stages cmx_pipeline_create builds) and refuses, with the reason, the engine
    that would take the device past its compute units -- instead of a time-out in the middle of a stream. Destroying an engine gives its
    share back."""
    import torch
    from cmix_amd import engine as E
    cus = torch.cuda.get_device_properties(0).multi_processor_count
    room = cus // 80
    pipes = []
    try:
        for _ in range(room):
            pipes.append(E.Pipeline(np.ones(256, np.uint8), 0, max_chunk_bytes=16))
        with pytest.raises(E.CmxError) as ei:
            E.Pipeline(np.ones(256, np.uint8), 0, max_chunk_bytes=16)
        assert "compute units" in str(ei.value) and "workgroups" in str(ei.value)
        pipes.pop().close()
        pipes.append(E.Pipeline(np.ones(256, np.uint8), 0, max_chunk_bytes=16))   # the released share is available again
    finally:
        for p in pipes:
            p.close()
