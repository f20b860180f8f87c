"""The fxcm stage on the MI355X, through the C ABI (cmx_fxcm_create / _run): cmx_fxcm_chunk_kernel + the host text parser
against the oracle (oracle/fxcm_model.c) and against layer-0 columns 3..433 of the golden traces recorded from the
unmodified reference predictor -- all 431 values per bit, bit for bit. Same cases as tests/test_fxcm_stage_host.py, which
runs the kernel's body on the host. (Named to sort after the other GPU tests: the stage was written after this round's
GPU budget was spent, so its first run on a device is the driver's.)"""
import numpy as np
import pytest

from conftest import load_golden
from test_fxcm_stage_host import compare, hints, oracle_rows

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(900)]


def run_device(data, lstmpr, lstmex, chunks, dictionary=None):
    import torch
    from cmix_amd import engine as E
    fx = E.Fxcm(dictionary, 0)
    data = np.ascontiguousarray(data, np.uint8)
    outs, pos = [], 0
    for n in chunks:
        n = min(n, len(data) - pos)
        if n <= 0:
            break
        pr = torch.from_numpy(np.ascontiguousarray(lstmpr[8 * pos:8 * (pos + n)], np.int16)).cuda()
        ex = torch.from_numpy(np.ascontiguousarray(lstmex[8 * pos:8 * (pos + n)], np.uint8)).cuda()
        probs = torch.full((8 * n, 434), -1.0, dtype=torch.float32, device="cuda")
        fx.run(data[pos:pos + n], pr, ex, probs)
        fx.sync()
        got = probs.cpu().numpy()
        assert (got[:, :3] == -1.0).all()          # columns outside 3..433 are not the stage's
        outs.append(got[:, 3:434])
        pos += n
    fx.close()
    return np.ascontiguousarray(np.concatenate(outs))


def test_text_vs_oracle_ragged_chunks():
    from cmix_amd import synth
    data = np.frombuffer(synth.enwik_like(6000, 31), np.uint8)
    pr, ex = hints(8 * len(data), 7)
    compare(run_device(data, pr, ex, [1, 1, 7, 100, 1000, 3, 2000, 4000]), oracle_rows(data, pr, ex), "enwik-like text")


def test_binary_and_runs_vs_oracle():
    r = np.random.default_rng(5)
    runs = np.concatenate([np.full(int(n), int(v), np.uint8) for n, v in zip(r.integers(1, 40, 150), r.integers(0, 256, 150))])[:2000]
    data = np.concatenate([r.integers(0, 256, 2000).astype(np.uint8), runs])
    pr, ex = hints(8 * len(data), 11)
    compare(run_device(data, pr, ex, [512] * 8), oracle_rows(data, pr, ex), "binary + runs")


def test_golden_columns():
    import make_golden as mg
    from oracle import oracle as O
    g = load_golden("text_96")
    probs, bits, stream = mg.unpack_probs(g), g["bits"], g["stream"]
    l = O.Lstm(g["vocab"])
    pr, ex = np.zeros(len(bits), np.int16), np.zeros(len(bits), np.uint8)
    t = 0
    for n in range(len(stream)):
        for j in range(7, -1, -1):
            l.bit_perceive((int(stream[n]) >> j) & 1)
            if j == 0:
                l.byte_update(g["ppmd_probs"][n + 1], stream[n])
            if t + 1 < len(bits):
                pr[t] = int(np.float32(1) + np.float32(4094) * np.float32(l.bit_predict()))
                ex[t] = int(l.ex())
            else:
                pr[t] = 2048
            t += 1
    compare(run_device(np.asarray(stream, np.uint8), pr, ex, [len(stream)]), np.ascontiguousarray(probs[:, 3:434]), "text_96")


def test_long_text_properties():
    """64 KB of text at full table sizes: every value is on the k / 4095 grid the model exports, the final probability
    column tracks the data (coding cost well under 8 bits per byte), and two runs of the same stream are identical."""
    from cmix_amd import synth
    data = np.frombuffer(synth.enwik_like(65536, 77), np.uint8)
    pr, ex = hints(8 * len(data), 3)
    a = run_device(data, pr, ex, [4096] * 16)
    b = run_device(data, pr, ex, [65536])
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    k = a[:, :429] * np.float32(4095)
    assert np.abs(k - np.round(k)).max() < 1e-3 and k.min() >= 0.999 and k.max() <= 4095.001
    bits = np.unpackbits(data)
    p1 = a[:, 428].astype(np.float64)                    # the model's own final probability (last AddPrediction)
    cost = -np.log2(np.where(bits == 1, p1, 1 - p1)).sum() / len(data)
    assert cost < 4.0, cost
