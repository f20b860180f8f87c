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
    k = a[8:, :429] * np.float32(4095)                   # rows 0..7: the first byte's layout, 0.5 where no map has a context yet
    assert np.abs(k - np.round(k)).max() < 1e-3 and k.min() >= 0 and k.max() <= 4095.001 and (a[:, 429:] == 0.5).all()
    bits = np.unpackbits(data)
    p1 = np.clip(a[:, 428].astype(np.float64), 1 / 4096, 1 - 1 / 4096)   # the model's own final probability (last AddPrediction); an APM may return 0
    cost = -np.log2(np.where(bits == 1, p1, 1 - p1)).sum() / len(data)
    assert cost < 4.0, cost


# ---- the stage inside the chunk pipeline: cmx_pipeline_enable_fxcm --------------------------------------------------
# oracle/_ref/cmix_lookahead with CMX_FXCM_DEVICE=1: the reference's preprocessor and paq8 objects on the host, fxcm as a
# device stage fed by the LSTM stage's hints on the device (pretraining over the dictionary included). The files must
# equal the reference binary's byte for byte (tests/golden/dropin_vectors.npz).

def _lookahead(mode, files, timeout=900):
    import os
    import subprocess
    import tempfile
    from test_gpu_dropin import LOOKAHEAD
    if not os.path.exists(LOOKAHEAD):
        pytest.skip("oracle/_ref/cmix_lookahead not built (make -C oracle lookahead)")
    with tempfile.TemporaryDirectory() as d:
        paths = []
        for name, data in files:
            p = os.path.join(d, name)
            with open(p, "wb") as f:
                f.write(data)
            paths.append(p)
        out = os.path.join(d, "out")
        r = subprocess.run([LOOKAHEAD, mode] + paths + [out], stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout,
                           env=dict(os.environ, CMX_FXCM_DEVICE="1"))
        assert r.returncode == 0, f"cmix_lookahead {mode} (fxcm on the device) failed: {r.stderr.decode(errors='replace')[-400:]}"
        with open(out, "rb") as f:
            return f.read()


def test_pipeline_with_device_fxcm_files_are_byte_identical():
    from test_gpu_dropin import _lookahead_vectors
    v = _lookahead_vectors()
    assert _lookahead("-n", [("in", v["raw_n_payload"])]) == v["raw_n_file"]
    assert _lookahead("-c", [("in", v["text_c_payload"])]) == v["text_c_file"]
    assert _lookahead("-c", [("in", v["text12k_c_payload"])]) == v["text12k_c_file"]


def test_pipeline_with_device_fxcm_dictionary_pretraining():
    from test_gpu_dropin import _lookahead_vectors
    v = _lookahead_vectors()
    assert _lookahead("-c", [("dict", v["dict_payload"]), ("in", v["dict_c_payload"])]) == v["dict_c_file"]


@pytest.mark.parametrize("name", ["fxcm_cols_wiki_16k", "fxcm_cols_dict_16k", "fxcm_cols_mixed_24k", "fxcm_cols_rich_16k"])
def test_reference_hashes_16k(name):
    """The device stage against the reference itself (not against its twin, the oracle): 16 KB of wiki markup and of
    dictionary-mode text, per-bit hashes of all 431 values recorded from the unmodified fxcmv1::Predictor
    (tests/golden/make_fxcm_hashes.py)."""
    import os
    from test_fxcm_stage_host import _reference_fixture
    data, pr, ex, dic, want, row_hash = _reference_fixture(name)
    try:
        got = row_hash(run_device(data, pr, ex, [4096, 1, 4095, 8192, 8192], dictionary=dic))
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (name, "first differing bit", int(bad[0]))
    finally:
        if dic:
            os.unlink(dic.decode())
