"""CPU: the oracle's restatement of the context stage and of the final mixing network against the UNMODIFIED reference predictor with its three heavy
members replaced by constant stand-ins at link time (oracle/ref_ctx_trace.cpp, `make -C oracle traces`: predictor.o, the context manager, every context,
Direct / DirectHash / Indirect / Match / Bracket, PPMD, Mixer and SSE are the reference's own objects; PAQ8 / FXCM return 0.5, the Lstm a uniform
distribution). 16 KB of the bench shard: the 54 small-model columns and 46 of the 47 mixer selectors by digest (the auxiliary-context selector averages
stand-in columns), and every final probability of the oracle's mixing network fed the reference's own rows. The golden traces pin the same things on
2 KB of the FULL predictor; this pins them on a stream eight times longer, and is the tool round 5's long-stream diagnosis used."""
import os
import subprocess
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "oracle", "_ref", "ref_ctx_mixnet_trace")


def _splitmix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


@pytest.mark.skipif(not os.path.exists(EXE), reason="oracle/_ref/ref_ctx_mixnet_trace not built (make -C oracle traces; needs /root/reference)")
def test_context_stage_and_mixing_network_on_16k_against_the_references_objects():
    from cmix_amd import synth
    from cmix_amd.pipeline import text_file_stream
    from oracle import oracle as O
    stream = np.frombuffer(bytes(text_file_stream(synth.enwik_like(16384, 1000, rich=True))), np.uint8)
    vocab = np.zeros(256, np.uint8)
    vocab[np.unique(stream)] = 1
    with tempfile.TemporaryDirectory() as d:
        stream.tofile(os.path.join(d, "s.bin"))
        vocab.tofile(os.path.join(d, "v.bin"))
        r = subprocess.run([EXE, os.path.join(d, "s.bin"), os.path.join(d, "v.bin"), os.path.join(d, "out.txt")], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-400:]
        f = open(os.path.join(d, "out.txt")).read().split()
    assert int(f[0]) == len(stream)
    ref_cols, ref_sel = f[132:132 + 54], f[187:187 + 47]
    assert f[-4:] == ["mixnet_bits_differing_so_far", "0", "first", "-1"], "the oracle's mixing network left the reference's Mixer / SSE objects: " + " ".join(f[-4:]) + r.stderr[-300:]
    with np.errstate(over="ignore"):
        B = _splitmix64(np.uint64(0x1000000) + np.arange(1 << 19, dtype=np.uint64)) | np.uint64(1)
        orc = O.CtxModels(vocab)
        p, sel = orc.run(stream.tobytes())
        orc.close()
        T = len(p)
        hc = ((p.view(np.uint32).astype(np.uint64) + np.uint64(1)) * B[:T, None]).sum(0, dtype=np.uint64)
        hs = ((sel + np.uint64(1)) * B[:T, None]).sum(0, dtype=np.uint64)
    cols = [0, 1, 2] + list(range(2025, 2076))
    bad = [cols[i] for i in range(54) if "%016x" % int(hc[i]) != ref_cols[i]]
    assert not bad, "small-model columns that differ from the reference's: %s" % bad
    bads = [k for k in range(47) if k != 12 and "%016x" % int(hs[k]) != ref_sel[k]]
    assert not bads, "mixer selectors that differ from the reference's: %s" % bads
