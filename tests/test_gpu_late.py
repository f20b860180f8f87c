"""GPU parity of the DECODER's form of the engine: the late-bit protocol (cmix_amd/csrc/cmx_late.h; include/cmix_amd.h section 4).

Decoder::Decode (reference src/coder/decoder.cpp:20-39) knows bit t only after Predictor::Predict() has returned p(t). The late
pipeline runs every model family on the device -- the same stage kernels a compressor's chunks use -- with the bits arriving one
at a time. These tests REPLAY the bits of golden traces of the unmodified reference through it (no arithmetic coder involved):
p(t) must equal Predictor::Predict()'s float bit for bit, and on the full traces all 2078 layer-0 inputs of every bit as well
(predictor.cpp:361-419); then real round trips: a file coded by the look-ahead engine is decoded by a fresh handle that is told
nothing but the bits it decodes itself.

Each case runs in its own process: the late pipeline needs every stage kernel of the stream running at the same time (14 HIP
streams on hardware queues of their own), which a process that has already created and destroyed other handles cannot promise.
"""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, bits_equal, load_golden

pytestmark = pytest.mark.gpu

_CHILD = r'''
import sys, os
os.environ["CMX_LATE_DEBUG"] = "1"   # the mixing network also writes its 47 Mixer::Mix values per bit (test hook)
import numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests")); sys.path.insert(0, os.path.join({root!r}, "tests", "golden"))
from conftest import bits_equal
import make_golden as mg
from cmix_amd import engine as E
name, rows = {name!r}, {rows!r}
with np.load(os.path.join({golden!r}, name + ".npz")) as z:
    g = {{k: z[k] for k in z.files}}
pipe = E.Pipeline(g["vocab"], 0, 4096)
pipe.enable_fxcm(None)
pipe.enable_paq8()
last = 0
if "pretrain" in g:
    pre = bytes(np.ascontiguousarray(g["pretrain"], np.uint8))
    pipe.pretrain(pre)
    last = pre[-1] & 1
pipe.late_start(last)
bits = np.ascontiguousarray(g["bits"], np.uint8)
ref_rows = mg.unpack_probs(g) if rows and "probs_q" in g else None
ref_sel = (g["sel"] & np.uint64(0xFFFFFFFF)).astype(np.uint32) if "sel" in g else None
ref_mix = np.ascontiguousarray(g["mix_out"], np.float32) if "mix_out" in g else None
pf = np.ascontiguousarray(g["p_final"], np.float32)
bad = None
for t in range(len(bits)):
    p = np.float32(pipe.late_predict())
    if ref_rows is not None:
        row, sel = pipe.late_row()
        d = np.nonzero(~bits_equal(row, ref_rows[t]))[0]
        if len(d):
            bad = "layer-0 input %d differs at bit %d: %r vs reference %r (%d inputs differ)" % (d[0], t, row[d[0]], ref_rows[t][d[0]], len(d))
            break
        if ref_sel is not None:
            d = np.nonzero((sel != ref_sel[t]) & (np.arange(47) != 12))[0]   # (12: auxiliary_context_, formed inside the mixing network from three of the inputs)
            if len(d):
                bad = "selector %d differs at bit %d: %r vs reference %r (%d differ)" % (d[0], t, sel[d[0]], ref_sel[t][d[0]], len(d))
                break
    if np.float32(p).view(np.uint32) != pf[t].view(np.uint32):
        bad = "p differs at bit %d: %r vs reference %r" % (t, p, pf[t])
        mix = pipe.late_mix() if ref_mix is not None and t > 0 else None   # diagnostic (of bit t - 1: the waves write them when they learn; read while they run)
        if mix is not None:
            d = np.nonzero(~bits_equal(mix, ref_mix[t - 1]))[0]
            bad += "; Mixer::Mix values of bit %d that differ from the reference's: mixers %s" % (t - 1, list(d[:16]))
        break
    pipe.late_perceive(int(bits[t]))
ms, n = pipe.late_host_ms()
pipe.late_stop()
pipe.close()
if bad:
    print("MISMATCH", bad); sys.exit(3)
print("OK bits", n, "host ms", {{k: round(v, 1) for k, v in ms.items()}}, "us/byte", round(1000 * sum(ms.values()) / max(n / 8, 1), 1))
'''


def _run_child(code, timeout=600):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout)
    sys.stdout.write(r.stdout[-3000:])
    assert r.returncode == 0, (r.stdout[-2000:] + "\n" + r.stderr[-3000:])
    return r.stdout


def _replay(name, rows=True):
    load_golden(name)  # skips when the fixture is absent
    return _run_child(_CHILD.format(root=ROOT, golden=GOLDEN, name=name, rows=rows))


def test_late_replay_text_96_all_inputs():
    _replay("text_96")


def test_late_replay_binary_64_all_inputs():
    _replay("binary_64")


def test_late_replay_pretrained_128():
    """Predictor::Pretrain over dictionary bytes through the chunk-mode stages, then the stream bit by bit"""
    _replay("pretrained_128")


def test_late_replay_brackets_1k_two_chunks():
    """1024 bytes = two late chunks of 512: the hand-over of every stage's state and distributions at the chunk boundary"""
    _replay("brackets_1k", rows=False)


def test_late_replay_text_2k_four_chunks():
    out = _replay("text_2k_nofull", rows=False)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "late_replay_text_2k.txt"), "w") as f:
        f.write(out)


_ROUNDTRIP = r'''
import sys, os, time
import numpy as np
sys.path.insert(0, {root!r})
from cmix_amd import engine as E, synth
payload = {payload}
vocab = np.zeros(256, np.uint8); vocab[list(set(payload))] = 1
enc_p = E.Predictor(vocab, 0)
enc_p.stage_input(payload)
enc = E.Encoder()
for B in payload:
    for j in range(8):
        bit = (B >> (7 - j)) & 1
        p = enc_p.Predict()
        enc.encode_bits(np.array([p], np.float32), np.array([bit], np.uint8))
        enc_p.Perceive(bit)
enc.flush()
code = enc.data()
enc_p.close()
t0 = time.time()
dec_p = E.Predictor(vocab, 0)          # nothing staged, no columns: the decoder's mode
dec = E.Decoder(code)
out = bytearray()
for i in range(len(payload)):
    B = 0
    for j in range(8):
        bit = dec.decode(dec_p.Predict())
        dec_p.Perceive(bit)
        B = B * 2 + bit
    out.append(B)
    if out[-1] != payload[i]:
        print("MISMATCH at byte", i); sys.exit(3)
assert dec_p.mode()[0] == 3
dt = time.time() - t0
dec_p.close()
print("OK", len(payload), "bytes ->", len(code), "decoded in", round(dt, 2), "s =", round(1e6 * dt / len(payload), 1), "us/byte (incl. construction)")
'''


def test_late_round_trip_text():
    """look-ahead engine codes 1500 bytes of rich text; a fresh handle decodes the file from its bits alone"""
    _run_child(_ROUNDTRIP.format(root=ROOT, payload="bytes(synth.enwik_like(1500, 77, rich=True))"))


def test_late_round_trip_binary():
    _run_child(_ROUNDTRIP.format(root=ROOT, payload="bytes(np.random.default_rng(5).integers(0, 256, 700, dtype=np.uint8))"))


_DECODE_STREAM = r'''
import sys, os, time
import numpy as np
sys.path.insert(0, {root!r})
from cmix_amd import engine as E, synth
payload = bytes(synth.enwik_like(1400, 4321, rich=True))
vocab = np.zeros(256, np.uint8); vocab[list(set(payload))] = 1
enc_p = E.Predictor(vocab, 0)
enc_p.stage_input(payload)
enc = E.Encoder()
for B in payload:
    for j in range(8):
        bit = (B >> (7 - j)) & 1
        enc.encode_bits(np.array([enc_p.Predict()], np.float32), np.array([bit], np.uint8))
        enc_p.Perceive(bit)
enc.flush()
code = enc.data()
enc_p.close()
dec_p = E.Predictor(vocab, 0)
t0 = time.time()
out = dec_p.decode_stream(code, len(payload))
dt = time.time() - t0
assert dec_p.mode()[0] == 3
dec_p.close()
assert out == payload, "cmx_decode_stream returned other bytes"
print("cmx_decode_stream: %d bytes in %.2f s (incl. the decoder's start-up) = %.0f us/byte" % (len(payload), dt, 1e6 * dt / len(payload)))
'''


def test_decode_stream_round_trip_inside_the_library():
    """cmx_decode_stream (round 6): a stream the look-ahead engine coded, decoded from its arithmetic code alone by a fresh handle -- Decoder::Decode
    (decoder.cpp:20-39) and the Decompress loop inside the library over the decoder's form of the engine (every model family a device stage; the LSTM's
    forward block as ONE launch per truncated-BPTT block, fourteen of which this stream crosses, and three decoder's chunks of 512 bytes)."""
    out = _run_child(_DECODE_STREAM.format(root=ROOT))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "decode_stream_time.txt"), "a") as f:
        f.write(out)
