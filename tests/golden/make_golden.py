#!/usr/bin/env python3
"""Generate golden traces from the UNMODIFIED reference (oracle/_ref/libcmixref.so).

Usage (dev container, where /root/reference exists and `make -C oracle ref` ran):
    python tests/golden/make_golden.py            # committed small fixtures
    python tests/golden/make_golden.py --big      # large local fixtures under oracle/_ref/golden/

Each trace records, per coded bit, what Predictor::Predict() saw and produced
(reference src/predictor.cpp:361-419) and, per byte, the byte-level state:
    bits        [T]       u8   coded bit (MSB first)
    probs_q     [T,2078]  u16  raw model outputs k where p == k*(1/4095) exactly (paq8/fxcm grid),
                               0xFFFF where the value is off-grid ...
    probs_off   [n_off]   f32  ... in which case the float is appended here in (t,i) order
    sel         [T,47]    u64  each mixer's selector (Mixer::context_)
    mix_out     [T,47]    f32  each mixer's own output (Mixer::p_)
    p_final     [T]       f32  value returned by Predict()
    regs        [N+1,25]  u64  ContextManager registers after each byte (row 0 = initial)
    ctx         [N+1,54]  u64  byte-level contexts after each byte
    bitctx      [T,8]     u64  bit-level contexts at Predict() time
    ppmd_probs  [N+1,256] f32  PPMd byte distribution after each byte
    lstm_probs  [N+1,256] f32  LSTM byte distribution after each byte
    bracket_probs [N+1,256] f32 Bracket model byte distribution after each byte
    small_probs [T,56]    f32  (traces without probs_q) layer-0 columns 0,1,2,2025..2077: the 54 small
                               native models, the PPMd and the LSTM bit predictions
The reference Predictor is one-per-process, so every trace is produced in a
fresh subprocess.
"""
import argparse
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

GRID = np.float32(1.0 / 4095)
SMALL_COLS = np.array([0, 1, 2] + list(range(2025, 2078)))


def text_block(payload: bytes) -> bytes:
    """What preprocessor::Encode writes for a TEXT block without dictionary
    (reference src/preprocess/preprocessor.cpp:536-537,443-449): type byte 4,
    big-endian length, WRT flag 0, then the bytes."""
    n = len(payload) + 1
    return bytes([4]) + n.to_bytes(4, "big") + b"\x00" + payload


def default_block(payload: bytes) -> bytes:
    """preprocessor::NoPreprocess (preprocessor.cpp:591-600): DEFAULT=0 header."""
    return bytes([0]) + len(payload).to_bytes(4, "big") + payload


def trace(stream: bytes, full=True, pretrain: bytes = b"", inject=None):
    """inject (tests/golden/make_wrap_traces.py): called with the fresh reference before the first bit -- places counters (state injection)."""
    from oracle import refharness as R
    r = R.Ref(R.vocab_of(stream))
    if inject:
        inject(r)
    for byte in pretrain:  # preprocessor::Pretrain (preprocessor.cpp:37-69): Predictor::Pretrain per bit, MSB first
        for j in range(7, -1, -1):
            r.pretrain((byte >> j) & 1)
    N = len(stream)
    T = 8 * N
    bits = np.empty(T, np.uint8)
    p_final = np.empty(T, np.float32)
    sel = np.empty((T, 47), np.uint64)
    mix_out = np.empty((T, 47), np.float32)
    bitctx = np.empty((T, r.n_bitctx), np.uint64)
    probs = np.empty((T, 2078), np.float32) if full else None
    regs = np.empty((N + 1, 25), np.uint64)
    ctx = np.empty((N + 1, r.n_ctx), np.uint64)
    ppmd = np.empty((N + 1, 256), np.float32)
    lstm = np.empty((N + 1, 256), np.float32)
    brk = np.empty((N + 1, 256), np.float32)
    small = None if full else np.empty((T, len(SMALL_COLS)), np.float32)
    regs[0], ctx[0], _ = r.manager()
    ppmd[0] = r.byte_probs(0)[0]
    lstm[0] = r.byte_probs(1)[0]
    brk[0] = r.byte_probs(2)[0]
    t = 0
    for n, byte in enumerate(stream):
        for j in range(7, -1, -1):
            bit = (byte >> j) & 1
            p_final[t] = r.predict()
            if full:
                probs[t] = r.model_probs()
            else:
                small[t] = r.model_probs()[SMALL_COLS]
            c0, o0 = r.mixers(0)
            c1, o1 = r.mixers(1)
            c2, o2 = r.mixers(2)
            sel[t] = np.concatenate([c0, c1, c2])
            mix_out[t] = np.concatenate([o0, o1, o2])
            bitctx[t] = r.manager()[2]
            r.perceive(bit)
            bits[t] = bit
            t += 1
        regs[n + 1], ctx[n + 1], _ = r.manager()
        ppmd[n + 1] = r.byte_probs(0)[0]
        lstm[n + 1] = r.byte_probs(1)[0]
        brk[n + 1] = r.byte_probs(2)[0]
    out = dict(stream=np.frombuffer(stream, np.uint8), bits=bits, p_final=p_final, sel=sel,
               mix_out=mix_out, bitctx=bitctx, regs=regs, ctx=ctx, ppmd_probs=ppmd,
               lstm_probs=lstm, bracket_probs=brk, vocab=R.vocab_of(stream), ctx_sizes=r.context_sizes())
    if not full:
        out["small_probs"] = small
    if pretrain:
        out["pretrain"] = np.frombuffer(pretrain, np.uint8)
    if full:
        q = np.rint(probs / GRID).astype(np.int64)
        on = (q >= 0) & (q <= 4095) & ((q.astype(np.float32) * GRID) == probs)
        pq = np.where(on, q, 0xFFFF).astype(np.uint16)
        out["probs_q"] = pq
        out["probs_off"] = probs[~on].astype(np.float32)
    return out


def unpack_probs(g):
    """Inverse of the probs_q/probs_off packing -> [T,2078] float32."""
    pq = g["probs_q"]
    probs = pq.astype(np.float32) * GRID
    off = pq == 0xFFFF
    probs[off] = g["probs_off"]
    return probs


def _child(kind, nbytes, seed, path, full):
    from cmix_amd import synth
    pre = b""
    if kind == "pretrained":  # a dictionary-like word list is pretrained, then text is coded
        words = synth.enwik_like(6000, seed + 1).split()
        pre = (b"\n".join(sorted(set(words))[:60]) + b"\n")[:300]
        kind = "text"
    if kind == "text":
        payload = synth.enwik_like(nbytes + 4096, seed)[4096:4096 + nbytes]
        stream = text_block(payload)
    elif kind == "random":  # incompressible bytes: fills the hashed tables fast (DirectHash evictions)
        stream = default_block(np.random.default_rng(seed).integers(0, 256, nbytes, dtype=np.uint8).tobytes())
    elif kind == "brackets":  # nested / unbalanced brackets and quotes, long repeats (Match length >= 32)
        rng = np.random.default_rng(seed)
        unit = bytes(rng.choice(np.frombuffer(b"(){}[]<>'\" ab\n", np.uint8), 97).tobytes())
        body = (unit * 3 + b"((((((((((((" + unit[:40] + b"))))" + bytes(300) + unit * 2)
        stream = text_block((body * (nbytes // len(body) + 1))[:nbytes])
    elif kind == "binary":
        rng = np.random.default_rng(seed)
        recs = rng.integers(0, 256, (nbytes // 16 + 1, 16), dtype=np.uint8)
        recs[:, :4] = np.arange(len(recs), dtype=np.uint32).view(np.uint8).reshape(-1, 4)
        stream = default_block(recs.tobytes()[:nbytes])
    else:
        raise ValueError(kind)
    g = trace(stream, full, pre)
    np.savez_compressed(path, **g)
    print("wrote", path, {k: v.shape for k, v in g.items()})


FIXTURES = [  # (name, kind, payload bytes, seed, full probs?)
    ("text_96", "text", 90, 1000, True),
    ("binary_64", "binary", 59, 7, True),
    ("text_2k_nofull", "text", 2042, 1001, False),
    ("brackets_1k", "brackets", 1018, 5, False),
    ("pretrained_128", "pretrained", 122, 21, False),
]
BIG = [
    ("text_4k", "text", 4090, 1000, True),
    ("text_32k", "text", 32762, 1002, True),
    ("random_160k", "random", 160000, 11, False),
]

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true")
    ap.add_argument("--child", nargs=5)
    a = ap.parse_args()
    if a.child:
        kind, nbytes, seed, path, full = a.child
        _child(kind, int(nbytes), int(seed), path, full == "1")
        sys.exit(0)
    outdir = os.path.join(ROOT, "oracle", "_ref", "golden") if a.big else os.path.dirname(os.path.abspath(__file__))
    os.makedirs(outdir, exist_ok=True)
    for name, kind, nbytes, seed, full in (BIG if a.big else FIXTURES):
        path = os.path.join(outdir, name + ".npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", kind,
                               str(nbytes), str(seed), path, "1" if full else "0"])
