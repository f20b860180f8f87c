#!/usr/bin/env python3
"""Golden traces of the UNMODIFIED reference across counter thresholds that a stream only reaches after hundreds of megabytes -- by state
injection (oracle/ref_harness.cpp: ref_debug_set_mixer_steps / ref_debug_set_history place the counters; everything that then runs is the
reference's own code). Round 5 found a defect that only showed when a 32-bit counter wrapped 8 MB into a stream; these fixtures pin the
engine (and the oracle) at the other thresholds of that kind without a run of that length:

    wrap_mixsteps_2p32   Mixer::steps_ (mixer.cpp:58,61; unsigned long long) passes 2^32 -- 512 MB into a stream, inside BASELINE config 3
    wrap_mixsteps_12m    ... passes 12 000 000: the decay schedule's pow() argument 1e-7 * steps + 0.8 crosses 2.0 (1.5 MB into a stream)
    wrap_mixsteps_2p24   ... passes 2^24: the last integer a float holds exactly (2 MB into a stream; the reference converts to double)
    wrap_history_100m    ContextManager::history_pos_ wraps at 100 000 000 (context-manager.cpp:24-27) while the Match models' own
                         counters run on (match.cpp:43-56: map_ holds them as 32-bit values, cur_match_ is reduced modulo the ring):
                         enwik8 crosses it by a few bytes, config 3 ten times

    python tests/golden/make_wrap_traces.py       # ~2 CPU-minutes; one subprocess per trace (one reference Predictor per process)
"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
HERE = os.path.dirname(os.path.abspath(__file__))

HISTORY_POS = 100000000 - 1500   # the ring's write position when the traced stream begins
HISTORY_TAIL = 2048              # bytes of "earlier stream" written to the ring in front of it
CASES = {
    "wrap_mixsteps_2p32": ("steps", (1 << 32) - 300),
    "wrap_mixsteps_12m": ("steps", 12000000 - 300),
    "wrap_mixsteps_2p24": ("steps", (1 << 24) - 300),
    "wrap_history_100m": ("history", HISTORY_POS),
}


WINDOW = (1506 - 150, 1506 + 250)   # stream bytes whose rows are kept in full (stream byte 1500 -- payload byte 1494 behind the block header -- lands on ring position 0)


def _splitmix64(x):
    x = x + np.uint64(0x9E3779B97F4A7C15)
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def row_digest(p54, sel):
    """One 64-bit digest per bit over the 54 small-model columns (as bit patterns) and the 47 selectors (as the 32-bit keys the mixers use; the
    auxiliary-context selector, which reads other stages' columns, left out): sum of (value + 1) * A[k] mod 2^64."""
    with np.errstate(over="ignore"):
        A = _splitmix64(np.arange(101, dtype=np.uint64)) | np.uint64(1)
        v = np.ascontiguousarray(p54, np.float32).view(np.uint32).astype(np.uint64) + np.uint64(1)
        s = (np.asarray(sel).astype(np.uint64) & np.uint64(0xFFFFFFFF)) + np.uint64(1)
        s[:, 12] = 0
        return (v * A[None, :54]).sum(1) + (s * A[None, 54:]).sum(1)


def history_payload():
    """3000 bytes of text block whose second half repeats phrases of the first (and of the injected tail), so that the Match models follow
    matches whose positions lie in front of the wrap while the ring's write position passes it."""
    from cmix_amd import synth
    base = synth.enwik_like(HISTORY_TAIL + 1400, 77, rich=True)
    tail, head = base[:HISTORY_TAIL], base[HISTORY_TAIL:HISTORY_TAIL + 1400]
    body = head + head[200:900] + tail[-600:] + head[:300]
    return tail, body[:3000]


def _child(name):
    import make_golden as mg
    kind, value = CASES[name]
    if kind == "steps":
        from cmix_amd import synth
        stream = mg.text_block(synth.enwik_like(4096 + 74, 1000)[4096:4096 + 74])   # 80 bytes = 640 bits: the counter passes the threshold at bit 300
        g = mg.trace(stream, True, inject=lambda r: r.set_mixer_steps(value))
        g["inject_mixer_steps"] = np.array([value], np.uint64)
    else:
        tail, body = history_payload()
        stream = mg.text_block(body)
        g = mg.trace(stream, False, inject=lambda r: r.set_history(value, tail))
        g["inject_history_pos"] = np.array([value], np.uint64)
        g["inject_history_tail"] = np.frombuffer(tail, np.uint8)
        for k in ("mix_out", "p_final", "ppmd_probs", "lstm_probs", "bracket_probs", "bitctx"):   # (not what this fixture pins: the context stage's columns, selectors, registers)
            g.pop(k)
        # every bit by digest, the rows themselves only around the wrap (the fixture stays under 1 MB)
        g["row_digest"] = row_digest(g["small_probs"][:, :54], g["sel"])
        lo, hi = 8 * WINDOW[0], 8 * WINDOW[1]
        g["window"] = np.array(WINDOW, np.int64)
        g["small_probs"] = g["small_probs"][lo:hi, :54].copy()
        g["sel"] = g["sel"][lo:hi].copy()
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        _child(sys.argv[2])
        sys.exit(0)
    for name in (sys.argv[1:] or CASES):
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", name])
