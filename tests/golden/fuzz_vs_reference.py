#!/usr/bin/env python3
"""Fuzz the CPU restatements (oracle/*.c), the host PPMd stage and the host coder against the UNMODIFIED reference
on seeded streams of adversarial flavours (not a pytest: ~1 min of reference time per stream, dev container only).
Every stream is traced through the reference Predictor bit by bit (tests/golden/make_golden.trace, one process per
stream) and the same checks the golden tests run are applied: mixing network + SSE (all 47 mixer outputs, final p),
contexts + 54 small models + 47 selectors + manager registers, LSTM byte mixer, PPMd byte model, coder.

    python tests/golden/fuzz_vs_reference.py [first_seed] [count]      # writes tests/golden/fuzz_log.txt
"""
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
OUT = "/tmp/cmx_fuzz"


def make_stream(seed):
    """Flavours aimed at paths the fixed fixtures touch lightly."""
    import make_golden as mg
    from cmix_amd import synth
    rng = np.random.default_rng(seed)
    n = int(rng.integers(700, 1500))
    kind = seed % 8
    text = synth.enwik_like(4000, seed)
    if kind == 0:    # WRT-style high bytes: runs of >= 0x80 (wrt_context_, interval maps), mixed with text
        parts = []
        while sum(map(len, parts)) < n:
            parts.append(text[len(parts) * 37:len(parts) * 37 + int(rng.integers(3, 40))])
            parts.append(bytes(rng.integers(0x80, 0x100, int(rng.integers(1, 6)), dtype=np.uint8)))
        body, name = b"".join(parts)[:n], "highbytes"
    elif kind == 1:  # very long lines (line_break_ saturates at 99) and empty lines
        body, name = (text.replace(b"\n", b" ")[:n // 2] + b"\n\n\n" + text[:n // 2]), "longlines"
    elif kind == 2:  # bracket stacks past every limit: 40 deep, distance > 255, unbalanced closers
        body = (b"(" * 40 + text[:300] + b")" * 50 + b"[{<" * 30 + bytes(300) + b">}]" * 5 + text[300:])[:n]
        name = "deepbrackets"
    elif kind == 3:  # long exact repeats (Match length saturation at 255 bits ... longest_match_ 7) then a break
        unit = text[:211]
        body, name = (unit * 6 + b"#" + unit * 2)[:n], "repeats"
    elif kind == 4:  # runs of one byte and of two alternating bytes (run_map states, Direct counts at limit)
        body, name = (b"a" * 400 + b"ab" * 200 + bytes(200) + b"\xff" * 200 + text)[:n], "runs"
    elif kind == 5:  # uniformly random bytes
        body, name = rng.integers(0, 256, n, dtype=np.uint8).tobytes(), "random"
    elif kind == 6:  # tiny alphabet (V small for the LSTM), digits and punctuation only
        body, name = bytes(rng.choice(np.frombuffer(b"0123456789.,;\n", np.uint8), n).tobytes()), "digits"
    else:            # plain text with a different seed
        body, name = text[:n], "text"
    block = mg.text_block(body) if kind in (0, 1, 2, 3, 7) else mg.default_block(body)
    return block, name


def child(seed):
    import make_golden as mg
    stream, name = make_stream(seed)
    g = mg.trace(stream, True)
    np.savez(os.path.join(OUT, f"fuzz_{seed}.npz"), **g)


def check(seed):
    import conftest
    conftest.GOLDEN_BIG = OUT
    conftest.GOLDEN = OUT
    import test_oracle_golden as T
    import test_ppmd_host as TP
    from oracle import oracle as O
    from oracle import refharness as R
    from cmix_amd import engine as E
    name = f"fuzz_{seed}"
    T.load_golden = lambda n, big=False: conftest.load_golden(n, True)
    TP.load_golden = T.load_golden
    T._check(name, big=True)            # mixing network + SSE
    T._check_ctxmodels(name, big=True)  # contexts, small models, selectors
    T._check_lstm(name)                 # LSTM byte mixer
    TP._check(name, big=True)           # host PPMd stage
    g = conftest.load_golden(name, True)
    code = R.ref_encode(g["p_final"], g["bits"])
    e = E.Encoder(); e.encode_bits(g["p_final"], g["bits"]); e.flush()
    assert e.data() == code == O.coder_encode(g["p_final"], g["bits"])
    return len(g["stream"])


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--child":
        os.makedirs(OUT, exist_ok=True)
        child(int(sys.argv[2]))
        sys.exit(0)
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    os.makedirs(OUT, exist_ok=True)
    log = open(os.path.join(ROOT, "tests", "golden", "fuzz_log.txt"), "a")
    for seed in range(first, first + count):
        t0 = time.time()
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", str(seed)])
        _, name = make_stream(seed)
        try:
            n = check(seed)
            line = f"seed {seed:3d} {name:12s} {n:5d} bytes: mixnet, ctxmodels, lstm, ppmd, coder bit-exact ({time.time() - t0:.0f} s)"
        except AssertionError as ex:
            line = f"seed {seed:3d} {name:12s} MISMATCH: {ex}"
        print(line, flush=True)
        log.write(line + "\n")
        log.flush()
        os.remove(os.path.join(OUT, f"fuzz_{seed}.npz"))
