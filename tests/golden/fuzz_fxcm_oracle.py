#!/usr/bin/env python3
"""Fuzz the assembled fxcm restatement (oracle/fxcm_model.c) against the reference's own fxcmv1::Predictor
(oracle/ref_fxcmcore.cpp -> oracle/_ref/libcmixreffxcm.so) on long seeded streams: the payload flavours of
fuzz_paq8_oracle.py plus wiki markup in cmix's WRT-swapped alphabet (the parser keys on the swapped punctuation), with
seeded random LSTM hints. All 431 outputs, the final probability and -- at every byte boundary -- the 256 context-slot
hashes are compared. Not a pytest (a minute per stream, dev container only); one process per stream because the
reference keeps the model in namespace-level globals.

    python tests/golden/fuzz_fxcm_oracle.py [first_seed] [count] [kbytes]     # appends to tests/golden/fuzz_log.txt
"""
import ctypes as C
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def swap(b):
    """cmix's WRT character swap (the inverse of fxcm's charSwap, reference src/models/fxcmv1.cpp:2281-2287) on raw text."""
    out = bytearray(b)
    for i, c in enumerate(out):
        if ord("{") <= c < 127:
            c += ord("P") - ord("{")
        elif ord("P") <= c < ord("T"):
            c -= ord("P") - ord("{")
        elif ord(":") <= c <= ord("?") or ord("J") <= c <= ord("O"):
            c ^= 0x70
        if c in (ord("X"), ord("`")):
            c ^= ord("X") ^ ord("`")
        out[i] = c
    return bytes(out)


def make_stream(seed, nbytes):
    import fuzz_paq8_oracle as F
    rng = np.random.default_rng(seed)
    wiki = (b"{{infobox|name=test|value=12}}\n{|\n|-\n| cell one || cell two\n|-\n| 3.14 || [[link|text]]\n|}\n* item one\n* item [[two]], three\n"
            b"== heading ==\n'''bold''' and ''italic'' text. see [http://example.org/page link] &amp; more; x &lt; y.\n\n<math>a^2</math> <ref>r</ref>\n"
            b"[[category:things]] [[image:x.png|thumb|caption here]]\n: indented line\n; term : definition\n")
    parts, total = [], 0
    while total < nbytes:
        kind = int(rng.integers(0, 10))
        n = int(rng.integers(1500, 9000))
        if kind < 8:
            body = F.payload(rng, kind, n, seed + len(parts))
        else:
            from cmix_amd import synth
            body = b"".join([wiki, synth.enwik_like(600, seed + len(parts)).lower()] * (n // 1200 + 1))[:n]
        if rng.random() < 0.5:
            body = swap(body.lower() if rng.random() < 0.5 else body)
        parts.append(body)
        total += len(body)
    return b"".join(parts)[:nbytes]


def child(seed, nbytes):
    import shutil
    import tempfile
    from oracle import oracle as O
    from oracle import refharness as R
    tmp = tempfile.mkdtemp()
    dst = os.path.join(tmp, "libcmixreffxcm_private.so")
    shutil.copy(R.FXCM_LIB_PATH, dst)
    L, lib = C.CDLL(dst), O.lib()
    P = C.c_void_p
    L.reffx_model_new.restype = P
    lib.orc_fx_model_new.restype = P
    L.reffx_model_update.argtypes = [P, C.c_int, C.c_int, C.c_int, P]
    lib.orc_fx_model_update.argtypes = [P, C.c_int, C.c_int, C.c_int, P]
    L.reffx_model_contexts.argtypes = [P]
    lib.orc_fx_model_contexts.argtypes = [P, P]
    data = make_stream(seed, nbytes)
    rng = np.random.default_rng(seed + 1)
    ref, got = L.reffx_model_new(), lib.orc_fx_model_new()
    a, b, ca, cb = np.zeros(431, np.float32), np.zeros(431, np.float32), np.zeros(256, np.uint32), np.zeros(256, np.uint32)
    for n, byte in enumerate(data):
        hints = rng.integers(0, 1 << 20, 8)
        for bpos in range(8):
            y = (byte >> (7 - bpos)) & 1
            hp, hx = 1 + int(hints[bpos]) % 4094, int(hints[bpos] >> 12) & 255
            pr, pg = L.reffx_model_update(ref, y, hp, hx, a.ctypes.data), lib.orc_fx_model_update(got, y, hp, hx, b.ctypes.data)
            bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
            if bad.size or pr != pg:
                print("MISMATCH at byte %d bit %d after %r: columns %s, p %d vs %d" % (n, bpos, data[max(0, n - 16):n], bad[:8], pr, pg))
                return 1
        L.reffx_model_contexts(ca.ctypes.data)
        lib.orc_fx_model_contexts(got, cb.ctypes.data)
        if (ca != cb).any():
            print("CONTEXT MISMATCH at byte %d after %r: slots %s" % (n, data[max(0, n - 16):n + 1], np.nonzero(ca != cb)[0][:8]))
            return 2
    print("ok %d bytes" % len(data))
    shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        sys.exit(child(int(sys.argv[2]), int(sys.argv[3])))
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 700
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    nbytes = (int(sys.argv[3]) if len(sys.argv) > 3 else 48) * 1024
    jobs = int(os.environ.get("JOBS", "4"))
    log = open(os.path.join(ROOT, "tests", "golden", "fuzz_log.txt"), "a")
    pending, running, failed = list(range(first, first + count)), [], 0
    while pending or running:
        while pending and len(running) < jobs:
            s = pending.pop(0)
            running.append((s, time.time(), subprocess.Popen([sys.executable, __file__, "--child", str(s), str(nbytes)], stdout=subprocess.PIPE, text=True)))
        for item in list(running):
            s, t0, pr = item
            if pr.poll() is None:
                continue
            running.remove(item)
            out = pr.stdout.read().strip().splitlines()
            line = "fxcm oracle seed %d: %s (%d s)" % (s, out[-1] if out else "no output, exit %d" % pr.returncode, time.time() - t0)
            if pr.returncode == 0:
                line = line.replace(": ok", ": all 431 outputs + final p + 256 context slots bit-exact,")
            else:
                failed += 1
            print(line, flush=True)
            log.write(line + "\n")
            log.flush()
        time.sleep(1)
    sys.exit(1 if failed else 0)
