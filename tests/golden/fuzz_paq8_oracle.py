#!/usr/bin/env python3
"""Fuzz the assembled paq8 restatement (oracle/paq8_predictor.c) against the reference's own paq8::Predictor
(oracle/ref_paq8core.cpp -> oracle/_ref/libcmixrefpaq8.so) on long seeded streams: blocks framed the way cmix's
preprocessor frames them (TEXT / DEFAULT / EXE / HDR headers) with payload flavours aimed at the table-replacement,
record-length, match and word-model paths that short fixtures touch lightly, at small memory levels so that the hash
tables overflow and evict. All 1591 outputs and the final probability are compared after every bit. Not a pytest
(minutes per stream, dev container only); one process per stream because the reference keeps its sub-models in
function-local statics.

    python tests/golden/fuzz_paq8_oracle.py [first_seed] [count] [kbytes]     # appends to tests/golden/fuzz_log.txt
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


def block(ftype, payload, info=None):
    hdr = bytes([ftype]) + len(payload).to_bytes(4, "big")
    if info is not None:
        hdr += info.to_bytes(4, "big")
    return hdr + payload


def payload(rng, kind, n, seed):
    from cmix_amd import synth
    text = synth.enwik_like(max(n, 4000), seed)
    if kind == 0:    # text
        return text[:n]
    if kind == 1:    # fixed-length records with counters, padding and a few text fields
        rl = int(rng.choice([12, 16, 24, 33, 48]))
        recs = [(i * 7).to_bytes(4, "little") + bytes([i % 5, 0]) + (b"%-*d" % (rl - 6, i % 97))[:rl - 6] for i in range(n // rl + 1)]
        return b"".join(recs)[:n]
    if kind == 2:    # x86-like opcode soup
        ops = [b"\x55\x8b\xec", b"\x83\xec\x10", b"\x8b\x45\x08", b"\xe8\x10\x00\x00\x00", b"\x0f\x84\x20\x01\x00\x00", b"\x48\x8b\x05\x10\x20\x00\x00",
               b"\xc3", b"\x90", b"\xff\x15\x00\x10\x40\x00", b"\xeb\xfe", b"\x89\x44\x24\x04", b"\x66\x0f\x1f\x44\x00\x00"]
        return b"".join(ops[int(k)] for k in rng.integers(0, len(ops), n // 2))[:n]
    if kind == 3:    # uniformly random bytes, minus the sequences that start a JPEG / RIFF / TGA / BMP detector
        b = bytearray(rng.integers(0, 256, n, dtype=np.uint8).tobytes())
        for i in range(len(b) - 1):
            if b[i] == 0xFF and b[i + 1] == 0xD8:
                b[i + 1] = 0xD7
        return bytes(b)
    if kind == 4:    # long exact repeats with breaks (match / sparse match lengths, run maps)
        unit = text[:int(rng.integers(100, 900))]
        return (unit * (n // len(unit) + 1))[:n // 2] + b"#" + text[1000:1000 + n // 4] + (unit * 3)[:n // 4]
    if kind == 5:    # XML-ish markup with attributes, comments, CDATA
        tags = [b"page", b"title", b"id", b"revision", b"text", b"contributor", b"username"]
        out = []
        while sum(map(len, out)) < n:
            t = tags[int(rng.integers(len(tags)))]
            out.append(b"<" + t + b' id="%d" xml:space="preserve">' % int(rng.integers(1, 99999)) + text[len(out) * 13:len(out) * 13 + int(rng.integers(5, 120))] +
                       b"</" + t + b">\n" + (b"<!-- c -->" if rng.random() < 0.1 else b"") + (b"<![CDATA[x]]>" if rng.random() < 0.05 else b""))
        return b"".join(out)[:n]
    if kind == 6:    # tables of numbers, dates and times in columns
        rows = [b"%4d\t%02d:%02d:%02d\t2018-%02d-%02d\t%7.3f\t%s\n" % (i, i % 24, (i * 7) % 60, (i * 13) % 60, 1 + i % 12, 1 + i % 28, i * 1.618,
                                                                    [b"alpha", b"beta", b"gamma"][i % 3]) for i in range(n // 40 + 1)]
        return b"".join(rows)[:n]
    # French / German / English mix with UTF-8 accents, quotes, abbreviations, hyphenated line breaks
    import test_oracle_paq8core as T
    corpus = bytes(T._text_corpus())
    return (corpus * (n // len(corpus) + 1))[:n]


def make_stream(seed, nbytes):
    rng = np.random.default_rng(seed)
    parts, total = [], 0
    while total < nbytes:
        kind = int(rng.integers(0, 8))
        n = int(rng.integers(1500, 9000))
        body = payload(rng, kind, n, seed + len(parts))
        ftype = {0: 4, 7: 4, 5: 4, 6: 4, 2: 3}.get(kind, 0)
        if rng.random() < 0.1:
            ftype = 1
        b = block(ftype, body, 0) if ftype == 4 else block(ftype, body)
        parts.append(b)
        total += len(b)
    return b"".join(parts)[:nbytes]


def child(seed, nbytes):
    import shutil
    import tempfile
    from oracle import oracle as O
    from oracle import refharness as R
    tmp = tempfile.mkdtemp()
    dst = os.path.join(tmp, "libcmixrefpaq8_private.so")
    shutil.copy(R.PAQ8_LIB_PATH, dst)
    L, lib = C.CDLL(dst), O.lib()
    for x, pre in ((L, "refp8"), (lib, "orc_p8")):
        getattr(x, pre + "_predictor_new").restype = C.c_void_p
        getattr(x, pre + "_predictor_new").argtypes = [C.c_int]
        getattr(x, pre + "_predictor_update").argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    level = seed % 3
    data = make_stream(seed, nbytes)
    lib.orc_p8_rnd_reset()
    ref, got = L.refp8_predictor_new(level), lib.orc_p8_predictor_new(level)
    a, b = np.zeros(1591, np.float32), np.zeros(1591, np.float32)
    for n, byte in enumerate(data):
        for bpos in range(8):
            y = (byte >> (7 - bpos)) & 1
            pr, pg = L.refp8_predictor_update(ref, y, a.ctypes.data), lib.orc_p8_predictor_update(got, y, b.ctypes.data)
            if pg < 0:
                print("REFUSED code %d at byte %d" % (pg, n))
                return 3
            bad = np.nonzero(a.view(np.uint32) != b.view(np.uint32))[0]
            if bad.size or pr != pg:
                print("MISMATCH at byte %d bit %d: columns %s, p %d vs %d" % (n, bpos, bad[:8], pr, pg))
                return 1
    print("ok level %d %d bytes" % (level, len(data)))
    shutil.rmtree(tmp, ignore_errors=True)
    return 0


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        sys.exit(child(int(sys.argv[2]), int(sys.argv[3])))
    first = int(sys.argv[1]) if len(sys.argv) > 1 else 500
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
            line = "paq8 oracle seed %d: %s (%d s)" % (s, out[-1] if out else "no output, exit %d" % pr.returncode, time.time() - t0)
            if pr.returncode == 0:
                line = line.replace(": ok", ": all 1591 outputs + final p bit-exact,")
            else:
                failed += 1
            print(line, flush=True)
            log.write(line + "\n")
            log.flush()
        time.sleep(1)
    sys.exit(1 if failed else 0)
