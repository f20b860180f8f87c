#!/usr/bin/env python3
"""Fuzz of the paq8 stage's media models against the UNMODIFIED reference (dev container only: needs oracle/_ref/libcmixref.so and
libcmixrefpaq8.so). Each case is a seeded random file -- images (BMP 1/4/8/24/32 bit, PBM/PGM/PPM/PAM, TGA), PCM audio (WAV 8/16 bit,
mono/stereo), JPEG (baseline gray/colour, 4:4:4 / 4:2:2 / 4:2:0, restart markers, optimised Huffman tables, progressive, cut off, with a
thumbnail) between pieces of text and binary data -- framed either by the reference's preprocessor (block path) or as one DEFAULT block
(paq8's own detectors). The 1591 values PAQ8::Predict() returns before every bit, from the reference's paq8::Predictor, are compared
(32-bit row hashes, make_paq8_hashes.row_hash) with the stage's host emulation (tests/host/p8stage_emul.cpp: the device kernels' own step
functions + the product's front end) run in random chunk sizes.

    python tests/golden/fuzz_paq8_media.py FIRST_SEED COUNT [WORKERS] [--late]   # appends to tests/golden/fuzz_media_log.txt
                                                                  # --late: the emulation in the decoder's order of operations
    python tests/golden/fuzz_paq8_media.py --case SEED                    # one case, in this process (one reference predictor per process)
"""
import os
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]
LOG = os.path.join(HERE, "fuzz_media_log.txt")


def build_case(seed):
    """-> (description, stream the predictor sees, may_be_refused)"""
    import make_paq8_hashes as M
    from cmix_amd import synth
    from make_golden import default_block
    r = np.random.default_rng(1000003 * seed + 17)
    text = synth.enwik_like(1200, 100 + seed)

    def filler():
        k = int(r.integers(0, 4))
        n = int(r.integers(20, 260))
        if k == 0:
            o = int(r.integers(0, len(text) - n))
            return text[o:o + n]
        if k == 1:
            return bytes(r.integers(0, 256, n, dtype=np.uint8))
        if k == 2:
            return bytes(r.integers(0, 4, n, dtype=np.uint8) * 63)
        return b""

    def picture(w, h, planes):
        k = int(r.integers(0, 5))
        if k == 0:
            return r.integers(0, 256, (h, w, planes), dtype=np.uint8)                      # noise
        if k == 1:
            return np.full((h, w, planes), int(r.integers(0, 256)), np.uint8)              # flat
        return M.photo(w, h, planes, int(r.integers(0, 1 << 30)))

    padded = False
    parts, names = [filler()], []
    for _ in range(int(r.integers(1, 4))):
        kind = ["bmp24", "bmp32", "bmp8g", "bmp8p", "bmp4", "bmp1", "pgm", "ppm", "pbm", "pam", "tga", "wav", "jpeg", "jpeg"][int(r.integers(0, 14))]
        if kind in ("bmp24", "bmp32"):
            planes = 3 if kind == "bmp24" else 4
            w = int(r.integers(2, 20)) * 4 if r.random() < 0.9 or planes == 4 else int(r.integers(9, 60))
            padded |= (w * planes) % 4 != 0
            h = int(r.integers(4, 40))
            parts.append(M.bmp_file(picture(w, h, planes)))
            names.append("%s %dx%d" % (kind, w, h))
        elif kind in ("bmp8g", "bmp8p"):
            w = int(r.integers(3, 24)) * 4 if r.random() < 0.9 else int(r.integers(9, 80))
            padded |= w % 4 != 0
            h = int(r.integers(4, 40))
            img = picture(w, h, 1)[:, :, 0]
            pal = [(i, i, i) for i in range(256)] if kind == "bmp8g" else r.integers(0, 256, (256, 3))
            parts.append(M.bmp8_file(img, pal))
            names.append("%s %dx%d" % (kind, w, h))
        elif kind == "bmp4":
            w = int(r.integers(2, 16)) * 8
            h = int(r.integers(4, 48))
            parts.append(M.bmp4_file((picture(w, h, 1)[:, :, 0] >> 4).astype(np.uint8), r.integers(0, 256, (16, 3))))
            names.append("bmp4 %dx%d" % (w, h))
        elif kind == "bmp1":
            w = int(r.integers(1, 8)) * 32
            h = int(r.integers(8, 64))
            parts.append(M.bmp1_file((picture(w, h, 1)[:, :, 0] > int(r.integers(60, 200))).astype(np.uint8)))
            names.append("bmp1 %dx%d" % (w, h))
        elif kind in ("pgm", "ppm"):
            planes = 1 if kind == "pgm" else 3
            w, h = int(r.integers(8, 72)), int(r.integers(4, 40))
            if planes == 3:
                w = (w + 3) & ~3
            parts.append((b"P5" if planes == 1 else b"P6") + b"\n%d %d\n255\n" % (w, h) + picture(w, h, planes).tobytes())
            names.append("%s %dx%d" % (kind, w, h))
        elif kind == "pbm":
            w, h = int(r.integers(1, 24)) * 8, int(r.integers(8, 64))
            parts.append(b"P4\n%d %d\n" % (w, h) + np.packbits(picture(w, h, 1)[:, :, 0] > 128, axis=1).tobytes())
            names.append("pbm %dx%d" % (w, h))
        elif kind == "pam":
            depth = int(r.choice([1, 3, 4]))
            w, h = int(r.integers(2, 16)) * 4, int(r.integers(4, 36))
            tt = {1: b"GRAYSCALE", 3: b"RGB", 4: b"RGB_ALPHA"}[depth]
            parts.append(b"P7\nWIDTH %d\nHEIGHT %d\nDEPTH %d\nMAXVAL 255\nTUPLTYPE %s\nENDHDR\n" % (w, h, depth, tt) + picture(w, h, depth).tobytes())
            names.append("pam%d %dx%d" % (depth, w, h))
        elif kind == "tga":
            tk = int(r.choice([1, 2, 2, 3]))
            w, h = int(r.integers(2, 16)) * 4, int(r.integers(4, 36))
            if tk == 2:
                img = picture(w, h, int(r.choice([3, 4])))
            else:
                img = picture(w, h, 1)[:, :, 0]
            parts.append(M.tga_file(img, tk))
            names.append("tga%d %dx%d" % (tk, w, h))
        elif kind == "wav":
            ch, bits = int(r.choice([1, 2])), int(r.choice([8, 16]))
            n = int(r.integers(200, 1500))
            parts.append(M.wav_file(n, ch, bits, int(r.integers(0, 1 << 30))))
            names.append("wav%d/%d x%d" % (bits, ch, n))
        else:
            gray = r.random() < 0.3
            w, h = int(r.integers(2, 12)) * 8, int(r.integers(2, 10)) * 8
            img = picture(w, h, 3)
            img = img[:, :, 0] if gray else img
            kw = {"quality": int(r.integers(25, 95))}
            if not gray:
                kw["subsampling"] = int(r.integers(0, 3))
            opt = int(r.integers(0, 6))
            if opt == 0:
                kw["restart_marker_rows"] = int(r.integers(1, 3))
            elif opt == 1:
                kw["restart_marker_blocks"] = int(r.integers(1, 6))
            elif opt == 2:
                kw["optimize"] = True
            elif opt == 3:
                kw["progressive"] = True
            if opt == 4 and not gray:
                j = M.jpeg_with_thumbnail(img, M.photo(16, 16, 3, seed))
            else:
                j = M.jpeg_file(img, **kw)
            if r.random() < 0.15:
                j = j[:int(r.integers(len(j) // 2, len(j)))]
            parts.append(j)
            names.append("jpeg%s %dx%d %s" % ("g" if gray else "c", w, h, ",".join("%s=%s" % kv for kv in sorted(kw.items()))))
        parts.append(filler())
    payload = b"".join(parts)
    if r.random() < 0.5:
        return "pre: " + "; ".join(names), M.preprocessed(payload), padded
    return "raw: " + "; ".join(names), default_block(payload), padded


def run_case(seed):
    import make_paq8_hashes as M
    import test_p8stage_host as T
    import ctypes as C
    desc, stream, padded = build_case(seed)
    stream = bytes(stream)[:9000]
    r = np.random.default_rng(seed)
    chunks = [int(r.integers(60, 1500)) for _ in range(3)]
    # the emulator first (as tests/test_p8stage_host.run_stage, but a refusal of the front end is a result, not an assertion): on a stream the
    # product refuses because the reference reads past an array there (image rows with padding, p8f_image.c), the reference may crash
    L = T.emul()
    data = np.ascontiguousarray(np.frombuffer(stream, np.uint8))
    h = L.p8s_create(11)
    if LATE:   # the decoder's order of operations, model steps included (p8stage_emul.cpp: p8s_set_late)
        L.p8s_set_late.argtypes = [C.c_void_p, C.c_int]
        L.p8s_set_late(h, 3)
    out = np.zeros((8 * len(data), 1591), np.float32)
    pos, k, rc = 0, 0, 0
    while pos < len(data):
        n = min(chunks[k % 3], len(data) - pos)
        k += 1
        rc = L.p8s_run(h, data[pos:].ctypes.data, n, out[8 * pos:8 * (pos + n)].ctypes.data)
        if rc:
            break
        pos += n
    L.p8s_destroy(h)
    head = "seed %d%s  %d bytes  chunks %s  [%s]  " % (seed, " (decoder's order)" if LATE else "", len(stream), chunks, desc)
    if rc:   # (said before the reference runs: should it crash, the driver loop logs this line with the crash)
        print(head + "REFUSED rc=%d in the chunk at byte %d%s; the reference on the same stream ..." % (rc, pos, " (padded rows: expected)" if padded else " (UNEXPECTED)"), flush=True)
    want = M.reference_hashes(stream)
    if rc:
        got = M.row_hash(out[:8 * pos])
        bad = np.nonzero(got != want[:8 * pos])[0]
        res = "REFUSED rc=%d in the chunk at byte %d%s%s" % (rc, pos, " (padded rows: expected)" if padded else " (UNEXPECTED)",
                                                               "" if bad.size == 0 else "; MISMATCH before it at step %d" % bad[0])
    else:
        bad = np.nonzero(M.row_hash(out) != want)[0]
        res = "ok" if bad.size == 0 else "MISMATCH at step %d (byte %d)" % (bad[0], bad[0] >> 3)
    return head + res


LATE = "--late" in sys.argv
if LATE:
    sys.argv.remove("--late")

if __name__ == "__main__":
    if sys.argv[1] == "--case":
        print(run_case(int(sys.argv[2])), flush=True)
        sys.exit(0)
    first, count = int(sys.argv[1]), int(sys.argv[2])
    workers = int(sys.argv[3]) if len(sys.argv) > 3 else 2
    seeds = list(range(first, first + count))
    running = {}
    while seeds or running:
        while seeds and len(running) < workers:
            s = seeds.pop(0)
            running[s] = subprocess.Popen([sys.executable, os.path.abspath(__file__), "--case", str(s)] + (["--late"] if LATE else []), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        for s, p in list(running.items()):
            if p.poll() is not None:
                o, e = p.communicate()
                mine = [l for l in o.splitlines() if l.startswith("seed ")]   # (the reference's preprocessor prints its block statistics too)
                line = mine[-1] if mine else "seed %d  CRASH rc=%s %s" % (s, p.returncode, e.strip().splitlines()[-1:])
                if p.returncode and mine:
                    line += " CRASHED (rc %d)" % p.returncode
                with open(LOG, "a") as f:
                    f.write(line + "\n")
                print(line, flush=True)
                del running[s]
        import time
        time.sleep(0.5)
