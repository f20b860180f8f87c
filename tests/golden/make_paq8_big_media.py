#!/usr/bin/env python3
"""A media stream at scale for the paq8 stage: 168 KB as the reference's preprocessor frames them -- a 160 x 120 24-bit BMP (IMAGE24 block, 57 KB: the
image model's segment runs through fourteen 4 KB chunks), a 16-bit stereo WAV (40 KB), a 320 x 240 JPEG, a 200 x 150 PGM (IMAGE8GRAY block), a 4-bit and
an 8-bit BMP, short pieces of text between them (longer ones would make the preprocessor type the surrounding block TEXT: refused). The
fixture holds the stream and one 64-bit digest per 256 steps of the 32-bit row hashes (make_paq8_hashes.row_hash) of the 1591 values the UNMODIFIED
reference's paq8::Predictor returns before every bit (oracle/_ref/libcmixrefpaq8.so).   python tests/golden/make_paq8_big_media.py"""
import ctypes as C
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]
STEPS = 256


def digest(h):
    """[T] u32 row hashes -> [ceil(T / 256)] u64: sum of hash * (2 i + 1) over each block of 256 steps (order-sensitive)"""
    n = (len(h) + STEPS - 1) // STEPS
    v = np.zeros(n * STEPS, np.uint64)
    v[:len(h)] = h
    w = (np.arange(STEPS, dtype=np.uint64) * np.uint64(2) + np.uint64(1))[None, :]
    with np.errstate(over="ignore"):
        return (v.reshape(n, STEPS) * w).sum(axis=1, dtype=np.uint64)


def payload():
    """the file itself (what `cmix -c` is given: tests/golden/make_dropin_media.py)"""
    import make_paq8_hashes as M
    from cmix_amd import synth
    t = synth.enwik_like(3000, 77)
    parts = [t[:200], M.bmp_file(M.photo(160, 120, 3, 101)), t[500:700], M.wav_file(10000, 2, 16, 102), t[900:1100],
             M.jpeg_file(M.photo(320, 240, 3, 103), quality=75), t[1300:1500], b"P5\n200 150\n255\n" + M.photo(200, 150, 1, 104)[:, :, 0].tobytes(), t[1700:1900],
             M.bmp4_file((M.photo(256, 128, 1, 105)[:, :, 0] >> 4).astype(np.uint8), np.random.default_rng(106).integers(0, 256, (16, 3))), t[2100:2300],
             M.bmp8_file(M.photo(120, 90, 1, 107)[:, :, 0], np.random.default_rng(108).integers(0, 256, (256, 3))), t[2500:2700]]
    return b"".join(parts)


def stream():
    """what the reference's preprocessor hands the predictor for it"""
    import make_paq8_hashes as M
    return bytes(M.preprocessed(payload()))


if __name__ == "__main__":
    import make_paq8_hashes as M
    s = stream()
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libcmixrefpaq8.so"))
    L.refp8_predictor_new.restype = C.c_void_p
    L.refp8_predictor_new.argtypes = [C.c_int]
    L.refp8_predictor_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    h = L.refp8_predictor_new(11)
    bits = np.unpackbits(np.frombuffer(s, np.uint8))
    hashes = np.zeros(len(bits), np.uint32)
    B = 8192
    buf = np.full((B, 1591), 0.5, np.float32)   # row 0: PAQ8::Predict() before the first Perceive
    fill, base = 1, 0
    for t in range(len(bits) - 1):
        if fill == B:
            hashes[base:base + B] = M.row_hash(buf)
            base += B
            fill = 0
        L.refp8_predictor_update(h, int(bits[t]), buf[fill].ctypes.data)
        fill += 1
    hashes[base:base + fill] = M.row_hash(buf[:fill])
    np.savez_compressed(os.path.join(HERE, "paq8_big_media_168k.npz"), stream=np.frombuffer(s, np.uint8), digest=digest(hashes))
    print(len(s), "bytes,", len(bits), "steps,", len(digest(hashes)), "digests")
