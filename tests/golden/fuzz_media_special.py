#!/usr/bin/env python3
"""Hand-made media headers and JPEG shapes for the paq8 stage's detectors and parsers, against the UNMODIFIED reference (dev container only):
WAV with 24 / 32-bit samples, four channels, a LIST chunk, an extensible fmt chunk; BMP top-down, V4 / V5 / OS/2 headers, RLE flag, 16 bits; TGA with an id
field, RLE types, other origins; PNM with comments, 16-bit maxval, width 1, ASCII variants; JPEG CMYK, odd sizes, 1200 x 16, 16 x 900, 1 x 1, comments,
two files back to back, nested thumbnails. Each inside a DEFAULT block between text; the stage's host emulation (tests/host/p8stage_emul.cpp) must
return the reference's 1591 values at every step -- whether the reference's detector takes the header or not.

    python tests/golden/fuzz_media_special.py            # every case, one process each -> tests/golden/fuzz_media_special_log.txt
"""
import io
import os
import struct
import subprocess
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]


def all_cases():
    import make_paq8_hashes as M
    from PIL import Image
    def wav(n, ch, bits, fmt_tag=1, extra=b"", fmt_extra=b""):
        r = np.random.default_rng(n)
        x = (np.sin(np.arange(n * ch) * 0.07) * 0.4 + r.normal(0, 0.01, n * ch))
        if bits == 8: data = (128 + 100 * x).clip(0, 255).astype(np.uint8).tobytes()
        elif bits == 16: data = (12000 * x).astype('<i2').tobytes()
        elif bits == 24: data = b"".join(int(v * 4e6).to_bytes(3, 'little', signed=True) for v in x)
        else: data = (x * 1e9).astype('<i4').tobytes()
        ba = ch * bits // 8
        fmt = struct.pack('<HHIIHH', fmt_tag, ch, 22050, 22050 * ba, ba, bits) + fmt_extra
        body = b"WAVE" + b"fmt " + struct.pack('<I', len(fmt)) + fmt + extra + b"data" + struct.pack('<I', len(data)) + data
        return b"RIFF" + struct.pack('<I', len(body)) + body
    def bmp(w, h, bpp, hdr=40, topdown=False, comp=0):
        row = ((w * bpp + 31) // 32) * 4
        r = np.random.default_rng(w * h)
        pix = bytes(r.integers(0, 256, row * h, dtype=np.uint8))
        pal = b"" if bpp > 8 else bytes(r.integers(0, 256, 4 << bpp, dtype=np.uint8))
        if hdr == 12:
            ih = struct.pack('<IHHHH', 12, w, h, 1, bpp); pal = pal[:3 * (1 << bpp)] if bpp <= 8 else b""
        else:
            ih = struct.pack('<IiiHHIIiiII', hdr, w, -h if topdown else h, 1, bpp, comp, len(pix), 2835, 2835, 0, 0) + bytes(hdr - 40)
        off = 14 + len(ih) + len(pal)
        return b"BM" + struct.pack('<IHHI', off + len(pix), 0, 0, off) + ih + pal + pix
    def tga(w, h, kind, bpp, idlen=0, desc=0):
        r = np.random.default_rng(w + h)
        hdr = bytes([idlen, 1 if kind in (1, 9) else 0, kind]) + (struct.pack('<HHB', 0, 256, 24) if kind in (1, 9) else bytes(5)) + struct.pack('<HHHHBB', 0, 0, w, h, bpp, desc)
        cmap = bytes(r.integers(0, 256, 768, dtype=np.uint8)) if kind in (1, 9) else b""
        return hdr + bytes(idlen) + cmap + bytes(r.integers(0, 256, w * h * bpp // 8, dtype=np.uint8))
    cases = [
     ("wav 24-bit stereo", wav(800, 2, 24)), ("wav 32-bit mono", wav(800, 1, 32)), ("wav 4 channels 16-bit", wav(500, 4, 16)),
     ("wav with LIST chunk before data", wav(900, 2, 16, extra=b"LIST" + struct.pack('<I', 12) + b"INFOISFT" + struct.pack('<I', 0))),
     ("wav extensible fmt", wav(700, 2, 16, fmt_tag=0xFFFE, fmt_extra=struct.pack('<HHI', 22, 16, 3) + bytes(16))),
     ("bmp top-down 24", bmp(40, 30, 24, topdown=True)), ("bmp V4 header 24", bmp(40, 30, 24, hdr=108)), ("bmp V5 header 32", bmp(32, 20, 32, hdr=124)),
     ("bmp OS/2 header 8", bmp(40, 30, 8, hdr=12)), ("bmp RLE8 flag", bmp(40, 30, 8, comp=1)), ("bmp 16-bit", bmp(40, 30, 16)),
     ("tga id field 24", tga(40, 30, 2, 24, idlen=17)), ("tga RLE type 10", tga(40, 30, 10, 24)), ("tga top-left origin 32", tga(32, 24, 2, 32, desc=0x28)),
     ("pgm with comment", b"P5\n# a comment\n40 30\n255\n" + bytes(np.random.default_rng(1).integers(0, 256, 1200, dtype=np.uint8))),
     ("pgm 16-bit maxval", b"P5\n40 30\n65535\n" + bytes(np.random.default_rng(2).integers(0, 256, 2400, dtype=np.uint8))),
     ("ppm width 1", b"P6\n1 200\n255\n" + bytes(np.random.default_rng(3).integers(0, 256, 600, dtype=np.uint8))),
     ("pbm ascii P1 (not binary)", b"P1\n8 8\n" + b"0 1 " * 16 + b"\n"),
    ]
    def jpg(img, mode=None, **kw):
        b = io.BytesIO(); im = Image.fromarray(img)
        if mode: im = im.convert(mode)
        im.save(b, "JPEG", **kw); return b.getvalue()
    cases += [
     ("cmyk 64x48", jpg(M.photo(64, 48, 3, 1), "CMYK", quality=70)),
     ("odd 37x29 4:2:0", jpg(M.photo(37, 29, 3, 2), quality=75)),
     ("wide 1200x16", jpg(M.photo(1200, 16, 3, 3), quality=60)),
     ("tall 16x900 gray", jpg(M.photo(16, 900, 1, 4)[:, :, 0], quality=60)),
     ("tiny 1x1", jpg(M.photo(1, 1, 3, 5), quality=90)),
     ("8x8 + comment", jpg(M.photo(8, 8, 3, 6), quality=50, comment=b"hello world comment")),
     ("4:2:2 odd 50x33 q10", jpg(M.photo(50, 33, 3, 7), quality=10, subsampling=1)),
     ("q100 4:4:4 40x40", jpg(M.photo(40, 40, 3, 8), quality=100, subsampling=0)),
     ("two jpegs back to back", jpg(M.photo(48, 32, 3, 9), quality=70) + jpg(M.photo(32, 48, 1, 10)[:, :, 0], quality=70)),
     ("jpeg inside jpeg app segment twice", M.jpeg_with_thumbnail(M.photo(48, 40, 3, 11), M.photo(16, 16, 3, 12))[:-2] + M.jpeg_with_thumbnail(M.photo(40, 32, 3, 13), M.photo(8, 8, 3, 14))),
    ]
    return cases


def run_case(k):
    import make_paq8_hashes as M
    import test_p8stage_host as T
    from cmix_amd import synth
    from make_golden import default_block
    t = synth.enwik_like(1500, 41)
    name, m = all_cases()[k]
    stream = bytes(default_block(t[:150] + m + t[150:350]))[:12000]
    data = np.frombuffer(stream, np.uint8).copy()
    L = T.emul()
    h = L.p8s_create(11)
    out = np.zeros((8 * len(data), 1591), np.float32)
    rc = L.p8s_run(h, data.ctypes.data, len(data), out.ctypes.data)
    L.p8s_destroy(h)
    want = M.reference_hashes(stream)
    bad = np.nonzero(M.row_hash(out) != want)[0] if rc == 0 else np.array([-1])
    return "case %d  %s  %d bytes  %s" % (k, name, len(data), "ok" if rc == 0 and bad.size == 0 else "rc %d, first differing step %d" % (rc, bad[0]))


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--case":
        print(run_case(int(sys.argv[2])), flush=True)
        sys.exit(0)
    n = len(all_cases())
    with open(os.path.join(HERE, "fuzz_media_special_log.txt"), "w") as f:
        for k in range(n):
            o = subprocess.run([sys.executable, os.path.abspath(__file__), "--case", str(k)], capture_output=True, text=True).stdout
            line = [l for l in o.splitlines() if l.startswith("case ")]
            line = line[-1] if line else "case %d  CRASH" % k
            f.write(line + "\n")
            print(line, flush=True)
