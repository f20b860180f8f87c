#!/usr/bin/env python3
"""Reference-derived fixtures for the paq8 stage on streams longer than the full traces can hold: per step a 32-bit
hash of the 1591 values PAQ8::Predict() returns (layer-0 columns 434..2024), from the UNMODIFIED reference's
paq8::Predictor compiled from /root/reference/src/models/paq8.cpp (oracle/_ref/libcmixrefpaq8.so, oracle/Makefile)
at cmix's level 11.   python tests/golden/make_paq8_hashes.py   ->  tests/golden/paq8_cols_*.npz
    stream [N] u8 (what cmix's preprocessor hands the predictor: block header + payload), hash [8 N] u32."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
HERE = os.path.dirname(os.path.abspath(__file__))


def multipliers():
    return np.random.default_rng(20260925).integers(1, 2**63, 1591, dtype=np.uint64) * np.uint64(2) + np.uint64(1)


def row_hash(rows):
    """rows [T, 1591] f32 -> [T] u32: sum over columns of (bit pattern + 1) * odd multiplier, top 32 bits."""
    v = np.ascontiguousarray(rows, np.float32).view(np.uint32).astype(np.uint64) + np.uint64(1)
    with np.errstate(over="ignore"):
        h = (v * multipliers()[None, :]).sum(axis=1, dtype=np.uint64)
    return (h >> np.uint64(32)).astype(np.uint32)


def header_lookalikes():
    """Binary data sprinkled with byte patterns paq8's image / JPEG detectors look at but do NOT accept -- the reference goes on
    as for ordinary data, after private state changes that must be followed exactly (imgModel :5393-5438, jpegModel :6058-6151):
    header-less DIB headers whose pixel area is <= 64 bytes (8 bpp with a 1 KB grayscale palette walked entry by entry, which rewrites
    Stats.Record for recordModel; 24 bpp; an icon-shaped 4 bpp one that comes back as a 1-bit mask), a 'BM' file header, SOI + APP0
    with a DQT holding a zero (parser reset), SOI + DQT + SOF without a scan header, an APP1 with an embedded SOI."""
    r = np.random.default_rng(31)

    def noise(n):
        return bytes(r.integers(0, 6, n, dtype=np.uint8) * 17)

    def dib(w, h, bpp, size=0, clr=0):
        return (40).to_bytes(4, "little") + w.to_bytes(4, "little") + h.to_bytes(4, "little") + (1).to_bytes(2, "little") + \
            bpp.to_bytes(2, "little") + bytes(4) + size.to_bytes(4, "little") + bytes(8) + bytes(4) + clr.to_bytes(4, "little")
    gray_palette = b"".join(bytes([i, i, i, 0]) for i in range(256))
    dqt_bad = b"\xff\xdb\x00\x43\x00" + bytes([3] * 20) + b"\x00" + bytes([5] * 43)
    dqt_ok = b"\xff\xdb\x00\x43\x00" + bytes(r.integers(1, 40, 64, dtype=np.uint8))
    sof = b"\xff\xc0\x00\x11\x08\x00\x10\x00\x10\x03\x01\x22\x00\x02\x11\x01\x03\x11\x01"
    parts = [noise(300), dib(4, 4, 8), gray_palette, noise(200), dib(3, 2, 24), noise(150),
             b"BM" + (1000).to_bytes(4, "little") + bytes(4) + (0x36).to_bytes(4, "little") + dib(2, 2, 24), noise(120),
             dib(8, 16, 4), bytes(r.integers(0, 256, 64, dtype=np.uint8)), noise(200),
             b"\xff\xd8\xff\xe0\x00\x10JFIF\x00\x01\x01\x00\x00\x01\x00\x01\x00\x00" + dqt_bad + noise(100),
             b"\xff\xd8" + dqt_ok + sof + noise(300),
             b"\xff\xd8\xff\xe1\x00\x40Exif\x00\x00" + b"\xff\xd8" + dqt_ok[:30] + noise(200), noise(500)]
    return b"".join(parts)


def photo(w, h, planes, seed):
    """a synthetic photograph: smooth gradients, a few edges, sensor-like noise; [h, w, planes] u8"""
    r = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    img = np.empty((h, w, planes))
    for c in range(planes):
        a, b, ph = r.uniform(0.5, 2.5), r.uniform(0.5, 2.5), r.uniform(0, 6)
        img[:, :, c] = 128 + 70 * np.sin(xx / w * 3 * a + ph) * np.cos(yy / h * 2 * b) + 40 * ((xx * 0.7 + yy) % 37 > 18) + r.normal(0, 2.5, (h, w))
    if planes == 4:
        img[:, :, 3] = 255
    return img.clip(0, 255).astype(np.uint8)


def bmp_file(img):
    """bottom-up BITMAPINFOHEADER file of an [h, w, 3 or 4] image (rows padded to 4 bytes)"""
    h, w, planes = img.shape
    row = (w * planes + 3) & ~3
    pix = b"".join(img[y].tobytes() + bytes(row - w * planes) for y in range(h - 1, -1, -1))
    return (b"BM" + (54 + len(pix)).to_bytes(4, "little") + bytes(4) + (54).to_bytes(4, "little") + (40).to_bytes(4, "little") + w.to_bytes(4, "little") +
            h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (8 * planes).to_bytes(2, "little") + bytes(4) + len(pix).to_bytes(4, "little") + bytes(16) + pix)


def bmp8_file(img, palette):
    """bottom-up 8-bit BMP of an [h, w] index image with a 256-entry palette ([256, 3] BGR)"""
    h, w = img.shape
    row = (w + 3) & ~3
    pix = b"".join(img[y].tobytes() + bytes(row - w) for y in range(h - 1, -1, -1))
    pal = b"".join(bytes([int(b), int(g), int(r), 0]) for b, g, r in palette)
    off = 54 + 1024
    return (b"BM" + (off + len(pix)).to_bytes(4, "little") + bytes(4) + off.to_bytes(4, "little") + (40).to_bytes(4, "little") + w.to_bytes(4, "little") +
            h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (8).to_bytes(2, "little") + bytes(4) + len(pix).to_bytes(4, "little") + bytes(16) + pal + pix)


def wav_file(nsamples, channels, bits, seed):
    """RIFF / WAVE, PCM: a few sine partials + noise per channel (the second channel mostly follows the first)"""
    r = np.random.default_rng(seed)
    t = np.arange(nsamples)
    base = sum(a * np.sin(t * f + ph) for a, f, ph in zip((0.5, 0.25, 0.12), r.uniform(0.01, 0.2, 3), r.uniform(0, 6, 3)))
    chans = [base + r.normal(0, 0.01, nsamples)] + [0.8 * base + 0.1 * np.sin(t * 0.05) + r.normal(0, 0.01, nsamples) for _ in range(channels - 1)]
    x = np.stack(chans, -1)
    if bits == 8:
        data = (128 + 100 * x).clip(0, 255).astype(np.uint8).tobytes()
    else:
        data = (12000 * x).clip(-32768, 32767).astype("<i2").tobytes()
    ba = channels * bits // 8
    fmt = (16).to_bytes(4, "little") + (1).to_bytes(2, "little") + channels.to_bytes(2, "little") + (22050).to_bytes(4, "little") + (22050 * ba).to_bytes(4, "little") + \
        ba.to_bytes(2, "little") + bits.to_bytes(2, "little")
    body = b"WAVE" + b"fmt " + fmt + b"data" + len(data).to_bytes(4, "little") + data
    return b"RIFF" + len(body).to_bytes(4, "little") + body


def bmp4_file(img, palette):
    """bottom-up 4-bit BMP of an [h, w] image of values 0..15 with a 16-entry palette"""
    h, w = img.shape
    row = ((w * 4 + 31) >> 5) * 4
    def pack(r):
        r = np.concatenate([r, np.zeros(w & 1, np.uint8)])
        return ((r[0::2] << 4) | r[1::2]).astype(np.uint8).tobytes().ljust(row, b"\0")
    pix = b"".join(pack(img[y]) for y in range(h - 1, -1, -1))
    pal = b"".join(bytes([int(b), int(g), int(r), 0]) for b, g, r in palette)
    off = 54 + 64
    return (b"BM" + (off + len(pix)).to_bytes(4, "little") + bytes(4) + off.to_bytes(4, "little") + (40).to_bytes(4, "little") + w.to_bytes(4, "little") +
            h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (4).to_bytes(2, "little") + bytes(4) + len(pix).to_bytes(4, "little") + bytes(16) + pal + pix)


def bmp1_file(bits):
    """bottom-up 1-bit BMP of an [h, w] 0/1 image (two palette entries, rows padded to 4 bytes)"""
    h, w = bits.shape
    row = ((w + 31) >> 5) * 4
    pix = b"".join(np.packbits(bits[y]).tobytes().ljust(row, b"\0") for y in range(h - 1, -1, -1))
    off = 54 + 8
    return (b"BM" + (off + len(pix)).to_bytes(4, "little") + bytes(4) + off.to_bytes(4, "little") + (40).to_bytes(4, "little") + w.to_bytes(4, "little") +
            h.to_bytes(4, "little") + (1).to_bytes(2, "little") + (1).to_bytes(2, "little") + bytes(4) + len(pix).to_bytes(4, "little") + bytes(16) +
            bytes([0, 0, 0, 0, 255, 255, 255, 0]) + pix)


def jpeg_file(img, **kw):
    """a baseline JPEG of an [h, w, 3] (or [h, w]) image (Pillow's encoder: JFIF APP0, DQT, SOF0, DHT, SOS, entropy-coded data, EOI)"""
    import io
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def tga_file(img, kind):
    """uncompressed TGA: kind 2 = true colour ([h, w, 3] or [h, w, 4]), 3 = grayscale ([h, w]), 1 = colour-mapped ([h, w] indices + a 256-entry 24-bit map)"""
    h, w = img.shape[:2]
    if kind == 1:
        cmap = np.random.default_rng(5).integers(0, 256, (256, 3), dtype=np.uint8).tobytes()
        hdr = bytes([0, 1, 1, 0, 0, 0, 1, 24]) + bytes(4) + w.to_bytes(2, "little") + h.to_bytes(2, "little") + bytes([8, 0])
        return hdr + cmap + img.tobytes()
    bpp = 8 if kind == 3 else 8 * img.shape[2]
    hdr = bytes([0, 0, kind, 0, 0, 0, 0, 0]) + bytes(4) + w.to_bytes(2, "little") + h.to_bytes(2, "little") + bytes([bpp, 8 if bpp == 32 else 0])
    return hdr + img.tobytes()


def jpeg_with_thumbnail(img, thumb):
    """a JPEG whose first segment after SOI is an APP1 that contains a complete small JPEG"""
    main, small = jpeg_file(img, quality=70), jpeg_file(thumb, quality=50)
    body = b"Exif\0\0" + small
    return main[:2] + b"\xff\xe1" + (len(body) + 2).to_bytes(2, "big") + body + main[2:]


def preprocessed(payload):
    """the stream the reference's preprocessor (preprocessor.cpp:568 Encode) hands the predictor for a file: block headers, detected
    types (HDR + IMAGE24 / IMAGE32 for a BMP), its transforms -- through oracle/_ref/libcmixref.so (oracle/ref_harness.cpp)"""
    import tempfile
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libcmixref.so"))
    L.ref_preprocess_encode.argtypes = [C.c_char_p] * 3
    d = tempfile.mkdtemp()
    open(os.path.join(d, "in"), "wb").write(payload)
    assert L.ref_preprocess_encode(os.path.join(d, "in").encode(), os.path.join(d, "out").encode(), os.path.join(d, "tmp").encode()) == 0
    return open(os.path.join(d, "out"), "rb").read()


def image_streams():
    """24 / 32-bit images (im24bitModel, paq8.cpp:5001-5353): as IMAGE24 / IMAGE32 blocks the preprocessor makes of BMP files between
    other data (block path, contextModel2 :8165-8166), and as a BMP file inside a DEFAULT block (`cmix -n`: no preprocessing; paq8's own
    header detector imgModel :5386-5504 switches the model on and off). Row widths are multiples of the pixel size (see p8f_image.c)."""
    from cmix_amd import synth
    from make_golden import default_block, text_block
    text = synth.enwik_like(700, 11)
    return {
        "bmp24_14k": preprocessed(text[:300] + bmp_file(photo(96, 48, 3, 1)) + text[300:]),
        "bmp32_8k": preprocessed(text[:200] + bmp_file(photo(48, 40, 4, 2)) + bytes(range(256))),
        "bmp24_raw_9k": default_block(text[:150] + bmp_file(photo(64, 44, 3, 3)) + text[150:400]),
        # 8-bit images (im8bitModel :4743-4999): a binary PGM, which the preprocessor turns into an IMAGE8GRAY block; BMP files with a gray ramp
        # palette (paq8's detector walks the palette, finds it gray: the grayscale face of the model) and with a colour palette (the palette face)
        "pgm8_4k": preprocessed(text[:250] + b"P5\n64 56\n255\n" + photo(64, 56, 1, 4)[:, :, 0].tobytes() + text[250:500]),
        "bmp8_gray_raw_5k": default_block(text[:100] + bmp8_file(photo(64, 52, 1, 5)[:, :, 0], [(i, i, i) for i in range(256)]) + text[100:300]),
        # 1-bit images (im1bitModel :4634-4673): a binary PBM, which the preprocessor turns into an IMAGE1 block, and a 1-bit BMP inside a DEFAULT block
        "pbm1_2k": preprocessed(text[:200] + b"P4\n128 96\n" + np.packbits(photo(128, 96, 1, 13)[:, :, 0] > 128, axis=1).tobytes() + text[200:400]),
        "bmp1_raw_2k": default_block(text[:100] + bmp1_file((photo(160, 80, 1, 14)[:, :, 0] > 120).astype(np.uint8)) + text[100:250]),
        # 4-bit images (im4bitModel :4675-4742, its 14 contexts on a HashTable<16>): a 16-colour BMP inside a DEFAULT block
        "bmp4_raw_3k": default_block(text[:100] + bmp4_file((photo(96, 56, 1, 15)[:, :, 0] >> 4).astype(np.uint8), np.random.default_rng(16).integers(0, 256, (16, 3))) + text[100:250]),
        # baseline JPEG (jpegModel :5911-6597): a colour picture with 4:2:0 chroma as the preprocessor frames it (a JPEG block), and a grayscale one with
        # restart markers every 2 MCU rows inside a DEFAULT block
        "jpeg_5k": preprocessed(text[:200] + jpeg_file(photo(96, 80, 3, 17), quality=70) + text[200:400]),
        "jpeg_rst_raw_3k": default_block(text[:100] + jpeg_file(photo(112, 64, 1, 18)[:, :, 0], quality=60, restart_marker_rows=2) + text[100:200]),
        # everything in one file, as the preprocessor frames it: text, a 24-bit BMP (IMAGE24 block), a WAV (paq8's own detector), a JPEG (JPEG block), a PGM
        # (IMAGE8GRAY block), text -- every switch between the generic models and a model with tables of its own, in chunks that cut anywhere
        "mixed_media_12k": preprocessed(text[:150] + bmp_file(photo(48, 32, 3, 41)) + text[150:220] + wav_file(500, 2, 16, 42) + text[220:300] +
                                        jpeg_file(photo(64, 48, 3, 43), quality=60) + text[300:360] + b"P5\n48 40\n255\n" + photo(48, 40, 1, 44)[:, :, 0].tobytes() + text[360:500]),
        # TGA payloads (imgModel's second detector :5441-5481): true colour as the preprocessor frames it, grayscale / colour-mapped / 32-bit inside DEFAULT blocks
        "tga24_5k": preprocessed(text[:120] + tga_file(photo(48, 32, 3, 51), 2) + text[120:300]),
        "tga_gray_map_32_raw_9k": default_block(text[:80] + tga_file(photo(56, 40, 1, 52)[:, :, 0], 3) + text[80:160] + tga_file((photo(48, 36, 1, 53)[:, :, 0] >> 2), 1) + text[160:220] +
                                                tga_file(photo(32, 28, 4, 54), 2) + text[220:300]),
        # more JPEG shapes: 4:4:4 chroma, a progressive file (SOF2: the model must stay off), a file cut off in the middle of its scan followed by text
        "jpeg_444_prog_cut_6k": preprocessed(text[:100] + jpeg_file(photo(64, 48, 3, 55), quality=80, subsampling=0) + text[100:160] +
                                             jpeg_file(photo(64, 48, 3, 56), quality=70, progressive=True) + text[160:220] + jpeg_file(photo(80, 64, 3, 57), quality=75)[:1100] + text[220:500]),
        # a 32-bit PAM (the preprocessor makes an IMAGE32 block: the block path with alpha) and a JPEG whose APP1 segment holds a thumbnail JPEG (the parser's
        # embedded-image stack :6058-6063) inside a DEFAULT block
        "pam32_thumb_8k": preprocessed(text[:90] + b"P7\nWIDTH 40\nHEIGHT 30\nDEPTH 4\nMAXVAL 255\nTUPLTYPE RGB_ALPHA\nENDHDR\n" + photo(40, 30, 4, 61).tobytes() + text[90:200]) +
                          default_block(text[200:260] + jpeg_with_thumbnail(photo(64, 48, 3, 62), photo(24, 16, 3, 63)) + text[260:400]),
        # PCM audio (audio8bModel :5552-5657, wavModel :5659-5804, each followed by recordModel): WAV files as the preprocessor frames them
        "wav16s_6k": preprocessed(text[:200] + wav_file(1400, 2, 16, 9) + text[200:450]),
        "wav8s_4k": preprocessed(text[:150] + wav_file(1800, 2, 8, 10) + text[150:300]),
        "wav16m_3k": preprocessed(text[:100] + wav_file(1300, 1, 16, 11) + text[100:200]),
        "wav8m_2k": preprocessed(text[:100] + wav_file(1700, 1, 8, 12) + text[100:200]),
        # models with tables of their own INSIDE A TEXT BLOCK (the preprocessor lets a TEXT block run past the end of the text; paq8's own detectors switch the
        # models on, Stats.Type stays TEXT, so every such step ends in the text chain of final APM stages, :8281-8296): a WAV, a JPEG, a 4-bit and a 1-bit BMP
        "media_in_text_9k": text_block(text[:300] + wav_file(700, 1, 8, 71) + text[300:420] + jpeg_file(photo(64, 48, 3, 72), quality=60) + text[420:520] +
                                       bmp4_file((photo(64, 40, 1, 73)[:, :, 0] >> 4).astype(np.uint8), np.random.default_rng(74).integers(0, 256, (16, 3))) + text[520:600] +
                                       bmp1_file((photo(96, 48, 1, 75)[:, :, 0] > 120).astype(np.uint8)) + wav_file(400, 2, 16, 76) + text[600:700]),
        "bmp8_pal_raw_5k": default_block(text[:120] + bmp8_file(((photo(64, 52, 1, 6)[:, :, 0] >> 3).astype(np.uint8) * np.uint8(5)),
                                                                np.random.default_rng(8).integers(0, 256, (256, 3))) + text[120:300]),
    }


def streams():
    from cmix_amd import synth
    from make_golden import default_block, text_block
    wiki = (b"== History ==\nThe '''town''' of [[Example, Ohio|Example]] was founded in [[1820]].<ref name=\"c\">{{cite web|url=http://x.org|title=T}}</ref>\n"
            b"{| class=\"wikitable\"\n|-\n! Year !! Pop.\n|-\n| 1900 || 1,204\n|-\n| 1910 || 1,377\n|}\n* [[Category:Towns]]\n&lt;br&gt; &amp; caf\xc3\xa9 na\xc3\xafve\n")
    r = np.random.default_rng(5)
    rec = b"".join(bytes([i & 255, (i >> 8) & 255, 0, 0]) + bytes(r.integers(32, 48, 12, dtype=np.uint8)) for i in range(512))
    return {
        "text_32k": text_block(synth.enwik_like(32768 - 6, 4242)),
        "wiki_12k": text_block((wiki * 60)[:12288 - 6]),
        "records_8k": default_block(rec[:8192 - 5]),
        # what the reference's preprocessor made of a mixed file (tests/golden/make_dropin_mixed.py): a TEXT block, then an EXE
        # block (x86-like calls with rewritten addresses, records, text) -- block headers, type switches, exeModel on real targets
        "mixed_24k": bytes(np.load(os.path.join(HERE, "dropin_mixed.npz"))["stream"]),
        # the head of the bench shard (round 3: rich alphabet, V = 205): multi-byte UTF-8 from 46 script blocks, all of ASCII
        "rich_16k": text_block(synth.enwik_like(16384 - 6, 1000, rich=True)),
        "hdrs_4k": default_block(header_lookalikes()),
        # state injection: the history ring's index passes 2^30 in the middle of this stream (INJECT_POS); text with repeats, so that the match models hold
        # positions from in front of the wrap
        "pos_1g_6k": text_block((synth.enwik_like(2600, 88, rich=True) * 3)[:6144 - 6]),
        **image_streams(),
    }


INJECT_POS = {"pos_1g_6k": (1 << 30) - 3000}   # state injection (round 6): the stream starts with paq8's byte position 3000 below the end of its 2^30-byte ring


def reference_hashes(stream, pos=None):
    L = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libcmixrefpaq8.so"))
    L.refp8_predictor_new.restype = C.c_void_p
    L.refp8_predictor_new.argtypes = [C.c_int]
    L.refp8_predictor_update.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    h = L.refp8_predictor_new(11)
    if pos is not None:
        L.refp8_set_pos.argtypes = [C.c_int]
        L.refp8_set_pos(pos)
    bits = np.unpackbits(np.frombuffer(stream, np.uint8))
    out = np.full((len(bits), 1591), 0.5, np.float32)   # PAQ8::Predict() before the first Perceive: 0.5 everywhere
    for t in range(len(bits) - 1):
        L.refp8_predictor_update(h, int(bits[t]), out[t + 1].ctypes.data)
    return row_hash(out)


if __name__ == "__main__":
    for name, s in streams().items():
        # one reference predictor per process (it keeps state in globals)
        if len(sys.argv) > 1 and sys.argv[1] == name:
            extra = {"inject_pos": np.array([INJECT_POS[name]], np.int64)} if name in INJECT_POS else {}
            np.savez_compressed(os.path.join(HERE, "paq8_cols_%s.npz" % name), stream=np.frombuffer(s, np.uint8), hash=reference_hashes(s, INJECT_POS.get(name)), **extra)
            print(name, len(s), "bytes")
        elif len(sys.argv) == 1:
            import subprocess
            subprocess.check_call([sys.executable, os.path.abspath(__file__), name])
