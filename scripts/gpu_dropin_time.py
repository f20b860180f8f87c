#!/usr/bin/env python3
"""Same box, same payload: the unmodified reference binary vs the drop-in build (reference CLI + integration/predictor.h
+ libcmixamd.so), compress and decompress, wall time. Writes gpurun_out/dropin_time.txt."""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
v = np.load(os.path.join(ROOT, "tests", "golden", "dropin_vectors.npz"))
payload = v["text12k_c_payload"].tobytes()
lines = []
with tempfile.TemporaryDirectory() as d:
    src = os.path.join(d, "in")
    open(src, "wb").write(payload)
    # process start-up (model allocation) measured on a 1-byte file, subtracted below
    one = os.path.join(d, "one")
    open(one, "wb").write(b"a")
    for exe in ("cmix_O3", "cmix_hybrid"):
        path = os.path.join(ROOT, "oracle", "_ref", exe)
        t0 = time.time(); subprocess.run([path, "-c", one, os.path.join(d, "o1")], check=True, capture_output=True); t_init = time.time() - t0
        out = os.path.join(d, exe + ".cmix")
        t0 = time.time(); subprocess.run([path, "-c", src, out], check=True, capture_output=True); t_c = time.time() - t0
        back = os.path.join(d, exe + ".back")
        t0 = time.time(); subprocess.run([path, "-d", out, back], check=True, capture_output=True); t_d = time.time() - t0
        ok = open(back, "rb").read() == payload and open(out, "rb").read() == v["text12k_c_file"].tobytes()
        n = len(payload)
        lines.append(f"{exe:12s} start-up {t_init:6.2f} s | compress {t_c:6.2f} s ({(t_c - t_init) / n * 1e3:.3f} ms/byte net) | "
                     f"decompress {t_d:6.2f} s ({(t_d - t_init) / n * 1e3:.3f} ms/byte net) | byte-identical+round-trip: {ok}")
    # the look-ahead compressor (chunk pipeline): compress only; its files are decompressed by the other two builds
    path = os.path.join(ROOT, "oracle", "_ref", "cmix_lookahead")
    if os.path.exists(path):
        from cmix_amd import synth
        big = os.path.join(d, "big")
        open(big, "wb").write(synth.enwik_like(50000, 91))
        t0 = time.time(); subprocess.run([path, "-c", one, os.path.join(d, "o1")], check=True, capture_output=True); t_init = time.time() - t0
        out = os.path.join(d, "la.cmix")
        t0 = time.time(); subprocess.run([path, "-c", src, out], check=True, capture_output=True); t_c = time.time() - t0
        ok = open(out, "rb").read() == v["text12k_c_file"].tobytes()
        t0 = time.time(); subprocess.run([path, "-c", big, os.path.join(d, "la50.cmix")], check=True, capture_output=True); t_b = time.time() - t0
        import hashlib
        ok50 = hashlib.sha256(open(os.path.join(d, "la50.cmix"), "rb").read()).digest() == v["text50k_c_sha256"].tobytes()
        lines.append(f"{'cmix_lookahead':12s} start-up {t_init:6.2f} s | compress  {t_c:6.2f} s ({(t_c - t_init) / len(payload) * 1e3:.3f} ms/byte net) | "
                     f"50 000 bytes: {t_b:6.2f} s ({(t_b - t_init) / 50000 * 1e3:.3f} ms/byte net) | byte-identical: {ok} / {ok50}")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
txt = f"payload: {len(payload)} bytes of seeded text (tests/golden/dropin_vectors.npz text12k_c), `cmix -c` / `cmix -d`\n" + "\n".join(lines) + "\n"
open(os.path.join(ROOT, "gpurun_out", "dropin_time.txt"), "w").write(txt)
print(txt)
