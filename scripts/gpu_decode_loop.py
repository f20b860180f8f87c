#!/usr/bin/env python3
"""How often does the decoder decode a file wrongly?  The 50 KB payload of tests/test_gpu_dropin.py's round trip is compressed once (cmix_dropin -c) and decoded
K times per setting (cmix_dropin -d); every wrong decode is reported with its first differing byte.

    python scripts/gpu_decode_loop.py 8 "" CMX_LATE_PULL=1        (K, then one setting per argument: "" = defaults, or NAME=VALUE[,NAME=VALUE])"""
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cmix_amd import synth  # noqa: E402

DROPIN = os.path.join(ROOT, "oracle", "_ref", "cmix_dropin")


def main():
    K = int(sys.argv[1])
    settings = sys.argv[2:] or [""]
    with np.load(os.path.join(ROOT, "tests", "golden", "dropin_vectors.npz")) as z:
        n, seed = (int(x) for x in z["text50k_c_seed"])
    payload = synth.enwik_like(n, seed)
    with tempfile.TemporaryDirectory() as d:
        src, blob, out = (os.path.join(d, x) for x in ("in", "blob", "out"))
        open(src, "wb").write(payload)
        subprocess.run([DROPIN, "-c", src, blob], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        print("compressed %d -> %d bytes" % (len(payload), os.path.getsize(blob)), flush=True)
        for st in settings:
            env = dict(os.environ)
            for kv in filter(None, st.split(",")):
                k, v = kv.split("=")
                env[k] = v
            bad = 0
            t0 = time.time()
            for r in range(K):
                p = subprocess.run([DROPIN, "-d", blob, out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=env)
                if p.returncode:
                    bad += 1
                    print("  [%s] run %d: exit code %d: %s" % (st or "defaults", r, p.returncode, p.stderr.decode(errors="replace")[-200:]), flush=True)
                    continue
                back = open(out, "rb").read()
                if back != payload:
                    bad += 1
                    m = min(len(back), len(payload))
                    k = next((i for i in range(m) if back[i] != payload[i]), m)
                    print("  [%s] run %d: WRONG from byte %d (decoded %d bytes)" % (st or "defaults", r, k, len(back)), flush=True)
            print("[%s] %d decodes of %d bytes, %d wrong, %.1f s each" % (st or "defaults", K, n, bad, (time.time() - t0) / K), flush=True)


if __name__ == "__main__":
    main()
