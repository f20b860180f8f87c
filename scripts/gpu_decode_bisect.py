#!/usr/bin/env python3
"""Which build decodes the 50 KB round trip of tests/test_gpu_dropin.py wrongly, and is it the same every time?

    python scripts/gpu_decode_bisect.py [libdir ...]     (each libdir holds a libcmixamd.so of another commit; the tree's own build is always run first, twice)

For every library: oracle/_ref/cmix_dropin -c of the payload (is the file the unmodified reference binary's? -- cmix_O3 -c runs beside it on a host core), then
cmix_dropin -d of that file against the payload: the first differing byte, if any."""
import hashlib
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
REF = os.path.join(ROOT, "oracle", "_ref", "cmix_O3")


def run(exe, mode, data, d, tag, env=None, wait=True):
    src, out = os.path.join(d, tag + ".in"), os.path.join(d, tag + ".out")
    with open(src, "wb") as f:
        f.write(data)
    e = dict(os.environ)
    e.update(env or {})
    p = subprocess.Popen([exe, mode, src, out], stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=e)
    if not wait:
        return p, out
    _, err = p.communicate(timeout=1500)
    if p.returncode:
        print("   %s %s failed: %s" % (os.path.basename(exe), mode, err.decode(errors="replace")[-300:]), flush=True)
        return None
    return open(out, "rb").read()


def main():
    with np.load(os.path.join(ROOT, "tests", "golden", "dropin_vectors.npz")) as z:
        n, seed = (int(x) for x in z["text50k_c_seed"])
    payload = synth.enwik_like(n, seed)
    libs = [("tree", None), ("tree (again)", None)] + [(os.path.basename(os.path.normpath(p)), os.path.abspath(p)) for p in sys.argv[1:]]
    extra = [("tree, CMX_LATE_PULL=1", None, {"CMX_LATE_PULL": "1"}), ("tree, CMX_LATE_LSTM_PER_BYTE=1", None, {"CMX_LATE_LSTM_PER_BYTE": "1"})]
    with tempfile.TemporaryDirectory() as d:
        refp = None
        if os.path.exists(REF):
            refp, refout = run(REF, "-c", payload, d, "ref", wait=False)
        blobs = {}
        for name, path, *more in [(a, b) for a, b in libs] + extra:
            env = dict(more[0]) if more else {}
            if path:
                env["LD_LIBRARY_PATH"] = path + ":" + os.environ.get("LD_LIBRARY_PATH", "")
            t0 = time.time()
            blob = run(DROPIN, "-c", payload, d, "c", env)
            t1 = time.time()
            if blob is None:
                continue
            blobs[name] = blob
            back = run(DROPIN, "-d", blob, d, "d", env)
            t2 = time.time()
            if back is None:
                continue
            if back == payload:
                verdict = "round trip OK"
            else:
                m = min(len(back), len(payload))
                k = next((i for i in range(m) if back[i] != payload[i]), m)
                verdict = "DECODED WRONGLY: first differing byte %d of %d (decoded length %d)" % (k, len(payload), len(back))
            print("%-32s -c %6d bytes sha %s (%.0f s)   -d %.0f s: %s" % (name, len(blob), hashlib.sha256(blob).hexdigest()[:16], t1 - t0, t2 - t1, verdict), flush=True)
        if refp:
            refp.communicate(timeout=1500)
            ref = open(refout, "rb").read()
            print("%-32s -c %6d bytes sha %s" % ("cmix_O3 (reference binary)", len(ref), hashlib.sha256(ref).hexdigest()[:16]))
            for k, b in blobs.items():
                print("   %-32s file %s the reference binary's" % (k, "==" if b == ref else "!="))
            # does the tree's decoder decode the REFERENCE's file?
            back = run(DROPIN, "-d", ref, d, "dref")
            print("   tree -d of the reference binary's file: %s" % ("OK" if back == payload else "WRONG"))


if __name__ == "__main__":
    main()
