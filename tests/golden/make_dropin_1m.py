#!/usr/bin/env python3
"""SHA-256 and size of the file the unmodified reference binary writes for the first 1 MiB of the bench shard
(synth.enwik_like(1 << 20, 1000, rich=True): V = 205 distinct bytes, `cmix -c`): tests/golden/dropin_1m.npz. The parity
check SURVEY.md 8d prescribes for 100 MB shards ("the separately compressed 1 MiB prefix file"). About 50 minutes on one core.

    python tests/golden/make_dropin_1m.py [nbytes [seed]]   # other sizes -> dropin_rich_<n>k.npz, other seeds -> ..._s<seed>.npz
                                                            # (8 MiB of shard 1000: about 7.5 hours on one core)
"""
import hashlib
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from cmix_amd import synth
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    payload = synth.enwik_like(n, seed, rich=True)
    with tempfile.TemporaryDirectory() as d:
        a, b = os.path.join(d, "in"), os.path.join(d, "out")
        open(a, "wb").write(payload)
        t0 = time.time()
        subprocess.run([os.path.join(ROOT, "oracle", "_ref", "cmix_O3"), "-c", a, b], check=True,
                       stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        dt = time.time() - t0
        blob = open(b, "rb").read()
    name = "dropin_1m.npz" if (n, seed) == (1 << 20, 1000) else "dropin_rich_%dk%s.npz" % (n >> 10, "" if seed == 1000 else "_s%d" % seed)
    np.savez(os.path.join(ROOT, "tests", "golden", name), sha256=np.frombuffer(hashlib.sha256(blob).digest(), np.uint8),
             size=np.array([len(blob)], np.int64), seed=np.array([n, seed], np.int64), ref_seconds=np.array([dt]),
             vocab=np.array([len(set(payload))], np.int64), rich=np.array([1], np.int64))
    print(n, "->", len(blob), "bytes in", round(dt), "s; V =", len(set(payload)))
