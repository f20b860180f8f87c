"""Per-chunk host timing of the engine's submit loop (debug aid): where does the submitting thread block?"""
import os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from cmix_amd import synth, shard
from cmix_amd.pipeline import EngineStream, text_file_stream
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 18
stream = text_file_stream(synth.enwik_like(n, 1000))
eng = EngineStream(0, stream, 4096)
torch.cuda.synchronize()
prev = eng.pipe.host_ms(); t0 = time.perf_counter(); tp = t0
k = 0
while eng.pos < len(stream):
    eng.feed(4096)
    now = time.perf_counter(); h = eng.pipe.host_ms()
    d = {a: h[a] - prev[a] for a in h}
    print("chunk %3d  t=%8.1f ms  dt=%7.1f  " % (k, (now - t0) * 1e3, (now - tp) * 1e3) + "  ".join("%s=%.1f" % (a[:9], v) for a, v in d.items()), flush=True)
    prev, tp, k = h, now, k + 1
blob = eng.finish()
print("total %.1f s, %d bytes out" % (time.perf_counter() - t0, len(blob)))
