"""Throughput mode with the FULL ensemble: S independent streams on ONE GPU, one host thread each (cmix_amd.multifile),
aggregate input bytes/s. Usage: python scripts/gpu_multistream_engine.py 1,2,3 [payload_bytes]"""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from cmix_amd import multifile, synth, shard
counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 17
out = {}
for S in counts:
    files = {"s%d" % i: synth.enwik_like(n, shard.shard_seed(0, i, 8)) for i in range(S)}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res, rep = multifile.compress_files(files, devices=[0] * S, step_bytes=1 << 15)
    dt = time.perf_counter() - t0
    # devices=[0]*S: the report is keyed by device, so only the last thread's entry survives; sizes are what matters here
    out[S] = {"streams": S, "bytes_each": n, "seconds": dt, "aggregate_bytes_per_s": S * n / dt, "sizes": [len(res[k]) for k in sorted(res)]}
    print(json.dumps(out[S]), flush=True)
