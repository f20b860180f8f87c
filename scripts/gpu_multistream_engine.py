"""Throughput mode with the FULL ensemble: S independent streams on ONE GPU, one host thread each, aggregate input bytes/s.
Every stream compresses the same payload (the bench shard's first N bytes), so every file can be checked against the one-stream result.
The engines are constructed before the clock starts.   Usage: python scripts/gpu_multistream_engine.py 1,2,3 [payload_bytes]
(CMX_MIXNET_SPEC=0 selects the one-workgroup mixing-network kernel: 68 instead of 94 workgroups per stream.)"""
import hashlib, json, os, sys, threading, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import torch
from cmix_amd import synth, shard
from cmix_amd.pipeline import EngineStream, text_file_stream
counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2").split(",")]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 17
payload = synth.enwik_like(n, shard.shard_seed(0), rich=True)
stream = text_file_stream(payload)
want = None
for S in counts:
    engines = [EngineStream(0, stream, 4096) for _ in range(S)]
    torch.cuda.synchronize()
    blobs = [None] * S
    start = threading.Barrier(S + 1)
    def work(i):
        start.wait()
        e = engines[i]
        fed = 0
        while fed < len(stream):
            e.feed(1 << 15); fed += 1 << 15
        blobs[i] = e.finish()
    th = [threading.Thread(target=work, args=(i,)) for i in range(S)]
    for t in th: t.start()
    start.wait()
    t0 = time.perf_counter()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    shas = [hashlib.sha256(b).hexdigest()[:16] for b in blobs]
    want = want or shas[0]
    print(json.dumps({"streams": S, "bytes_each": len(stream), "seconds": round(dt, 3), "aggregate_bytes_per_s": round(S * len(stream) / dt), "mixnet_spec": os.environ.get("CMX_MIXNET_SPEC", "1"),
                      "file_bytes": len(blobs[0]), "sha256_16": shas[0], "all_files_identical": all(s == want for s in shas)}), flush=True)
    for e in engines:
        e.close() if hasattr(e, "close") else None
    del engines
    torch.cuda.empty_cache()
