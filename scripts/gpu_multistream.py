"""Aggregate throughput of S independent streams on ONE GPU (each stream = its own StreamPipeline: PPMd host
stage + three device stages, one CU per persistent stage kernel). Streams are independent, so this
is what a GPU does when it is given several files (config 4: 12 Silesia files on 8 GPUs) -- not the single-stream
`value` bench.py reports. One host thread per stream (the C ABI calls release the GIL)."""
import json, os, sys, threading, time
# every stage kernel is persistent for a whole chunk and pins a hardware queue: with the default 4 queues the
# streams of different pipelines queue up behind each other's 90 ms kernels
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import numpy as np, torch
from cmix_amd.pipeline import StreamPipeline

counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
steps, warm, chunk = 4, 1, 1024
res = []
for S in counts:
    pipes = [StreamPipeline(0, 1000 + s, chunk, steps + warm) for s in range(S)]
    torch.cuda.synchronize()
    def drive(p, lo, hi):
        for i in range(lo, hi):
            p.step(i)
    for p in pipes:
        drive(p, 0, warm)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    th = [threading.Thread(target=drive, args=(p, warm, warm + steps)) for p in pipes]
    for t in th: t.start()
    for t in th: t.join()
    t_host = time.perf_counter() - t0
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for p in pipes: p.sync()
    mix = float(np.mean([p.last_stage_ms()['mixnet'] for p in pipes]))
    r = {"streams": S, "bytes_per_s": S * steps * chunk / dt, "per_stream_bytes_per_s": steps * chunk / dt,
         "host_enqueue_s": t_host, "wall_s": dt, "mixnet_ms_per_chunk": mix,
         "algorithmic_GBps_mixnet": S * 3617849344 / (mix / 1e3) / 1e9}
    print(json.dumps(r), flush=True)
    res.append(r)
    for p in pipes: p.close()
    del pipes
    torch.cuda.empty_cache()
json.dump(res, open(os.path.join(R, "gpurun_out", "multistream.json"), "w"), indent=1)
