"""Throughput mode with one PROCESS per stream (no shared HIP runtime locks, no GIL): S independent streams on ONE
GPU, each its own StreamPipeline (PPMd host stage + three device stages). Aggregate bytes/s = S * bytes / wall of
the slowest stream between a common start barrier and its last chunk. Companion of gpu_multistream.py (threads)."""
import json, os, sys, time
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R)
import multiprocessing as mp

STEPS, WARM, CHUNK = 4, 1, 1024


def child(seed, barrier, q):
    if os.environ.get("CMX_SPREAD") == "1":  # stream s: mixing network on XCD s, LSTM on XCD s+4 (mod 8)
        os.environ["CMX_MIXNET_XCD"] = str(seed % 8)
        os.environ["CMX_LSTM_XCD"] = str((seed + 4) % 8)
    import torch
    from cmix_amd.pipeline import StreamPipeline
    p = StreamPipeline(0, seed, CHUNK, STEPS + WARM)
    for i in range(WARM):
        p.step(i)
    p.sync()
    torch.cuda.synchronize()
    barrier.wait()
    t0 = time.perf_counter()
    for i in range(WARM, WARM + STEPS):
        p.step(i)
    p.sync()
    dt = time.perf_counter() - t0
    q.put((seed, dt, p.last_stage_ms()))
    p.close()


if __name__ == "__main__":
    counts = [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "2,4,8,16").split(",")]
    ctx = mp.get_context("spawn")
    res = []
    for S in counts:
        barrier, q = ctx.Barrier(S), ctx.Queue()
        ps = [ctx.Process(target=child, args=(1000 + s, barrier, q)) for s in range(S)]
        for p in ps: p.start()
        out = [q.get(timeout=600) for _ in ps]
        for p in ps: p.join()
        slow = max(o[1] for o in out)
        r = {"streams": S, "mode": "one process per stream", "bytes_per_s": S * STEPS * CHUNK / slow,
             "slowest_stream_s": slow, "fastest_stream_s": min(o[1] for o in out),
             "mixnet_ms_per_chunk": sum(o[2]["mixnet"] for o in out) / S, "lstm_ms_per_chunk": sum(o[2]["lstm"] for o in out) / S}
        print(json.dumps(r), flush=True)
        res.append(r)
    json.dump(res, open(os.path.join(R, "gpurun_out", "multiproc.json"), "w"), indent=1)
