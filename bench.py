#!/usr/bin/env python3
"""bench.py -- throughput of the device-resident part of cmix's per-bit prediction path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched under torch.distributed.run, one rank per GPU. Prints ONE JSON line on rank 0.

What a "step" is: one pass of the hot path over one chunk of `--chunk-bytes` input
bytes (8 Predict+Perceive pairs per byte) of an enwik8-shaped stream, all operands
resident in HBM before the timed region. Streams are independent: rank r
processes its own shard (seed 1000+r) on its own GPU, no collective on the data path
("scaling": "weak"); value = bytes processed by all ranks / max-over-ranks time.

Device stages covered: (1) ContextManager + the 54 byte contexts + the 54 small native models
(reference src/context-manager.cpp, src/contexts, src/models/{direct,direct-hash,indirect,match,
bracket}.cpp) -> layer-0 columns 0-2, 2025-2075 and the 47 mixer selectors; (2) the byte-level
LSTM byte mixer (src/mixer/{byte-mixer,lstm,lstm-layer}.cpp) -> column 2077; (3) the final mixing
network -- stretch, 26+20+1 gated logistic mixers with online update, squash, SSE
(src/predictor.cpp:388-418,432-437). Each stage runs on its own HIP stream; a chunk's mixing
network starts when the chunk's other two stages have written their columns. PPMd (byte distribution
feeding the LSTM and column 2076) is the engine's host stage (cmix_amd/csrc/ppmd_host.cpp): it runs
on one host core inside the timed loop. paq8 and fxcm (layer-0 columns 3..2024) have no stage yet: a
seeded stand-in of the same shape and value grid (k/4095) replaces them, and `config.workload` says so. The number is the throughput of these device stages, not of a whole
predictor.

roofline: HBM-bound accounting per SURVEY.md 8(d)(i): 55 172 f32 weights x 8 B (read +
write) per bit = 3.53 MB per input byte for the final mixers, + 4 SSE cache lines per bit.
cpu_baseline: the plain-C oracle of the same three stages (oracle/*.c, "port") timed on one host
core over bounded prefixes of the same operands (us/bit of the stages add up on a CPU); `cpu_reference_full` additionally
times the unmodified reference binary (whole predictor, oracle/_ref/cmix_O3) on a short
prefix of the same shard for context.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALGO_BYTES_PER_INPUT_BYTE = 55172 * 8 * 8 + 4 * 64 * 8  # SURVEY.md 8(d)(i) + SSE lines
HBM_PEAK_GBS = 8000.0


def cpu_baseline_port(probs, bits, text, vocab, budget_s=14.0):
    """Time the plain-C oracle of the same three device stages on one host core over a bounded prefix.
    cmix is single-threaded, so the stages run back to back on the CPU: us/bit adds up."""
    from oracle import oracle as O
    from cmix_amd import engine as E
    nb = min(len(text), 512)
    ctx = O.CtxModels(vocab)
    t0 = time.perf_counter()
    _, sel = ctx.run(bytes(text[:nb]))
    us_ctx = (time.perf_counter() - t0) / (8 * nb) * 1e6
    n = 8 * nb
    p = probs[:n].cpu().numpy()
    b = bits[:n].cpu().numpy()
    net = O.MixNet()
    t0 = time.perf_counter()
    done = 0
    while done < n and time.perf_counter() - t0 < budget_s / 2:
        net.step(p[done], sel[done], b[done])
        done += 1
    us_mix = (time.perf_counter() - t0) / done * 1e6
    pp = E.Ppmd(vocab).run(bytes(text[:nb]))  # the engine's own host stage supplies the LSTM's input
    lstm = O.Lstm(vocab)
    t0 = time.perf_counter()
    k = 0
    while k < nb and time.perf_counter() - t0 < budget_s / 2:
        lstm.byte_update(pp[k], text[k])
        k += 1
    us_lstm = (time.perf_counter() - t0) / (8 * k) * 1e6
    tot = us_mix + us_ctx + us_lstm
    return {"value": 1e6 / (8 * tot), "unit": "input bytes/s", "cores": 1, "kind": "port",
            "sample": f"oracle/*.c of the same three device stages on one core: mixing network {done} bits "
                      f"({us_mix:.1f} us/bit), contexts+small models {8 * nb} bits ({us_ctx:.1f} us/bit), "
                      f"LSTM {k} bytes ({us_lstm:.1f} us/bit); stages run back to back on a CPU",
            "us_per_bit": {"mixnet": us_mix, "ctxmodels": us_ctx, "lstm": us_lstm}}


def measured_traffic(chunk_bytes):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same command
    (profiles/r01_pmc_bench.json: `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate runs,
    scripts/gpu_pmc_bench.sh). FETCH_SIZE is doubled: on gfx950 it reports half of the bytes of 16 B/lane
    reads (MI355X_MICROARCH.md, HBM section); WRITE_SIZE matched a known 503 MB fill to 1.5 %. Counters cannot
    be collected inside a timed run, so this is a recorded measurement, valid for 1024-byte chunks."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_bench.json")
    if chunk_bytes != 1024 or not os.path.exists(path):
        return None
    k = json.load(open(path)).get("cmx_mixnet_chunk_kernel")
    if not k:
        return None
    rd = 2.0 * k["FETCH_SIZE"]["sum_kb"] * 1024 / k["FETCH_SIZE"]["launches"]
    wr = k["WRITE_SIZE"]["sum_kb"] * 1024 / k["WRITE_SIZE"]["launches"]
    return rd + wr


def cpu_reference_full(text, nbytes=4096):
    exe = os.path.join(ROOT, "oracle", "_ref", "cmix_O3")
    if not os.path.exists(exe):
        return None
    with tempfile.TemporaryDirectory() as d:
        src, dst = os.path.join(d, "in"), os.path.join(d, "out")
        with open(src, "wb") as f:
            f.write(bytes(text[:nbytes]))
        t0 = time.perf_counter()
        try:
            subprocess.run([exe, "-c", src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=120, check=True)
        except Exception as e:  # noqa: BLE001
            return {"error": str(e)}
        dt = time.perf_counter() - t0
        return {"value": nbytes / dt, "unit": "input bytes/s", "cores": 1, "kind": "reference",
                "sample": f"cmix_O3 -c on the first {nbytes} bytes of the same shard, whole predictor, "
                          f"wall of main() incl. ~4 s construction; compressed to {os.path.getsize(dst)} bytes"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=48)   # 48 KB per stream, ~5 s timed: the ~95 ms pipeline fill
    ap.add_argument("--warmup", type=int, default=2)   # (PPMd + LSTM of the first timed chunk) weighs 2 %
    ap.add_argument("--chunk-bytes", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fxcm-device", action="store_true",
                    help="also run the fxcm family as a device stage (columns 3..433) instead of its stand-in; off by default until the stage has been timed on a GPU")
    ap.add_argument("--streams-per-gpu", type=int, default=1,
                    help="independent input streams per GPU (throughput mode for multi-file jobs); the headline "
                         "metric is 1 stream per GPU")
    a = ap.parse_args()
    if a.streams_per_gpu > 1:  # persistent stage kernels pin hardware queues: give every stream's stages their own
        os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
        if a.streams_per_gpu > 5:  # ~16 queues run unsliced (profiles/r01_multiproc.txt): two HIP streams per pipeline
            os.environ.setdefault("CMX_PIPELINE_STREAMS", "2")

    import torch
    import torch.distributed as dist
    from cmix_amd import engine as E
    from cmix_amd import shard

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    nsteps = a.warmup + a.steps
    cb = a.chunk_bytes * 8
    from cmix_amd.pipeline import StreamPipeline
    S = a.streams_per_gpu
    pipes = [StreamPipeline(local, shard.shard_seed(rank, s, S), a.chunk_bytes, nsteps, fxcm_device=a.fxcm_device) for s in range(S)]
    pipe = pipes[0]  # rank r = GPU r owns streams r*S .. r*S+S-1

    def step(i):
        if S == 1:
            pipe.step(i)
            return
        import threading  # one host thread per stream: PPMd and the launches of different streams overlap
        th = [threading.Thread(target=p.step, args=(i,)) for p in pipes]
        for t in th:
            t.start()
        for t in th:
            t.join()
    torch.cuda.synchronize()

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    for p in pipes:
        p.sync()
        p.stage_totals(reset=True)  # the stage timers below cover exactly the timed chunks
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.warmup, nsteps):
        step(i)  # everything is enqueued asynchronously: chunk i+1's context/LSTM stages run under chunk i's mixing
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    for p in pipes:
        p.sync()
    total_bytes, dt, _ = shard.aggregate_throughput(S * a.chunk_bytes * a.steps, dt, dev)  # sum of bytes / max of times
    st = pipe.stage_totals()  # HIP events around each stage of every timed chunk, on the stage's own stream: means
    mix_ms, ctx_ms, lstm_ms = st["mixnet"], st["ctxmodels"], st["lstm"]
    assert st["chunks"] == a.steps, st

    if rank == 0:
        avg_kernel_s = mix_ms / 1e3
        algo = ALGO_BYTES_PER_INPUT_BYTE * a.chunk_bytes
        achieved = algo / avg_kernel_s / 1e9
        out = {
            "metric": "input bytes/s on enwik8-shaped text (engine stages built so far, see config.workload)",
            "value": total_bytes / dt, "unit": "input bytes/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "S-enwik8 shard (seed 1000+rank), %d-byte chunks through the three device stages: "
                            "contexts + 54 small models (layer-0 columns 0-2, 2025-2075, all 47 selectors), "
                            "byte-level LSTM (column 2077) and the final mixing network (stretch + 26/20/1 mixers + "
                            "SSE), strict bit-exact mode, fed by the PPMd host stage (one host core, inside the timed loop). "
                            "No stage yet, replaced by a seeded stand-in of the same shape: the paq8 and fxcm "
                            "columns (3..2024)" % a.chunk_bytes,
                "chunk_bytes": a.chunk_bytes, "streams_per_gpu": S,
                "parallelism": "%d stream%s per GPU, no collective" % (S, "" if S == 1 else "s")},
            "us_per_bit": dt / (a.steps * cb) * 1e6,  # wall per bit of ONE stream
            "stage_us_per_bit": {"mixnet": avg_kernel_s / cb * 1e6, "ctxmodels": ctx_ms * 1e3 / cb,
                                 "lstm": lstm_ms * 1e3 / cb,
                                 "note": "mean HIP-event time of each stage over the timed chunks (stages overlap on separate streams)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": measured_traffic(a.chunk_bytes),
                         "kernel": "cmx_mixnet_chunk_kernel", "avg_kernel_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": algo},
        }
        if a.fxcm_device:
            out["stage_us_per_bit"]["fxcm"] = pipe.pipe.fxcm_total_ms() / a.steps * 1e3 / cb
            out["config"]["workload"] += "; --fxcm-device: fxcm's columns (3..433) come from the fxcm device stage (host text parser + cmx_fxcm_chunk_kernel), only paq8's are a stand-in"
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_port(pipe.probs, pipe.bits, pipe.text, pipe.vocab)
            ref = cpu_reference_full(pipe.text)
            if ref:
                out["cpu_reference_full"] = ref
        print(json.dumps(out))
    for p in pipes:
        p.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
