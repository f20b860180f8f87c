#!/usr/bin/env python3
"""bench.py -- throughput of the device-resident part of cmix's per-bit prediction path.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is
launched under torch.distributed.run, one rank per GPU. Prints ONE JSON line on rank 0.

What a "step" is: one pass of the hot path over one chunk of `--chunk-bytes` input
bytes (8 Predict+Perceive pairs per byte) of an enwik8-shaped stream, all operands
resident in HBM before the timed region. Streams are independent: rank r
processes its own shard (seed 1000+r) on its own GPU, no collective on the data path
("scaling": "weak"); value = bytes processed by all ranks / max-over-ranks time.

Device stages covered this round: the final mixing network -- stretch, 26+20+1 gated
logistic mixers with online update, squash, SSE (reference src/predictor.cpp:388-418,
432-437). The model families that feed it (paq8, fxcm, ppmd, lstm, small models) are not
on the device yet, so their 2078-wide per-bit prediction stream is a seeded synthetic
stand-in with the reference's value grid (k/4095) and enwik8-like selector locality;
`config.workload` says so. The number is therefore the throughput of this stage, not of a
whole predictor.

roofline: HBM-bound accounting per SURVEY.md 8(d)(i): 55 172 f32 weights x 8 B (read +
write) per bit = 3.53 MB per input byte for the final mixers, + 4 SSE cache lines per bit.
cpu_baseline: the plain-C oracle of the same stage (oracle/mixnet_oracle.c, "port") timed on
one host core over a bounded prefix of the same operands; `cpu_reference_full` additionally
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


def make_operands(nbytes, seed, device):
    """Seeded stand-in for the upstream model stages, generated on the device."""
    import torch
    from cmix_amd import synth
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    T = nbytes * 8
    text = np.frombuffer(synth.enwik_like(nbytes + 8, seed), np.uint8)[:nbytes]
    bits_np = np.unpackbits(text)  # MSB first, as runner.cpp:106-108 feeds the coder
    bits = torch.from_numpy(bits_np).to(device)
    k = torch.randint(0, 4096, (T, 2078), generator=g, device=device, dtype=torch.int32)
    conf = torch.rand((T, 2078), generator=g, device=device) < 0.5
    side = torch.rand((T, 2078), generator=g, device=device) < 0.5
    k = torch.where(conf, torch.where(side, k % 200, 4095 - (k % 200)), k)
    probs = k.to(torch.float32) * np.float32(1.0 / 4095)
    probs[:, 2025:2078] = torch.rand((T, 53), generator=g, device=device)
    probs[:, 432:434] = 0.5
    # selector keys from the actual bytes: order-0/1/2 partial-byte contexts, byte classes ...
    b = text.astype(np.int64)
    prev1 = np.concatenate([[0], b[:-1]])
    prev2 = np.concatenate([[0, 0], b[:-2]])
    prev3 = np.concatenate([[0, 0, 0], b[:-3]])
    lbc = np.ones(T, np.int64)
    for j in range(1, 8):
        lbc[j::8] = lbc[j - 1::8] * 2 + bits_np[j - 1::8]
    rep = lambda a: np.repeat(a, 8)
    sel = np.zeros((T, 47), np.int64)
    percol = {0: lbc, 1: lbc, 2: (rep(prev1) << 8) + lbc, 3: (rep(prev1) << 8) + lbc,
              4: (rep(prev1 & 15) << 12) + (rep(prev2 & 15) << 8) + lbc, 5: (rep(prev1 & 3) << 8) + lbc,
              6: rep(prev3), 7: rep(prev3), 8: 0 * lbc, 9: rep(np.arange(nbytes, dtype=np.int64) % 100),
              10: rep(prev1 & 7), 11: rep((prev1 << 8) + prev2), 13: rep(prev1 >> 3), 14: rep(prev1 >> 2),
              15: rep(prev1 >> 5), 16: (rep(prev1 >> 5) << 8) + lbc, 17: rep((prev1 >> 6) + 4 * (prev2 >> 6)),
              18: rep((prev1 >> 6) + 4 * (prev2 >> 6) + 16 * (prev3 >> 6)), 19: (rep(prev1 >> 6) << 8) + lbc,
              20: rep(prev1 >> 5), 21: rep((prev1 >> 5) + 8 * (prev2 >> 5)), 22: (rep(prev1 >> 5) << 8) + lbc,
              23: (rep(prev2) << 8) + lbc, 24: rep(prev1 + 256 * prev2), 25: rep(prev2 + 256 * prev3)}
    for m, v in percol.items():
        sel[:, m] = v
    l1 = [0 * lbc, 0 * lbc, lbc, lbc, lbc, rep(prev1), rep(prev2), rep(prev3), rep(prev1 & 7),
          rep((prev1 << 8) + prev2), rep(prev1 >> 3), rep(prev1 >> 2), rep(prev1 >> 5),
          rep((prev1 >> 6) + 4 * (prev2 >> 6)), rep((prev1 >> 6) + 4 * (prev2 >> 6) + 16 * (prev3 >> 6)),
          rep(prev1 >> 5), rep((prev1 >> 5) + 8 * (prev2 >> 5)), (rep(prev1 >> 6) << 8) + lbc,
          (rep(prev1 >> 5) << 8) + lbc, (rep(prev1 >> 5) << 8) + lbc]
    for j, v in enumerate(l1):
        sel[:, 26 + j] = v
    sel32 = torch.from_numpy((sel & 0xFFFFFFFF).astype(np.uint32).view(np.int32)).to(device)
    return probs.contiguous(), sel32.contiguous(), bits.contiguous(), text


def cpu_baseline_port(probs, sel32, bits, budget_s=12.0):
    """Time the plain-C oracle of the same stage on one host core over a bounded prefix."""
    from oracle import oracle as O
    n = min(len(bits), 4096)
    p = probs[:n].cpu().numpy()
    s = sel32[:n].cpu().numpy().view(np.uint32).astype(np.uint64)
    b = bits[:n].cpu().numpy()
    net = O.MixNet()
    t0 = time.perf_counter()
    done = 0
    while done < n and time.perf_counter() - t0 < budget_s:
        net.step(p[done], s[done], b[done])
        done += 1
    dt = time.perf_counter() - t0
    return {"value": (done / 8.0) / dt, "unit": "input bytes/s", "cores": 1, "kind": "port",
            "sample": f"first {done} bits of the same stage operands through oracle/mixnet_oracle.c "
                      f"(includes ~{5e-6 * done / dt * 100:.0f}% ctypes call overhead)"}


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
    ap.add_argument("--steps", type=int, default=4)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--chunk-bytes", type=int, default=1024)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cmix_amd import engine as E

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
    probs, sel, bits, text = make_operands(a.chunk_bytes * nsteps, 1000 + rank, dev)
    cb = a.chunk_bytes * 8
    net = E.MixNet(local)
    stream = torch.cuda.current_stream(dev)
    p_out = torch.empty(cb * nsteps, dtype=torch.float32, device=dev)

    def step(i):
        s = slice(i * cb, (i + 1) * cb)
        net.run(probs[s], sel[s], bits[s], p_out[s])

    for i in range(a.warmup):
        step(i)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    kernel_ms = []
    t0 = time.perf_counter()
    for i in range(a.warmup, nsteps):
        step(i)
        kernel_ms.append(net.last_kernel_ms())  # HIP events on the launch stream (syncs this chunk)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    if rank == 0:
        total_bytes = a.chunk_bytes * a.steps * world
        avg_kernel_s = float(np.mean(kernel_ms)) / 1e3
        algo = ALGO_BYTES_PER_INPUT_BYTE * a.chunk_bytes
        achieved = algo / avg_kernel_s / 1e9
        out = {
            "metric": "input bytes/s on enwik8-shaped text (device stages only, see config.workload)",
            "value": total_bytes / dt, "unit": "input bytes/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "S-enwik8 shard (seed 1000+rank), %d-byte chunks; device stage = final mixing "
                            "network (stretch + 26/20/1 mixers + SSE, strict bit-exact mode); model-prediction "
                            "stream (paq8/fxcm/ppmd/lstm/small models) is a seeded synthetic stand-in: those "
                            "stages are not on the device yet" % a.chunk_bytes,
                "chunk_bytes": a.chunk_bytes, "streams_per_gpu": 1, "parallelism": "1 stream per GPU, no collective"},
            "us_per_bit": dt / (a.steps * cb) * 1e6,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "cmx_mixnet_chunk_kernel", "avg_kernel_ms": avg_kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": algo},
        }
        if not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_port(probs, sel, bits)
            ref = cpu_reference_full(text)
            if ref:
                out["cpu_reference_full"] = ref
        print(json.dumps(out))
    net.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
