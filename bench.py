#!/usr/bin/env python3
"""bench.py -- input bytes/s of the whole cmix v21 predictor on enwik8-shaped text, on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU. Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1], "enwik8 full model ensemble on 1 MI355X", at the size the parity fixture holds):
rank r compresses the first --payload-bytes (default 1 MiB) of its own S-enwik8 shard (cmix_amd.synth.enwik_like,
seed 1000 + r) exactly as `cmix -c` would: the stream the reference's preprocessor hands the predictor (one TEXT
block), every bit through the FULL ensemble -- all 2078 layer-0 inputs come from engine stages on the GPU (contexts + 54
small models, LSTM, fxcm's 431 and paq8's 1591 outputs; PPMd and the two text parsers are the engine's host stages,
inside the timed loop) -- and the final mixing network + SSE, strict bit-exact mode; then the arithmetic coder. A
"step" is 1/K of the stream (fed in 4 KB sub-chunks so that the stages of consecutive sub-chunks overlap). Inputs are
in host memory when the timed region starts (the boundary hands over bytes; 1 byte in, 1 float out per bit crosses
PCIe); value = stream bytes of all ranks / max-over-ranks time of the K steps + coder. The W warm-up steps run the same
code on a throw-away engine instance over the head of the same shard. Streams are independent: no
collective on the data path ("scaling": "weak").

verified: the timed run's OUTPUT FILE (header + code) is compared with the SHA-256 / size of the file the unmodified
reference binary wrote for the same payload (tests/golden/dropin_*.npz, tests/golden/make_dropin_1m.py) -- compressed
-size parity in its strongest form -- on rank 0; a payload size without a fixture reports the size only.

roofline: per SURVEY.md 8(d): algorithmic HBM bytes of the slowest stage's dominant kernel / its HIP-event time.
cpu_baseline (kind "reference"): the unmodified reference binary (oracle/_ref/cmix_O3 -c) on a bounded prefix of the
same shard, construction excluded by differencing two runs, on one pinned host core.
"""
import argparse
import hashlib
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
# every stage of a stream runs on its own HIP stream (9 per stream); HIP multiplexes streams onto 4 hardware queues by
# default, which would serialise stage kernels that are meant to overlap: ask for one queue per stream before HIP starts
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0
# algorithmic HBM bytes per input byte: weights touched per bit x 8 B (read + write) x 8 bits
ALGO = {
    "mixnet": 55172 * 8 * 8 + 4 * 64 * 8,                 # final mixers (f32) + 4 SSE lines per bit
    "paq8": (28 * 1552 + 32) * 2 * 2 * 8 + 267 * 3 * 64 * 2,  # paq8 mixer rows (i16, read + write) + bucket probes
    "fxcm": (10 * 512 + 2 * 16) * 2 * 2 * 8 + 81 * 3 * 64 * 2,  # fxcm mixers + bucket probes
    "lstm": 8.86e6,                                       # gate weights forward + BPTT share per byte
    "ctxmodels": 54 * 8 * 64,
}
# HBM traffic per input byte of each stage's kernels, from the PMC passes of this bench command (profiles/r02_pmc_bench_64k.json:
# rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, KB summed over the launches; FETCH_SIZE doubled as
# /opt/skills/guides/MI355X_MICROARCH.md prescribes for 16-byte-per-lane loads on gfx950); that run processed 81 926 stream bytes
# (65 542 timed + 16 384 warm-up) in 22 launches per stage
PMC_KB = {"mixnet": (28218.2e3, 54742.7e3, 22), "fxcm": (1442.8e3, 2832.8e3, 22), "lstm": (46764.0e3 + 7836.3e3 + 4565.0e3, 904.9e3 + 10470.9e3 + 21056.7e3, 22),
          "paq8": (9444.5e3, 26357.6e3, 22), "ctxmodels": (379.5e3, 679.1e3, 22)}   # paq8: its mixer kernel (the role that sets the stage's period)


def pmc_traffic_per_byte(stage):
    f, w, _launches = PMC_KB[stage]
    return (2.0 * f + w) * 1024.0 / 81926.0


KERNEL = {"mixnet": "cmx_mixnet_chunk_kernel", "paq8": "cmx_p8s_mix2_kernel / cmx_p8s_fam2_kernel (slowest role)", "fxcm": "cmx_fxcm_chunk_kernel",
          "lstm": "cmx_lstm_fwd / cmx_lstm_bptt_*", "ctxmodels": "cmx_ctxmodels_kernel"}


def cpu_reference(payload, nbytes, core):
    """The unmodified reference binary on the first nbytes of the shard, minus a 256-byte run (construction: ~4 s)."""
    exe = os.path.join(ROOT, "oracle", "_ref", "cmix_O3")
    if not os.path.exists(exe):
        return None
    pin = ["taskset", "-c", str(core)] if subprocess.call(["which", "taskset"], stdout=subprocess.DEVNULL) == 0 else []

    def run(n):
        with tempfile.TemporaryDirectory() as d:
            src, dst = os.path.join(d, "in"), os.path.join(d, "out")
            with open(src, "wb") as f:
                f.write(bytes(payload[:n]))
            t0 = time.perf_counter()
            subprocess.run(pin + [exe, "-c", src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600, check=True)
            return time.perf_counter() - t0, os.path.getsize(dst)
    try:
        t_small, _ = run(256)
        t_big, size = run(256 + nbytes)
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}
    dt = max(t_big - t_small, 1e-9)
    return {"value": nbytes / dt, "unit": "input bytes/s", "cores": 1, "kind": "reference",
            "sample": f"oracle/_ref/cmix_O3 -c (unmodified reference, g++ -O3) on the first {256 + nbytes} bytes of the same shard "
                      f"minus a 256-byte run ({t_small:.1f} s: construction), whole predictor + coder, "
                      f"{'taskset -c %d' % core if pin else 'unpinned'}; {dt:.1f} s for {nbytes} bytes; file {size} bytes"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--payload-bytes", type=int, default=1 << 20, help="bytes of the shard each rank compresses (fixtures: 65536, 262144, 1048576)")
    ap.add_argument("--sub-chunk", type=int, default=4096, help="bytes per pipeline submit (stages of consecutive sub-chunks overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-bytes", type=int, default=16384)
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    from cmix_amd import shard, synth
    from cmix_amd.pipeline import EngineStream, text_file_stream

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:  # bookkeeping only (barrier, max of times): gloo, the data path has no collective
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback exists for the product path)")
    torch.cuda.set_device(local)

    payload = synth.enwik_like(a.payload_bytes, shard.shard_seed(rank))
    stream = text_file_stream(payload)
    n = len(stream)
    # a step = 1/K of the stream's sub-chunks (whole sub-chunks: a ragged step would put a few-byte chunk into the pipeline)
    nsub_total = -(-n // a.sub_chunk)
    edges = [min(n, (i * nsub_total // a.steps) * a.sub_chunk) for i in range(a.steps)] + [n]
    step_sizes = [edges[i + 1] - edges[i] for i in range(a.steps)]
    step_bytes = max(step_sizes)

    # ---- warm-up: the same code on a throw-away engine over the head of the shard ----
    if a.warmup > 0:
        wn = min(n, a.warmup * step_bytes)
        w = EngineStream(local, stream[:wn], a.sub_chunk)
        for _ in range(a.warmup):
            w.feed(step_bytes)
        w.finish()
        w.close()
        del w
    eng = EngineStream(local, stream, a.sub_chunk)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(a.steps):
        eng.feed(step_sizes[k])
    blob = eng.finish()   # sync, p[] back, arithmetic coder
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    total_bytes, dt, _ = shard.aggregate_throughput(n, dt, None)
    host = eng.pipe.host_ms()
    st = eng.pipe.stage_totals()
    nsub = max(st["chunks"], 1)
    bits_per_sub = 8.0 * n / nsub
    roles, rchunks = eng.pipe.paq8_role_ms()
    p8_roles = {k: v / max(rchunks, 1) * 1e3 / bits_per_sub for k, v in roles.items()}
    # the paq8 stage's role kernels run on streams of their own: its period is the slowest role, its latency their chain
    us = {"mixnet": st["mixnet"] * 1e3 / bits_per_sub, "ctxmodels": st["ctxmodels"] * 1e3 / bits_per_sub, "lstm": st["lstm"] * 1e3 / bits_per_sub,
          "fxcm": eng.pipe.fxcm_total_ms() / nsub * 1e3 / bits_per_sub, "paq8": max(p8_roles.values())}
    p8_span = eng.pipe.paq8_total_ms() / nsub * 1e3 / bits_per_sub

    if rank == 0:
        sha = hashlib.sha256(blob).hexdigest()
        verified = {"output_bytes": len(blob), "sha256": sha, "fixture": None, "identical_to_reference_file": None}
        name = "dropin_1m.npz" if a.payload_bytes == 1 << 20 else "dropin_%dk.npz" % (a.payload_bytes >> 10)
        fx = os.path.join(ROOT, "tests", "golden", name)
        if os.path.exists(fx):
            with np.load(fx) as z:
                want_sha, want_size, seed = z["sha256"].tobytes().hex(), int(z["size"][0]), z["seed"]
            if int(seed[0]) == a.payload_bytes and int(seed[1]) == shard.shard_seed(0):
                verified.update(fixture="tests/golden/" + name, reference_bytes=want_size, identical_to_reference_file=bool(sha == want_sha and len(blob) == want_size))
        dom = max(us, key=us.get)
        algo_launch = ALGO[dom] * n / nsub
        kernel_s = us[dom] * bits_per_sub / 1e6
        achieved = algo_launch / kernel_s / 1e9
        out = {
            "metric": "input bytes/s on enwik8 at 1 GPU; compressed-size parity vs CPU ref",
            "value": total_bytes / dt, "unit": "input bytes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "enwik8-shaped shard (cmix_amd.synth.enwik_like, seed 1000 + rank), first %d bytes, compressed as `cmix -c` does: "
                            "TEXT-block stream through the FULL model ensemble (2078 layer-0 inputs: contexts + 54 small models, PPMd host stage, "
                            "LSTM, fxcm 431, paq8 1591 -- every column produced by an engine stage, no stand-ins) + final mixing network + SSE, "
                            "strict bit-exact mode, + arithmetic coder; output file checked against the reference binary's" % a.payload_bytes,
                "payload_bytes": a.payload_bytes, "stream_bytes": n, "sub_chunk_bytes": a.sub_chunk,
                "parallelism": "1 stream per GPU, no collective"},
            "us_per_bit": dt / (8.0 * n) * 1e6,
            "stage_us_per_bit": dict(us, note="mean HIP-event time per bit of each stage's kernel(s) over the timed run; the stages overlap on their own streams, "
                                              "so the stream's period is the slowest one (paq8 = its slowest role kernel)"),
            "paq8_role_us_per_bit": dict(p8_roles, span=p8_span, note="role kernels of the paq8 stage on their own streams; span = first launch to end of its mixer, per chunk"),
            "host_us_per_byte": dict({k: v * 1e3 / n for k, v in host.items()}, note="wall time of the submitting thread per stream byte (slot_wait = blocked on the device)"),
            "verified": verified,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": pmc_traffic_per_byte(dom) * n / nsub,
                         "traffic_source": "profiles/r02_pmc_bench_64k.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command at 64 KB; 2 x FETCH_SIZE + WRITE_SIZE, "
                                           "scaled to this launch size)",
                         "kernel": KERNEL[dom], "stage": dom, "avg_launch_ms": kernel_s * 1e3,
                         "algorithmic_bytes_per_launch": algo_launch,
                         "note": "every stage is a latency-bound dependent chain per stream (DESIGN.md 4); strict-mode ceiling of the final mixing network: "
                                 "2078 dependent f32 adds per bit"},
        }
        if not a.no_cpu_baseline:
            ref = cpu_reference(payload, a.cpu_baseline_bytes, core=max((os.cpu_count() or 2) - 1, 0))
            if ref:
                out["cpu_baseline"] = ref
                if "value" in ref:
                    out["speedup_vs_cpu_reference"] = out["value"] / ref["value"]
        print(json.dumps(out))
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
