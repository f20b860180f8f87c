#!/usr/bin/env python3
"""bench.py -- input bytes/s of the whole cmix v21 predictor on enwik8-shaped text, on MI355X.

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N > 1 it is launched under
torch.distributed.run, one rank per GPU. Prints ONE JSON line on rank 0.

Workload (BASELINE.json configs[1], "enwik8 full model ensemble on 1 MI355X", at the size the parity fixture holds):
rank r compresses the first --payload-bytes (default 1 MiB) of its own S-enwik8 shard (cmix_amd.synth.enwik_like,
seed 1000 + r, rich alphabet: V = 205 distinct bytes as in enwik8) exactly as `cmix -c` would: the stream the reference's preprocessor hands the predictor (one TEXT
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

roofline: per SURVEY.md 8(d): algorithmic HBM bytes of the slowest stage's dominant kernel / its HIP-event time;
`traffic` from the PMC passes committed under profiles/ (read at run time, null when the file is missing).
end_to_end: the second figure SURVEY.md 8d asks for -- payload bytes / (engine construction + framing + the timed run);
the warm-up engine's run is NOT in it (round 3 counted it as construction).
cpu_baseline (kind "reference"): the unmodified reference binary (oracle/_ref/cmix_O3) on a BOUNDED PREFIX of the SAME payload --
its first --cpu-baseline-bytes (default 128 KB; the whole 1 MiB costs the reference ~55 minutes, the bench contract asks for a
bounded sample; `config.cpu_baseline_rule` states the rule in the line) --, `-c` (the GPU run's mode, end to end) and `-n` (no
preprocessing: predictor only), each minus a 256-byte run (construction), each on its own pinned host core, running WHILE the
GPU run is timed (the bench thread is kept off those cores; `cpu_baseline.concurrent_with_gpu_run` says so; --cpu-baseline-serial
runs it after the timed section instead, +3 minutes). The reference slows down by ~10 % between 128 KB and 1 MiB as its tables
fill (tests/golden fixtures' ref_seconds), so the prefix figure flatters the reference, not the engine.
At N > 1 every rank also verifies ITS OWN shard: after the timed run it codes the first 128 KB of its shard on a fresh engine and
compares the file with the reference binary's for that seed (tests/golden/dropin_rich_128k_s<seed>.npz); rank 0 reports all ranks.
--tolerance: the mixing network's tolerance mode (an explicit API switch; not bit-exact; `config.mode` comes from the library).
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
# the line is a PREFIX of a 100 MB-shaped shard; how far file identity with the reference has been checked on this shard (records under profiles/)
FULL_SHARD_PARITY = ("unverified beyond 16 MiB: the file equals the unmodified reference binary's through 8 MiB (tests/golden/dropin_rich_8192k.npz, profiles/r05_long_run_8m_fixed.json; "
                     "4 MiB also through the drop-in, profiles/r06_dropin_4mib.txt); 16 MiB: the reference's model families run piecewise against the engine's per-64-KB column digests "
                     "(profiles/r06_long_run_16m.txt); counters / thresholds a longer stream reaches are pinned by state injection (tests/test_wraps_and_thresholds.py). "
                     "OPEN: two digest-harness runs of one 50 MB stream had equal column digests and different final probabilities from 33.9 MB on "
                     "(profiles/r06_two_runs_50m.txt, DESIGN.md 8 item 0): a rare nondeterminism no file comparison has shown")
# algorithmic HBM bytes per input byte: weights touched per bit x 8 B (read + write) x 8 bits
ALGO = {
    "mixnet": 55172 * 8 * 8 + 4 * 64 * 8,                 # final mixers (f32) + 4 SSE lines per bit
    "paq8": (28 * 1552 + 32) * 2 * 2 * 8 + 267 * 3 * 64 * 2,  # paq8 mixer rows (i16, read + write) + bucket probes
    "fxcm": (10 * 512 + 2 * 16) * 2 * 2 * 8 + 81 * 3 * 64 * 2,  # fxcm mixers + bucket probes
    "ctxmodels": 54 * 8 * 64,
}


def lstm_algo_bytes(V, C=200, H=100):
    """forward + output layer + BPTT accumulators + output-layer BPTT read + Adam share, per input byte."""
    g = 3 * C * (2 * V + 3 * C + 4)
    return 4 * g + 3 * 4 * V * (2 * C + 1) + 8 * g + 4 * V * 2 * C + 28 * 3 * C * (4 * V + 3 * C + 2) / H


_DECODE_CHILD = r'''
import json, sys, time
import numpy as np
sys.path.insert(0, {root!r})
from cmix_amd import engine as E, synth
from cmix_amd.pipeline import EngineStream, text_file_stream
payload = synth.enwik_like({n}, {seed}, rich=True)
stream = bytes(text_file_stream(payload))
eng = EngineStream({dev}, stream, 4096)
eng.feed(len(stream))
blob = eng.finish()
eng.close()
length, dic, vocab, hl = E.header_read(blob)
p = E.Predictor(vocab, {dev})
t0 = time.perf_counter()
head = p.decode_stream(blob[hl:], 64)          # start-up: the decoder's kernels are built and launched with the first bit
t1 = time.perf_counter()
p.close()
p = E.Predictor(vocab, {dev})
t2 = time.perf_counter()
out = p.decode_stream(blob[hl:], length)
t3 = time.perf_counter()
p.close()
marg = ((t3 - t2) - (t1 - t0)) / max(1, length - 64)
print(json.dumps({{"bytes": length, "seconds": t3 - t2, "startup_s": (t1 - t0) - 64 * marg, "us_per_byte": 1e6 * marg, "round_trip_ok": bool(out == stream and head == stream[:64])}}))
'''


def decode_leg(payload, nbytes, dev):
    """the decoder's form of the engine (late-bit protocol) on the head of the same shard, in a child process (a decoder needs
    every stage kernel of the stream co-resident from its first bit: a fresh process): the look-ahead engine codes the first nbytes, cmx_decode_stream decodes the
    code; marginal time per byte = (whole - a 64-byte run) / the bytes between."""
    try:
        r = subprocess.run([sys.executable, "-c", _DECODE_CHILD.format(root=ROOT, n=nbytes, seed=1000, dev=dev)], capture_output=True, text=True, timeout=300)
        line = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not line:
            return {"error": (r.stderr or r.stdout)[-300:]}
        d = json.loads(line[-1])
        d["unit"] = "us per decoded byte (marginal)"
        d["note"] = ("cmx_decode_stream: the same stage kernels made patient, the arithmetic decoder and the per-step host stages on one host thread; the reference binary decodes at "
                     "~1300 us/byte on this box's core (profiles/r05_decode_time.txt)")
        return d
    except Exception as e:   # the decode leg never takes the bench line down
        return {"error": repr(e)[:300]}


def _pmc_file():
    """the newest committed PMC summary of the bench command (profiles/rNN_pmc_bench.json)"""
    import glob
    fs = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_bench.json")))
    return fs[-1] if fs else os.path.join(ROOT, "profiles", "r03_pmc_bench.json")


PMC_FILE = _pmc_file()
PMC_KERNELS = {"mixnet": ["cmx_mixnet_spec_kernel"], "fxcm": ["cmx_fxcm_roles_kernel"], "ctxmodels": ["cmx_ctxmodels_kernel"],
               "lstm": ["cmx_lstm_fwdblk", "cmx_lstm_bpttblk", "cmx_lstm_bptt_acc", "cmx_lstm_bptt_gb"], "paq8": ["cmx_p8s_fam2_kernel", "cmx_p8s_mix4_kernel"]}


def pmc_traffic_per_byte(stage):
    """HBM bytes per stream byte of the stage's kernels from the committed PMC passes of this command (scripts/gpu_measure.sh:
    rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs with --kernel-trace only; the counters are KB summed over the
    launches; FETCH_SIZE doubled as /opt/skills/guides/MI355X_MICROARCH.md prescribes for 16-byte-per-lane loads on gfx950).
    None when the file is missing or does not list the kernel: a stale constant is worse than no number."""
    try:
        with open(PMC_FILE) as f:
            z = json.load(f)
        nbytes = float(z["_meta"]["stream_bytes_processed"])
        tot = 0.0
        for k in PMC_KERNELS[stage]:
            match = [v for name, v in z.items() if name.startswith(k)]
            if not match:
                return None
            for v in match:
                tot += (2.0 * v["FETCH_SIZE"]["sum"] + v["WRITE_SIZE"]["sum"]) * 1024.0
        return tot / nbytes
    except Exception:  # noqa: BLE001
        return None


KERNEL = {"mixnet": "cmx_mixnet_spec_kernel", "paq8": "cmx_p8s_fam2_kernel / cmx_p8s_mix4_kernel (slowest role)", "fxcm": "cmx_fxcm_roles_kernel",
          "lstm": "cmx_lstm_fwdblk / cmx_lstm_bpttblk / cmx_lstm_bptt_*", "ctxmodels": "cmx_ctxmodels_kernel"}


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


class CpuReference:
    """The unmodified reference binary on the first nbytes of the SAME payload, in the background: `-c` (end to end, the GPU run's
    mode) and `-n` (predictor only: no preprocessing), each as (run on 256 + nbytes bytes) - (run on 256 bytes:
    construction, ~4 s), each on its own pinned core. start() before the GPU work, result() after it."""

    def __init__(self, payload, nbytes, cores, modes=("-c", "-n")):
        self.payload, self.nbytes, self.cores, self.modes = payload, nbytes, cores, modes
        self.exe = os.path.join(ROOT, "oracle", "_ref", "cmix_O3")
        self.pin = subprocess.call(["which", "taskset"], stdout=subprocess.DEVNULL) == 0
        self.res, self.threads = {}, []

    def _one(self, mode, core):
        pin = ["taskset", "-c", str(core)] if self.pin else []

        def run(n):
            with tempfile.TemporaryDirectory() as d:
                src, dst = os.path.join(d, "in"), os.path.join(d, "out")
                with open(src, "wb") as f:
                    f.write(bytes(self.payload[:n]))
                t0 = time.perf_counter()
                subprocess.run(pin + [self.exe, mode, src, dst], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=1500, check=True)
                return time.perf_counter() - t0, os.path.getsize(dst)
        try:
            t_small, _ = run(256)
            t_big, size = run(256 + self.nbytes)
            self.res[mode] = {"seconds": max(t_big - t_small, 1e-9), "construct_s": t_small, "wall_s": t_big, "file_bytes": size, "core": core}
        except Exception as e:  # noqa: BLE001
            self.res[mode] = {"error": str(e)}

    def start(self):
        if not os.path.exists(self.exe):
            return False
        import threading
        for mode, core in zip(self.modes, self.cores):
            t = threading.Thread(target=self._one, args=(mode, core), daemon=True)
            t.start()
            self.threads.append(t)
        return True

    def result(self):
        for t in self.threads:
            t.join()
        c, n = self.res.get("-c", {}), self.res.get("-n", {})
        if "seconds" not in c:
            return {"error": c.get("error", "reference binary missing")}
        out = {"value": self.nbytes / c["seconds"], "unit": "input bytes/s", "cores": 1, "kind": "reference", "cpu_model": cpu_model(),
               "host_cores": os.cpu_count(), "concurrent_with_gpu_run": not getattr(self, "serial", False),
               "end_to_end": (256 + self.nbytes) / c["wall_s"],
               "sample": f"oracle/_ref/cmix_O3 -c (unmodified reference, g++ -O3) on the first {256 + self.nbytes} bytes of the same payload minus a 256-byte "
                         f"run ({c['construct_s']:.1f} s: construction), whole predictor + coder, {'taskset -c %d' % c['core'] if self.pin else 'unpinned'}, concurrent "
                         f"with the GPU run; {c['seconds']:.1f} s for {self.nbytes} bytes; file {c['file_bytes']} bytes; end_to_end = bytes / wall of main() incl. construction"}
        if "seconds" in n:
            out["predictor_only"] = self.nbytes / n["seconds"]
            out["sample"] += f"; predictor_only = the same with -n (no preprocessing) on core {n['core']}: {n['seconds']:.1f} s"
        return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--payload-bytes", type=int, default=1 << 20, help="bytes of the shard each rank compresses (fixtures: 65536, 262144, 1048576)")
    ap.add_argument("--sub-chunk", type=int, default=4096, help="bytes per pipeline submit (stages of consecutive sub-chunks overlap)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-bytes", type=int, default=131072)
    ap.add_argument("--cpu-baseline-serial", action="store_true", help="run the reference binary after the timed section instead of beside it")
    ap.add_argument("--decode-bytes", type=int, default=4096, help="after the timed run, rank 0 decodes this many bytes of the shard in a child process (cmx_decode_stream over the decoder's "
                    "form of the engine) and reports `decode`; 0 = skip")
    ap.add_argument("--tolerance", action="store_true", help="the mixing network's tolerance mode (NOT bit-exact; the output is not the reference's file)")
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
    if os.environ.get("CMX_BENCH_SAME_DEVICE") == "1":   # dry runs of the N > 1 path on a one-GPU box: every rank on device 0
        local = 0
    torch.cuda.set_device(local)

    payload = synth.enwik_like(a.payload_bytes, shard.shard_seed(rank), rich=True)
    ncpu = os.cpu_count() or 2
    cpu = None
    # N > 1 (north_star: the reference CPU cmix on the node's own host cores in the same run, at 1 / 2 / 4 / 8 GPUs): one pinned reference process per rank,
    # each on the prefix of ITS rank's shard (`-c`: the GPU run's mode), as many as the host's memory holds at ~32 GB each; rank 0 reports their sum
    # (cores = the number of processes). They run beside the timed GPU run on the highest-numbered cores; the bench threads are kept off those.
    cpu_n = None
    if world > 1 and not a.no_cpu_baseline and ncpu >= 2 * world + 2:
        try:
            mem_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2 ** 30
        except (ValueError, OSError):
            mem_gb = 64.0
        nref = max(1, min(world, int(mem_gb // 32)))
        if rank < nref:
            cpu_n = CpuReference(payload, min(a.cpu_baseline_bytes, a.payload_bytes), (ncpu - 1 - rank,), modes=("-c",))
            if not cpu_n.start():
                cpu_n = None
        try:
            os.sched_setaffinity(0, set(range(ncpu - world)))
        except (AttributeError, OSError):
            pass
    if rank == 0 and world == 1 and not a.no_cpu_baseline and ncpu >= 4:
        cpu = CpuReference(payload, min(a.cpu_baseline_bytes, a.payload_bytes), (ncpu - 1, ncpu - 2))
        cpu.serial = a.cpu_baseline_serial
        if a.cpu_baseline_serial:
            if not os.path.exists(cpu.exe):
                cpu = None
        elif cpu.start():
            try:
                os.sched_setaffinity(0, set(range(ncpu - 2)))   # the bench thread (PPMd, text parsers, submits) stays off the two cores
            except (AttributeError, OSError):
                pass
        else:
            cpu = None
    t_c0 = time.perf_counter()
    stream = text_file_stream(payload)
    t_framing = time.perf_counter() - t_c0
    n = len(stream)
    # a step = 1/K of the stream's sub-chunks (whole sub-chunks: a ragged step would put a few-byte chunk into the pipeline)
    nsub_total = -(-n // a.sub_chunk)
    edges = [min(n, (i * nsub_total // a.steps) * a.sub_chunk) for i in range(a.steps)] + [n]
    step_sizes = [edges[i + 1] - edges[i] for i in range(a.steps)]
    step_bytes = max(step_sizes)

    # ---- warm-up: the same code on a throw-away engine over the head of the shard ----
    wn = 0
    t_cold = None   # construction of the FIRST engine of the process: the cold-start figure (HIP context, ~20 GB of tables mapped and initialised for the first time)
    if a.warmup > 0:
        wn = min(n, a.warmup * step_bytes)
        t_w0 = time.perf_counter()
        w = EngineStream(local, stream[:wn], a.sub_chunk)
        torch.cuda.synchronize()
        t_cold = time.perf_counter() - t_w0
        if a.tolerance:
            w.pipe.set_tolerance(True)
        for _ in range(a.warmup):
            w.feed(step_bytes)
        w.finish()
        w.close()
        del w
    t_c1 = time.perf_counter()
    eng = EngineStream(local, stream, a.sub_chunk)
    if a.tolerance:
        eng.pipe.set_tolerance(True)
    torch.cuda.synchronize()
    t_warm = time.perf_counter() - t_c1
    if t_cold is None:
        t_cold = t_warm
    t_construct = t_framing + max(t_cold, t_warm)   # framing + the SLOWER of the process's two engine constructions (the first -- HIP context, tables mapped for the first time -- and the timed engine's own: `construct_first_s` / `construct_warm_s`)
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

    mode_name = "tolerance" if eng.pipe.mixnet_mode() == 1 else "strict"
    # ---- every rank verifies its own shard (N > 1): the first 128 KB on a fresh engine against the reference binary's file for ITS seed ----
    rank_checks = None
    if world > 1:
        mine = {"rank": rank, "seed": shard.shard_seed(rank), "fixture": None, "identical_to_reference_file": None}
        # (rank 0's WHOLE file is compared below; its entry is filled from that)
        fxr = os.path.join(ROOT, "tests", "golden", "dropin_rich_128k_s%d.npz" % shard.shard_seed(rank))
        if rank and os.path.exists(fxr) and a.payload_bytes >= 131072 and mode_name == "strict":
            with np.load(fxr) as z:
                w_sha, w_size = z["sha256"].tobytes().hex(), int(z["size"][0])
            eng.close()
            chk = EngineStream(local, text_file_stream(payload[:131072]), a.sub_chunk)
            chk.feed(1 << 30)
            b2 = chk.finish()
            chk.close()
            mine.update(fixture=os.path.relpath(fxr, ROOT), identical_to_reference_file=bool(hashlib.sha256(b2).hexdigest() == w_sha and len(b2) == w_size))
        rank_checks = [None] * world
        dist.all_gather_object(rank_checks, mine)
    cpu_all = None
    if world > 1:
        mine_cpu = None
        if cpu_n is not None:
            r = cpu_n.result()
            mine_cpu = {"rank": rank, "value": r.get("value"), "core": ncpu - 1 - rank, "error": r.get("error"), "sample": r.get("sample"), "cpu_model": r.get("cpu_model")}
        cpu_all = [None] * world
        dist.all_gather_object(cpu_all, mine_cpu)
    if rank == 0:
        sha = hashlib.sha256(blob).hexdigest()
        verified = {"output_bytes": len(blob), "sha256": sha, "fixture": None, "identical_to_reference_file": None,
                    "full_shard_parity": FULL_SHARD_PARITY}
        name = "dropin_1m.npz" if a.payload_bytes == 1 << 20 else "dropin_rich_%dk.npz" % (a.payload_bytes >> 10)
        fx = os.path.join(ROOT, "tests", "golden", name)
        if os.path.exists(fx):
            with np.load(fx) as z:
                want_sha, want_size, seed, fx_rich = z["sha256"].tobytes().hex(), int(z["size"][0]), z["seed"], "rich" in z.files
            if int(seed[0]) == a.payload_bytes and int(seed[1]) == shard.shard_seed(0) and fx_rich:
                verified.update(fixture="tests/golden/" + name, reference_bytes=want_size, identical_to_reference_file=bool(sha == want_sha and len(blob) == want_size))
        period_stage = max(us, key=us.get)
        V = int(eng.vocab.sum())
        ALGO["lstm"] = lstm_algo_bytes(V)
        # the roofline kernel: the one with the largest algorithmic traffic on the path (the final mixing network: 3.53 of the 14.1 MB per
        # input byte), which the reviews name; the stage that sets the stream's period is reported beside it
        dom = "mixnet"
        algo_launch = ALGO[dom] * n / nsub
        traffic_pb = pmc_traffic_per_byte(dom)
        kernel_s = us[dom] * bits_per_sub / 1e6
        achieved = algo_launch / kernel_s / 1e9
        out = {
            "metric": "input bytes/s on enwik8 at 1 GPU; compressed-size parity vs CPU ref",
            "value": total_bytes / dt, "unit": "input bytes/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt / a.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {
                "workload": "enwik8-shaped shard (cmix_amd.synth.enwik_like, seed 1000 + rank, rich alphabet), first %d bytes, compressed as `cmix -c` does: "
                            "TEXT-block stream through the FULL model ensemble (2078 layer-0 inputs: contexts + 54 small models, PPMd host stage, "
                            "LSTM, fxcm 431, paq8 1591 -- every column produced by an engine stage, no stand-ins) + final mixing network + SSE, "
                            "strict bit-exact mode, + arithmetic coder; output file checked against the reference binary's" % a.payload_bytes,
                "payload_bytes": a.payload_bytes, "stream_bytes": n, "sub_chunk_bytes": a.sub_chunk, "vocab": V, "warmup_stream_bytes": wn,
                "parallelism": "1 stream per GPU, no collective",
                "cpu_baseline_rule": "the reference binary is timed on the first %d bytes of the same payload (a bounded prefix: the whole payload would cost it ~%d minutes), "
                                     "minus a 256-byte run; it slows down by ~10 %% between 128 KB and 1 MiB, so the prefix figure favours the reference" % (min(a.cpu_baseline_bytes, a.payload_bytes), a.payload_bytes // 320 // 60),
                "mode": ("tolerance (--tolerance -> cmx_pipeline_set_tolerance: tree-sum dot products in the final mixing network; NOT bit-exact, the output is not the "
                         "reference's file and `verified` says so)") if mode_name == "tolerance" else "strict (bit-exact; the default and the only mode that claims stream parity)"},
            "us_per_bit": dt / (8.0 * n) * 1e6,
            "end_to_end": {"value": a.payload_bytes * world / (dt + t_construct), "unit": "input bytes/s", "construct_s": t_construct, "construct_first_s": t_framing + t_cold, "construct_warm_s": t_framing + t_warm,
                           "note": "payload bytes / (framing + engine construction: ~20 GB of tables allocated and initialised; the SLOWER of the process's first construction -- the warm-up engine's, "
                                   "HIP context included: construct_first_s -- and the timed engine's own, construct_warm_s -- + the timed run incl. the coder); "
                                   "`value` above is the predictor-only figure of SURVEY.md 8d (stream bytes / the Compress() loop)"},
            "mfma": {"instructions": None, "note": "tolerance mode: the LSTM's weight-update contraction runs as v_mfma_f32_16x16x4_f32 tiles (113 100 SQ_INSTS_VALU_MFMA_F32 per BPTT round, "
                                                   "profiles/r04_lstm_mfma_tolerance.txt); nothing else on the path issues one"} if mode_name == "tolerance" else
                    {"instructions": 0, "note": "strict mode: every dot product on the path is an ordered chain of separately rounded f32 (or wrapping int16-pair) operations; an MFMA "
                                                "step fuses the multiply-add and fixes a blocked K order, so no kernel of the product issues one (SQ_INSTS_VALU_MFMA_* = 0)"},
            "stage_us_per_bit": dict(us, note="mean HIP-event time per bit of each stage's kernel(s) over the timed run; the stages overlap on their own streams, "
                                              "so the stream's period is the slowest one (paq8 = its slowest role kernel)"),
            "paq8_role_us_per_bit": dict(p8_roles, span=p8_span, note="role kernels of the paq8 stage on their own streams; span = first launch to end of its mixer, per chunk"),
            "host_us_per_byte": dict({k: v * 1e3 / n for k, v in host.items()}, note="wall time of the submitting thread per stream byte (slot_wait = blocked on the device)"),
            "verified": verified,
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None if traffic_pb is None else traffic_pb * n / nsub,
                         "traffic_source": os.path.relpath(PMC_FILE, ROOT) + " read at run time (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command, kernel-trace only; "
                                           "2 x FETCH_SIZE + WRITE_SIZE per stream byte, scaled to this launch size); null if the file is missing",
                         "kernel": KERNEL[dom], "stage": dom, "avg_launch_ms": kernel_s * 1e3,
                         "period_stage": {"stage": period_stage, "kernel": KERNEL[period_stage], "us_per_bit": us[period_stage],
                                          "achieved_GBps": ALGO[period_stage] / (8.0 * us[period_stage] * 1e-6) / 1e9,
                                          "note": "the stage whose kernel sets the stream's period this run; its algorithmic traffic per input byte is %.2f MB" % (ALGO[period_stage] / 1e6)},
                         "whole_path": {"achieved_GBps": (total_bytes / dt) * (5.28e6 + ALGO["lstm"]) / 1e9,
                                        "frac": (total_bytes / dt) * (5.28e6 + ALGO["lstm"]) / 1e9 / HBM_PEAK_GBS,
                                        "note": "SURVEY.md 8d's formula: input bytes/s x (5.28 MB + LSTM(V)) algorithmic bytes per input byte / 8 TB/s"},
                         "algorithmic_bytes_per_launch": algo_launch,
                         "note": "every stage is a latency-bound dependent chain per stream (DESIGN.md 4): the HBM roof is the wrong roof by construction"},
        }
        if rank_checks is not None:
            rank_checks[0].update(fixture=verified["fixture"], identical_to_reference_file=verified["identical_to_reference_file"], whole_file=True)
            out["verified"]["ranks"] = rank_checks
        if cpu_all is not None:
            good = [c for c in cpu_all if c and c.get("value")]
            if good:
                out["cpu_baseline"] = {"value": sum(c["value"] for c in good), "unit": "input bytes/s", "cores": len(good), "kind": "reference", "cpu_model": good[0]["cpu_model"],
                                       "host_cores": ncpu, "concurrent_with_gpu_run": True, "per_rank": [{k: c[k] for k in ("rank", "value", "core")} for c in good],
                                       "sample": "%d concurrent pinned processes of oracle/_ref/cmix_O3 -c, one per rank on the first %d bytes of that rank's shard (minus a 256-byte run: "
                                                 "construction), beside the timed GPU run; value = the sum of their bytes/s; rank 0's: %s" % (len(good), min(a.cpu_baseline_bytes, a.payload_bytes), good[0]["sample"])}
                out["speedup_vs_cpu_reference"] = out["value"] / out["cpu_baseline"]["value"]
            else:
                out["cpu_baseline"] = {"error": "no rank's reference run finished: %s" % [c.get("error") if c else None for c in cpu_all]}
        if world == 1 and a.decode_bytes > 0 and mode_name == "strict":
            eng.close()
            out["decode"] = decode_leg(payload, a.decode_bytes, local)
        if cpu is not None:
            if a.cpu_baseline_serial:
                cpu.start()
            ref = cpu.result()
            out["cpu_baseline"] = ref
            if "value" in ref:
                out["speedup_vs_cpu_reference"] = out["value"] / ref["value"]
                if "predictor_only" in ref:
                    out["speedup_vs_cpu_reference_predictor_only"] = out["value"] / ref["predictor_only"]
                out["end_to_end"]["speedup_vs_cpu_reference_end_to_end"] = out["end_to_end"]["value"] / ref["end_to_end"]
        print(json.dumps(out))
    if rank_checks is None or not rank_checks[rank].get("fixture"):
        eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
