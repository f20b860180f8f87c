#!/usr/bin/env python3
"""Does anything else running on the device change what the engine computes?  (DESIGN.md 5 "Long streams": round 5's digest run saw the final
probabilities leave the reference's 1.44 MB into a stream while torch kernels shared the device; every clean run is bit-exact.)

Part A -- the mixing network alone (cmx_mixnet_run on synthetic rows, 4096-bit launches): a clean run, then the same inputs through a fresh handle while
          foreign work of several kinds runs on the null stream between / under the launches; p and all 47 mixer outputs compared bit for bit.
Part B -- the whole engine (every stage, EngineStream) on the head of the bench shard: a clean run, then runs under the digest computation of
          scripts/gpu_stage_hashes.py (the round-5 situation) and under other loads; p and the 47 mixer outputs of every bit compared on the device,
          the first differing bit / mixer reported.

    python scripts/gpu_foreign_load.py --part A --bits 65536
    python scripts/gpu_foreign_load.py --part B --bytes 1048576 --loads digest,digest
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")


class Foreign:
    """Foreign work on torch's current (null) stream. kinds: digest (gpu_stage_hashes.py's int64 digest of a 4 KB chunk's layer-0 rows: 0.5 GB of
    temporaries), copy (HBM-bound device-to-device copies), lds (sorts / scans: LDS-heavy kernels on many compute units), tiny (hundreds of
    one-block kernels), alloc (cache-emptying allocation churn: hipMalloc / hipFree under the running kernels), all (a mix)."""

    def __init__(self, kind, dev):
        import torch
        self.torch, self.kind, self.dev = torch, kind, dev
        self.l0 = torch.rand((8 * 4096, 2078), device=dev)
        self.A = torch.randint(0, 1 << 62, (2080,), dtype=torch.int64, device=dev)
        self.B = torch.randint(0, 1 << 62, (8 * 4096,), dtype=torch.int64, device=dev)
        self.H = torch.zeros(131, dtype=torch.int64, device=dev)
        self.big = torch.empty(64 << 20, dtype=torch.float32, device=dev)
        self.n = 0

    def step(self):
        torch = self.torch
        k = self.kind if self.kind != "all" else ("digest", "copy", "lds", "tiny", "alloc")[self.n % 5]
        self.n += 1
        if k == "digest":
            v = (self.l0.view(torch.int32).to(torch.int64) & 0xffffffff) + 1
            v = torch.nn.functional.pad(v, (0, 2))
            w = (v * self.A[None, :]).view(-1, 130, 16).sum(2)
            self.H[:130] += (w * self.B[:, None]).sum(0)
        elif k == "copy":
            for _ in range(4):
                self.big[:32 << 20].copy_(self.big[32 << 20:])
        elif k == "lds":
            x = self.l0[:4096].flatten()
            torch.sort(x)
            torch.cumsum(x, 0)
            self.l0[:8192].sum(1)
        elif k == "tiny":
            x = self.H[:8]
            for _ in range(300):
                x = x + 1
        elif k == "alloc":
            t = [torch.empty(256 << 20, dtype=torch.uint8, device=self.dev) for _ in range(4)]
            t[0].zero_()
            del t
            torch.cuda.empty_cache()
        elif k != "none":
            raise ValueError(k)


def first_diff(a, b):
    """a, b: CUDA float tensors of one shape: (number of differing rows, first differing row, the columns that differ there)."""
    import torch
    bad = a.view(torch.int32) != b.view(torch.int32)
    if bad.dim() == 1:
        bad = bad[:, None]
    rows = bad.any(1)
    n = int(rows.sum())
    if not n:
        return 0, None, []
    t = int(torch.nonzero(rows)[0])
    return n, t, torch.nonzero(bad[t]).flatten().tolist()


def part_a(a):
    import torch
    from conftest import synth_mixnet_inputs
    from cmix_amd import engine as E
    dev = torch.device("cuda", 0)
    T, C = a.bits, 4096
    probs, sel, bits = synth_mixnet_inputs(T, seed=7)
    d_probs = torch.from_numpy(probs).to(dev)
    d_sel = torch.from_numpy((sel & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)).to(dev)
    d_bits = torch.from_numpy(bits).to(dev)
    out = {}
    report = []
    for kind in ["none"] + a.loads.split(","):
        envset = None
        if kind.startswith("env:"):   # a library switch instead of foreign work: "env:CMX_MIXNET_ROTATE=1" (the roles' workgroups change XCD with every launch)
            envset = kind[4:].split("=")
            os.environ[envset[0]] = envset[1]
        F = Foreign(kind, dev) if kind != "none" and not envset else None
        net = E.MixNet(0)
        if envset:
            os.environ.pop(envset[0], None)
        p = torch.zeros(T, dtype=torch.float32, device=dev)
        mix = torch.zeros((T, 47), dtype=torch.float32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for lo in range(0, T, C):
            hi = min(T, lo + C)
            net.run(d_probs[lo:hi], d_sel[lo:hi], d_bits[lo:hi], p[lo:hi], mix[lo:hi])
            if F:
                for _ in range(a.reps):
                    F.step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        net.close()
        if kind == "none":
            out["p"], out["mix"] = p, mix
            report.append({"load": kind, "us_per_bit": dt * 1e6 / T})
            print("A clean: %d bits %.2f us/bit" % (T, dt * 1e6 / T), flush=True)
            continue
        n_p, t_p, _ = first_diff(p, out["p"])
        n_m, t_m, cols = first_diff(mix, out["mix"])
        r = {"load": kind, "us_per_bit": dt * 1e6 / T, "p_bits_differing": n_p, "first_p": t_p, "mix_rows_differing": n_m, "first_mix": t_m, "first_mix_mixers": cols}
        report.append(r)
        print("A %-7s %.2f us/bit: p differs in %d bits (first %s); mixer outputs differ in %d bits (first %s, mixers %s)" % (kind, dt * 1e6 / T, n_p, t_p, n_m, t_m, cols), flush=True)
    return report


def part_b(a):
    import torch
    from cmix_amd import engine as E, synth
    from cmix_amd.pipeline import EngineStream, text_file_stream
    dev = torch.device("cuda", 0)
    payload = synth.enwik_like(a.bytes, a.seed, rich=True)
    stream = text_file_stream(payload)
    n = len(stream)
    T = 8 * n
    keep = {}
    report = []
    for kind in ["none"] + a.loads.split(","):
        F = Foreign(kind, dev) if kind != "none" else None
        eng = EngineStream(0, stream, 4096)
        mix = torch.zeros((T, 47), dtype=torch.float32, device=dev)
        eng.pipe.debug_mix_out(mix)
        sub = eng.sub
        nsub = -(-n // sub)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(nsub):
            if F and k >= E.PIPELINE_SLOTS:      # gpu_stage_hashes.py's order: wait for the chunk 8 back, digest it on the null stream, submit the next
                eng.pipe.wait(k - E.PIPELINE_SLOTS)
                for _ in range(a.reps):
                    F.step()
            m = min(sub, n - eng.pos)
            eng.pipe.submit(eng.stream[eng.pos:eng.pos + m], eng.layer0[k % E.PIPELINE_SLOTS][:8 * m], eng.p_dev[8 * eng.pos:8 * (eng.pos + m)])
            eng.pos += m
            eng.nsub += 1
        eng.pipe.sync()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        p = eng.p_dev.clone()
        eng.close()
        del eng
        torch.cuda.empty_cache()
        if kind == "none":
            keep["p"], keep["mix"] = p, mix
            report.append({"load": kind, "bytes_per_s": n / dt})
            print("B clean: %d bytes, %.0f B/s" % (n, n / dt), flush=True)
            continue
        n_p, t_p, _ = first_diff(p, keep["p"])
        n_m, t_m, cols = first_diff(mix, keep["mix"])
        r = {"load": kind, "bytes_per_s": n / dt, "p_bits_differing": n_p, "first_p": t_p, "mix_rows_differing": n_m, "first_mix": t_m, "first_mix_mixers": cols}
        if t_m is not None:
            r["first_mix_values"] = {"loaded": [float(mix[t_m, c]) for c in cols[:8]], "clean": [float(keep["mix"][t_m, c]) for c in cols[:8]],
                                     "loaded_hex": ["%08x" % (int(mix[t_m, c].view(torch.int32)) & 0xffffffff) for c in cols[:8]],
                                     "clean_hex": ["%08x" % (int(keep["mix"][t_m, c].view(torch.int32)) & 0xffffffff) for c in cols[:8]]}
            # per mixer: the first bit at which it differs (which mixer left first, and how the difference spread)
            bad = mix.view(torch.int32) != keep["mix"].view(torch.int32)
            firsts = {}
            for c in range(47):
                nz = torch.nonzero(bad[:, c])
                if len(nz):
                    firsts[c] = int(nz[0])
            r["first_bit_by_mixer"] = firsts
        report.append(r)
        print("B %-7s %.0f B/s: p differs in %d bits (first %s = byte %s); mixer outputs differ in %d bits (first %s, mixers %s)"
              % (kind, n / dt, n_p, t_p, None if t_p is None else t_p // 8, n_m, t_m, cols), flush=True)
        del p, mix
    return report


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--part", default="A")
    ap.add_argument("--bits", type=int, default=65536)
    ap.add_argument("--bytes", type=int, default=1 << 20)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--loads", default="digest,copy,lds,tiny,alloc,all")
    ap.add_argument("--reps", type=int, default=1, help="foreign steps per chunk")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "foreign_load.json"))
    a = ap.parse_args()
    rep = {}
    if "A" in a.part:
        rep["A"] = part_a(a)
    if "B" in a.part:
        rep["B"] = part_b(a)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(rep, f, indent=1)


if __name__ == "__main__":
    main()
