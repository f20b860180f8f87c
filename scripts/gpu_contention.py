#!/usr/bin/env python3
"""Who slows whom: every stage timed alone and next to each other stage on one GPU (round 5; the LSTM runs at 2.2 us/bit alone and at
4.6 in the pipeline). One thread per stage, each with its own handle and HIP stream, looping over its own stand-alone workload; the
measured stage's time per bit is taken while the partner loops.   python scripts/gpu_contention.py [partners]
CMX_LSTM_SLEEP=1 / CMX_MIXNET_SLEEP=1 / CMX_MIXNET_PAD=1 select the poll variants of the library."""
import os
import sys
import threading
import time

os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import numpy as np
import torch

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [R, os.path.join(R, "tests"), os.path.join(R, "tests", "golden")]
from conftest import load_golden, synth_mixnet_inputs  # noqa: E402
from cmix_amd import engine as E, synth  # noqa: E402


class LstmJob:
    name = "lstm"

    def __init__(self):
        g = load_golden("text_2k_nofull")
        vocab = np.zeros(256, np.uint8)
        vocab[np.unique(g["stream"])] = 1
        extra = [i for i in range(256) if not vocab[i]][: max(0, 205 - int(vocab.sum()))]
        vocab[extra] = 1
        self.l = E.Lstm(vocab, 0)
        self.N = 2000
        self.d_in = torch.from_numpy(g["ppmd_probs"][1:self.N + 1].copy()).cuda()
        self.d_b = torch.from_numpy(g["stream"][:self.N].copy()).cuda()
        self.bits = 8 * self.N
        self.s = torch.cuda.Stream()

    def once(self):
        with torch.cuda.stream(self.s):
            self.l.run(self.d_in, self.d_b)
        self.s.synchronize()


class MixnetJob:
    name = "mixnet"

    def __init__(self):
        T = 4096
        probs, sel, bits = synth_mixnet_inputs(T, seed=1)
        self.net = E.MixNet(0)
        self.dp = torch.from_numpy(probs).cuda()
        self.ds = torch.from_numpy((sel & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)).cuda()
        self.db = torch.from_numpy(bits).cuda()
        self.bits = T
        self.s = torch.cuda.Stream()

    def once(self):
        with torch.cuda.stream(self.s):
            self.net.run(self.dp, self.ds, self.db)
        self.s.synchronize()


class FxcmJob:
    name = "fxcm"

    def __init__(self):
        C = 1024
        self.C = C
        self.data = np.frombuffer(synth.enwik_like(C * 8, 1000), np.uint8)
        r = np.random.default_rng(1)
        self.pr = torch.from_numpy(r.integers(1, 4096, 8 * C).astype(np.int16)).cuda()
        self.ex = torch.from_numpy(r.integers(0, 256, 8 * C).astype(np.uint8)).cuda()
        self.probs = torch.full((8 * C, 2078), 0.5, dtype=torch.float32, device="cuda")
        self.fx = E.Fxcm(None, 0)
        self.i = 0
        self.bits = 8 * C
        self.s = torch.cuda.Stream()

    def once(self):
        i = self.i % 8
        self.i += 1
        with torch.cuda.stream(self.s):
            self.fx.run(self.data[i * self.C:(i + 1) * self.C], self.pr, self.ex, self.probs)
        self.s.synchronize()


class P8Job:
    name = "paq8"

    def __init__(self):
        self.data = synth.enwik_like(8 * 1024, 1000)
        self.st = E.P8Stage(0)
        self.i = 0
        self.bits = 8192
        self.s = torch.cuda.Stream()
        self.out = torch.empty((8192, 1591), dtype=torch.float32, device="cuda:0")

    def once(self):
        i = self.i % 8
        self.i += 1
        with torch.cuda.stream(self.s):
            self.st.run(self.data[1024 * i:1024 * (i + 1)], out=self.out)
        self.s.synchronize()   # (the caller's stream waits for the chunk's last role kernel)


def timed(job, reps):
    job.once()
    t0 = time.perf_counter()
    for _ in range(reps):
        job.once()
    return (time.perf_counter() - t0) * 1e6 / (reps * job.bits)


def main():
    want = (sys.argv[1] if len(sys.argv) > 1 else "lstm,mixnet,fxcm,paq8").split(",")
    mk = {"lstm": LstmJob, "mixnet": MixnetJob, "fxcm": FxcmJob, "paq8": P8Job}
    jobs = {n: mk[n]() for n in want}
    reps = {"lstm": 8, "mixnet": 6, "fxcm": 4, "paq8": 4}
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("CMX_")})
    alone = {n: timed(j, reps[n]) for n, j in jobs.items()}
    print("alone (us/bit, wall incl. launch):", {n: round(v, 2) for n, v in alone.items()})
    for a in want:
        row = []
        for b in want:
            if a == b:
                row.append("   -  ")
                continue
            stop = threading.Event()

            def spin(j=jobs[b]):
                while not stop.is_set():
                    j.once()

            th = threading.Thread(target=spin)
            th.start()
            time.sleep(0.05)
            v = timed(jobs[a], reps[a])
            stop.set()
            th.join()
            row.append("%6.2f" % v)
        print("%-7s next to %s: %s   (alone %.2f)" % (a, want, " ".join(row), alone[a]))


def crowd():
    """CMX_CROWD=a,b,..: the first stage timed while ALL the others loop, and again with each of them left out in turn."""
    want = os.environ["CMX_CROWD"].split(",")
    mk = {"lstm": LstmJob, "mixnet": MixnetJob, "fxcm": FxcmJob, "paq8": P8Job}
    jobs = {n: mk[n]() for n in want}
    reps = {"lstm": 8, "mixnet": 6, "fxcm": 4, "paq8": 4}
    a = want[0]
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("CMX_")})
    print("%s alone: %.2f us/bit" % (a, timed(jobs[a], reps[a])))
    for leave in [None] + want[1:]:
        others = [n for n in want[1:] if n != leave]
        stop = threading.Event()
        ths = []
        for b in others:
            def spin(j=jobs[b]):
                while not stop.is_set():
                    j.once()
            t = threading.Thread(target=spin); t.start(); ths.append(t)
        time.sleep(0.1)
        v = timed(jobs[a], reps[a])
        stop.set()
        for t in ths:
            t.join()
        print("%s next to %s: %.2f us/bit" % (a, "+".join(others) or "-", v))


if __name__ == "__main__":
    if os.environ.get("CMX_CROWD"):
        crowd()
    else:
        main()
