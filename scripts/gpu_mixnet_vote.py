#!/usr/bin/env python3
"""Catch the rare event in which the mixing network computes other probabilities from identical inputs (round 5's digest run at 1.44 MB; round 6's two
independent 50 MB runs of one stream, whose 130 column digests agree in all 767 blocks and whose final-probability digests part at 33.9 MB).

The whole engine codes a long stream as usual; every chunk's layer-0 rows, selectors and bits are ALSO run through K extra, independent mixing-network handles
(cmx_mixnet_run on streams of their own, one chunk behind the engine's), and the K + 1 probability vectors of the chunk are compared on the device. As long as
they agree nothing is kept. At the first chunk in which an instance leaves the majority: the first differing bit, all 47 mixer outputs of every instance in a
window around it, the selectors and the coded bits there go to the report, the instance is retired, the run goes on with the others.

    python scripts/gpu_mixnet_vote.py --bytes 60000000 --extra 3 --seconds 3000 [--load digest]
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "scripts"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bytes", type=int, default=60000000)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--extra", type=int, default=3, help="extra mixing-network instances beside the engine's own")
    ap.add_argument("--seconds", type=float, default=3000.0, help="stop submitting after this much wall time")
    ap.add_argument("--load", default="digest", help="foreign work per chunk on the null stream (gpu_foreign_load.Foreign kinds; none = nothing)")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "mixnet_vote.json"))
    a = ap.parse_args()
    import torch
    from cmix_amd import engine as E, synth
    from cmix_amd.pipeline import EngineStream, text_file_stream
    from gpu_foreign_load import Foreign
    dev = torch.device("cuda", 0)
    L = E.lib()
    L.cmx_pipeline_debug_slot.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
    stream = text_file_stream(synth.enwik_like(a.bytes, a.seed, rich=True))
    n = len(stream)
    eng = EngineStream(0, stream, 4096)
    sub = eng.sub
    T = 8 * sub
    K = a.extra
    nets = [E.MixNet(0) for _ in range(K)]
    streams = [torch.cuda.Stream(dev) for _ in range(K)]
    RING = 2 * E.PIPELINE_SLOTS
    p_x = [torch.zeros((RING, T), dtype=torch.float32, device=dev) for _ in range(K)]
    m_x = [torch.zeros((RING, T, 47), dtype=torch.float32, device=dev) for _ in range(K)]
    # private copies of a chunk's rows, selectors and bits: the engine reuses the slot's buffers for chunk k + 8 as soon as chunk k has left ITS network, while the
    # extra instances still need them for a whole chunk period
    l0c = torch.zeros((RING, T, E.N_INPUTS), dtype=torch.float32, device=dev)
    selc = torch.zeros((RING, T * 47), dtype=torch.int32, device=dev)
    bitc = torch.zeros((RING, T), dtype=torch.uint8, device=dev)
    hip = C.cdll.LoadLibrary("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    m_a = torch.zeros((RING * T, 47), dtype=torch.float32, device=dev)     # the engine's own mixer outputs, a ring of RING chunks (cmx_pipeline_debug_mix_out is re-armed per lap)
    F = Foreign(a.load, dev) if a.load != "none" else None
    alive = [True] * (K + 1)      # instance 0 = the engine's own network
    events = []
    done_ev = [[None] * RING for _ in range(K)]
    nsub = -(-n // sub)
    t0 = time.perf_counter()
    checked = 0

    def launch_extras(k):      # chunk k has left the engine's network: the same rows through the extra instances
        sel_p, bits_p, nb = C.c_void_p(), C.c_void_p(), C.c_size_t()
        if L.cmx_pipeline_debug_slot(eng.pipe.h, k, C.byref(sel_p), C.byref(bits_p), C.byref(nb)):
            raise RuntimeError(E.last_error())
        tb = 8 * nb.value
        r = k % RING
        l0c[r][:tb].copy_(eng.layer0[k % E.PIPELINE_SLOTS][:tb])
        if hip.hipMemcpy(selc[r].data_ptr(), sel_p, tb * 47 * 4, 3) or hip.hipMemcpy(bitc[r].data_ptr(), bits_p, tb, 3):
            raise RuntimeError("hipMemcpy failed")
        cur = torch.cuda.current_stream(dev)
        for j in range(K):
            if not alive[j + 1]:
                continue
            streams[j].wait_stream(cur)
            rc = L.cmx_mixnet_run(nets[j].h, l0c[r].data_ptr(), selc[r].data_ptr(), bitc[r].data_ptr(), tb, p_x[j][k % RING].data_ptr(), m_x[j][k % RING].data_ptr(), C.c_void_p(streams[j].cuda_stream))
            if rc:
                raise RuntimeError(E.last_error())
            ev = torch.cuda.Event()
            ev.record(streams[j])
            done_ev[j][k % RING] = ev
        return tb, sel_p, bits_p

    meta = {}

    def check(k):              # compare chunk k across the instances
        nonlocal checked
        tb, sel_p, bits_p = meta[k]
        lo = k * sub
        for j in range(K):
            if alive[j + 1]:
                done_ev[j][k % RING].synchronize()
        ps = [eng.p_dev[8 * lo:8 * lo + tb]] + [p_x[j][k % RING][:tb] for j in range(K)]
        live = [i for i in range(K + 1) if alive[i]]
        ref = live[0]
        same = {i: bool(torch.equal(ps[i].view(torch.int32), ps[ref].view(torch.int32))) for i in live}
        checked += 1
        if all(same.values()):
            return
        # who is the odd one out: majority among the live instances
        groups = {}
        for i in live:
            key = next((g for g in groups if torch.equal(ps[i].view(torch.int32), ps[g].view(torch.int32))), None)
            groups.setdefault(key if key is not None else i, []).append(i)
        major = max(groups.values(), key=len)
        rep = {"chunk": k, "stream_byte": lo, "live": live, "groups": list(groups.values()), "seconds": time.perf_counter() - t0}
        mixes = [m_a[(k % RING) * T:(k % RING) * T + tb]] + [m_x[j][k % RING][:tb] for j in range(K)]
        for g in groups.values():
            if g is major:
                continue
            i, r = g[0], major[0]
            bad = mixes[i].view(torch.int32) != mixes[r].view(torch.int32)
            rows = torch.nonzero(bad.any(1)).flatten()
            pb = torch.nonzero(ps[i].view(torch.int32) != ps[r].view(torch.int32)).flatten()
            d = {"instance": g, "against": r, "mix_rows_differing": int(len(rows)), "p_bits_differing": int(len(pb)), "first_p_bit": int(pb[0]) if len(pb) else None}
            if len(rows):
                t = int(rows[0])
                cols = torch.nonzero(bad[t]).flatten().tolist()
                w0, w1 = max(0, t - 3), min(tb, t + 4)
                d.update(first_mix_bit=t, first_mix_mixers=cols,
                         odd={"mix": mixes[i][w0:w1].cpu().numpy().view(np.uint32).tolist(), "p": ps[i][w0:w1].cpu().numpy().view(np.uint32).tolist()},
                         major={"mix": mixes[r][w0:w1].cpu().numpy().view(np.uint32).tolist(), "p": ps[r][w0:w1].cpu().numpy().view(np.uint32).tolist()},
                         window_from_bit=w0,
                         per_mixer_first_bit={c: int(torch.nonzero(bad[:, c])[0]) for c in range(47) if bool(bad[:, c].any())})
                selv = selc[k % RING][:tb * 47].view(tb, 47)
                d["selectors_at_first_bit"] = selv[t].cpu().numpy().view(np.uint32).tolist()
                d["selectors_bit_before"] = selv[max(0, t - 1)].cpu().numpy().view(np.uint32).tolist()
                d["bits_window"] = bitc[k % RING][w0:w1].cpu().tolist()
            rep.setdefault("details", []).append(d)
            for q in g:
                alive[q] = False
        events.append(rep)
        print("EVENT", json.dumps(rep)[:3000], flush=True)
        with open(a.out, "w") as f:
            json.dump({"events": events, "chunks_checked": checked}, f)

    pending = []
    for k in range(nsub):
        if time.perf_counter() - t0 > a.seconds or sum(alive) < 2:
            break
        if k % RING == 0:
            eng.pipe.debug_mix_out(m_a)       # re-arm: the next RING chunks' mixer outputs into the ring from its start
        if k >= E.PIPELINE_SLOTS:
            kk = k - E.PIPELINE_SLOTS
            eng.pipe.wait(kk)
            meta[kk] = launch_extras(kk)
            pending.append(kk)
            if F:
                F.step()
        while pending and pending[0] <= k - RING + 2:     # its ring slot is about to be reused
            check(pending.pop(0))
            meta.pop(k - RING, None)
        m = min(sub, n - eng.pos)
        eng.pipe.submit(eng.stream[eng.pos:eng.pos + m], eng.layer0[k % E.PIPELINE_SLOTS][:8 * m], eng.p_dev[8 * eng.pos:8 * (eng.pos + m)])
        eng.pos += m
        eng.nsub += 1
        if k and k % 2048 == 0:
            print("  %7.1f MiB  %6.0f s  %d chunks checked, %d events, live %s" % (eng.pos / 2**20, time.perf_counter() - t0, checked, len(events), [i for i in range(K + 1) if alive[i]]), flush=True)
    while pending:
        check(pending.pop(0))
    eng.pipe.sync()
    dt = time.perf_counter() - t0
    with open(a.out, "w") as f:
        json.dump({"events": events, "chunks_checked": checked, "stream_bytes_coded": eng.pos, "seconds": dt, "instances": K + 1, "load": a.load}, f)
    print("%d bytes in %.0f s (%.0f B/s), %d chunks checked on %d instances, %d events" % (eng.pos, dt, eng.pos / dt, checked, K + 1, len(events)))
    for x in nets:
        x.close()
    eng.close()


if __name__ == "__main__":
    main()
