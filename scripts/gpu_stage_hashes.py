#!/usr/bin/env python3
"""Per-64-KB digests of the layer-0 columns (130 groups of 16 consecutive columns) and of the final probability over a long stream, on the
device -- the engine's side of oracle/ref_long_trace.cpp (same digest: sum over bits t and columns c of (bits(p[t][c]) + 1) * A[c] * B[t mod 2^19]
mod 2^64). Compared block by block with the unmodified reference Predictor's digests this tells WHICH columns leave the reference's values first,
and where.

    python scripts/gpu_stage_hashes.py --bytes 8388608 --out gpurun_out/stage_hashes_8m.txt [--ref ref_hashes.txt]
    python scripts/gpu_stage_hashes.py --compare mine.txt ref.txt [g0 g1] (no GPU: compare two digest files, optionally groups g0..g1 only)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
NG = 130


def group_name(g):
    if g == NG:
        return "final p"
    lo, hi = 16 * g, min(2078, 16 * g + 16) - 1
    def owner(c):
        return ("contexts" if c < 3 or 2025 <= c < 2076 else "PPMd" if c == 2076 else "LSTM" if c == 2077 else "fxcm" if c < 434 else "paq8[%d]" % (c - 434))
    a, b = owner(lo), owner(hi)
    return "cols %d..%d (%s)" % (lo, hi, a if a == b else a + " .. " + b)


def compare(mine_path, ref_path, glo=0, ghi=NG):
    mine = [l.split() for l in open(mine_path) if l.strip()]
    ref = [l.split() for l in open(ref_path) if l.strip()]
    n = min(len(mine), len(ref))
    for b in range(n):
        bad = [g for g in range(glo, ghi + 1) if mine[b][1 + g] != ref[b][1 + g]]
        if bad:
            print("compared %d blocks: FIRST DIFFERENCE in block %d (bytes %d..%s): %d groups differ" % (n, b, b * 65536, mine[b][0], len(bad)))
            for g in bad[:40]:
                print("   ", group_name(g))
            later = {}
            for bb in range(b, n):
                for g in range(glo, ghi + 1):
                    if mine[bb][1 + g] != ref[bb][1 + g]:
                        later.setdefault(g, bb)
            print("    first block each group differs in:", {group_name(g): v for g, v in sorted(later.items(), key=lambda kv: kv[1])[:60]})
            return b
    print("compared %d blocks, groups %d..%d: all digests equal" % (n, glo, ghi))
    return None


def splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
    return x ^ (x >> np.uint64(31))


def main():
    if len(sys.argv) >= 4 and sys.argv[1] == "--compare":   # [first group, last group]: oracle/ref_paq8_trace.cpp fills groups 28..125 only
        compare(sys.argv[2], sys.argv[3], *[int(x) for x in sys.argv[4:6]])
        return
    ap = argparse.ArgumentParser()
    ap.add_argument("--bytes", type=int, default=8 << 20)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "stage_hashes.txt"))
    ap.add_argument("--ref", default=None, help="oracle/_ref/ref_long_trace's output: compare and report the first block / group that differs")
    ap.add_argument("--detail-block", type=int, default=-1, help="stop after this 64 KB block and compare its rows with the reference's detail files (ref_long_trace's "
                    "<out>.block<k>.g<g>.f32 / .p.f32 under --detail-dir): first differing bit and column of every dumped group, both values")
    ap.add_argument("--detail-dir", default=os.path.join(ROOT, "tmp_longref"))
    ap.add_argument("--lean", action="store_true", help="with --detail-block: no digests, only the detail block's comparison; the report is written (and the process ends) at the "
                    "first difference -- for a run that has seconds of GPU time; --head-file / --vocab-file: the stream's head and the whole file's vocabulary from files")
    ap.add_argument("--stop-block", type=int, default=-1, help="run only the stream's first N + 1 blocks of 64 KB (the engine is still built for the whole stream)")
    ap.add_argument("--selfcheck", action="store_true", help="round 6 (DESIGN.md 5): is a difference of the final probability's digest the ENGINE's or the READ-BACK's? Keeps a copy of p as "
                    "each digest read it, recomputes the digests from the finished run's p, then runs the same bytes through a second engine WITHOUT any digest work and compares the two runs' p")
    ap.add_argument("--finish", action="store_true", help="after the run: the arithmetic coder over every p (size and SHA-256 of the file `cmix -c` would write), the mixers' row counts, "
                    "the speculation statistics, PPMd's arena -- what scripts/gpu_long_run.py reports; with it the digests so far are also written every 16 MiB (<out>.partial)")
    ap.add_argument("--inputs-digest", action="store_true", help="round 6 (DESIGN.md 8 item 0): also digest what the mixing network reads BESIDE the 2078 columns -- the 47 selectors of "
                    "every bit and the coded bits, from the chunk's slot buffers (cmx_pipeline_debug_slot) -- per 64 KB block into <out>.inputs (47 + 1 digests per block; two "
                    "runs are compared with --compare a.inputs b.inputs 0 47). Not yet run on a device (written after the round's last GPU minute).")
    ap.add_argument("--head-file", default=None)
    ap.add_argument("--vocab-file", default=None)
    a = ap.parse_args()
    import glob
    import re
    import torch
    from cmix_amd import engine as E, synth
    from cmix_amd.pipeline import EngineStream, text_file_stream
    with np.errstate(over="ignore"):
        A = splitmix64(np.arange(2078, dtype=np.uint64)) | np.uint64(1)
        B = splitmix64(np.uint64(0x1000000) + np.arange(1 << 19, dtype=np.uint64)) | np.uint64(1)
    dev = torch.device("cuda", 0)
    At = torch.from_numpy(A.view(np.int64)).to(dev)
    Bt = torch.from_numpy(B.view(np.int64)).to(dev)
    At = torch.cat([At, torch.zeros(16 * NG - 2078, dtype=torch.int64, device=dev)])
    if a.head_file:
        stream = np.fromfile(a.head_file, np.uint8).tobytes()
    else:
        payload = synth.enwik_like(a.bytes, a.seed, rich=True)
        stream = text_file_stream(payload)
    detail = {}
    if a.detail_block >= 0:   # the reference's rows of the block, by group (and "p")
        for f in glob.glob(os.path.join(a.detail_dir, "*.block%d.*.f32" % a.detail_block)):
            m = re.search(r"\.block%d\.(g(\d+)|p)\.f32$" % a.detail_block, f)
            if m:
                key = "p" if m.group(1) == "p" else int(m.group(2))
                arr = np.fromfile(f, np.float32)
                detail[key] = torch.from_numpy(arr if key == "p" else arr.reshape(-1, 16)).to(dev)
        print("detail files of block %d: %s" % (a.detail_block, sorted(map(str, detail))))
    n = len(stream)
    if a.detail_block >= 0:   # the engine is built for the WHOLE stream (its vocabulary is the file's), only the head of it is run
        n = min(n, (a.detail_block + 1) * 65536)
    if a.stop_block >= 0:
        n = min(n, (a.stop_block + 1) * 65536)
    whole_vocab = np.zeros(256, np.uint8)
    whole_vocab[np.unique(np.frombuffer(stream, np.uint8))] = 1
    report = {}
    eng = EngineStream(0, stream, 4096, vocab=np.fromfile(a.vocab_file, np.uint8) if a.vocab_file else whole_vocab if a.stop_block >= 0 else None)
    psnap = torch.zeros(8 * n, dtype=torch.float32, device=dev) if a.selfcheck else None
    sub = eng.sub
    nsub = -(-n // sub)
    blocks = -(-n // 65536)
    H = torch.zeros((blocks, NG + 1), dtype=torch.int64, device=dev)
    HX = selbuf = bitbuf = hip = None
    if a.inputs_digest:
        import ctypes as C
        HX = torch.zeros((blocks, 48), dtype=torch.int64, device=dev)
        selbuf = torch.zeros(8 * sub * 47, dtype=torch.int32, device=dev)
        bitbuf = torch.zeros(8 * sub, dtype=torch.uint8, device=dev)
        hip = C.cdll.LoadLibrary("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        E.lib().cmx_pipeline_debug_slot.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]

    def digest(k):   # sub-chunk k is complete: fold its rows into its 64 KB block's digests
        lo, hi = k * sub, min(n, (k + 1) * sub)
        if a.lean and lo // 65536 != a.detail_block:
            return   # (submit() itself waits for a free slot)
        eng.pipe.wait(k)
        l0 = eng.layer0[k % E.PIPELINE_SLOTS][:8 * (hi - lo)]
        t = torch.arange(8 * lo, 8 * hi, device=dev) & ((1 << 19) - 1)
        v = (l0.view(torch.int32).to(torch.int64) & 0xffffffff) + 1
        v = torch.nn.functional.pad(v, (0, 16 * NG - 2078))
        w = (v * At[None, :]).view(-1, NG, 16).sum(2)          # (int64 arithmetic wraps: the digest is mod 2^64; no int64 matrix product on the device)
        bt = Bt[t]
        H[lo // 65536, :NG] += (w * bt[:, None]).sum(0)
        pv = (eng.p_dev[8 * lo:8 * hi].view(torch.int32).to(torch.int64) & 0xffffffff) + 1
        H[lo // 65536, NG] += (pv * bt).sum()
        if HX is not None:   # the chunk's selectors and bits as they lie in its slot (what the mixing network's kernel read)
            sel_p, bits_p, nb = C.c_void_p(), C.c_void_p(), C.c_size_t()
            if E.lib().cmx_pipeline_debug_slot(eng.pipe.h, k, C.byref(sel_p), C.byref(bits_p), C.byref(nb)):
                raise RuntimeError(E.last_error())
            tb = 8 * nb.value
            if tb != 8 * (hi - lo) or hip.hipMemcpy(selbuf.data_ptr(), sel_p, tb * 47 * 4, 3) or hip.hipMemcpy(bitbuf.data_ptr(), bits_p, tb, 3):
                raise RuntimeError("slot %d: size %d / copy failed" % (k, tb))
            sv = (selbuf[:tb * 47].view(tb, 47).to(torch.int64) & 0xffffffff) + 1
            HX[lo // 65536, :47] += (sv * bt[:, None]).sum(0)
            HX[lo // 65536, 47] += ((bitbuf[:tb].to(torch.int64) + 1) * bt).sum()
        if psnap is not None:
            psnap[8 * lo:8 * hi] = eng.p_dev[8 * lo:8 * hi]
        if detail and lo // 65536 == a.detail_block:
            r0 = 8 * (lo - a.detail_block * 65536)
            for key, ref in detail.items():
                mine = eng.p_dev[8 * lo:8 * hi].view(-1, 1) if key == "p" else l0[:, 16 * key:16 * key + 16]
                if ref.shape[0] < r0 + mine.shape[0]:
                    continue   # (the reference's file ends inside this piece)
                want = ref[r0:r0 + mine.shape[0]].view(mine.shape[0], -1)[:, :mine.shape[1]]
                bad = (mine.view(torch.int32) != want.view(torch.int32))
                rep = report.setdefault(key, {"bits_differing": 0, "per_column": [0] * mine.shape[1], "first": None})
                rep["bits_differing"] += int(bad.any(1).sum())
                pc = bad.sum(0).tolist()
                rep["per_column"] = [x + y for x, y in zip(rep["per_column"], pc)]
                if rep["first"] is None and bool(bad.any()):
                    t = int(torch.nonzero(bad.any(1))[0])
                    cols = torch.nonzero(bad[t]).flatten().tolist()
                    w0, w1 = max(0, t - 24), min(mine.shape[0], t + 40)
                    c = cols[0]
                    rep["first"] = {"bit_in_stream": 8 * lo + t, "byte": lo + t // 8, "bit_of_byte": t % 8, "columns": [(0 if key == "p" else 16 * key) + x for x in cols],
                                    "engine": [float(mine[t, x]) for x in cols], "reference": [float(want[t, x]) for x in cols],
                                    "engine_hex": ["%08x" % (int(mine[t, x].view(torch.int32)) & 0xffffffff) for x in cols],
                                    "reference_hex": ["%08x" % (int(want[t, x].view(torch.int32)) & 0xffffffff) for x in cols],
                                    "window_from_bit": 8 * lo + w0, "window_engine": [float(v) for v in mine[w0:w1, c]], "window_reference": [float(v) for v in want[w0:w1, c]],
                                    "bytes_before": bytes(stream[max(0, lo + t // 8 - 48):lo + t // 8 + 1]).decode("latin1")}
                    if a.lean:   # seconds of GPU time left: the report NOW, then out
                        import json
                        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
                        with open(a.out + ".detail.json", "w") as f:
                            json.dump({str(kk): vv for kk, vv in report.items()}, f, indent=1)
                            f.flush()
                            os.fsync(f.fileno())
                        print("first difference: %s" % json.dumps(rep["first"])[:1200], flush=True)
                        os._exit(0)

    t0 = time.perf_counter()
    t_mark, pos_mark = t0, 0
    for k in range(nsub):
        if k >= E.PIPELINE_SLOTS:
            digest(k - E.PIPELINE_SLOTS)          # its layer-0 slot is about to be reused
        if a.finish and k and k % 4096 == 0:       # every 16 MiB: the rate of the piece, and the digests of the complete blocks so far to <out>.partial
            now = time.perf_counter()
            print("  %7.1f MiB  %8.0f B/s" % (eng.pos / 2**20, (eng.pos - pos_mark) / (now - t_mark)), flush=True)
            t_mark, pos_mark = now, eng.pos
            done_blocks = max(0, (eng.pos - (E.PIPELINE_SLOTS + 1) * sub) // 65536)
            Hp = H[:done_blocks].cpu().numpy().view(np.uint64)
            os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
            with open(a.out + ".partial", "w") as f:
                for b in range(done_blocks):
                    f.write("%d %s\n" % ((b + 1) * 65536, " ".join("%016x" % int(x) for x in Hp[b])))
        m = min(sub, n - eng.pos)
        eng.pipe.submit(eng.stream[eng.pos:eng.pos + m], eng.layer0[k % E.PIPELINE_SLOTS][:8 * m], eng.p_dev[8 * eng.pos:8 * (eng.pos + m)])
        eng.pos += m
        eng.nsub += 1
    for k in range(max(0, nsub - E.PIPELINE_SLOTS), nsub):
        digest(k)
    eng.pipe.sync()
    dt = time.perf_counter() - t0
    Hh = H.cpu().numpy().view(np.uint64)
    os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
    with open(a.out, "w") as f:
        for b in range(blocks):
            f.write("%d %s\n" % (min(n, (b + 1) * 65536), " ".join("%016x" % int(x) for x in Hh[b])))
    if HX is not None:
        Hx = HX.cpu().numpy().view(np.uint64)
        with open(a.out + ".inputs", "w") as f:
            for b in range(blocks):
                f.write("%d %s\n" % (min(n, (b + 1) * 65536), " ".join("%016x" % int(x) for x in Hx[b])))
    print("%d bytes in %.1f s (%.0f B/s), %d blocks of 64 KB -> %s" % (n, dt, n / dt, blocks, a.out))
    if detail:
        import json
        with open(a.out + ".detail.json", "w") as f:
            json.dump({str(k): v for k, v in report.items()}, f, indent=1)
        for k, v in sorted(report.items(), key=lambda kv: str(kv[0])):
            print("group %s: %d bits differ in block %d; first: %s" % (k, v["bits_differing"], a.detail_block, json.dumps(v["first"])[:1500]))
    if a.ref and os.path.exists(a.ref):
        compare(a.out, a.ref)
    if a.finish:
        import ctypes as C
        import hashlib
        import json
        L = E.lib()
        for fn in ("cmx_pipeline_mixnet_rows", "cmx_pipeline_spec_stats", "cmx_pipeline_ppmd_arena"):
            getattr(L, fn).argtypes = [C.c_void_p, C.c_void_p]
        rows = np.zeros(47, np.uint32); spec = np.zeros(5, np.uint64); arena = np.zeros(3, np.uint64)
        L.cmx_pipeline_mixnet_rows(eng.pipe.h, rows.ctypes.data)
        L.cmx_pipeline_spec_stats(eng.pipe.h, spec.ctypes.data)
        L.cmx_pipeline_ppmd_arena(eng.pipe.h, arena.ctypes.data)
        tc = time.perf_counter()
        eng.pos = n
        blob = eng.finish()          # p[] back, the arithmetic coder (host)
        rep = {"stream_bytes": n, "seconds": dt, "bytes_per_s": n / dt, "coder_seconds": time.perf_counter() - tc, "output_bytes": len(blob), "sha256": hashlib.sha256(blob).hexdigest(),
               "bits_per_byte": 8.0 * len(blob) / n, "mixer_rows": rows.tolist(), "mixers_at_row_cap": int((rows >= 10000).sum()),
               "speculation": {"segments": int(spec[0]), "hits": int(spec[1]), "hit_rate": float(spec[1]) / max(1, int(spec[0])), "reruns": [int(x) for x in spec[2:5]]},
               "ppmd_arena_bytes": {"reserved": int(arena[0]), "untouched": int(arena[1]), "in_use": int(arena[2])}}
        with open(a.out + ".run.json", "w") as f:
            json.dump(rep, f, indent=1)
        print(json.dumps({k: v for k, v in rep.items() if k != "mixer_rows"}), flush=True)
    if a.selfcheck:
        torch.cuda.synchronize()
        p_loaded = eng.p_dev[:8 * n].clone()
        # (1) what each digest read against what the finished run holds: a difference = the digest read p before it was written
        late = torch.nonzero(psnap.view(torch.int32) != p_loaded.view(torch.int32)).flatten()
        print("selfcheck 1: p as the digests read it vs p after the run: %d bits differ%s" % (len(late), "" if not len(late) else "; first bit %d (block %d)" % (int(late[0]), int(late[0]) >> 19)))
        # (2) the digests recomputed from the finished run's p
        H2 = torch.zeros(blocks, dtype=torch.int64, device=dev)
        for b in range(blocks):
            lo, hi = b << 19, min(8 * n, (b + 1) << 19)
            pv = (p_loaded[lo:hi].view(torch.int32).to(torch.int64) & 0xffffffff) + 1
            H2[b] = (pv * Bt[:hi - lo]).sum()
        badb = torch.nonzero(H2 != H[:, NG]).flatten().tolist()
        print("selfcheck 2: final-p digests taken during the run vs recomputed afterwards: blocks that differ: %s" % badb)
        eng.close()
        del eng
        torch.cuda.empty_cache()
        # (3) the same bytes through a second engine with nothing else on the device
        eng2 = EngineStream(0, stream, 4096, vocab=whole_vocab if a.stop_block >= 0 else None)
        eng2.feed(n)
        eng2.pipe.sync()
        torch.cuda.synchronize()
        p_clean = eng2.p_dev[:8 * n]
        d = torch.nonzero(p_clean.view(torch.int32) != p_loaded.view(torch.int32)).flatten()
        print("selfcheck 3: p of the digest run vs p of a clean run of the same %d bytes: %d bits differ%s" % (n, len(d), "" if not len(d) else "; first bit %d (byte %d, block %d)" % (int(d[0]), int(d[0]) >> 3, int(d[0]) >> 19)))
        H3 = []
        for b in range(blocks):
            lo, hi = b << 19, min(8 * n, (b + 1) << 19)
            pv = (p_clean[lo:hi].view(torch.int32).to(torch.int64) & 0xffffffff) + 1
            H3.append(int((pv * Bt[:hi - lo]).sum()) & 0xFFFFFFFFFFFFFFFF)
        with open(a.out + ".clean_p_digests.txt", "w") as f:
            for b in range(blocks):
                f.write("%d %016x\n" % (min(n, (b + 1) * 65536), H3[b]))
        eng2.close()
        return
    eng.close()


if __name__ == "__main__":
    main()
