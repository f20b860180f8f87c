#!/usr/bin/env python3
"""A long stream through the whole engine on one MI355X: throughput per MiB and the counters that only move late in a stream.

    python scripts/gpu_long_run.py --bytes 8388608 [--out gpurun_out/long_run_8m.json]

The first --bytes of the bench shard (cmix_amd.synth.enwik_like(n, 1000, rich=True)) compressed as `cmix -c` does (bench.py's path:
EngineStream, 4 KB sub-chunks, 8 in flight). Reported: input bytes/s of every MiB; rows allocated by each of the 47 final mixers
against the 10 000-row cap (reference src/mixer/mixer.cpp:16-36: past it every new context shares one overflow row); the mixing
network's speculation hit rate; the paq8 family kernel's share of steps that left the one-pass path (CMX_P8FAM_PROFILE=1); PPMd's
arena use; every stage's time-out flag (cmx_pipeline_sync fails on any). If tests/golden holds the reference binary's file for this
size (dropin_rich_<n>k.npz / dropin_1m.npz, tests/golden/make_dropin_1m.py), the output file is compared with its SHA-256.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
os.environ.setdefault("CMX_P8FAM_PROFILE", "1")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--bytes", type=int, default=8 << 20)
    ap.add_argument("--seed", type=int, default=1000)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "long_run.json"))
    ap.add_argument("--golden", default=os.path.join(ROOT, "tests", "golden"), help="where the reference binary's file fixtures are (a second build under test keeps its own tree)")
    a = ap.parse_args()
    import torch
    from cmix_amd import engine as E, synth
    from cmix_amd.pipeline import EngineStream, text_file_stream
    t0 = time.time()
    payload = synth.enwik_like(a.bytes, a.seed, rich=True)
    stream = text_file_stream(payload)
    t_gen = time.time() - t0
    n = len(stream)
    eng = EngineStream(0, stream, 4096)
    torch.cuda.synchronize()
    L = E.lib()
    for f in ("cmx_pipeline_mixnet_rows", "cmx_pipeline_spec_stats", "cmx_pipeline_paq8_profile", "cmx_pipeline_ppmd_arena"):
        getattr(L, f).argtypes = [C.c_void_p, C.c_void_p]
    per_mib = []
    t_start = time.perf_counter()
    t_prev, pos = t_start, 0
    step = 1 << 20
    while pos < n:
        m = min(step, n - pos)
        eng.feed(m)
        pos += m
        now = time.perf_counter()
        per_mib.append({"upto_bytes": pos, "bytes_per_s": m / (now - t_prev)})
        t_prev = now
        print("  %6.1f MiB  %8.0f B/s" % (pos / 2**20, per_mib[-1]["bytes_per_s"]), flush=True)
    blob = eng.finish()
    dt = time.perf_counter() - t_start
    # SHA-256 of the probabilities of every MiB of input (float32, as the arithmetic coder sees them): two builds that differ can be compared MiB by MiB
    p_all = eng.p_dev[:8 * n].cpu().numpy()
    p_sha = [hashlib.sha256(p_all[8 * i:8 * min(n, i + step)].tobytes()).hexdigest()[:16] for i in range(0, n, step)]
    rows = np.zeros(47, np.uint32)
    spec = np.zeros(5, np.uint64)
    prof = np.zeros(128, np.uint64)
    arena = np.zeros(3, np.uint64)
    L.cmx_pipeline_mixnet_rows(eng.pipe.h, rows.ctypes.data)
    L.cmx_pipeline_spec_stats(eng.pipe.h, spec.ctypes.data)
    have_prof = L.cmx_pipeline_paq8_profile(eng.pipe.h, prof.ctypes.data) == 0
    L.cmx_pipeline_ppmd_arena(eng.pipe.h, arena.ctypes.data)
    st = eng.pipe.stage_totals()
    nsub = max(st["chunks"], 1)
    bits_per_sub = 8.0 * n / nsub
    roles, rch = eng.pipe.paq8_role_ms()
    out = {
        "what": "first %d bytes of the bench shard (seed %d, rich alphabet) through the whole engine, strict mode, one stream on one MI355X" % (a.bytes, a.seed),
        "payload_bytes": a.bytes, "stream_bytes": n, "output_bytes": len(blob), "sha256": hashlib.sha256(blob).hexdigest(),
        "seconds": dt, "bytes_per_s": n / dt, "per_mib": per_mib, "p_sha256_16_per_mib": p_sha, "payload_generation_s": t_gen,
        "stage_us_per_bit": {"mixnet": st["mixnet"] * 1e3 / bits_per_sub, "ctxmodels": st["ctxmodels"] * 1e3 / bits_per_sub, "lstm": st["lstm"] * 1e3 / bits_per_sub,
                             "fxcm": eng.pipe.fxcm_total_ms() / nsub * 1e3 / bits_per_sub, "paq8_roles": {k: v / max(rch, 1) * 1e3 / bits_per_sub for k, v in roles.items()}},
        "mixer_rows": {"allocated": [int(x) for x in rows], "cap": 10000, "mixers_at_cap": int((rows >= 10000).sum()),
                       "note": "rows 0..25 layer 0, 26..45 layer 1, 46 layer 2; a mixer at the cap sends every NEW selector value to its shared overflow row (mixer.cpp:16-36)"},
        "speculation": {"segments": int(spec[0]), "resolved_from_a_candidate_lane": int(spec[1]), "hit_rate": float(spec[1]) / max(int(spec[0]), 1),
                        "reruns_by_segment": [int(x) for x in spec[2:5]]},
        "ppmd_arena": {"reserved_bytes": int(arena[0]), "untouched_bytes": int(arena[1]), "in_use_bytes": int(arena[2]),
                       "note": "the model restart of ppmd.cpp:686-727 (not implemented: the engine stops with an error) is reached when untouched_bytes hits 0"},
        "stage_failures": "none (cmx_pipeline_sync checks the LSTM, fxcm, mixing-network and paq8-mixer time-out flags and the context stage's error word)",
        "host_us_per_byte": {k: v * 1e3 / n for k, v in eng.pipe.host_ms().items()},
    }
    if have_prof:
        steps = prof[80:88].astype(np.float64)
        rounds = prof[64:72].astype(np.float64)
        out["paq8_family"] = {"steps_by_bit_position": [int(x) for x in steps], "steps_that_left_the_one_pass_path": [int(x) for x in rounds],
                              "share_by_bit_position": [float(r / s) if s else 0.0 for r, s in zip(rounds, steps)], "share_overall": float(rounds.sum() / max(steps.sum(), 1))}
    name = "dropin_1m.npz" if (a.bytes, a.seed) == (1 << 20, 1000) else "dropin_rich_%dk%s.npz" % (a.bytes >> 10, "" if a.seed == 1000 else "_s%d" % a.seed)
    fx = os.path.join(a.golden, name)
    if os.path.exists(fx):
        with np.load(fx) as z:
            want_sha, want_size = z["sha256"].tobytes().hex(), int(z["size"][0])
        out["verified"] = {"fixture": "tests/golden/" + name, "reference_bytes": want_size,
                           "identical_to_reference_file": bool(out["sha256"] == want_sha and len(blob) == want_size)}
    else:
        out["verified"] = {"fixture": None, "identical_to_reference_file": None, "note": "no reference file for this size: throughput and counters only"}
    eng.close()
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: out[k] for k in ("payload_bytes", "output_bytes", "seconds", "bytes_per_s", "verified")}))


if __name__ == "__main__":
    main()
