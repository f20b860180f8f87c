"""One process, one driver thread per GPU, many files (BASELINE configs 4 and 5).

Streams are independent -- no shared model state, no exchange step, separate output files -- so a node is used by giving
every GPU whole files: longest-first greedy assignment (shard.assign_streams), one host thread per GPU that takes its
files in turn, no collective. Two drivers:

compress_paths(jobs, devices, exe)
    FILES ON DISK THROUGH THE REFERENCE'S OWN FRAMING. `exe` is an engine command line built against the reference's
    preprocessor (integration/compress_engine.cpp or the drop-in build of runner.cpp: `exe -c [dict] in out`), so type
    detection, block headers, the e8e9 and WRT transforms are the reference's code, not a restatement; each file is one
    process bound to its GPU through CMIX_DEVICE. This is the driver for mixed corpora (Silesia: text, executables,
    tables, images).

compress_files(files, devices, open_stream)
    PAYLOADS IN MEMORY THROUGH IN-PROCESS ENGINE STREAMS. `open_stream(device, payload)` builds the per-file stream object
    (feed / finish / close) and is responsible for the framing: there is deliberately no default, because the only
    in-process framing the package has is pipeline.text_file_stream (ONE TEXT block), which equals the reference's
    only for files its detector classifies as text (the enwik8-shaped shards of config 5; bench.py pins that case on a
    reference-written file) -- `open_text_stream` is that opener, for callers who know their inputs are such files.
    The host stages of a stream (PPMd, the fxcm parser, paq8's front end: ~13 us per input byte) run on the stream's own
    thread inside the C library, outside the GIL, so eight threads feed eight GPUs.
"""
import os
import subprocess
import threading
import time

from . import shard


def open_text_stream(device, payload):
    """In-process stream for a payload the reference's detector classifies as text (>= 95 % TEXT blocks): one TEXT block."""
    from .pipeline import EngineStream, text_file_stream
    return EngineStream(device, text_file_stream(payload))


def _run_plan(plan, devices, work):
    """One thread per device; thread k runs work(dev, i) for the indices of plan[k] in order. Returns (report, errors)."""
    report, errors = {}, []
    lock = threading.Lock()

    def worker(slot):
        dev = devices[slot]
        rep = {"files": [], "bytes": 0, "seconds": 0.0}
        t0 = time.perf_counter()
        for i in plan[slot]:
            try:
                name, nbytes = work(dev, i)
                rep["files"].append(name)
                rep["bytes"] += nbytes
            except Exception as e:  # noqa: BLE001 -- reported by the caller, the GPU goes on with its next file
                with lock:
                    errors.append((i, dev, e))
        rep["seconds"] = time.perf_counter() - t0
        with lock:
            report[dev] = rep

    threads = [threading.Thread(target=worker, args=(k,), name="cmix-gpu%d" % devices[k]) for k in range(len(devices))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    return report, errors


def compress_paths(jobs, devices, exe, mode="-c", dictionary=None, timeout=None):
    """jobs: [(input_path, output_path), ...]; devices: GPU indices; exe: the engine command line (see the module docstring).
    Every file is compressed by `exe mode [dictionary] input output` with CMIX_DEVICE set to its GPU. Returns report[device] =
    {"files": [input paths in processing order], "bytes": n, "seconds": s}. A failing file does not stop the other GPUs; the
    first failure is re-raised after every thread has finished."""
    devices = list(devices)
    if not devices:
        raise ValueError("compress_paths: no device")
    if not os.path.exists(exe):
        raise FileNotFoundError("compress_paths: engine command line %r not built" % exe)
    jobs = list(jobs)
    sizes = [os.path.getsize(src) for src, _ in jobs]
    plan = shard.assign_streams(sizes, len(devices))

    def work(dev, i):
        src, dst = jobs[i]
        env = dict(os.environ, CMIX_DEVICE=str(dev))
        cmd = [exe, mode] + ([dictionary] if dictionary else []) + [src, dst]
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout)
        if r.returncode != 0:
            raise RuntimeError("%s failed (rc %d): %s" % (" ".join(cmd), r.returncode, r.stderr.decode(errors="replace")[-300:]))
        return src, sizes[i]

    report, errors = _run_plan(plan, devices, work)
    if errors:
        i, dev, e = errors[0]
        raise RuntimeError("compress_paths: %r failed on GPU %d (%d of %d files failed)" % (jobs[i][0], dev, len(errors), len(jobs))) from e
    return report


def compress_files(files, devices, open_stream, step_bytes=1 << 16, progress=None):
    """files: {name: payload bytes}; devices: GPU indices; open_stream(device, payload) -> stream object with feed(nbytes) /
    finish() -> container bytes / close() -- the caller's choice of framing (module docstring). Returns ({name: container bytes},
    report) where report[device] = {"files": [names in processing order], "bytes": n, "seconds": s}. A failing file does not stop
    the other GPUs; the first exception is re-raised after every thread has finished."""
    names = list(files)
    devices = list(devices)
    if not devices:
        raise ValueError("compress_files: no device")
    if open_stream is None:
        raise ValueError("compress_files: open_stream is required (the framing is the caller's: see multifile.__doc__)")
    plan = shard.assign_streams([len(files[n]) for n in names], len(devices))
    out = {}
    lock = threading.Lock()

    def work(dev, i):
        name, payload = names[i], files[names[i]]
        st = open_stream(dev, payload)
        try:
            fed = 0
            while fed < len(payload) + 64:   # a stream is at most a few header bytes longer than its payload
                st.feed(step_bytes)
                fed += step_bytes
                if progress:
                    progress(dev, name, min(fed, len(payload)), len(payload))
            blob = st.finish()
        finally:
            st.close()
        with lock:
            out[name] = blob
        return name, len(payload)

    report, errors = _run_plan(plan, devices, work)
    if errors:
        i, dev, e = errors[0]
        raise RuntimeError("compress_files: %r failed on GPU %d (%d of %d files failed)" % (names[i], dev, len(errors), len(names))) from e
    return out, report
