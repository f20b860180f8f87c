"""One process, one driver thread per GPU, many files (BASELINE configs 4 and 5).

Streams are independent -- no shared model state, no exchange step, separate output files -- so a node is used by giving
every GPU whole files: longest-first greedy assignment (shard.assign_streams), one host thread per GPU that takes its
files in turn through one engine stream each (the engine's state is per handle, nothing is process-global), no
collective. The host stages of a stream (PPMd, the fxcm parser, paq8's front end: ~20 us per input byte) run on the
stream's own thread inside the C library, outside the GIL, so eight threads feed eight GPUs.

    files = {"dickens": open(...).read(), ...}
    out = compress_files(files, devices=range(8))          # {"dickens": b"<.cmix container>", ...}

`open_stream(device, payload)` builds the per-file stream object (feed / finish / close); the default is the whole
engine on the TEXT-block framing of pipeline.text_file_stream. Tests inject fakes."""
import threading
import time

from . import shard


def _default_open(device, payload):
    from .pipeline import EngineStream, text_file_stream
    return EngineStream(device, text_file_stream(payload))


def compress_files(files, devices, open_stream=_default_open, step_bytes=1 << 16, progress=None):
    """files: {name: payload bytes}; devices: GPU indices. Returns ({name: container bytes}, report) where report[device] =
    {"files": [names in processing order], "bytes": n, "seconds": s}. A failing file does not stop the other GPUs; the
    first exception is re-raised after every thread has finished."""
    names = list(files)
    devices = list(devices)
    if not devices:
        raise ValueError("compress_files: no device")
    plan = shard.assign_streams([len(files[n]) for n in names], len(devices))
    out, report, errors = {}, {}, []
    lock = threading.Lock()

    def worker(slot):
        dev = devices[slot]
        rep = {"files": [], "bytes": 0, "seconds": 0.0}
        t0 = time.perf_counter()
        for i in plan[slot]:
            name, payload = names[i], files[names[i]]
            st = None
            try:
                st = open_stream(dev, payload)
                fed = 0
                while fed < len(payload) + 64:   # a stream is at most a few header bytes longer than its payload
                    st.feed(step_bytes)
                    fed += step_bytes
                    if progress:
                        progress(dev, name, min(fed, len(payload)), len(payload))
                blob = st.finish()
                with lock:
                    out[name] = blob
                rep["files"].append(name)
                rep["bytes"] += len(payload)
            except Exception as e:  # noqa: BLE001 -- reported below, the GPU goes on with its next file
                with lock:
                    errors.append((name, dev, e))
            finally:
                if st is not None:
                    st.close()
        rep["seconds"] = time.perf_counter() - t0
        with lock:
            report[dev] = rep

    threads = [threading.Thread(target=worker, args=(k,), name="cmix-gpu%d" % devices[k]) for k in range(len(devices))]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        name, dev, e = errors[0]
        raise RuntimeError("compress_files: %r failed on GPU %d (%d of %d files failed)" % (name, dev, len(errors), len(names))) from e
    return out, report
