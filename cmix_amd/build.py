"""Build libcmixamd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only dev container;
the built .so travels to the GPU box with the snapshot. Every source is compiled
to its own object (in parallel, rebuilt only when it or a header changed), then linked.
"""
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libcmixamd.so")
OBJ = os.path.join(HERE, "lib", "obj")
SOURCES = ["cmx_api.hip", "mixnet_kernels.hip", "mixnet_chunk.hip", "lstm_api.hip", "lstm_kernels.hip", "lstm_block.hip",
           "ctxmodels_api.hip", "ctxmodels_kernels.hip", "ppmd_host.cpp", "pipeline_api.hip", "coder_host.cpp", "engine_api.hip",
           "fxcm_stage.hip", "fxcm_parser_host.cpp", "p8stage.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value"]
# the paq8 stage's host front end: plain C (gcc), every allocation tracked through the force-included p8f_alloc.h
FRONT_DIR = os.path.join(CSRC, "p8front")
FRONT_FLAGS = ["-std=gnu11", "-O2", "-fPIC", "-ffp-contract=off", "-w"]


def _headers():
    hs = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    hs += [os.path.join(FRONT_DIR, f) for f in os.listdir(FRONT_DIR) if f.endswith(".h")]
    hs.append(os.path.join(os.path.dirname(HERE), "include", "cmix_amd.h"))
    return hs


def _jobs():
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    jobs = []
    for s in SOURCES:
        jobs.append((os.path.join(CSRC, s), os.path.join(OBJ, s.rsplit(".", 1)[0] + ".o"),
                     [hipcc] + FLAGS + (["-x", "hip"] if s.endswith(".cpp") else []) + ["-c", os.path.join(CSRC, s)]))
    for f in sorted(os.listdir(FRONT_DIR)):
        if f.endswith(".c"):
            jobs.append((os.path.join(FRONT_DIR, f), os.path.join(OBJ, f[:-2] + ".o"),
                         ["gcc"] + FRONT_FLAGS + ["-include", os.path.join(FRONT_DIR, "p8f_alloc.h"), "-c", os.path.join(FRONT_DIR, f)]))
    return hipcc, jobs


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    hipcc, jobs = _jobs()
    hdr_t = max(os.path.getmtime(h) for h in _headers())
    todo = []
    for src, obj, cmd in jobs:
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            todo.append((obj, cmd + ["-o", obj]))
    if not todo and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(o) for _, o, _ in jobs):
        return LIB

    def run(job):
        if verbose:
            print(" ".join(job[1]))
        subprocess.check_call(job[1], cwd=CSRC)

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 2)) as ex:
        list(ex.map(run, todo))
    cmd = [hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB] + [o for _, o, _ in jobs]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
