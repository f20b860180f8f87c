"""Build libcmixamd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the CPU-only dev container;
the built .so travels to the GPU box with the snapshot.
"""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "lib", "libcmixamd.so")
SOURCES = ["cmx_api.hip", "mixnet_kernels.hip", "mixnet_chunk.hip", "lstm_api.hip", "lstm_kernels.hip",
           "ctxmodels_api.hip", "ctxmodels_kernels.hip", "ppmd_host.cpp", "pipeline_api.hip", "coder_host.cpp", "engine_api.hip", "p8mixer.hip",
           "fxcm_stage.hip", "fxcm_parser_host.cpp", "p8cm2.hip", "p8cm.hip", "p8dmc.hip", "p8match.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value"]


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)]
    deps.append(os.path.join(os.path.dirname(HERE), "include", "cmix_amd.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not _stale():
        return LIB
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    hipcc = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else "hipcc"
    cmd = [hipcc] + FLAGS + ["-o", LIB] + [os.path.join(CSRC, s) for s in SOURCES]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=CSRC)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
