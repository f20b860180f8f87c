"""cmix_amd -- MI355X (gfx950) per-bit prediction engine for cmix v21.

Only what the hot path needs lives here:
  csrc/     hand-written HIP kernels + the C ABI (include/cmix_amd.h)
  lib/      the built libcmixamd.so (in-tree, git-ignored)
  engine.py ctypes binding used by tests / bench (mirrors the C++ shim in INTEGRATION.md)
  synth.py  seeded enwik8-shaped synthetic input (no corpus is available offline)
"""
__version__ = "0.1"
