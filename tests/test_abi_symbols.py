"""CPU: libcmixamd.so loads and exports every entry point include/cmix_amd.h declares;
without a GPU every create call fails loudly (no CPU fallback in the product)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, "include", "cmix_amd.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cmx_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from cmix_amd import build, engine
    build.build()
    L = C.CDLL(engine.LIB_PATH)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/cmix_amd.h but not exported"


def test_no_cpu_fallback():
    from cmix_amd import engine as E
    if E.device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(E.CmxError, match="no HIP device"):
        E.MixNet(0)
    assert not E.lib().cmx_create(np.ones(256, np.uint8).ctypes.data, None, 0)
    assert "no HIP device" in E.last_error()
    with pytest.raises(E.CmxError, match="no HIP device"):
        E.Pipeline(np.ones(256, np.uint8), 0, 64)


def test_product_never_imports_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    for dirpath, _, files in os.walk(os.path.join(ROOT, "cmix_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dirpath, f), errors="replace").read()
                assert "oracle" not in txt.replace("-- the oracle", "") or f == "__init__.py" or \
                    "import oracle" not in txt and "from oracle" not in txt and "oracle/" not in txt, f


def test_glibc_rand_clone_matches_libc():
    """The LSTM weight draw re-implements glibc rand() (TYPE_3) so the library never touches the
    process-global generator; it must reproduce srand(0xDEADBEEF); rand()... exactly."""
    from cmix_amd import engine as E
    libc = C.CDLL("libc.so.6")
    for seed in (0xDEADBEEF, 1, 12345):
        libc.srand(seed)
        ref = np.array([libc.rand() for _ in range(5000)], np.int32)
        assert np.array_equal(ref, E.glibc_rand(seed, 5000))
