"""GPU: device expf/tanhf/logistic (cmix_amd/csrc/cmx_libm.h) must return the very float
the host glibc returns (the libm the -O3 reference binary calls)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _host(fn, x):
    libm = C.CDLL("libm.so.6")
    f = getattr(libm, fn)
    f.restype = C.c_float
    f.argtypes = [C.c_float]
    return np.array([f(float(v)) for v in x], np.float32)


def _samples(n, seed):
    rng = np.random.default_rng(seed)
    u = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    x = u.view(np.float32).copy()
    edge = np.array([0.0, -0.0, 1.0, -1.0, 88.72, 88.73, -103.9, -104.0, 22.0, -22.0, 0.5, 1e-8,
                     np.inf, -np.inf, 12.2, -12.2, 0.3465, 1.04, 87.0, -87.0, 3e-39, -3e-39], np.float32)
    dense = rng.normal(0, 4, n).astype(np.float32)  # the range the mixers actually produce
    return np.concatenate([x, edge, dense])


@pytest.mark.parametrize("which,fn", [(0, "expf"), (1, "tanhf")])
def test_device_libm_matches_glibc(which, fn):
    from cmix_amd import engine as E
    x = _samples(200000, 1 + which)
    y = E.probe_libm(which, x)
    ref = _host(fn, x)
    same = (y.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(y) & np.isnan(ref))
    bad = np.nonzero(~same)[0]
    assert len(bad) == 0, f"{fn}({x[bad[0]]!r}) device {y[bad[0]]!r} host {ref[bad[0]]!r} ({len(bad)} mismatches)"


def test_device_logistic_matches_oracle():
    from cmix_amd import engine as E
    from oracle import oracle as O
    x = _samples(100000, 9)
    y = E.probe_libm(2, x)
    ref = np.array([O.logistic(v) for v in x], np.float32)
    same = (y.view(np.uint32) == ref.view(np.uint32)) | (np.isnan(y) & np.isnan(ref))
    assert same.all()
