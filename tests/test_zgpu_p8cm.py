"""The ContextMap family kernel on the MI355X through the C ABI (cmx_p8cm_create / _run) against the oracle: the case of
tests/test_p8cm_host.py (which runs the kernel's body on the host). Written after round 1's GPU budget was spent: sorted
after the other GPU tests, first device run is the driver's."""
import numpy as np
import pytest

from test_p8cm2_host import tables
from test_p8cm_host import COUNTS, SIZES, family_contexts, hashed, oracle_rows, stream

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def test_family_vs_oracle():
    import torch
    from cmix_amd import engine as E
    data = stream(3200)
    cxs = family_contexts(data, COUNTS, 3)
    want, _ = oracle_rows(SIZES, COUNTS, data, cxs)
    nex, stretch, ilog = tables()
    fam = E.P8ContextMapFamily(SIZES, COUNTS, nex, stretch, ilog, 0)
    c32, k16 = hashed(SIZES, COUNTS, cxs)
    bits = np.unpackbits(np.ascontiguousarray(data))
    outs, pos = [], 0
    for n in [1, 9, 700, 5000]:
        n = min(n, len(data) - pos)
        if n <= 0:
            break
        o = fam.run(torch.from_numpy(c32[pos:pos + n].view(np.int32).copy()).cuda(), torch.from_numpy(k16[pos:pos + n].view(np.int16).copy()).cuda(),
                    torch.from_numpy(bits[8 * pos:8 * (pos + n)].copy()).cuda())
        torch.cuda.synchronize()
        outs.append(o.cpu().numpy())
        pos += n
    fam.close()
    got = np.concatenate(outs)
    bad = np.argwhere(got != want)
    assert bad.size == 0, ("first mismatch (step, input):", bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
