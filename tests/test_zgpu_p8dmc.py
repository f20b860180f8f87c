"""The DMC forest kernel on the MI355X through the C ABI (cmx_p8dmc_create / _run) against the oracle: the cases of
tests/test_p8dmc_host.py (which runs the kernel's body on the host). Written after round 1's GPU budget was spent: sorted
after the other GPU tests, first device run is the driver's."""
import numpy as np
import pytest

from test_p8cm2_host import tables
from test_p8dmc_host import CASES, oracle_rows

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.mark.parametrize("level,nbytes", CASES)
def test_vs_oracle(level, nbytes):
    import torch
    from cmix_amd import engine as E, synth
    data = np.frombuffer(synth.enwik_like(nbytes, 29), np.uint8)
    want = oracle_rows(level, data)
    nex, stretch, _ = tables()
    f = E.P8DmcForest(level, nex, stretch, 0)
    bits = np.unpackbits(np.ascontiguousarray(data))
    outs, pos = [], 0
    for n in [3, 13, 8000, 1 << 30]:
        n = min(n, len(bits) - pos)
        if n <= 0:
            break
        o = f.run(torch.from_numpy(bits[pos:pos + n].copy()).cuda())
        torch.cuda.synchronize()
        outs.append(o.cpu().numpy())
        pos += n
    f.close()
    got = np.concatenate(outs)
    bad = np.argwhere(got != want)
    assert bad.size == 0, ("first mismatch (bit, input):", bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
