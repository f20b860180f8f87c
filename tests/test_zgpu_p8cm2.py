"""paq8's ContextMap2 kernel on the MI355X through the C ABI (cmx_p8cm2_create / _run) against the oracle: the cases of
tests/test_p8cm2_host.py (which runs the kernel's body on the host). Written after round 1's GPU budget was spent: sorted
after the other GPU tests, first device run is the driver's."""
import numpy as np
import pytest

from test_p8cm2_host import CASES, contexts, oracle_rows, tables

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


@pytest.mark.parametrize("size_bytes,count,nbytes,flavour", CASES)
def test_vs_oracle(size_bytes, count, nbytes, flavour):
    import torch
    from cmix_amd import engine as E, synth
    data = np.frombuffer(synth.enwik_like(nbytes, 13), np.uint8)
    cx = contexts(data, count, flavour)
    want = oracle_rows(size_bytes, count, data, cx)
    nex, stretch, ilog = tables()
    cm = E.P8ContextMap2(size_bytes, count, nex, stretch, ilog, 0)
    c32, k16 = cm.hash(cx)
    bits = np.unpackbits(np.ascontiguousarray(data))
    outs, pos = [], 0
    for n in [1, 7, 500, 1000, 4000]:
        n = min(n, len(data) - pos)
        if n <= 0:
            break
        o = cm.run(torch.from_numpy(c32[pos:pos + n].view(np.int32).copy()).cuda(), torch.from_numpy(k16[pos:pos + n].view(np.int16).copy()).cuda(),
                   torch.from_numpy(bits[8 * pos:8 * (pos + n)].copy()).cuda())
        torch.cuda.synchronize()
        outs.append(o.cpu().numpy())
        pos += n
    cm.close()
    got = np.concatenate(outs)
    bad = np.argwhere(got != want)
    assert bad.size == 0, (flavour, "first mismatch (step, input):", bad[0], got[tuple(bad[0])], want[tuple(bad[0])])
