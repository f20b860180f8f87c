"""The match-models kernel on the MI355X through the C ABI (cmx_p8match_create / _run) against the oracle: the case of
tests/test_p8match_host.py (which runs the kernel's body on the host). Written after round 1's GPU budget was spent: sorted
after the other GPU tests, first device run is the driver's."""
import numpy as np
import pytest

from test_p8cm2_host import tables
from test_p8match_host import HIST_LOG2, MATCH_BYTES, SPARSE_BYTES, ilog_table, oracle_rows, stream

pytestmark = [pytest.mark.gpu, pytest.mark.timeout(600)]


def test_vs_oracle():
    import torch
    from cmix_amd import engine as E
    data = stream()
    want = oracle_rows(data)
    nex, stretch, _ = tables()
    mm = E.P8MatchModels(MATCH_BYTES, SPARSE_BYTES, HIST_LOG2, nex, stretch, ilog_table(), 0)
    parts, pos = [], 0
    for n in [1, 5, 2000, 1 << 30]:
        n = min(n, len(data) - pos)
        if n <= 0:
            break
        r = mm.run(torch.from_numpy(data[pos:pos + n].copy()).cuda())
        torch.cuda.synchronize()
        parts.append([x.cpu().numpy() for x in r])
        pos += n
    mm.close()
    for k, name in enumerate(("inputs", "stats", "selectors")):
        g = np.concatenate([p[k] for p in parts])
        bad = np.argwhere(g != want[k])
        assert bad.size == 0, (name, "first mismatch (bit, column):", bad[0], g[tuple(bad[0])], want[k][tuple(bad[0])])
