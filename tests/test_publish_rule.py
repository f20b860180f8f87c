"""A row count may only move when EVERY wave's stores of the row have landed.

On gfx950 `__syncthreads()` is a bare `s_barrier` when no LDS traffic is pending: it does not wait for the other waves' global stores (checked in the ISA,
profiles/r06_roundtrip_failure.txt). A kernel in which several waves store parts of a row, meet at a barrier and let one thread publish the row's counter
(cmx_late.h `late_publish`, lstm_block.hip `wg_signal`) is only right when every thread executes `s_waitcnt vmcnt(0)` BEFORE that barrier. Round 6 found seven
places without it; this test keeps the rule: in the product's kernels, a `__syncthreads()` directly in front of a one-thread `late_publish` has the drain in
front of it."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _code_lines(path):
    out = []
    for no, line in enumerate(open(path, encoding="utf-8", errors="replace"), 1):
        code = line.split("//")[0].strip()
        if code:
            out.append((no, code, line))
    return out


def test_every_barrier_in_front_of_a_row_count_is_drained():
    bad, seen = [], 0
    for path in sorted(glob.glob(os.path.join(ROOT, "cmix_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "cmix_amd", "csrc", "*.h"))):
        lines = _code_lines(path)
        for k, (no, code, _) in enumerate(lines):
            if "late_publish(" not in code or code.startswith("__device__"):
                continue
            # walk back over the publish's own guard / bookkeeping lines to the barrier, if there is one right in front
            j = k - 1
            while j >= 0 and k - j <= 3 and "__syncthreads()" not in lines[j][1]:
                j -= 1
            if j < 0 or k - j > 3 or "__syncthreads()" not in lines[j][1]:
                continue      # published by the wave that stored (late_publish drains that wave itself) or nothing stored
            seen += 1
            before = lines[j - 1][1] if j > 0 else ""
            same = lines[j][1]
            if not any(x in y for x in ("s_waitcnt vmcnt(0)", "wave_mem_sync()") for y in (before, same)):
                bad.append("%s:%d: __syncthreads() in front of the late_publish of line %d without every thread's s_waitcnt vmcnt(0)" % (os.path.relpath(path, ROOT), lines[j][0], no))
    assert seen >= 8, "the scan no longer finds the publish sites (%d)" % seen
    assert not bad, "\n".join(bad)


def test_wg_signal_drains_before_its_barrier():
    src = open(os.path.join(ROOT, "cmix_amd", "csrc", "lstm_block.hip"), encoding="utf-8").read()
    m = re.search(r"void wg_signal\(unsigned\* p\) \{(.*?)\n\}", src, re.S)
    assert m, "wg_signal not found"
    body = m.group(1)
    assert body.index("s_waitcnt vmcnt(0)") < body.index("__syncthreads()") < body.index("__hip_atomic_fetch_add")
