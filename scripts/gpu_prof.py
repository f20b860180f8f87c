"""Per-phase shader-clock breakdown of the mixnet kernel (debug aid; not the bench)."""
import sys, time, os
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests')]
from conftest import synth_mixnet_inputs
from cmix_amd import engine as E

T = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
probs, sel, bits = synth_mixnet_inputs(T, seed=1)
net = E.MixNet(0)
dp = torch.from_numpy(probs).cuda()
ds = torch.from_numpy((sel & np.uint64(0xFFFFFFFF)).astype(np.uint32).view(np.int32)).cuda()
db = torch.from_numpy(bits).cuda()
net.run(dp, ds, db); torch.cuda.synchronize()
print('plain: kernel %.2f ms  %.2f us/bit' % (net.last_kernel_ms(), net.last_kernel_ms() * 1e3 / T))
net.profile(True)
net.run(dp, ds, db); torch.cuda.synchronize()
ms = net.last_kernel_ms()
pr = net.profile(False)
if os.environ.get('CMX_MIXNET_V1') == '1':
    names = ['stretch+B1', 'select_row', 'B2+ew load', 'B3 wait (row loads+prod h0)', 'chain h0', 'B4,B5 (prod h1)',
             'chain h1', 'extras chain L0', 'L1', 'L2+SSE', 'perceive scalars+L1/L2 upd', 'B6,B7 (L0 row update)']
else:
    names = ['wait scout', 'wait staged seg 0', 'chain (4 segs)', 'extras chain', 'u + publish',
             'tail handoff + extras upd', 'P0: wait scout', 'P0: wait u', 'P0: serial window (upd/swap 0,1 + stage 0)',
             'P0: upd/swap chunks 2..8', 'P0: stage 1..3 (incl. waits)', 'P0: loop top', 'wait tail_done(t-1)',
             'wait staged seg 1', 'wait staged seg 2', 'wait staged seg 3']
if os.environ.get('CMX_MIXNET_V1') != '1' and os.environ.get('CMX_MIXNET_SPEC', '1') != '0':   # cmx_mixnet_spec_kernel: the gather wave's phases
    names = ['wait scout', 'row state + decay', 'wait the 26 sums (helpers)', 'extras chain', 'u + publish (global)',
             'tail handoff + extras upd', 'P0: -', 'P0: -', 'P0: -', 'P0: -', 'P0: -', 'P0: -', 'wait tail_done(t-1)', '-', '-', '-']
    print('speculation:', net.spec_stats())
if int(os.environ.get('CMX_MIXNET_DBG', '0')) & 2:
    names[6:12] = ['SCOUT: wait consumed', 'SCOUT: probs load + stretch LUT + xs', 'SCOUT: aux + select_row', 'SCOUT: prefetch drain+issue', 'SCOUT: rest + publish', 'SCOUT: loop top']
if int(os.environ.get('CMX_MIXNET_DBG', '0')) & 4:
    names[6:12] = ['TAIL: prefetch rows + SSE cells', 'TAIL: wait tail_in', 'TAIL: layer 1 (dot + chain)', 'TAIL: layer 2 + SSE + L2 perceive scalars', 'TAIL: updates + publish', 'TAIL: loop top + wait scout']
    if os.environ.get('CMX_MIXNET_SPEC', '1') != '0':   # the tail on two waves
        names[6:12] = ['TAIL A: row switch + prefetch', 'TAIL A: wait slot + tail_in', 'TAIL A: layer 1 (dot + chain) + hand-over', '-', 'TAIL A: layer-1 updates', 'TAIL A: loop top + wait scout']
        names[13:16] = ['TAIL B: SSE touches + wait hand-over', 'TAIL B: layer 2 + SSE + output', 'TAIL B: layer-2 update + publish']
tot = sum(pr[:6]) + pr[12] + (sum(pr[13:16]) if os.environ.get('CMX_MIXNET_SPEC', '1') == '0' else 0) if os.environ.get('CMX_MIXNET_V1') != '1' else sum(pr[:12])
print('profiled: kernel %.2f ms  %.2f us/bit; total ticks/bit %.0f' % (ms, ms * 1e3 / T, tot / T))
for n, v in zip(names, pr):
    if n == 'simd ids':
        continue
    print('  %-34s %9.0f ticks/bit  %5.1f%%' % (n, v / T, 100.0 * v / tot))
