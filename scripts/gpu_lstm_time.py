import sys, time, numpy as np, torch
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, 'tests'), os.path.join(R, 'tests', 'golden')]
from conftest import load_golden
from cmix_amd import engine as E
g = load_golden('text_2k_nofull')
N = min(int(sys.argv[1]) if len(sys.argv) > 1 else 2000, len(g["stream"]), len(g["ppmd_probs"]) - 1)   # (round 5: N used to exceed the golden trace (2000 bytes) while the time was divided by N -- the "2.2 us/bit alone" of rounds 2-4 was 4.4)
vocab = np.zeros(256, np.uint8); vocab[np.unique(g['stream'])] = 1
# V ~ 205 like enwik8: add unused symbols to the vocabulary
extra = [i for i in range(256) if not vocab[i]][: max(0, 205 - int(vocab.sum()))]
vocab[extra] = 1
l = E.Lstm(vocab, 0)
print('V =', l.V)
d_in = torch.from_numpy(g['ppmd_probs'][1:N + 1].copy()).cuda()
d_b = torch.from_numpy(g['stream'][:N].copy()).cuda()
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.time()
    out, bp, bx = l.run(d_in, d_b)
    torch.cuda.synchronize(); dt = time.time() - t0
    print('rep %d: %d bytes in %.1f ms -> %.1f us/byte = %.2f us/bit' % (rep, N, dt * 1e3, dt / N * 1e6, dt / N / 8 * 1e6))
