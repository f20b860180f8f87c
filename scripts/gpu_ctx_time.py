"""Timing of the context / small-model stage kernel (debug aid; not the bench)."""
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from cmix_amd import engine as E, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
data = np.frombuffer(synth.enwik_like(N + 4096, 1000)[4096:4096 + N], np.uint8).copy()
vocab = np.zeros(256, np.uint8); vocab[np.unique(data)] = 1
c = E.CtxModels(vocab, 0)
d = torch.from_numpy(data).cuda()
probs = torch.empty((8 * N, 2078), dtype=torch.float32, device='cuda')
sel = torch.empty((8 * N, 47), dtype=torch.int32, device='cuda')
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.time()
    c.run(d, probs, sel); c.sync()
    dt = time.time() - t0
    print('rep %d: %d bytes in %.1f ms -> %.2f us/byte = %.3f us/bit' % (rep, N, dt * 1e3, dt / N * 1e6, dt / N / 8 * 1e6))
