"""Host-side driver of one stream's stage pipeline (what `Predictor` becomes once every stage exists):
PPMd on a host core, then per chunk the context/small-model stage, the LSTM byte mixer and the final
mixing network on their own HIP streams, each writing its columns of the chunk's layer-0 matrix in place.
Used by bench.py, scripts/gpu_multistream.py; tests/test_gpu_pipeline.py does the same by hand.

Until the fxcm and paq8 stages exist their columns (3..2024) are a seeded stand-in with the
reference's value grid (k/4095), generated on the device (`standin_columns`)."""
import numpy as np

from . import engine as E
from . import synth


def standin_columns(nbytes, seed, device):
    """Seeded stand-in for the fxcm/paq8 model columns (generated on the device) + the text and its bits."""
    import torch
    from cmix_amd import synth
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    T = nbytes * 8
    text = np.frombuffer(synth.enwik_like(nbytes + 8, seed), np.uint8)[:nbytes]
    bits_np = np.unpackbits(text)  # MSB first, as runner.cpp:106-108 feeds the coder
    bits = torch.from_numpy(bits_np).to(device)
    k = torch.randint(0, 4096, (T, 2078), generator=g, device=device, dtype=torch.int32)
    conf = torch.rand((T, 2078), generator=g, device=device) < 0.5
    side = torch.rand((T, 2078), generator=g, device=device) < 0.5
    k = torch.where(conf, torch.where(side, k % 200, 4095 - (k % 200)), k)
    probs = k.to(torch.float32) * np.float32(1.0 / 4095)
    probs[:, 2025:2078] = torch.rand((T, 53), generator=g, device=device)
    probs[:, 432:434] = 0.5
    return probs.contiguous(), bits.contiguous(), text


class StreamPipeline:
    """One input stream on one GPU: buffers for `nchunks` chunks, three HIP streams, step(i) enqueues chunk i."""

    def __init__(self, device_index, seed, chunk_bytes, nchunks):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device_index)
        self.local = device_index
        self.chunk_bytes, self.cb, self.nchunks = chunk_bytes, chunk_bytes * 8, nchunks
        nbytes = chunk_bytes * nchunks
        self.probs, self.bits, text = standin_columns(nbytes, seed, self.dev)
        self.text = np.ascontiguousarray(text)
        self.vocab = np.zeros(256, np.uint8)
        self.vocab[np.unique(self.text)] = 1
        self.d_bytes = torch.from_numpy(self.text.copy()).to(self.dev)
        # PPMd = host stage: runs on a host core inside the timed loop and ships 1 KB per byte to HBM
        self.host_ppmd = E.Ppmd(self.vocab)
        self.ppmd = torch.empty((nbytes + 1, 256), dtype=torch.float32, device=self.dev)
        self.ppmd[0] = 1.0 / 256  # ByteModel constructor (byte-model.cpp:5-6)
        self.pp_host = torch.empty((nbytes, 256), dtype=torch.float32).pin_memory()
        self.sel = torch.zeros((nbytes * 8, 47), dtype=torch.int32, device=self.dev)
        self.net, self.ctx, self.lstm = E.MixNet(device_index), E.CtxModels(self.vocab, device_index), E.Lstm(self.vocab, device_index)
        self.st_mix, self.st_ctx, self.st_lstm = (torch.cuda.Stream(self.dev) for _ in range(3))
        self.p_out = torch.empty(self.cb * nchunks, dtype=torch.float32, device=self.dev)
        self.lstm_out = torch.empty((chunk_bytes, 256), dtype=torch.float32, device=self.dev)
        self.ev = [[torch.cuda.Event(enable_timing=True) for _ in range(6)] for _ in range(nchunks)]

    def step(self, i):
        """Chunk i through all stages, asynchronously: chunk i+1's context/LSTM stages run under chunk i's
        mixing network; the mixing network consumes a chunk once the other stages have written their columns."""
        torch, cb = self.torch, self.cb
        r = slice(i * cb, (i + 1) * cb)
        n0, n1 = i * self.chunk_bytes, (i + 1) * self.chunk_bytes
        ev = self.ev[i]
        self.pp_host[n0:n1] = torch.from_numpy(self.host_ppmd.run(self.text[n0:n1]))
        with torch.cuda.stream(self.st_lstm):
            self.ppmd[n0 + 1:n1 + 1].copy_(self.pp_host[n0:n1], non_blocking=True)
            ev_pp = torch.cuda.Event()
            ev_pp.record(self.st_lstm)
        self.st_ctx.wait_event(ev_pp)
        ev[0].record(self.st_ctx)
        self.ctx.run(self.d_bytes[n0:n1], self.probs[r], self.sel[r], stream=self.st_ctx.cuda_stream)
        E.bytemodel_bits(self.ppmd[n0], self.ppmd[n0 + 1:n1 + 1], self.d_bytes[n0:n1], self.probs[r], 2076, self.local,
                         self.st_ctx.cuda_stream)
        ev[1].record(self.st_ctx)
        ev[2].record(self.st_lstm)
        self.lstm.run(self.ppmd[n0 + 1:n1 + 1], self.d_bytes[n0:n1], layer0=self.probs[r], out=self.lstm_out,
                      stream=self.st_lstm.cuda_stream)
        ev[3].record(self.st_lstm)
        self.st_mix.wait_event(ev[1])
        self.st_mix.wait_event(ev[3])
        ev[4].record(self.st_mix)  # HIP events on the stream the mixing-network kernel is launched on
        self.net.run(self.probs[r], self.sel[r], self.bits[r], self.p_out[r], stream=self.st_mix.cuda_stream)
        ev[5].record(self.st_mix)

    def stage_ms(self, first, last):
        """Mean HIP-event time per chunk of (mixnet, ctxmodels, lstm) over chunks [first, last)."""
        m = lambda a, b: float(np.mean([self.ev[i][a].elapsed_time(self.ev[i][b]) for i in range(first, last)]))
        return m(4, 5), m(0, 1), m(2, 3)

    def sync(self):
        self.ctx.sync()
        self.net.sync()

    def close(self):
        for o in (self.net, self.ctx, self.lstm, self.host_ppmd):
            o.close()
