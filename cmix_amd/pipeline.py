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
    """One input stream on one GPU: operand buffers for `nchunks` chunks; step(i) hands chunk i to the native
    orchestration (cmx_pipeline_submit: PPMd host stage + three device stages on their own HIP streams)."""

    def __init__(self, device_index, seed, chunk_bytes, nchunks):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device_index)
        self.chunk_bytes, self.cb, self.nchunks = chunk_bytes, chunk_bytes * 8, nchunks
        nbytes = chunk_bytes * nchunks
        self.probs, self.bits, text = standin_columns(nbytes, seed, self.dev)
        self.text = np.ascontiguousarray(text)
        self.vocab = np.zeros(256, np.uint8)
        self.vocab[np.unique(self.text)] = 1
        self.p_out = torch.empty(self.cb * nchunks, dtype=torch.float32, device=self.dev)
        torch.cuda.synchronize(self.dev)  # the stand-in columns are complete before any submit
        self.pipe = E.Pipeline(self.vocab, device_index, chunk_bytes)
        self.stage_log = []

    def step(self, i):
        """Chunk i through all stages, asynchronously (two chunks in flight): chunk i+1's PPMd / context / LSTM
        stages run under chunk i's mixing network."""
        cb = self.cb
        r = slice(i * cb, (i + 1) * cb)
        self.pipe.submit(self.text[i * self.chunk_bytes:(i + 1) * self.chunk_bytes], self.probs[r], self.p_out[r])

    def sync(self):
        self.pipe.sync()

    def last_stage_ms(self):
        return self.pipe.last_stage_ms()

    def close(self):
        self.pipe.close()
