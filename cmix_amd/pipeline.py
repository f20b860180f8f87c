"""Host-side drivers of one stream's stage pipeline (what `Predictor` is for a compressor): `EngineStream` feeds a stream through
the whole engine a sub-chunk at a time (every model family a stage: contexts + small models, PPMd host stage, LSTM, fxcm, paq8, and
the final mixing network, each on its own HIP stream, each writing its columns of the sub-chunk's layer-0 matrix in place) and codes
the probabilities; `compress_stream` does the same for callers that bring some columns themselves. Used by bench.py,
cmix_amd.multifile, scripts/gpu_multistream_engine.py; tests/test_gpu_pipeline.py."""
import numpy as np

from . import engine as E
from . import synth


def compress_stream(stream, vocab, layer0, device_index=0, chunk_bytes=4096, pretrain=None, dictionary_used=False):
    """What runner.cpp's RunCompression does after preprocessing (runner.cpp:196-208): header, then every byte of
    the (already preprocessed) stream through the predictor and the arithmetic coder. The prediction runs a chunk
    at a time on the device; p[] then comes back to the host, where the coder (a few ns per bit against the
    predictor's microseconds) consumes it. `layer0` is the stream's [8n, 2078] device matrix holding the columns no stage produces yet
    (fxcm, paq8); the stages fill in theirs. Returns the container bytes (header + code)."""
    import torch
    stream = np.ascontiguousarray(np.frombuffer(bytes(stream), np.uint8))
    n = len(stream)
    if tuple(layer0.shape) != (8 * n, 2078) or layer0.dtype != torch.float32 or not layer0.is_cuda:
        raise E.CmxError("compress_stream: layer0 must be a float32 [8n, 2078] device matrix")
    dev = layer0.device
    pipe = E.Pipeline(vocab, device_index, max(1, min(chunk_bytes, n)))
    enc = E.Encoder()
    try:
        if pretrain:
            pipe.pretrain(pretrain)
        p_dev = torch.empty(8 * n, dtype=torch.float32, device=dev)
        p_host = torch.empty(8 * n, dtype=torch.float32).pin_memory()
        edges = list(range(0, n, chunk_bytes)) + [n]
        torch.cuda.synchronize(dev)
        for a, b in zip(edges[:-1], edges[1:]):
            pipe.submit(stream[a:b], layer0[8 * a:8 * b], p_dev[8 * a:8 * b])
        pipe.sync()
        p_host.copy_(p_dev)
        enc.encode_bytes(p_host.numpy(), stream)
        enc.flush()
        return E.header_write(n, vocab, dictionary_used) + enc.data()
    finally:
        enc.close()
        pipe.close()


def text_file_stream(payload):
    """What `cmix -c` without a dictionary hands the predictor for a file its detector classifies as (>= 95 %) text:
    preprocessor::Encode writes ONE block -- type byte TEXT = 4, the file length big-endian -- and encode_text a WRT flag 0
    followed by the bytes (reference src/preprocess/preprocessor.cpp:533-540, 443-449). The enwik8-shaped bench shards
    are such files from 64 KB on; bench.py checks the resulting .cmix file against the reference binary's
    (tests/golden/dropin_*.npz), which pins this framing. (Mixed files are split into typed blocks by the reference's
    detector; the engine compressor integration/compress_engine.cpp runs the reference's own preprocessor.)"""
    n = len(payload)
    return bytes([4]) + n.to_bytes(4, "big") + b"\x00" + bytes(payload)


class EngineStream:
    """One input stream through the WHOLE engine on one GPU: every model family has a stage (contexts + small models,
    PPMd host stage, LSTM, fxcm, paq8) and the final mixing network consumes their 2078 columns -- what `Predictor`
    is for a compressor (reference src/predictor.cpp:361-469), a sub-chunk of known bytes at a time. feed() enqueues;
    finish() returns the container bytes (header + arithmetic code), identical to the reference binary's file."""

    def __init__(self, device_index, stream, sub_chunk=4096, dictionary_used=False, vocab=None):
        import torch
        self.torch = torch
        self.dev = torch.device("cuda", device_index)
        self.stream = np.ascontiguousarray(np.frombuffer(bytes(stream), np.uint8))
        n = len(self.stream)
        self.sub = max(1, min(sub_chunk, n))
        self.vocab = np.ones(256, np.uint8)
        if vocab is not None:   # the HEAD of a longer stream is run (diagnosis scripts): the vocabulary is the whole file's
            self.vocab = np.ascontiguousarray(vocab, np.uint8).copy()
        elif n >= 10000:  # kMinVocabFileSize (runner.cpp:14,196-199)
            self.vocab = np.zeros(256, np.uint8)
            self.vocab[np.unique(self.stream)] = 1
        self.header = E.header_write(n, self.vocab, dictionary_used)
        self.layer0 = [torch.empty((8 * self.sub, E.N_INPUTS), dtype=torch.float32, device=self.dev) for _ in range(E.PIPELINE_SLOTS)]
        self.p_dev = torch.empty(8 * n, dtype=torch.float32, device=self.dev)
        self.pipe = E.Pipeline(self.vocab, device_index, self.sub)
        self.pipe.enable_fxcm(None)
        self.pipe.enable_paq8()
        self.pos = 0
        self.nsub = 0
        torch.cuda.synchronize(self.dev)

    def feed(self, nbytes):
        """The next nbytes bytes of the stream, in sub-chunks (asynchronous; up to E.PIPELINE_SLOTS sub-chunks in flight)."""
        end = min(self.pos + nbytes, len(self.stream))
        while self.pos < end:
            m = min(self.sub, end - self.pos)
            l0 = self.layer0[self.nsub % E.PIPELINE_SLOTS][:8 * m]
            self.pipe.submit(self.stream[self.pos:self.pos + m], l0, self.p_dev[8 * self.pos:8 * (self.pos + m)])
            self.pos += m
            self.nsub += 1

    def finish(self):
        """Wait for the device, bring p[] back, run the arithmetic coder (host: a multiply-add and a compare per bit)."""
        self.pipe.sync()
        p = self.p_dev[:8 * self.pos].cpu().numpy()
        enc = E.Encoder()
        try:
            enc.encode_bytes(p, self.stream[:self.pos])
            enc.flush()
            return self.header + enc.data()
        finally:
            enc.close()

    def close(self):
        self.pipe.close()
