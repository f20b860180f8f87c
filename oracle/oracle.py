"""oracle/oracle.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding for oracle/_build/libcmixoracle.so (the plain-C restatement in
oracle/*.c).  Importers allowed: tests/, __graft_entry__.smoke(), bench.py's
cpu_baseline leg.  The product package cmix_amd/ must never import this.
"""
import contextlib
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_build", "libcmixoracle.so")

N_IN0, N_MIX = 2078, 47


def build(force=False):
    srcs = [f for f in os.listdir(HERE) if f.endswith((".c", ".h"))]
    newest = max(os.path.getmtime(os.path.join(HERE, f)) for f in srcs)
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < newest:
        subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(LIB_PATH)
        L.orc_logistic.restype = C.c_float
        L.orc_logistic.argtypes = [C.c_float]
        L.orc_stretch.restype = C.c_float
        L.orc_stretch.argtypes = [C.c_float]
        L.orc_mixnet_create.restype = C.c_void_p
        L.orc_mixnet_destroy.argtypes = [C.c_void_p]
        L.orc_mixnet_step.restype = C.c_float
        L.orc_mixnet_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_mixnet_aux_context.restype = C.c_uint64
        L.orc_mixnet_aux_context.argtypes = [C.c_void_p]
        L.orc_sse_create.restype = C.c_void_p
        L.orc_sse_destroy.argtypes = [C.c_void_p]
        L.orc_sse_predict.restype = C.c_float
        L.orc_sse_predict.argtypes = [C.c_void_p, C.c_float]
        L.orc_sse_perceive.argtypes = [C.c_void_p, C.c_int]
        L.orc_coder_encode.restype = C.c_size_t
        L.orc_coder_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        L.orc_coder_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.orc_header_write.restype = C.c_size_t
        L.orc_header_write.argtypes = [C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
        for name in ("orc_p8_mixer_new", "orc_p8_apm1_new", "orc_p8_statemap_new", "orc_p8_statemap32_new", "orc_p8_apm_new"):
            getattr(L, name).restype = C.c_void_p
        L.orc_p8_mixer_new.argtypes = [C.c_int] * 4
        L.orc_p8_mixer_free.argtypes = [C.c_void_p]
        L.orc_p8_mixer_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                        C.c_void_p, C.c_void_p]
        L.orc_p8_apm1_new.argtypes = [C.c_int]
        L.orc_p8_apm1_p.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_p8_statemap_p.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_p8_statemap32_new.argtypes = [C.c_int]
        L.orc_p8_statemap32_p.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_p8_apm_new.argtypes = [C.c_int]
        L.orc_p8_apm_p.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        L.orc_p8_cm2_new.restype = C.c_void_p
        L.orc_p8_cm2_new.argtypes = [C.c_uint64, C.c_uint32]
        L.orc_p8_cm2_free.argtypes = [C.c_void_p]
        L.orc_p8_cm2_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_p8_cm_new.restype = C.c_void_p
        L.orc_p8_cm_new.argtypes = [C.c_uint64, C.c_int]
        L.orc_p8_cm_free.argtypes = [C.c_void_p]
        L.orc_p8_cm_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_p8_rnd_next.restype = C.c_uint32
        L.orc_p8_rcm_new.restype = C.c_void_p
        L.orc_p8_rcm_new.argtypes = [C.c_int]
        L.orc_p8_rcm_set.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
        L.orc_p8_rcm_mix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_p8_dmc_new.restype = C.c_void_p
        L.orc_p8_dmc_new.argtypes = [C.c_int]
        L.orc_p8_dmc_free.argtypes = [C.c_void_p]
        L.orc_p8_dmc_mix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_p8_lpm_new.restype = C.c_void_p
        L.orc_p8_lpm_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_p8_ctxmodel_new.restype = C.c_void_p
        L.orc_p8_ctxmodel_new.argtypes = [C.c_int, C.c_int]
        L.orc_p8_ctxmodel_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                           C.c_void_p]
        L.orc_p8_hash3.restype = C.c_uint64
        L.orc_p8_hash3.argtypes = [C.c_uint64] * 3
        L.orc_p8_hash6.restype = C.c_uint64
        L.orc_p8_hash6.argtypes = [C.c_uint64] * 6
        L.orc_p8_sparse_new.restype = C.c_void_p
        L.orc_p8_sparse_new.argtypes = [C.c_int, C.c_int]
        L.orc_p8_sparse_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_p8_match_new.restype = C.c_void_p
        L.orc_p8_match_new.argtypes = [C.c_uint32]
        L.orc_p8_match_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        L.orc_p8_sparsematch_new.restype = C.c_void_p
        L.orc_p8_sparsematch_new.argtypes = [C.c_uint64]
        L.orc_p8_sparsematch_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p,
                                              C.c_void_p, C.c_void_p]
        L.orc_p8_small_new.restype = C.c_void_p
        L.orc_p8_small_new.argtypes = [C.c_int]
        L.orc_p8_small_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                                        C.c_uint32, C.c_int, C.c_void_p]
        L.orc_p8_record_new.restype = C.c_void_p
        L.orc_p8_record_new.argtypes = [C.c_int]
        L.orc_p8_record_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32,
                                         C.c_int, C.c_void_p, C.c_void_p]
        L.orc_p8_xml_new.restype = C.c_void_p
        L.orc_p8_xml_new.argtypes = [C.c_int]
        L.orc_p8_xml_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_uint32, C.c_int, C.c_void_p,
                                      C.c_void_p]
        L.orc_p8_exe_new.restype = C.c_void_p
        L.orc_p8_exe_new.argtypes = [C.c_int]
        L.orc_p8_exe_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_uint32, C.c_int,
                                      C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_p8_word_new.restype = C.c_void_p
        L.orc_p8_word_new.argtypes = [C.c_int]
        L.orc_p8_word_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                       C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_p8_dmap_new.restype = C.c_void_p
        L.orc_p8_dmap_new.argtypes = [C.c_int] * 4
        L.orc_p8_dmap_set_direct.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_p8_dmap_set.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_p8_dmap_mix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        L.orc_p8_hash2.restype = C.c_uint64
        L.orc_p8_hash2.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_p8_combine64.restype = C.c_uint64
        L.orc_p8_combine64.argtypes = [C.c_uint64, C.c_uint64]
        L.orc_p8_finalize64.restype = C.c_uint32
        L.orc_p8_finalize64.argtypes = [C.c_uint64, C.c_int]
        L.orc_p8_checksum64.restype = C.c_uint64
        L.orc_p8_checksum64.argtypes = [C.c_uint64, C.c_int, C.c_int]
        L.orc_lstm_create.restype = C.c_void_p
        L.orc_lstm_create.argtypes = [C.c_void_p, C.c_int]
        L.orc_lstm_destroy.argtypes = [C.c_void_p]
        L.orc_lstm_vocab_size.argtypes = [C.c_void_p]
        L.orc_lstm_gate_rowlen.argtypes = [C.c_void_p, C.c_int]
        L.orc_lstm_gate_weights.restype = C.POINTER(C.c_float)
        L.orc_lstm_gate_weights.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.orc_lstm_byte_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.orc_lstm_bit_predict.restype = C.c_float
        L.orc_lstm_bit_predict.argtypes = [C.c_void_p]
        L.orc_lstm_bit_perceive.argtypes = [C.c_void_p, C.c_int]
        L.orc_lstm_probs.restype = C.POINTER(C.c_float)
        L.orc_lstm_probs.argtypes = [C.c_void_p]
        L.orc_lstm_ex.argtypes = [C.c_void_p]
        L.orc_ctx_create.restype = C.c_void_p
        L.orc_ctx_create.argtypes = [C.c_void_p]
        L.orc_ctx_destroy.argtypes = [C.c_void_p]
        L.orc_ctx_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ctx_perceive.argtypes = [C.c_void_p, C.c_int]
        L.orc_ctx_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.orc_ctx_get_manager.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_ctx_bracket_probs.restype = C.POINTER(C.c_float)
        L.orc_ctx_bracket_probs.argtypes = [C.c_void_p]
        L.orc_ctx_indirect_offset.restype = C.c_uint64
        L.orc_ctx_indirect_offset.argtypes = [C.c_void_p, C.c_int]
        L.orc_ctx_model_column.argtypes = [C.c_void_p, C.c_int]
        _lib = L
    return _lib


# ---- lifetime of the oracle's models (orc_alloc.h): they are C structs without destructors ----------------------------------------
@contextlib.contextmanager
def scope():
    """Everything the oracle allocates inside the block and has not freed itself is freed at its end. Do not create MixNet / SSE /
    Lstm / CtxModels objects inside (their __del__ frees through the oracle's own destroy functions)."""
    L = lib()
    L.orc_scope_begin.restype = C.c_uint32
    L.orc_scope_begin()
    try:
        yield
    finally:
        L.orc_scope_end()


def new_owned(ctor, *args):
    """(handle, tag) of a model built by `ctor(*args)`: release(tag) frees every table the constructor allocated."""
    L = lib()
    L.orc_scope_begin.restype = C.c_uint32
    tag = L.orc_scope_begin()
    try:
        h = ctor(*args)
    finally:
        L.orc_scope_pause()
    return h, tag


def release(tag):
    L = lib()
    L.orc_scope_free.argtypes = [C.c_uint32]
    L.orc_scope_free(tag)


def logistic(x):
    return np.float32(lib().orc_logistic(C.c_float(float(x))))


def stretch(p):
    return np.float32(lib().orc_stretch(C.c_float(float(p))))


def logit_table():
    out = np.empty(100001, np.float32)
    lib().orc_logit_table(out.ctypes.data_as(C.c_void_p))
    return out


def sse_tables():
    st = np.empty(32768, np.uint16)
    sq = np.empty(32768, np.uint16)
    lib().orc_sse_tables(st.ctypes.data_as(C.c_void_p), sq.ctypes.data_as(C.c_void_p))
    return st, sq


class MixNet:
    """Layers 0-2 + SSE. step() = one Predict()+Perceive(bit)."""

    def __init__(self):
        self.h = lib().orc_mixnet_create()
        self._mix = np.empty(N_MIX, np.float32)

    def step(self, probs, sel, bit, want_mix=False):
        probs = np.ascontiguousarray(probs, np.float32)
        sel = np.ascontiguousarray(sel, np.uint64)
        assert probs.shape == (N_IN0,) and sel.shape == (N_MIX,)
        p = lib().orc_mixnet_step(self.h, probs.ctypes.data_as(C.c_void_p),
                                  sel.ctypes.data_as(C.c_void_p), int(bit),
                                  self._mix.ctypes.data_as(C.c_void_p) if want_mix else None)
        return (np.float32(p), self._mix.copy()) if want_mix else np.float32(p)

    def run(self, probs, sel, bits):
        """probs [T,2078] f32, sel [T,47] u64, bits [T] -> p [T] f32"""
        T = len(bits)
        out = np.empty(T, np.float32)
        for t in range(T):
            out[t] = self.step(probs[t], sel[t], bits[t])
        return out

    def aux_context(self):
        return int(lib().orc_mixnet_aux_context(self.h))

    def set_steps(self, steps):
        """State injection: Mixer::steps_ of all 47 mixers (tests/golden/make_wrap_traces.py)."""
        lib().orc_mixnet_set_steps.argtypes = [C.c_void_p, C.c_uint64]
        lib().orc_mixnet_set_steps(self.h, int(steps))

    def close(self):
        if self.h:
            lib().orc_mixnet_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


class SSE:
    def __init__(self):
        self.h = lib().orc_sse_create()

    def predict(self, p):
        return np.float32(lib().orc_sse_predict(self.h, C.c_float(float(p))))

    def perceive(self, bit):
        lib().orc_sse_perceive(self.h, int(bit))

    def __del__(self):
        if self.h:
            lib().orc_sse_destroy(self.h)
            self.h = None


class Lstm:
    """Byte-level LSTM byte mixer + its ByteModel bit interface."""
    SKIP_RAND = 31  # Indirect constructors before the LSTM in Predictor::Predictor

    def __init__(self, vocab, skip_rand=SKIP_RAND):
        vocab = np.ascontiguousarray(vocab, np.uint8)
        self.h = lib().orc_lstm_create(vocab.ctypes.data, skip_rand)
        self.V = lib().orc_lstm_vocab_size(self.h)

    def gate_weights(self, layer, gate):
        n = lib().orc_lstm_gate_rowlen(self.h, layer)
        p = lib().orc_lstm_gate_weights(self.h, layer, gate)
        return np.ctypeslib.as_array(p, shape=(200, n)).copy()

    def byte_update(self, in256, byte):
        in256 = np.ascontiguousarray(in256, np.float32)
        lib().orc_lstm_byte_update(self.h, in256.ctypes.data, int(byte))
        return self.probs()

    def probs(self):
        return np.ctypeslib.as_array(lib().orc_lstm_probs(self.h), shape=(256,)).copy()

    def bit_predict(self):
        return np.float32(lib().orc_lstm_bit_predict(self.h))

    def bit_perceive(self, bit):
        lib().orc_lstm_bit_perceive(self.h, int(bit))

    def ex(self):
        return lib().orc_lstm_ex(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_lstm_destroy(self.h)
            self.h = None


SMALL_COLS = np.array([0, 1, 2] + list(range(2025, 2076)))  # layer-0 columns of the 54 small models


class CtxModels:
    """ContextManager + contexts + the 54 small native models. run(bytes) walks Predict/Perceive."""

    def __init__(self, vocab):
        vocab = np.ascontiguousarray(vocab, np.uint8)
        self.h = lib().orc_ctx_create(vocab.ctypes.data)

    def predict(self, want_sel=True):
        p = np.empty(54, np.float32)
        s = np.empty(47, np.uint64)
        lib().orc_ctx_predict(self.h, p.ctypes.data, s.ctypes.data if want_sel else None)
        return p, s

    def perceive(self, bit):
        lib().orc_ctx_perceive(self.h, int(bit))

    def manager(self):
        regs = np.empty(25, np.uint64)
        ctx = np.empty(54, np.uint64)
        bctx = np.empty(8, np.uint64)
        lib().orc_ctx_get_manager(self.h, regs.ctypes.data, ctx.ctypes.data, bctx.ctypes.data)
        return regs, ctx, bctx

    def bracket_probs(self):
        return np.ctypeslib.as_array(lib().orc_ctx_bracket_probs(self.h), shape=(256,)).copy()

    def set_history(self, pos, tail):
        """State injection: the history ring's write position, the Match models' byte counters, the bytes in front of it."""
        tail = np.ascontiguousarray(np.frombuffer(bytes(tail), np.uint8))
        lib().orc_ctx_set_history.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        lib().orc_ctx_set_history(self.h, int(pos), tail.ctypes.data, len(tail))

    def run(self, data):
        """data: bytes -> (probs [T,54] f32, sel [T,47] u64)"""
        data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
        T = 8 * len(data)
        probs = np.empty((T, 54), np.float32)
        sel = np.empty((T, 47), np.uint64)
        lib().orc_ctx_run(self.h, data.ctypes.data, len(data), probs.ctypes.data, sel.ctypes.data)
        return probs, sel

    def close(self):
        if getattr(self, "h", None):
            lib().orc_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()


def coder_encode(p, bits):
    """Encoder::Encode over (p[t], bits[t]) then Flush (encoder.cpp) -> code bytes."""
    p = np.ascontiguousarray(p, np.float32)
    bits = np.ascontiguousarray(bits, np.uint8)
    assert p.shape == bits.shape
    out = np.empty(len(p) * 2 + 16, np.uint8)
    n = lib().orc_coder_encode(p.ctypes.data, bits.ctypes.data, len(p), out.ctypes.data, len(out))
    assert n != C.c_size_t(-1).value
    return out[:n].tobytes()


def coder_decode(p, code):
    """Decoder::Decode replaying p[] (decoder.cpp) -> bits."""
    p = np.ascontiguousarray(p, np.float32)
    code = np.ascontiguousarray(np.frombuffer(bytes(code), np.uint8))
    bits = np.empty(len(p), np.uint8)
    lib().orc_coder_decode(p.ctypes.data, len(p), code.ctypes.data, len(code), bits.ctypes.data)
    return bits


def header_write(length, vocab, dictionary_used=False):
    out = np.zeros(37, np.uint8)
    vocab = np.ascontiguousarray(vocab, np.uint8)
    n = lib().orc_header_write(int(length), vocab.ctypes.data, int(bool(dictionary_used)), out.ctypes.data)
    return out[:n].tobytes()
