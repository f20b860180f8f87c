"""oracle/refharness.py -- TEST INFRASTRUCTURE ONLY.

ctypes binding for oracle/_ref/libcmixref.so: the unmodified reference
Predictor (reference src/predictor.cpp) plus oracle/ref_harness.cpp's state
read-out.  Used to (a) generate the golden fixtures under tests/golden/ and
(b) pin the C restatement in oracle/*.c.  Never imported by the product
(cmix_amd/) -- see DESIGN.md "oracle".

The reference keeps paq8/fxcm state in process globals, so one Predictor per
process: callers that need several runs use `run_in_subprocess`.
"""
import ctypes as C
import os
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "_ref", "libcmixref.so")

N_IN0, N_IN1, N_IN2 = 2078, 29, 49
N_MIX = (26, 20, 1)


def available():
    return os.path.exists(LIB_PATH)


class Ref:
    def __init__(self, vocab=None, dict_path=None):
        self.lib = C.CDLL(LIB_PATH)
        L = self.lib
        L.ref_predict.restype = C.c_float
        L.ref_logistic.restype = C.c_float
        L.ref_logistic.argtypes = [C.c_float]
        L.ref_logit.restype = C.c_float
        L.ref_logit.argtypes = [C.c_float]
        L.ref_mixer_steps.restype = C.c_uint64
        L.ref_mixer_rows.restype = C.c_uint64
        L.ref_mixer_lr.restype = C.c_float
        L.ref_context_size.restype = C.c_uint64
        if vocab is None:
            vocab = np.ones(256, np.uint8)
        vocab = np.ascontiguousarray(vocab, np.uint8)
        rc = L.ref_create(vocab.ctypes.data_as(C.c_void_p),
                          (dict_path or "").encode())
        if rc != 0:
            raise RuntimeError("reference Predictor already created in this process")
        self.n_in = [L.ref_num_inputs(i) for i in range(3)]
        self.n_mix = [L.ref_num_mixers(i) for i in range(3)]
        self.n_ctx = L.ref_num_contexts()
        self.n_bitctx = L.ref_num_bit_contexts()
        self.aux = [L.ref_auxiliary(i) for i in range(3)]

    def predict(self):
        return np.float32(self.lib.ref_predict())

    def perceive(self, bit):
        self.lib.ref_perceive(int(bit))

    def pretrain(self, bit):
        self.lib.ref_pretrain(int(bit))

    def model_probs(self):
        out = np.empty(self.n_in[0], np.float32)
        n = self.lib.ref_get_model_probs(out.ctypes.data_as(C.c_void_p))
        assert n == self.n_in[0]
        return out

    def layer_inputs(self, layer):
        out = np.empty(self.n_in[layer], np.float32)
        self.lib.ref_get_layer_inputs(layer, out.ctypes.data_as(C.c_void_p))
        return out

    def mixers(self, layer):
        ctx = np.empty(self.n_mix[layer], np.uint64)
        out = np.empty(self.n_mix[layer], np.float32)
        self.lib.ref_get_mixers(layer, ctx.ctypes.data_as(C.c_void_p),
                                out.ctypes.data_as(C.c_void_p))
        return ctx, out

    def mixer_row(self, layer, k):
        w = np.zeros(self.n_in[layer], np.float32)
        ew = np.zeros(max(k, 1), np.float32)
        steps = C.c_uint64(0)
        n = self.lib.ref_get_mixer_row(layer, k, w.ctypes.data_as(C.c_void_p),
                                       ew.ctypes.data_as(C.c_void_p), C.byref(steps))
        return (w, ew[:k], steps.value) if n >= 0 else None

    def manager(self):
        regs = np.empty(25, np.uint64)
        ctx = np.empty(self.n_ctx, np.uint64)
        bctx = np.empty(self.n_bitctx, np.uint64)
        self.lib.ref_get_manager(regs.ctypes.data_as(C.c_void_p),
                                 ctx.ctypes.data_as(C.c_void_p),
                                 bctx.ctypes.data_as(C.c_void_p))
        return regs, ctx, bctx

    # ---- state injection (round 6; oracle/ref_harness.cpp): counters placed where a stream would take them after hundreds of megabytes ----
    def set_mixer_steps(self, steps):
        self.lib.ref_debug_set_mixer_steps.argtypes = [C.c_uint64]
        return self.lib.ref_debug_set_mixer_steps(int(steps))

    def set_history(self, pos, tail):
        tail = np.ascontiguousarray(np.frombuffer(bytes(tail), np.uint8))
        self.lib.ref_debug_set_history.argtypes = [C.c_uint64, C.c_void_p, C.c_uint64]
        return self.lib.ref_debug_set_history(int(pos), tail.ctypes.data_as(C.c_void_p), len(tail))

    def context_sizes(self):
        return np.array([self.lib.ref_context_size(i) for i in range(self.n_ctx)], np.uint64)

    def byte_probs(self, which):
        out = np.empty(256, np.float32)
        tbe = (C.c_int * 3)()
        self.lib.ref_get_byte_probs(which, out.ctypes.data_as(C.c_void_p), tbe)
        return out, tuple(tbe)

    def lstm_dims(self):
        d = (C.c_int * 6)()
        self.lib.ref_lstm_dims(d)
        return dict(input_size=d[0], output_size=d[1], cells=d[2], layers=d[3],
                    horizon=d[4], epoch=d[5])

    def lstm_hidden(self):
        d = self.lstm_dims()
        out = np.empty(d["cells"] * d["layers"] + 1, np.float32)
        self.lib.ref_lstm_hidden(out.ctypes.data_as(C.c_void_p))
        return out

    def lstm_gate_weights(self, layer, gate):
        d = self.lstm_dims()
        n = self.lib.ref_lstm_gate_row_len(layer)
        out = np.empty((d["cells"], n), np.float32)
        self.lib.ref_lstm_gate_weights(layer, gate, out.ctypes.data_as(C.c_void_p))
        return out

    def lstm_output_layer(self, epoch):
        d = self.lstm_dims()
        out = np.empty((d["output_size"], d["cells"] * d["layers"] + 1), np.float32)
        self.lib.ref_lstm_output_layer(epoch, out.ctypes.data_as(C.c_void_p))
        return out

    def logistic(self, x):
        return np.float32(self.lib.ref_logistic(C.c_float(float(x))))

    def logit(self, p):
        return np.float32(self.lib.ref_logit(C.c_float(float(p))))


def vocab_of(data: bytes):
    """Reference src/runner.cpp:88-94,196-202: all 256 below 10000 bytes."""
    v = np.zeros(256, np.uint8)
    if len(data) < 10000:
        v[:] = 1
    else:
        v[np.unique(np.frombuffer(data, np.uint8))] = 1
    return v


CODER_LIB_PATH = os.path.join(HERE, "_ref", "libcmixrefcoder.so")


def coder_available():
    return os.path.exists(CODER_LIB_PATH)


def _coder_lib():
    L = C.CDLL(CODER_LIB_PATH)
    L.refcoder_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_char_p]
    L.refcoder_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_char_p, C.c_void_p]
    return L


def ref_encode(p, bits):
    """The reference's own Encoder (compiled by oracle/ref_coder.cpp) over a recorded p[] -> code bytes."""
    import tempfile
    p = np.ascontiguousarray(p, np.float32)
    bits = np.ascontiguousarray(bits, np.uint8)
    with tempfile.NamedTemporaryFile(suffix=".cmix") as f:
        assert _coder_lib().refcoder_encode(p.ctypes.data, bits.ctypes.data, len(p), f.name.encode()) == 0
        with open(f.name, "rb") as g:
            return g.read()


def ref_decode(p, code):
    import tempfile
    p = np.ascontiguousarray(p, np.float32)
    bits = np.empty(len(p), np.uint8)
    with tempfile.NamedTemporaryFile(suffix=".cmix") as f:
        f.write(bytes(code))
        f.flush()
        assert _coder_lib().refcoder_decode(p.ctypes.data, len(p), f.name.encode(), bits.ctypes.data) == 0
    return bits


PAQ8_LIB_PATH = os.path.join(HERE, "_ref", "libcmixrefpaq8.so")


def paq8core_available():
    return os.path.exists(PAQ8_LIB_PATH)


def paq8core_lib():
    """The reference's own paq8 building blocks (oracle/ref_paq8core.cpp)."""
    L = C.CDLL(PAQ8_LIB_PATH)
    for name in ("refp8_mixer_new", "refp8_apm1_new", "refp8_statemap_new", "refp8_statemap32_new", "refp8_apm_new"):
        getattr(L, name).restype = C.c_void_p
    L.refp8_tables.argtypes = [C.c_void_p] * 4
    L.refp8_mixer_new.argtypes = [C.c_int] * 4
    L.refp8_mixer_free.argtypes = [C.c_void_p]
    L.refp8_mixer_step.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int,
                                   C.c_void_p, C.c_void_p]
    L.refp8_apm1_new.argtypes = [C.c_int]
    L.refp8_apm1_p.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.refp8_statemap_p.argtypes = [C.c_void_p, C.c_int, C.c_int]
    L.refp8_statemap32_new.argtypes = [C.c_int]
    L.refp8_statemap32_p.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
    L.refp8_apm_new.argtypes = [C.c_int]
    L.refp8_apm_p.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
    L.refp8_ilog_table.argtypes = [C.c_void_p]
    L.refp8_word_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]
    L.refp8_exe_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.refp8_xml_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p]
    L.refp8_record_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.refp8_small_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    L.refp8_sparsematch_new.restype = C.c_void_p
    L.refp8_sparsematch_new.argtypes = [C.c_uint64]
    L.refp8_sparsematch_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.refp8_match_new.restype = C.c_void_p
    L.refp8_match_new.argtypes = [C.c_uint32]
    L.refp8_match_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
    L.refp8_sparse_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int,
                                    C.c_void_p]
    L.refp8_ctxmodel_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_int, C.c_void_p,
                                      C.c_int, C.c_void_p]
    L.refp8_hash3.restype = C.c_uint64
    L.refp8_hash3.argtypes = [C.c_uint64] * 3
    L.refp8_hash6.restype = C.c_uint64
    L.refp8_hash6.argtypes = [C.c_uint64] * 6
    L.refp8_lpm_step.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    L.refp8_dmc_new.restype = C.c_void_p
    L.refp8_dmc_new.argtypes = [C.c_int]
    L.refp8_dmc_free.argtypes = [C.c_void_p]
    L.refp8_dmc_mix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.refp8_rcm_new.restype = C.c_void_p
    L.refp8_rcm_new.argtypes = [C.c_int]
    L.refp8_rcm_set.argtypes = [C.c_void_p, C.c_uint64, C.c_int]
    L.refp8_rcm_mix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
    L.refp8_cm_new.restype = C.c_void_p
    L.refp8_cm_new.argtypes = [C.c_uint64, C.c_int]
    L.refp8_cm_free.argtypes = [C.c_void_p]
    L.refp8_cm_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    L.refp8_rnd_next.restype = C.c_uint32
    L.refp8_cm2_new.restype = C.c_void_p
    L.refp8_cm2_new.argtypes = [C.c_uint64, C.c_uint32]
    L.refp8_cm2_free.argtypes = [C.c_void_p]
    L.refp8_cm2_step.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
    for name in ("refp8_sscm_new", "refp8_smap_new", "refp8_imap_new"):
        getattr(L, name).restype = C.c_void_p
    L.refp8_sscm_new.argtypes = [C.c_int, C.c_int]
    L.refp8_sscm_set.argtypes = [C.c_void_p, C.c_uint32]
    L.refp8_sscm_mix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.refp8_smap_new.argtypes = [C.c_int, C.c_int, C.c_int]
    L.refp8_smap_set_direct.argtypes = [C.c_void_p, C.c_uint32]
    L.refp8_smap_set.argtypes = [C.c_void_p, C.c_uint64]
    L.refp8_smap_mix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.refp8_imap_new.argtypes = [C.c_int, C.c_int]
    L.refp8_imap_set_direct.argtypes = [C.c_void_p, C.c_uint32]
    L.refp8_imap_set.argtypes = [C.c_void_p, C.c_uint64]
    L.refp8_imap_mix.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    for name, rt, at in (("refp8_hash2", C.c_uint64, [C.c_uint64, C.c_uint64]), ("refp8_combine64", C.c_uint64, [C.c_uint64, C.c_uint64]),
                         ("refp8_finalize64", C.c_uint32, [C.c_uint64, C.c_int]), ("refp8_checksum64", C.c_uint64, [C.c_uint64, C.c_int, C.c_int])):
        getattr(L, name).restype = rt
        getattr(L, name).argtypes = at
    L.refp8_init_dt()
    return L


FXCM_LIB_PATH = os.path.join(HERE, "_ref", "libcmixreffxcm.so")


def fxcmcore_available():
    return os.path.exists(FXCM_LIB_PATH)


def fxcmcore_lib():
    """The reference's own fxcm building blocks (oracle/ref_fxcmcore.cpp)."""
    L = C.CDLL(FXCM_LIB_PATH)
    L.reffx_init()
    return L
