"""Host-side Python binding of the C ABI in include/cmix_amd.h (ctypes).

This mirrors, for test and bench drivers, what the C++ shim in INTEGRATION.md
does for the reference's `class Predictor` (reference src/predictor.h:17-22):
it only forwards to libcmixamd.so.  There is no Python or CPU implementation of
the per-bit path behind it -- if the HIP library is missing or no gfx950 device
is visible, construction raises.

torch is used purely as the device-memory allocator for chunk operands
(`tensor.data_ptr()` is what crosses the ABI).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libcmixamd.so")

N_INPUTS, N_MIXERS = 2078, 47
PIPELINE_SLOTS = 8   # CMX_PIPELINE_SLOTS (include/cmix_amd.h): chunks in flight per stream


class CmxError(RuntimeError):
    pass


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CmxError(f"{LIB_PATH} is missing: run `python -m cmix_amd.build` "
                           "(there is no non-HIP fallback)")
        # Load the HIP runtime that PyTorch-ROCm bundles first, so that libcmixamd.so binds to the
        # same libamdhip64 instance (two HIP runtimes in one process cannot both own the GPU).
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
        L = C.CDLL(LIB_PATH)
        L.cmx_last_error.restype = C.c_char_p
        L.cmx_version.restype = C.c_char_p
        L.cmx_p8stage_create.restype = C.c_void_p
        L.cmx_p8stage_create.argtypes = [C.c_int]
        L.cmx_p8stage_destroy.argtypes = [C.c_void_p]
        L.cmx_p8stage_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p]
        L.cmx_p8stage_sync.argtypes = [C.c_void_p]
        L.cmx_p8stage_set_generator_counter.argtypes = [C.c_void_p, C.c_uint32]
        L.cmx_fxcm_create.restype = C.c_void_p
        L.cmx_fxcm_create.argtypes = [C.c_char_p, C.c_int]
        L.cmx_fxcm_destroy.argtypes = [C.c_void_p]
        L.cmx_fxcm_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.cmx_fxcm_sync.argtypes = [C.c_void_p]
        L.cmx_fxcm_failed.argtypes = [C.c_void_p]
        L.cmx_mixnet_create.restype = C.c_void_p
        L.cmx_mixnet_create.argtypes = [C.c_int]
        L.cmx_mixnet_destroy.argtypes = [C.c_void_p]
        L.cmx_mixnet_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmx_mixnet_predict.restype = C.c_float
        L.cmx_mixnet_predict.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmx_mixnet_perceive.argtypes = [C.c_void_p, C.c_int]
        L.cmx_mixnet_sync.argtypes = [C.c_void_p]
        L.cmx_mixnet_bits_done.argtypes = [C.c_void_p, C.c_void_p]
        L.cmx_mixnet_profile.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.cmx_mixnet_last_kernel_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.cmx_mixnet_spec_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.cmx_lstm_create.restype = C.c_void_p
        L.cmx_lstm_create.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cmx_lstm_destroy.argtypes = [C.c_void_p]
        L.cmx_lstm_vocab_size.argtypes = [C.c_void_p]
        L.cmx_lstm_set_tolerance.argtypes = [C.c_void_p, C.c_int]
        L.cmx_lstm_run.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p,
                                   C.c_size_t, C.c_void_p, C.c_void_p]
        L.cmx_bytemodel_bits_run.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                             C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.cmx_lstm_get_gate_weights.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.cmx_lstm_gate_rowlen.argtypes = [C.c_void_p, C.c_int]
        L.cmx_lstm_failed.argtypes = [C.c_void_p]
        L.cmx_glibc_rand_selftest.argtypes = [C.c_uint32, C.c_int, C.c_void_p]
        L.cmx_ctxmodels_create.restype = C.c_void_p
        L.cmx_ctxmodels_create.argtypes = [C.c_void_p, C.c_int]
        L.cmx_ctxmodels_destroy.argtypes = [C.c_void_p]
        L.cmx_ctxmodels_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p,
                                        C.c_void_p]
        L.cmx_ctxmodels_pretrain.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.cmx_ctxmodels_sync.argtypes = [C.c_void_p]
        L.cmx_ctxmodels_get_manager.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmx_ppmd_create.restype = C.c_void_p
        L.cmx_ppmd_create.argtypes = [C.c_void_p]
        L.cmx_ppmd_create_ex.restype = C.c_void_p
        L.cmx_ppmd_create_ex.argtypes = [C.c_void_p, C.c_int, C.c_int]
        L.cmx_ppmd_destroy.argtypes = [C.c_void_p]
        L.cmx_ppmd_run.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.cmx_pipeline_create.restype = C.c_void_p
        L.cmx_pipeline_create.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
        L.cmx_pipeline_destroy.argtypes = [C.c_void_p]
        L.cmx_pipeline_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.cmx_create.restype = C.c_void_p
        L.cmx_create.argtypes = [C.c_void_p, C.c_char_p, C.c_int]
        L.cmx_destroy.argtypes = [C.c_void_p]
        L.cmx_predict.restype = C.c_float
        L.cmx_predict.argtypes = [C.c_void_p]
        L.cmx_perceive.argtypes = [C.c_void_p, C.c_int]
        L.cmx_pretrain.argtypes = [C.c_void_p, C.c_int]
        L.cmx_set_model_outputs.argtypes = [C.c_void_p, C.c_void_p]
        L.cmx_get_lstm_hint.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmx_stage_input.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.cmx_mode.argtypes = [C.c_void_p, C.c_void_p]
        L.cmx_debug_last_row.restype = C.c_void_p
        L.cmx_debug_last_row.argtypes = [C.c_void_p]
        L.cmx_ctxmodels_debug_slow_bytes.argtypes = [C.c_void_p, C.c_void_p]
        L.cmx_ctxmodels_peek.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
        L.cmx_encoder_create.restype = C.c_void_p
        L.cmx_encoder_destroy.argtypes = [C.c_void_p]
        L.cmx_encoder_encode_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.cmx_encoder_encode_bytes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        L.cmx_encoder_flush.argtypes = [C.c_void_p]
        L.cmx_encoder_size.restype = C.c_size_t
        L.cmx_encoder_size.argtypes = [C.c_void_p]
        L.cmx_encoder_data.restype = C.c_void_p
        L.cmx_encoder_data.argtypes = [C.c_void_p]
        L.cmx_decoder_create.restype = C.c_void_p
        L.cmx_decoder_create.argtypes = [C.c_void_p, C.c_size_t]
        L.cmx_decoder_destroy.argtypes = [C.c_void_p]
        L.cmx_decoder_decode.argtypes = [C.c_void_p, C.c_float]
        L.cmx_decoder_decode_bits.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.cmx_header_write.restype = C.c_size_t
        L.cmx_header_write.argtypes = [C.c_uint64, C.c_void_p, C.c_int, C.c_void_p]
        L.cmx_header_read.restype = C.c_size_t
        L.cmx_header_read.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmx_pipeline_begin.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        L.cmx_pipeline_hints.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmx_pipeline_finish.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmx_pipeline_stage_totals.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        L.cmx_pipeline_pretrain.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        L.cmx_pipeline_sync.argtypes = [C.c_void_p]
        L.cmx_pipeline_enable_fxcm.argtypes = [C.c_void_p, C.c_char_p]
        L.cmx_pipeline_fxcm_enabled.argtypes = [C.c_void_p]
        L.cmx_pipeline_finish_cols.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.cmx_pipeline_fxcm_total_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.cmx_pipeline_enable_paq8.argtypes = [C.c_void_p]
        L.cmx_pipeline_wait.argtypes = [C.c_void_p, C.c_uint64]
        L.cmx_pipeline_debug_mix_out.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        L.cmx_pipeline_paq8_total_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.cmx_pipeline_host_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.cmx_pipeline_paq8_role_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmx_pipeline_last_stage_ms.argtypes = [C.c_void_p, C.c_void_p]
        L.cmx_probe_libm.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t]
        L.cmx_pipeline_set_tolerance.argtypes = [C.c_void_p, C.c_int]
        L.cmx_pipeline_mixnet_mode.argtypes = [C.c_void_p]
        L.cmx_mixnet_set_tolerance.argtypes = [C.c_void_p, C.c_int]
        L.cmx_mixnet_mode.argtypes = [C.c_void_p]
        L.cmx_pipeline_late_start.argtypes = [C.c_void_p, C.c_int]
        L.cmx_pipeline_late_predict.restype = C.c_float
        L.cmx_pipeline_late_predict.argtypes = [C.c_void_p]
        L.cmx_pipeline_late_perceive.argtypes = [C.c_void_p, C.c_int]
        L.cmx_pipeline_late_stop.argtypes = [C.c_void_p]
        L.cmx_pipeline_late_host_ms.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.cmx_pipeline_late_debug_row.restype = C.c_void_p
        L.cmx_pipeline_late_debug_row.argtypes = [C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


def last_error():
    return lib().cmx_last_error().decode()


def device_count():
    return lib().cmx_device_count()


def probe_libm(which, x, device=0):
    """which: 0 expf, 1 tanhf, 2 logistic; x: float32 ndarray -> float32 ndarray (device-evaluated)."""
    x = np.ascontiguousarray(x, np.float32)
    y = np.empty_like(x)
    if lib().cmx_probe_libm(device, which, x.ctypes.data, y.ctypes.data, x.size):
        raise CmxError(last_error())
    return y


class MixNet:
    """Final mixing network stage (layers 0-2 + SSE) of one stream on one GPU."""

    def __init__(self, device=0):
        self.h = lib().cmx_mixnet_create(device)
        if not self.h:
            raise CmxError(last_error())
        self.device = device

    def set_tolerance(self, on=True):
        """Tolerance mode (NOT bit-exact: layer-0 dot products as f64 tree sums): explicit, before the handle's first bit."""
        if lib().cmx_mixnet_set_tolerance(self.h, int(bool(on))):
            raise CmxError(last_error())

    def close(self):
        if getattr(self, "h", None):
            lib().cmx_mixnet_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, probs, sel, bits, p_out=None, mix_out=None, stream=None):
        """Chunk mode on HBM-resident torch tensors.
        probs [T,2078] f32, sel [T,47] int32 (u32 keys), bits [T] u8 -> p_out [T] f32."""
        import torch
        T = int(bits.numel())
        assert probs.is_cuda and sel.is_cuda and bits.is_cuda
        assert probs.dtype == torch.float32 and probs.is_contiguous() and probs.numel() == T * N_INPUTS
        assert sel.dtype == torch.int32 and sel.is_contiguous() and sel.numel() == T * N_MIXERS
        assert bits.dtype == torch.uint8 and bits.is_contiguous()
        if p_out is None:
            p_out = torch.empty(T, dtype=torch.float32, device=probs.device)
        if stream is None:
            stream = torch.cuda.current_stream(probs.device).cuda_stream
        rc = lib().cmx_mixnet_run(self.h, probs.data_ptr(), sel.data_ptr(), bits.data_ptr(), T,
                                  p_out.data_ptr(), mix_out.data_ptr() if mix_out is not None else None,
                                  C.c_void_p(stream))
        if rc:
            raise CmxError(last_error())
        return p_out

    def predict(self, probs, sel):
        """Bit-synchronous Predict(): host arrays in, float out."""
        probs = np.ascontiguousarray(probs, np.float32)
        sel = np.ascontiguousarray(sel, np.uint32)
        assert probs.size == N_INPUTS and sel.size == N_MIXERS
        p = lib().cmx_mixnet_predict(self.h, probs.ctypes.data, sel.ctypes.data)
        if p < 0:
            raise CmxError(last_error())
        return np.float32(p)

    def perceive(self, bit):
        if lib().cmx_mixnet_perceive(self.h, int(bit)):
            raise CmxError(last_error())

    def sync(self):
        if lib().cmx_mixnet_sync(self.h):
            raise CmxError(last_error())

    def bits_done(self):
        v = C.c_uint64(0)
        if lib().cmx_mixnet_bits_done(self.h, C.byref(v)):
            raise CmxError(last_error())
        return v.value

    def profile(self, enable=True):
        out = (C.c_uint64 * 16)()
        if lib().cmx_mixnet_profile(self.h, int(enable), out):
            raise CmxError(last_error())
        return list(out)

    def last_kernel_ms(self):
        v = C.c_float(0)
        if lib().cmx_mixnet_last_kernel_ms(self.h, C.byref(v)):
            raise CmxError(last_error())
        return v.value

    def debug_set_steps(self, steps):
        """Test hook (state injection): the network as after `steps` bits -- Mixer::steps_ of all 47 mixers."""
        lib().cmx_mixnet_debug_set_steps.argtypes = [C.c_void_p, C.c_uint64]
        if lib().cmx_mixnet_debug_set_steps(self.h, int(steps)):
            raise CmxError(last_error())

    def spec_stats(self):
        """Speculative segment-parallel chain (cmx_mixnet_spec_kernel): segments run, resolved from a candidate, re-runs of segment 1..3."""
        out = (C.c_uint64 * 5)()
        if lib().cmx_mixnet_spec_stats(self.h, out):
            raise CmxError(last_error())
        v = list(out)
        return {"segments": v[0], "hits": v[1], "reruns": v[2:5], "hit_rate": (v[1] / v[0]) if v[0] else None}


class Lstm:
    """Byte-level LSTM byte mixer stage of one stream on one GPU (chunk mode)."""
    SKIP_RAND = 31

    def __init__(self, vocab, device=0, skip_rand=SKIP_RAND):
        vocab = np.ascontiguousarray(vocab, np.uint8)
        assert vocab.size == 256
        self.h = lib().cmx_lstm_create(vocab.ctypes.data, skip_rand, device)
        if not self.h:
            raise CmxError(last_error())
        self.V = lib().cmx_lstm_vocab_size(self.h)

    def set_tolerance(self, on=True):
        """TOLERANCE mode (NOT bit-exact): the BPTT round's weight-update contraction as v_mfma_f32_16x16x4_f32 tiles"""
        if lib().cmx_lstm_set_tolerance(self.h, 1 if on else 0):
            raise CmxError(last_error())

    def close(self):
        if getattr(self, "h", None):
            lib().cmx_lstm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, in_probs, data, want_bits=True, stream=None, layer0=None, out=None):
        """in_probs [N,256] f32 cuda, data [N] u8 cuda -> (out_probs [N,256], bit_p [N,8], bit_ex [N,8]).
        With layer0 = the [8N,2078] layer-0 matrix, the bit predictions are written in place into its
        column 2077 (and bit_p is None)."""
        import torch
        N = int(data.numel())
        assert in_probs.is_cuda and in_probs.dtype == torch.float32 and in_probs.is_contiguous()
        assert in_probs.numel() == N * 256 and data.dtype == torch.uint8 and data.is_contiguous()
        if out is None:
            out = torch.empty((N, 256), dtype=torch.float32, device=in_probs.device)
        bp, bp_ptr, stride = None, None, 1
        if layer0 is not None:
            assert layer0.dtype == torch.float32 and layer0.is_contiguous() and layer0.shape == (8 * N, N_INPUTS)
            bp_ptr, stride = layer0.data_ptr() + 4 * (N_INPUTS - 1), N_INPUTS
        elif want_bits:
            bp = torch.empty((N, 8), dtype=torch.float32, device=in_probs.device)
            bp_ptr = bp.data_ptr()
        bx = torch.empty((N, 8), dtype=torch.int32, device=in_probs.device) if bp_ptr else None
        if stream is None:
            stream = torch.cuda.current_stream(in_probs.device).cuda_stream
        rc = lib().cmx_lstm_run(self.h, in_probs.data_ptr(), data.data_ptr(), N, out.data_ptr(),
                                bp_ptr, stride, bx.data_ptr() if bx is not None else None, C.c_void_p(stream))
        if rc:
            raise CmxError(last_error())
        return out, bp, bx

    def gate_weights(self, layer, gate):
        n = lib().cmx_lstm_gate_rowlen(self.h, layer)
        out = np.empty((200, n), np.float32)
        if lib().cmx_lstm_get_gate_weights(self.h, layer, gate, out.ctypes.data):
            raise CmxError("cmx_lstm_get_gate_weights failed")
        return out


SMALL_COLS = [0, 1, 2] + list(range(2025, 2076))  # layer-0 columns the ctx/small-model stage writes


class CtxModels:
    """Context plumbing + the 54 small native models of one stream on one GPU (chunk mode)."""

    def __init__(self, vocab, device=0):
        vocab = np.ascontiguousarray(vocab, np.uint8)
        assert vocab.size == 256
        self.h = lib().cmx_ctxmodels_create(vocab.ctypes.data, device)
        if not self.h:
            raise CmxError(last_error())

    def close(self):
        if getattr(self, "h", None):
            lib().cmx_ctxmodels_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, data, probs=None, sel=None, stream=None):
        """data [N] u8 cuda -> writes columns SMALL_COLS of probs [8N,2078] f32 and sel [8N,47] i32 (u32 keys)."""
        import torch
        N = int(data.numel())
        assert data.is_cuda and data.dtype == torch.uint8 and data.is_contiguous()
        if probs is None:
            probs = torch.full((8 * N, N_INPUTS), 0.5, dtype=torch.float32, device=data.device)
        if sel is None:
            sel = torch.zeros((8 * N, N_MIXERS), dtype=torch.int32, device=data.device)
        assert probs.dtype == torch.float32 and probs.is_contiguous() and probs.shape[0] == 8 * N
        assert sel.dtype == torch.int32 and sel.is_contiguous() and sel.numel() == 8 * N * N_MIXERS
        if stream is None:
            stream = torch.cuda.current_stream(data.device).cuda_stream
        rc = lib().cmx_ctxmodels_run(self.h, data.data_ptr(), N, probs.data_ptr(), probs.shape[1], sel.data_ptr(),
                                     C.c_void_p(stream))
        if rc:
            raise CmxError(last_error())
        return probs, sel

    def pretrain(self, data, stream=None):
        """Predictor::Pretrain over data [N] u8 cuda (dictionary warm-up): state only, no outputs."""
        import torch
        assert data.is_cuda and data.dtype == torch.uint8 and data.is_contiguous()
        if stream is None:
            stream = torch.cuda.current_stream(data.device).cuda_stream
        if lib().cmx_ctxmodels_pretrain(self.h, data.data_ptr(), int(data.numel()), C.c_void_p(stream)):
            raise CmxError(last_error())

    def peek(self, byte, probs8, sel8, stream=None):
        """Bit-synchronous mode: the 8 rows the stage would write if the next byte were byte[0] (u8 cuda), leaving
        all state untouched; row j is exact when the top j bits of byte[0] are the coded ones."""
        import torch
        assert byte.is_cuda and byte.dtype == torch.uint8 and probs8.shape == (8, N_INPUTS) and sel8.numel() == 8 * N_MIXERS
        if stream is None:
            stream = torch.cuda.current_stream(byte.device).cuda_stream
        if lib().cmx_ctxmodels_peek(self.h, byte.data_ptr(), -1, probs8.data_ptr(), N_INPUTS, sel8.data_ptr(),
                                    C.c_void_p(stream)):
            raise CmxError(last_error())

    def debug_set_history(self, pos, tail):
        """Test hook (state injection): the stage as after `pos` bytes of a stream that ended in `tail` -- history ring position, Match counters."""
        tail = np.ascontiguousarray(np.frombuffer(bytes(tail), np.uint8))
        lib().cmx_ctxmodels_debug_set_history.argtypes = [C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        if lib().cmx_ctxmodels_debug_set_history(self.h, int(pos), tail.ctypes.data, len(tail)):
            raise CmxError(last_error())

    def slow_bytes(self):
        out = np.zeros(2, np.uint64)
        if lib().cmx_ctxmodels_debug_slow_bytes(self.h, out.ctypes.data):
            raise CmxError(last_error())
        return int(out[0]), int(out[1])

    def sync(self):
        if lib().cmx_ctxmodels_sync(self.h):
            raise CmxError(last_error())

    def manager(self):
        regs = np.empty(25, np.uint64)
        ctx = np.empty(54, np.uint64)
        bctx = np.empty(8, np.uint64)
        if lib().cmx_ctxmodels_get_manager(self.h, regs.ctypes.data, ctx.ctypes.data, bctx.ctypes.data):
            raise CmxError(last_error())
        return regs, ctx, bctx


def bytemodel_bits(dist0, dist_rest, data, layer0, col, device=0, stream=None):
    """ByteModel::Predict/Perceive along known bytes for any byte model (PPMd: col 2076): dist0 [256] is the
    distribution going into byte 0, dist_rest [N-1.., 256] the one after each byte; writes layer0[:, col]."""
    import torch
    N = int(data.numel())
    assert layer0.dtype == torch.float32 and layer0.is_contiguous() and layer0.shape == (8 * N, N_INPUTS)
    assert dist0.is_contiguous() and dist_rest.is_contiguous() and dist_rest.numel() >= (N - 1) * 256
    if stream is None:
        stream = torch.cuda.current_stream(data.device).cuda_stream
    rc = lib().cmx_bytemodel_bits_run(device, dist0.data_ptr(), dist_rest.data_ptr(), data.data_ptr(), N,
                                      layer0.data_ptr() + 4 * col, N_INPUTS, None, C.c_void_p(stream))
    if rc:
        raise CmxError(last_error())


class Pipeline:
    """One stream through every stage built so far (native orchestration in libcmixamd.so)."""

    def __init__(self, vocab, device=0, max_chunk_bytes=4096):
        vocab = np.ascontiguousarray(vocab, np.uint8)
        assert vocab.size == 256
        self.h = lib().cmx_pipeline_create(vocab.ctypes.data, device, max_chunk_bytes)
        if not self.h:
            raise CmxError(last_error())

    def submit(self, data, layer0, p_out):
        """data: bytes / u8 array (host); layer0 [8n,2078] f32 cuda with columns 3..2024 filled and synchronised;
        p_out [8n] f32 cuda. Asynchronous."""
        import torch
        data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
        n = len(data)
        assert layer0.is_cuda and layer0.dtype == torch.float32 and layer0.is_contiguous() and layer0.shape == (8 * n, N_INPUTS)
        assert p_out.is_cuda and p_out.dtype == torch.float32 and p_out.is_contiguous() and p_out.numel() == 8 * n
        if lib().cmx_pipeline_submit(self.h, data.ctypes.data, n, layer0.data_ptr(), p_out.data_ptr()):
            raise CmxError(last_error())

    def enable_fxcm(self, dictionary_path=None):
        """Run the fxcm family as a device stage (columns 3..433) instead of taking its columns from the caller; before the first chunk."""
        if lib().cmx_pipeline_enable_fxcm(self.h, dictionary_path.encode() if isinstance(dictionary_path, str) else dictionary_path):
            raise CmxError(last_error())

    def wait(self, index):
        if lib().cmx_pipeline_wait(self.h, index):
            raise CmxError(last_error())

    def debug_mix_out(self, d_mix):
        """Diagnosis: d_mix = a CUDA float32 tensor [bits][47] that receives all 47 mixer outputs of every bit from now on (None: off)."""
        if lib().cmx_pipeline_debug_mix_out(self.h, d_mix.data_ptr() if d_mix is not None else None, d_mix.shape[0] if d_mix is not None else 0):
            raise CmxError(last_error())

    def enable_paq8(self):
        """Run the paq8 family as a device stage (columns 434..2024); before the first chunk."""
        if lib().cmx_pipeline_enable_paq8(self.h):
            raise CmxError(last_error())

    def paq8_total_ms(self):
        ms = C.c_double(0)
        lib().cmx_pipeline_paq8_total_ms(self.h, C.byref(ms))
        return ms.value

    def paq8_role_ms(self):
        """(dict of the paq8 role kernels' summed HIP-event ms, chunks collected) since the last totals reset."""
        v, c = (C.c_double * 7)(), C.c_uint64(0)
        lib().cmx_pipeline_paq8_role_ms(self.h, v, C.byref(c))
        return dict(zip(("family", "mixer", "cm2_order_n", "cm2_text", "cm2_exe", "lanes", "dmc"), [float(x) for x in v])), int(c.value)

    def host_ms(self):
        """Calling-thread wall time inside begin / finish since the last totals reset: dict of ms."""
        v = (C.c_double * 6)()
        lib().cmx_pipeline_host_ms(self.h, v)
        return dict(zip(("slot_wait", "ppmd", "ctx_lstm_enqueue", "fxcm_parser_enqueue", "paq8_front_enqueue", "mixnet_enqueue"), [float(x) for x in v]))

    def finish_cols(self, cols, first_col, p_out):
        """finish() with host rows covering layer-0 columns first_col .. first_col + cols.shape[1] - 1 only."""
        cols = np.ascontiguousarray(cols, np.float32)
        if lib().cmx_pipeline_finish_cols(self.h, cols.ctypes.data, first_col, cols.shape[1], p_out.data_ptr()):
            raise CmxError(last_error())

    def fxcm_total_ms(self):
        ms = C.c_double(0)
        lib().cmx_pipeline_fxcm_total_ms(self.h, C.byref(ms))
        return ms.value

    def pretrain(self, data):
        data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
        if lib().cmx_pipeline_pretrain(self.h, data.ctypes.data, len(data)):
            raise CmxError(last_error())

    def sync(self):
        if lib().cmx_pipeline_sync(self.h):
            raise CmxError(last_error())

    def last_stage_ms(self):
        ms = (C.c_float * 3)()
        lib().cmx_pipeline_last_stage_ms(self.h, ms)
        return {"ctxmodels": ms[0], "lstm": ms[1], "mixnet": ms[2]}

    def begin(self, data, layer0):
        """First step of a chunk (PPMd, contexts, LSTM); up to PIPELINE_SLOTS chunks may be begun and not finished."""
        data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
        self._n_begun = getattr(self, "_n_begun", [])
        self._n_begun.append(len(data))
        if lib().cmx_pipeline_begin(self.h, data.ctypes.data, len(data), layer0.data_ptr()):
            raise CmxError(last_error())

    def hints(self, nbytes):
        """The LSTM byte mixer's per-bit ByteModel::Predict value and `ex` for the oldest begun chunk (8n+1 entries)."""
        p = np.empty(8 * nbytes + 1, np.float32)
        ex = np.empty(8 * nbytes + 1, np.int32)
        if lib().cmx_pipeline_hints(self.h, p.ctypes.data, ex.ctypes.data):
            raise CmxError(last_error())
        return p, ex

    def finish(self, cols, p_out):
        """Last step of the oldest begun chunk: host rows cols [8n, 2022] (or None) + the mixing network."""
        if cols is not None:
            cols = np.ascontiguousarray(cols, np.float32)
            assert cols.shape[1] == 2022
        if lib().cmx_pipeline_finish(self.h, cols.ctypes.data if cols is not None else None, p_out.data_ptr()):
            raise CmxError(last_error())

    def stage_totals(self, reset=False):
        """Mean HIP-event ms per chunk of each stage over the chunks finished since the last reset (call after sync)."""
        ms = (C.c_double * 3)()
        n = C.c_uint64(0)
        lib().cmx_pipeline_stage_totals(self.h, ms, C.byref(n), int(reset))
        k = max(n.value, 1)
        return {"ctxmodels": ms[0] / k, "lstm": ms[1] / k, "mixnet": ms[2] / k, "chunks": n.value}

    def set_tolerance(self, on=True):
        """The mixing network's tolerance mode (NOT bit-exact): an explicit switch, before the first chunk."""
        if lib().cmx_pipeline_set_tolerance(self.h, int(bool(on))):
            raise CmxError(last_error())

    def mixnet_mode(self):
        """0 strict (bit-exact, the default), 1 tolerance -- as the library reports it"""
        return lib().cmx_pipeline_mixnet_mode(self.h)

    # ---- the decoder's form (late-bit protocol, include/cmix_amd.h section 4): fxcm and paq8 must be enabled ----
    def late_start(self, last_bit=0):
        if lib().cmx_pipeline_late_start(self.h, int(last_bit)):
            raise CmxError(last_error())

    def late_predict(self):
        p = lib().cmx_pipeline_late_predict(self.h)
        if p < 0:
            raise CmxError(last_error())
        return p

    def late_perceive(self, bit):
        if lib().cmx_pipeline_late_perceive(self.h, int(bit)):
            raise CmxError(last_error())

    def late_replay(self, bits):
        """predict / perceive over known bits in one native call -> p before each bit"""
        bits = np.ascontiguousarray(bits, np.uint8)
        p = np.empty(len(bits), np.float32)
        lib().cmx_pipeline_late_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        if lib().cmx_pipeline_late_replay(self.h, bits.ctypes.data, len(bits), p.ctypes.data):
            raise CmxError(last_error())
        return p

    def late_stop(self):
        if lib().cmx_pipeline_late_stop(self.h):
            raise CmxError(last_error())

    def late_row(self):
        """(layer-0 row [2078] f32, selectors [47] u32) of the bit predicted last, copied out of the host-coherent chunk buffers."""
        sel = C.c_void_p(0)
        r = lib().cmx_pipeline_late_debug_row(self.h, C.byref(sel))
        if not r:
            raise CmxError("no pending late predict")
        row = np.ctypeslib.as_array(C.cast(r, C.POINTER(C.c_float)), (N_INPUTS,)).copy()
        s = np.ctypeslib.as_array(C.cast(sel.value, C.POINTER(C.c_uint32)), (47,)).copy()
        return row, s

    def late_mix(self):
        """the 47 Mixer::Mix values of the bit BEFORE the one predicted last (CMX_LATE_DEBUG=1 when the handle was built), or None"""
        out = np.zeros(47, np.float32)
        lib().cmx_pipeline_late_debug_mix.argtypes = [C.c_void_p, C.c_void_p]
        return None if lib().cmx_pipeline_late_debug_mix(self.h, out.ctypes.data) else out

    def late_host_ms(self):
        v, n = (C.c_double * 6)(), C.c_uint64(0)
        lib().cmx_pipeline_late_host_ms(self.h, v, C.byref(n))
        return dict(zip(("wait_p", "ppmd", "paq8_front", "fxcm_parser", "lstm_launch", "chunk_launch"), [float(x) for x in v])), int(n.value)

    def close(self):
        if getattr(self, "h", None):
            lib().cmx_pipeline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Ppmd:
    """HOST stage: PPMd order-25 byte model (runs on a host core ahead of the device pipeline)."""

    def __init__(self, vocab, order=25, memory_mb=14000):
        vocab = np.ascontiguousarray(vocab, np.uint8)
        assert vocab.size == 256
        self.h = lib().cmx_ppmd_create_ex(vocab.ctypes.data, order, memory_mb)
        if not self.h:
            raise CmxError(last_error())

    def run(self, data):
        """data: bytes / u8 array -> [N,256] f32: the byte distribution after each byte."""
        data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
        out = np.empty((len(data), 256), np.float32)
        if lib().cmx_ppmd_run(self.h, data.ctypes.data, len(data), out.ctypes.data):
            raise CmxError(last_error())
        return out

    def close(self):
        if getattr(self, "h", None):
            lib().cmx_ppmd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def glibc_rand(seed, n):
    out = np.empty(n, np.int32)
    lib().cmx_glibc_rand_selftest(seed, n, out.ctypes.data)
    return out


class Encoder:
    """Encoder::Encode / Flush (src/coder/encoder.cpp) over probabilities a pipeline chunk produced (host arrays)."""

    def __init__(self):
        self.h = lib().cmx_encoder_create()

    def encode_bits(self, p, bits):
        p = np.ascontiguousarray(p, np.float32)
        bits = np.ascontiguousarray(bits, np.uint8)
        if p.shape != bits.shape:
            raise CmxError("Encoder.encode_bits: p and bits differ in length")
        if lib().cmx_encoder_encode_bits(self.h, p.ctypes.data, bits.ctypes.data, len(p)):
            raise CmxError(last_error())

    def encode_bytes(self, p, data):
        p = np.ascontiguousarray(p, np.float32)
        data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
        if len(p) != 8 * len(data):
            raise CmxError("Encoder.encode_bytes: need 8 probabilities per byte")
        if lib().cmx_encoder_encode_bytes(self.h, p.ctypes.data, data.ctypes.data, len(data)):
            raise CmxError(last_error())

    def flush(self):
        if lib().cmx_encoder_flush(self.h):
            raise CmxError(last_error())

    def data(self):
        n = lib().cmx_encoder_size(self.h)
        return C.string_at(lib().cmx_encoder_data(self.h), n) if n else b""

    def close(self):
        if self.h:
            lib().cmx_encoder_destroy(self.h)
            self.h = None

    __del__ = close


class Decoder:
    """Decoder::Decode (src/coder/decoder.cpp)."""

    def __init__(self, code):
        self._code = np.ascontiguousarray(np.frombuffer(bytes(code), np.uint8))  # kept alive: the decoder reads it
        self.h = lib().cmx_decoder_create(self._code.ctypes.data if len(self._code) else None, len(self._code))
        if not self.h:
            raise CmxError(last_error())

    def decode(self, p):
        b = lib().cmx_decoder_decode(self.h, float(p))
        if b < 0:
            raise CmxError(last_error())
        return b

    def decode_bits(self, p):
        p = np.ascontiguousarray(p, np.float32)
        bits = np.empty(len(p), np.uint8)
        if lib().cmx_decoder_decode_bits(self.h, p.ctypes.data, len(p), bits.ctypes.data):
            raise CmxError(last_error())
        return bits

    def close(self):
        if self.h:
            lib().cmx_decoder_destroy(self.h)
            self.h = None

    __del__ = close


def header_write(length, vocab, dictionary_used=False):
    """WriteHeader (src/runner.cpp:34-52)."""
    out = np.zeros(37, np.uint8)
    vocab = np.ascontiguousarray(vocab, np.uint8)
    n = lib().cmx_header_write(int(length), vocab.ctypes.data, int(bool(dictionary_used)), out.ctypes.data)
    if n == 0:
        raise CmxError(last_error())
    return out[:n].tobytes()


def header_read(buf):
    """ReadHeader (src/runner.cpp:62-84) -> (length, dictionary_used, vocab[256], header bytes consumed)."""
    b = np.ascontiguousarray(np.frombuffer(bytes(buf[:37]), np.uint8))
    length, dic, vocab = C.c_uint64(0), C.c_int(0), np.zeros(256, np.uint8)
    n = lib().cmx_header_read(b.ctypes.data if len(b) else None, len(b), C.byref(length), C.byref(dic), vocab.ctypes.data)
    if n == 0:
        raise CmxError(last_error())
    return length.value, bool(dic.value), vocab, n


class Predictor:
    """`class Predictor` (src/predictor.h:17-22) over cmx_create .. cmx_destroy: Predict / Perceive / Pretrain in the
    reference's strict per-bit protocol. Two modes, decided by the first call that needs device state: after
    `stage_input` (a compressor knows its bytes) the handle runs the chunk pipeline with every model family on the device and
    Predict / Perceive pop and check; without it the per-bit stages are built (what a Decoder needs) and the fxcm / paq8
    columns of each bit come from the caller (`set_model_outputs`)."""

    def __init__(self, vocab, device=0, dict_path=None):
        vocab = np.ascontiguousarray(vocab, np.uint8)
        assert vocab.size == 256
        self.h = lib().cmx_create(vocab.ctypes.data, dict_path.encode() if dict_path else None, device)
        if not self.h:
            raise CmxError(last_error())

    def set_model_outputs(self, cols):
        cols = np.ascontiguousarray(cols, np.float32)
        if cols.shape != (2022,):
            raise CmxError("Predictor.set_model_outputs: need the 2022 layer-0 columns 3..2024")
        if lib().cmx_set_model_outputs(self.h, cols.ctypes.data):
            raise CmxError(last_error())

    def Predict(self):
        p = lib().cmx_predict(self.h)
        if p < 0:
            raise CmxError(last_error())
        return np.float32(p)

    def Perceive(self, bit):
        if lib().cmx_perceive(self.h, int(bit)):
            raise CmxError(last_error())

    def Pretrain(self, bit):
        if lib().cmx_pretrain(self.h, int(bit)):
            raise CmxError(last_error())

    def stage_input(self, data, end=True):
        """Look-ahead: the bytes about to be coded (appended to what is already staged); end=True marks the end of the input."""
        b = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
        if len(b) and lib().cmx_stage_input(self.h, b.ctypes.data, len(b)):
            raise CmxError(last_error())
        if end and lib().cmx_stage_input(self.h, None, 0):
            raise CmxError(last_error())

    def mode(self):
        n = C.c_int(0)
        return lib().cmx_mode(self.h, C.byref(n)), n.value

    def decode_stream(self, code, nbytes):
        """Decoder::Decode over a whole stream inside the library (cmx_decode_stream): the arithmetic code behind the container header -> the nbytes
        bytes the predictor saw. On a fresh handle."""
        code = np.ascontiguousarray(np.frombuffer(bytes(code), np.uint8))
        out = np.zeros(nbytes, np.uint8)
        lib().cmx_decode_stream.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
        if lib().cmx_decode_stream(self.h, code.ctypes.data if len(code) else None, len(code), out.ctypes.data if nbytes else None, nbytes):
            raise CmxError(last_error())
        return out.tobytes()

    def lstm_hint(self):
        a, b = C.c_int(0), C.c_int(0)
        if lib().cmx_get_lstm_hint(self.h, C.byref(a), C.byref(b)):
            raise CmxError(last_error())
        return a.value, b.value

    def close(self):
        if self.h:
            lib().cmx_destroy(self.h)
            self.h = None

    __del__ = close



class Fxcm:
    """The fxcm stage of one stream on one GPU (chunk mode): layer-0 columns 3..433. The text parser half runs on the
    calling thread inside run(); the learned tables live on the device (include/cmix_amd.h section 2f)."""

    def __init__(self, dictionary_path=None, device=0):
        self.h = lib().cmx_fxcm_create(dictionary_path.encode() if isinstance(dictionary_path, str) else dictionary_path, device)
        if not self.h:
            raise CmxError(last_error())

    def close(self):
        if getattr(self, "h", None):
            lib().cmx_fxcm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def run(self, data_host, lstmpr, lstmex, probs=None, stream=None):
        """data_host [N] u8 numpy; lstmpr [8N] i16 cuda, lstmex [8N] u8 cuda -> writes columns 3..433 of probs [8N, >=434] f32."""
        import torch
        data_host = np.ascontiguousarray(data_host, np.uint8)
        N = int(data_host.size)
        dev = lstmpr.device
        assert lstmpr.is_cuda and lstmpr.dtype == torch.int16 and lstmpr.is_contiguous() and lstmpr.numel() == 8 * N
        assert lstmex.is_cuda and lstmex.dtype == torch.uint8 and lstmex.is_contiguous() and lstmex.numel() == 8 * N
        d_bytes = torch.from_numpy(data_host.copy()).to(dev)
        if probs is None:
            probs = torch.full((8 * N, N_INPUTS), 0.5, dtype=torch.float32, device=dev)
        assert probs.dtype == torch.float32 and probs.is_contiguous() and probs.shape[0] == 8 * N and probs.shape[1] >= 434
        if stream is None:
            stream = torch.cuda.current_stream(dev).cuda_stream
        rc = lib().cmx_fxcm_run(self.h, data_host.ctypes.data, d_bytes.data_ptr(), N, lstmpr.data_ptr(), lstmex.data_ptr(), probs.data_ptr(),
                                probs.shape[1], C.c_void_p(stream))
        if rc:
            raise CmxError(last_error())
        self._keep = d_bytes   # the kernel reads it asynchronously
        return probs

    def sync(self):
        if lib().cmx_fxcm_sync(self.h):
            raise CmxError(last_error())


class P8Stage:
    """The paq8 stage of one stream on one GPU (include/cmix_amd.h section 2f): bytes in (host), per bit the 1591 values
    PAQ8::Predict() returns out (device)."""

    def __init__(self, device=0):
        self.device = device
        self.h = lib().cmx_p8stage_create(device)
        if not self.h:
            raise CmxError(last_error())

    def run(self, data, out=None, col0=0, stream=None):
        """data: bytes-like (host). out: cuda f32 [8 n, >= col0 + 1591] (default: a fresh [8 n, 1591]); the values go to
        columns col0 .. col0 + 1590 of row t."""
        import torch
        data = np.ascontiguousarray(np.frombuffer(bytes(data), np.uint8))
        n = len(data)
        if out is None:
            out = torch.empty((8 * n, 1591), dtype=torch.float32, device="cuda:%d" % self.device)
        assert out.is_cuda and out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] == 8 * n and out.shape[1] >= col0 + 1591
        if stream is None:
            stream = torch.cuda.current_stream(out.device).cuda_stream
        if lib().cmx_p8stage_run(self.h, data.ctypes.data, n, out.data_ptr() + 4 * col0, out.shape[1], C.c_void_p(stream)):
            raise CmxError(last_error())
        return out

    def sync(self):
        if lib().cmx_p8stage_sync(self.h):
            raise CmxError(last_error())

    def debug_set_pos(self, pos):
        """Test hook (before the first byte): the front end's byte position (the index into paq8's 2^30-byte history ring)."""
        lib().cmx_p8stage_debug_set_pos.argtypes = [C.c_void_p, C.c_int]
        if lib().cmx_p8stage_debug_set_pos(self.h, int(pos)):
            raise CmxError(last_error())

    def set_generator_counter(self, counter):
        """Test hook (before the first byte): the counter of the ContextMap family's shared generator, a multiple of 64."""
        if lib().cmx_p8stage_set_generator_counter(self.h, int(counter) & 0xFFFFFFFF):
            raise CmxError(last_error())

    def close(self):
        if getattr(self, "h", None):
            lib().cmx_p8stage_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass




