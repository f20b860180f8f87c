/* cmix_amd.h -- C ABI of libcmixamd.so, the MI355X (gfx950) per-bit prediction
 * engine for cmix v21.
 *
 * Drop-in boundary: the reference's `class Predictor`
 *   Predictor(const std::vector<bool>& vocab); float Predict();
 *   void Perceive(int bit); void Pretrain(int bit);
 * (reference src/predictor.h:17-22), whose only callers are Encoder::Encode
 * (src/coder/encoder.cpp:14-30), Decoder::Decode (src/coder/decoder.cpp:20-39),
 * preprocessor::Pretrain (src/preprocess/preprocessor.cpp:37-69) and the two
 * construction sites src/runner.cpp:205,246.  INTEGRATION.md shows the C++ shim
 * a cmix maintainer adds to bind these entry points.
 *
 * Two granularities are exported:
 *   1. the whole-predictor surface (cmx_create .. cmx_destroy), one handle per
 *      input stream, any number of handles per process (one per GPU);
 *   2. stage-level entry points (cmx_mixnet_*, cmx_lstm_*, ...) -- the device
 *      pipeline stages the predictor is assembled from.  A stage consumes and
 *      produces plain arrays, either one bit at a time (bit-synchronous: what a
 *      decoder needs, and what a hybrid build uses while some model families
 *      still run on the host) or a whole chunk of already-known bits at once
 *      (compression look-ahead: every input bit is known in advance, SURVEY.md
 *      7.1) with all operands resident in HBM.
 *
 * All functions return 0 / non-NULL on success; on failure they return
 * nonzero / NULL and cmx_last_error() describes why.  Nothing here falls back
 * to a CPU implementation: without a gfx950 device every create call fails.
 *
 * Plain C: no C++ or torch types cross this boundary.
 */
#ifndef CMIX_AMD_H
#define CMIX_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CMX_N_INPUTS 2078 /* layer-0 width, SURVEY.md Appendix A.1 */
#define CMX_N_MIXERS 47   /* 26 + 20 + 1, predictor.cpp:193-356 */
#define CMX_N_MIX0 26
#define CMX_N_MIX1 20
#define CMX_AUX_MIXER 12 /* layer-0 mixer keyed by auxiliary_context_ */

const char* cmx_last_error(void);
/* Library / device identification: fills name (<=256 bytes) and returns the
 * number of visible HIP devices (0 if none; never fails). */
int cmx_device_count(void);
const char* cmx_version(void);
/* Memory helpers, so that a host program written against this header needs no HIP headers: device memory,
 * page-locked host memory, and a synchronous device-to-host copy ordered after all prior work of the device. */
void* cmx_device_alloc(int device, size_t bytes);
void cmx_device_free(int device, void* p);
void* cmx_host_alloc(size_t bytes);
void cmx_host_free(void* p);
int cmx_copy_to_host(int device, void* dst, const void* d_src, size_t bytes);

/* ------------------------------------------------------------------------
 * 1. Whole-predictor surface (replaces class Predictor, predictor.h:17-22)
 * ------------------------------------------------------------------------ */
typedef struct cmx_engine cmx_t;
/* vocab[i] != 0 iff byte value i occurs in the (preprocessed) input
 * (runner.cpp:196-202). dict_path replaces the global `dictionary_path`
 * (runner.cpp:17) read by fxcm; may be NULL. */
cmx_t* cmx_create(const uint8_t vocab[256], const char* dict_path, int device);   /* checks the device; stages are built on first use */
float cmx_predict(cmx_t*);             /* Predictor::Predict,  predictor.cpp:361; < 0 on error */
int cmx_perceive(cmx_t*, int bit);     /* Predictor::Perceive, predictor.cpp:421 */
int cmx_pretrain(cmx_t*, int bit);     /* Predictor::Pretrain, predictor.cpp:471 */
/* Decoder::Decode (coder/decoder.cpp:20-39) + the Decompress loop (runner.cpp:214-246) over a whole stream inside the library: `code` = the arithmetic code behind
 * the container header, nbytes = the stream's length from that header, out[nbytes] = the bytes the predictor saw (the preprocessed stream; the caller runs the
 * reference's preprocessor::Decode over them). Call on a fresh handle (after cmx_pretrain, if there is a dictionary). Equivalent to nbytes x 8 rounds of
 * cmx_predict / cmx_decoder_decode / cmx_perceive; the reference's own decoder.cpp over the ABI remains the parity path (integration/predictor_dropin.h). */
int cmx_decode_stream(cmx_t*, const uint8_t* code, size_t code_len, uint8_t* out, size_t nbytes);
/* TEST HOOK (the third mode of a handle; rounds 1-3 decoded this way): fxcm and paq8 have been device stages since round 2 -- a compressor (cmx_stage_input)
 * and a decoder (cmx_predict without staged input) take no column from the caller. A handle that is given columns BEFORE its first cmx_predict() runs the
 * per-bit stages instead and takes the two families' outputs of every bit from the caller -- layer-0 columns 3..2024 in the reference's order (431 fxcm values,
 * then 1591 paq8 values; predictor.cpp:363-369) -- which is how tests/test_gpu_predictor.py isolates the other stages. cmx_predict() fails loudly when the
 * columns of a bit are missing: nothing is computed on the CPU in their place. */
int cmx_set_model_outputs(cmx_t*, const float cols_3_to_2024[2022]);
/* The hidden globals `lstmpr`, `lstmex` (predictor.cpp:359,462-465) as they stand after the last cmx_perceive():
 * what the caller's fxcm reads in its own Perceive. cmx_perceive() only enqueues the device work; this call (or the
 * next cmx_predict()) waits for it, so host work placed between the two overlaps the device. */
int cmx_get_lstm_hint(cmx_t*, int* lstmpr, int* lstmex);
/* Introspection used by the parity tests: the 2078 layer-0 inputs of the last cmx_predict() (valid until the next). */
const float* cmx_debug_last_row(cmx_t*);
/* Look-ahead for compression: the next n bytes the caller is going to code, in order; may be called
 * repeatedly (bytes are appended); n == 0 marks the end of the input (the ragged last chunk is then submitted at once).
 * The first call must precede the first cmx_predict() and puts the handle in LOOK-AHEAD MODE: it builds the chunk
 * pipeline (cmx_pipeline_*: every model family a device stage, fxcm and paq8 included -- no cmx_set_model_outputs),
 * trains it on the bytes cmx_pretrain() has collected, and keeps CMX_PIPELINE_SLOTS chunks of 4 KB in flight.
 * cmx_predict() then pops the next probability of the chunk that has left the mixing network (same float the per-bit
 * surface would return) and cmx_perceive(bit) checks the bit against the staged one -- a mismatch is an error, the
 * device has already learnt the staged bit. With it the reference's unmodified Encoder (encoder.cpp:14-30) and
 * Compress loop (runner.cpp:101-119) run at the pipeline's speed: integration/predictor_dropin.h.
 * Without it the handle builds the per-bit stages on first use (what a Decoder needs). */
int cmx_stage_input(cmx_t*, const uint8_t* bytes, size_t n);
/* 0 = undecided (nothing built yet), 1 = per-bit stages, 2 = look-ahead pipeline; *chunks_in_flight (may be NULL). */
int cmx_mode(cmx_t*, int* chunks_in_flight);
void cmx_destroy(cmx_t*);

/* ------------------------------------------------------------------------
 * 2a. Stage: final mixing network = MixerInput stretch -> 26+20+1 gated
 *     logistic mixers -> squash -> SSE  (predictor.cpp:388-418,432-437;
 *     src/mixer/{mixer,mixer-input,sigmoid,sse}.cpp)
 * ------------------------------------------------------------------------ */
typedef struct cmx_mixnet cmx_mixnet_t;
cmx_mixnet_t* cmx_mixnet_create(int device);
void cmx_mixnet_destroy(cmx_mixnet_t*);

/* The stream the handle's host-to-device copies (the decay schedule of a chunk) go on; NULL-less default: a stream the handle
 * creates. A copy must never sit behind a long kernel in stream order: on this hardware it then holds up later copies of the
 * whole process. cmx_pipeline_* hands ONE upload stream to all its stages. Same for cmx_fxcm_* (records) and cmx_p8stage_*. */
/* Tolerance mode of the 27-workgroup kernel (NOT bit-exact: layer-0 dot products as f64 tree sums rounded once). An explicit
 * switch of the handle, before its first bit -- never an environment variable: a stream coded with it is not the reference's
 * and cannot be decoded by this library's (strict) decoder. cmx_mixnet_mode: 0 strict (default), 1 tolerance. */
int cmx_mixnet_set_tolerance(cmx_mixnet_t*, int on);
const int* cmx_mixnet_error_flag(cmx_mixnet_t*);   /* DEVICE address of the sticky "an in-launch wait ran out" word (see cmx_lstm_fail_flag) */
int cmx_mixnet_mode(cmx_mixnet_t*);
int cmx_mixnet_set_upload_stream(cmx_mixnet_t*, void* stream);
/* Chunk mode. All pointers are DEVICE pointers (HBM-resident operands):
 *   d_probs [nbits][2078] f32  raw model outputs (Model::Predict values)
 *   d_sel   [nbits][47]   u32  each mixer's selector key (its 64-bit context
 *                              truncated to 32 bits exactly as the reference's
 *                              unordered_map<unsigned int,...> key does,
 *                              mixer.h:33); entry 12 is ignored (derived on
 *                              device from the inputs, predictor.cpp:388-393)
 *   d_bits  [nbits]       u8   the coded bits
 *   d_p_out [nbits]       f32  OUT: value Predictor::Predict() returns
 *   d_mix_out [nbits][47] f32  OUT (may be NULL): every Mixer::p_
 * Processes the bits strictly in order inside one persistent kernel launch on
 * `stream` (a hipStream_t passed as void*, NULL = default stream) and returns
 * without synchronising. Every chunk of a handle must be enqueued on the SAME stream (the chunks train one state in order
 * and share one staging buffer); a call with a different stream is refused. */
int cmx_mixnet_run(cmx_mixnet_t*, const float* d_probs, const uint32_t* d_sel,
                   const uint8_t* d_bits, size_t nbits, float* d_p_out, float* d_mix_out,
                   void* stream);

/* Bit-synchronous mode. HOST pointers; synchronous. predict() may be followed
 * only by perceive() (same protocol as the reference). */
float cmx_mixnet_predict(cmx_mixnet_t*, const float* probs2078, const uint32_t* sel47);
int cmx_mixnet_perceive(cmx_mixnet_t*, int bit);

/* Bit-synchronous mode with DEVICE operands, asynchronous on `stream`: the same kernel and protocol without the
 * host round trips. The operands must stay untouched until the matching perceive has run; *d_p receives the
 * probability. */
int cmx_mixnet_predict_async(cmx_mixnet_t*, const float* d_probs2078, const uint32_t* d_sel47, float* d_p, void* stream);
int cmx_mixnet_perceive_async(cmx_mixnet_t*, int bit, void* stream);

/* Waits for all work of this handle and reports device-side failures (a bounded
 * in-kernel wait that timed out). */
int cmx_mixnet_sync(cmx_mixnet_t*);

/* Introspection used by the parity tests. */
int cmx_mixnet_bits_done(const cmx_mixnet_t*, uint64_t* out);
/* Phase timers: enable != 0 turns on in-kernel shader-clock accumulation for the
 * following chunks; out16 (may be NULL) receives the 16 accumulators gathered
 * since the previous call, which are then cleared. Synchronises the device. */
int cmx_mixnet_profile(cmx_mixnet_t*, int enable, uint64_t* out16);
/* Elapsed device time (ms, HIP events on the launch stream) of the last
 * cmx_mixnet_run kernel; synchronises with it. */
int cmx_mixnet_last_kernel_ms(cmx_mixnet_t*, float* ms);
/* Chunk mode runs cmx_mixnet_spec_kernel: 26 helper workgroups cut each layer-0 mixer's ordered 2078-term sum into four segments
 * that run at once, segments 1..3 speculatively from 64 candidate start values (exact: the lane whose candidate equals the true
 * start holds the reference's result, otherwise the segment is re-run). Statistics since creation: out[0] speculative segments,
 * [1] resolved from a candidate, [2..4] re-runs of segment 1 / 2 / 3. Synchronises the device. */
int cmx_mixnet_spec_stats(cmx_mixnet_t*, uint64_t out[5]);
/* Test hook (state injection): the network as after `steps` bits of a stream -- Mixer::steps_ (mixer.cpp:58,61) of all 47 mixers. The wrap / threshold
 * fixtures of tests/golden/make_wrap_traces.py place the reference's counters the same way (oracle/ref_harness.cpp). Between chunks only. */
int cmx_mixnet_debug_set_steps(cmx_mixnet_t*, uint64_t steps);

/* ------------------------------------------------------------------------
 * 2b. Stage: byte-level LSTM byte mixer = ByteMixer + Lstm + LstmLayer + its ByteModel bit
 *     interface (src/mixer/{byte-mixer,lstm,lstm-layer}.cpp, src/models/byte-model.cpp;
 *     wired at predictor.cpp:189-191,378-387,450-467)
 * ------------------------------------------------------------------------ */
typedef struct cmx_lstm cmx_lstm_t;
/* vocab as for cmx_create. skip_rand = number of rand() values the reference consumes after
 * srand(0xDEADBEEF) before it constructs the LSTM (31 in Predictor::Predictor: one per Indirect
 * model, indirect.cpp:10); the weights are then drawn exactly as lstm-layer.cpp:52-59 does. */
cmx_lstm_t* cmx_lstm_create(const uint8_t vocab[256], int skip_rand, int device);
void cmx_lstm_destroy(cmx_lstm_t*);
int cmx_lstm_vocab_size(const cmx_lstm_t*);
/* Chunk mode over nbytes already-known bytes. DEVICE pointers:
 *   d_in_probs  [nbytes][256] f32  the byte model's (PPMd) distribution after each byte
 *                                  (ByteModel::BytePredict, predictor.cpp:450-457)
 *   d_bytes     [nbytes]      u8   the bytes coded
 *   d_out_probs [nbytes][256] f32  OUT: the LSTM's distribution after each byte (ByteMixer::probs_)
 *   d_bit_p     [nbytes][8]   f32  OUT (may be NULL): ByteModel::Predict value for each bit of byte n,
 *                                  i.e. layer-0 input 2077 (for n = 0: from the state before this call);
 *                                  bit k of byte n goes to d_bit_p[(8n+k)*bit_p_stride] -- stride 1 for a
 *                                  dense array, 2078 to write column 2077 of the layer-0 matrix in place
 *   d_bit_ex    [nbytes][8]   i32  OUT (may be NULL): ByteModel::ex per bit (-> lstmex, predictor.cpp:465)
 * Asynchronous on `stream`. Truncated BPTT + Adam run every 100 bytes as in lstm.cpp:93-110. */
int cmx_lstm_run(cmx_lstm_t*, const float* d_in_probs, const uint8_t* d_bytes, size_t nbytes,
                 float* d_out_probs, float* d_bit_p, size_t bit_p_stride, int* d_bit_ex, void* stream);
/* ByteModel::Predict/Perceive for any byte model (PPMd, Bracket, LSTM): per-bit predictions along
 * the known bytes. dist for byte 0 = d_dist0[256]; for byte n >= 1 = d_dist_rest[(n-1)*256 ...]. */
int cmx_bytemodel_bits_run(int device, const float* d_dist0, const float* d_dist_rest,
                           const uint8_t* d_bytes, size_t nbytes, float* d_bit_p, size_t bit_p_stride,
                           int* d_bit_ex, void* stream);
/* Bit-synchronous mode: ByteModel::Predict for bit k (0..7) of the byte whose top k bits (in *d_byte) are the coded
 * ones: d_bit_p[k * stride] and *d_p_copy (may be NULL) receive the value, d_bit_ex[k] (may be NULL) `ex`. */
int cmx_bytemodel_bit_run(int device, const float* d_dist, const uint8_t* d_byte, int k, float* d_bit_p,
                          size_t bit_p_stride, int* d_bit_ex, float* d_p_copy, void* stream);
/* Test hooks. */
int cmx_lstm_get_gate_weights(cmx_lstm_t*, int layer, int gate, float* out_host);
int cmx_lstm_gate_rowlen(const cmx_lstm_t*, int layer);
/* 1 if a bounded in-launch wait of the multi-workgroup kernels (lstm_block.hip) ran out -- the stream's LSTM output
 * is void from that point --, 0 otherwise. Synchronises the device. */
int cmx_lstm_failed(cmx_lstm_t*);
/* TOLERANCE mode (NOT bit-exact, measurement only): the BPTT round's weight-update contraction (200 x rowlen x 100 per gate,
 * reference src/mixer/lstm-layer.cpp:182-186) as v_mfma_f32_16x16x4_f32 tiles instead of the ordered, separately rounded chain */
int cmx_lstm_set_tolerance(cmx_lstm_t*, int on);
/* DEVICE address of the sticky flag cmx_lstm_failed reads (4 bytes): copy it back in stream order behind the stage's
 * kernels to learn of a timed-out hand-off without synchronising the device. */
const unsigned* cmx_lstm_fail_flag(cmx_lstm_t*);
int cmx_glibc_rand_selftest(uint32_t seed, int n, int* out);

/* ------------------------------------------------------------------------
 * 2c. Stage: context plumbing + the 54 small native models = ContextManager, the 54 byte / 8 bit
 *     contexts, Direct / DirectHash / Indirect / Match / Bracket (src/context-manager.cpp,
 *     src/contexts/ *.cpp, src/states/ *.cpp, src/models/{direct,direct-hash,indirect,match,bracket,
 *     byte-model}.cpp; wired at predictor.cpp:90-178,199-356,361-369,421-446)
 * ------------------------------------------------------------------------ */
typedef struct cmx_ctxmodels cmx_ctxmodels_t;
cmx_ctxmodels_t* cmx_ctxmodels_create(const uint8_t vocab[256], int device);
void cmx_ctxmodels_destroy(cmx_ctxmodels_t*);
/* Chunk mode over nbytes already-known bytes. DEVICE pointers:
 *   d_bytes [nbytes]            u8   the bytes coded
 *   d_probs [8*nbytes][pstride] f32  the layer-0 matrix the mixing network reads (pstride >= 2078);
 *                                    OUT: columns 0,1,2 and 2025..2075 of every row (Model::Predict of
 *                                    the 54 models); other columns untouched
 *   d_sel   [8*nbytes][47]      u32  OUT: every mixer's selector at each Predict() (entry 12 = 0: the
 *                                    mixing-network stage derives auxiliary_context_ itself)
 * Asynchronous on `stream`. */
int cmx_ctxmodels_run(cmx_ctxmodels_t*, const uint8_t* d_bytes, size_t nbytes, float* d_probs, size_t pstride,
                      uint32_t* d_sel, void* stream);
/* Predictor::Pretrain (predictor.cpp:471-487) over nbytes dictionary bytes for the models of this
 * stage: same state transitions as cmx_ctxmodels_run, no outputs (mixers/SSE/LSTM/PPMd are not
 * trained during pretraining). */
/* Bit-synchronous mode: the 8 rows (outputs + selectors) the stage would produce if the next byte were *d_byte,
 * without changing any state. Row j depends only on the top j bits of *d_byte: with j bits of the byte coded, row j
 * is what Predict() uses for the next bit. */
int cmx_ctxmodels_peek(cmx_ctxmodels_t*, const uint8_t* d_byte, int bit_index /* 0..7: the row needed; -1: all */,
                       float* d_probs8, size_t probs_stride, uint32_t* d_sel8, void* stream);
int cmx_ctxmodels_pretrain(cmx_ctxmodels_t*, const uint8_t* d_bytes, size_t nbytes, void* stream);
/* Test introspection: bytes that went through the serial path taken when two Indirect models' 256-byte windows of
 * the shared map overlap (indirect.cpp:16-31) -- [0] committed, [1] in dry (peek) passes. Synchronises. */
int cmx_ctxmodels_debug_slow_bytes(cmx_ctxmodels_t*, uint64_t out2[2]);
/* Test hook (state injection): the stage as after `pos` bytes of a stream whose last n bytes were `tail` (HOST) -- ContextManager::history_pos_ (modulo the
 * 100 000 000-byte ring, context-manager.cpp:24-27), every Match model's own byte counter (match.cpp:43-46), the ring's bytes in front of the position. */
int cmx_ctxmodels_debug_set_history(cmx_ctxmodels_t*, uint64_t pos, const uint8_t* tail, uint64_t n);
/* Waits for the handle's work; reports device-side failures. */
int cmx_ctxmodels_sync(cmx_ctxmodels_t*);
/* Test hook: ContextManager registers (25), byte contexts (54), bit contexts (8) between bytes. */
int cmx_ctxmodels_get_manager(cmx_ctxmodels_t*, uint64_t* regs25, uint64_t* ctx54, uint64_t* bitctx8);

/* ------------------------------------------------------------------------
 * 2f. Stage: the vendored fxcm model = FXCM::Predict / FXCM::Perceive (src/models/fxcm.cpp:14-33) around
 *     fxcmv1::Predictor::update1 + modelPrediction (src/models/fxcmv1.cpp:4758-4833, :3798-4757); cmix wires it at
 *     predictor.cpp:98-99 (construction, dictionary path), :462-468 (lstmpr / lstmex set, then Perceive last).
 *     431 values per bit = layer-0 columns 3..433. The text parser half (everything the model computes at a byte
 *     boundary from the bytes alone) runs on the calling host thread, the learned tables on the device.
 * ------------------------------------------------------------------------ */
typedef struct cmx_fxcm cmx_fxcm_t;
/* dictionary_path: cmix's WRT dictionary (the reference's global `dictionary_path`, runner.cpp:17; fxcmv1.cpp:412-428),
 * NULL or "" without one. */
cmx_fxcm_t* cmx_fxcm_create(const char* dictionary_path, int device);
void cmx_fxcm_destroy(cmx_fxcm_t*);
/* Chunk mode over nbytes already-known bytes (whole bytes; chunks of one stream in order).
 *   bytes    [nbytes]            u8  HOST   the bytes coded (parsed on the calling thread before the call returns)
 *   d_bytes  [nbytes]            u8  DEVICE the same bytes
 *   d_lstmpr [8*nbytes]          i16 DEVICE lstmpr as cmix sets it before FXCM::Perceive of each bit (1 + 4094 * p, predictor.cpp:463)
 *   d_lstmex [8*nbytes]          u8  DEVICE lstmex, same (the LSTM's likeliest byte, predictor.cpp:464)
 *   d_probs  [8*nbytes][pstride] f32 DEVICE OUT columns 3..433 of every row: row q = FXCM::Predict() before bit q of the
 *                                    chunk is coded (row 0 of a stream's first chunk = the constructor's 0.5); other
 *                                    columns untouched; pstride >= 434
 * Asynchronous on `stream`. */
int cmx_fxcm_run(cmx_fxcm_t*, const uint8_t* bytes, const uint8_t* d_bytes, size_t nbytes, const int16_t* d_lstmpr, const uint8_t* d_lstmex,
                 float* d_probs, size_t pstride, void* stream);
int cmx_fxcm_sync(cmx_fxcm_t*);
/* 1 if a bounded in-launch wait of the roles kernel (context maps on two workgroups, units, mixers: four workgroups) ran out (the stream's
 * fxcm columns are void from there); syncs. */
int cmx_fxcm_failed(cmx_fxcm_t*);
const unsigned* cmx_fxcm_fail_flag(cmx_fxcm_t*);   /* as cmx_lstm_fail_flag */
int cmx_fxcm_set_upload_stream(cmx_fxcm_t*, void* stream);   /* see cmx_mixnet_set_upload_stream */
/* diagnostics (CMX_FXCM_PROFILE=1 at create time): thread 0's clocks per phase of each role since creation, out[16 role + k] (role 0 = the
 * context maps' first wavefront, 1 = units, 2 = mixers; scripts/gpu_fxcm_time.py names the phases) */
int cmx_fxcm_profile(cmx_fxcm_t*, unsigned long long out128[128]);   /* [64 + 8 bpos + k]: role M by bit position */

/* ---- callers of the path: arithmetic coder + container header (HOST code) -------------------------------------
 * Replaces Encoder::Encode/Flush (src/coder/encoder.cpp:10-39), Decoder::Decoder/Decode (src/coder/decoder.cpp:3-39)
 * and WriteHeader/ReadHeader (src/runner.cpp:34-84). The probabilities are the p[] a pipeline chunk produced
 * (copied to the host by the caller); p[t] = P(bit t = 1). Bytes are coded MSB first (runner.cpp:106-108). */
typedef struct cmx_encoder cmx_encoder_t;
typedef struct cmx_decoder cmx_decoder_t;
cmx_encoder_t* cmx_encoder_create(void);
void cmx_encoder_destroy(cmx_encoder_t*);
int cmx_encoder_encode_bits(cmx_encoder_t*, const float* p, const uint8_t* bits, size_t nbits);
int cmx_encoder_encode_bytes(cmx_encoder_t*, const float* p /* [8*nbytes] */, const uint8_t* bytes, size_t nbytes);
int cmx_encoder_flush(cmx_encoder_t*);                 /* Encoder::Flush; further encode calls fail */
size_t cmx_encoder_size(const cmx_encoder_t*);          /* code bytes produced so far */
const uint8_t* cmx_encoder_data(const cmx_encoder_t*);  /* valid until the next encode/flush/destroy */
/* `code` must stay valid for the decoder's lifetime; reads past its end return zeros like the reference. */
cmx_decoder_t* cmx_decoder_create(const uint8_t* code, size_t len);
void cmx_decoder_destroy(cmx_decoder_t*);
int cmx_decoder_decode(cmx_decoder_t*, float p);        /* the next bit (0/1), -1 on a bad argument */
int cmx_decoder_decode_bits(cmx_decoder_t*, const float* p, size_t nbits, uint8_t* bits_out); /* replay of known p[] */
#define CMX_HEADER_MAX 37
#define CMX_MIN_VOCAB_FILE_SIZE 10000                   /* runner.cpp:14 */
/* 5 bytes of length (bit 39 = dictionary flag) + the 32-byte vocabulary bitmap when length >= 10000. Return the
 * number of header bytes written / consumed, 0 on error. */
size_t cmx_header_write(uint64_t length, const uint8_t vocab[256], int dictionary_used, uint8_t out[CMX_HEADER_MAX]);
size_t cmx_header_read(const uint8_t* in, size_t len, uint64_t* length, int* dictionary_used, uint8_t vocab[256]);

/* ------------------------------------------------------------------------
 * 2d. HOST stage: PPMd order-25 byte model = PPMD::PPMD / PPMD::ByteUpdate (src/models/ppmd.cpp:
 *     1322-1338 and everything below it; constructed with (25, 14000 MB) at predictor.cpp:101).
 *     A pure function of the byte stream, 0.4 % of the reference's CPU time: it runs ahead of the device
 *     pipeline on one host core (SURVEY.md 8 note 2). HOST pointers.
 * ------------------------------------------------------------------------ */
typedef struct cmx_ppmd cmx_ppmd_t;
cmx_ppmd_t* cmx_ppmd_create(const uint8_t vocab[256]);
cmx_ppmd_t* cmx_ppmd_create_ex(const uint8_t vocab[256], int order, int memory_mb); /* test hook: other (order, MB) */
void cmx_ppmd_destroy(cmx_ppmd_t*);
/* Feeds nbytes bytes; out_probs [nbytes][256] f32 receives ByteModel::probs_ after each byte (what
 * predictor.cpp:450-457 hands to the byte mixer and what column 2076 is formed from). Fails (nonzero) if the
 * model would need the reference's memory-exhaustion path (ppmd.cpp:686-727), which is not implemented. */
int cmx_ppmd_run(cmx_ppmd_t*, const uint8_t* bytes, size_t nbytes, float* out_probs);

/* ------------------------------------------------------------------------
 * 3. One stream through every stage built so far: the native orchestration behind the Predictor
 *    surface, a chunk of already-known bytes at a time (Predictor::Predict/Perceive, predictor.cpp:
 *    361-469, in compression look-ahead form). PPMd runs on the calling host thread; the context /
 *    small-model stage, the LSTM and the mixing network run on three internal HIP streams; up to
 *    CMX_PIPELINE_SLOTS chunks may be in flight (a submit blocks while the chunk that many submits back is still running:
 *    a chunk's latency through the stages is several chunk periods, the depth keeps every stage busy).
 * ------------------------------------------------------------------------ */
/* Environment read by the engine (all optional):
 *   GPU_MAX_HW_QUEUES      (HIP's own) must be >= 14 before the first HIP call for the stages of one stream to overlap: an
 *                          engine uses 14 HIP streams (5 stages, upload, 7 paq8 roles, paq8 hand-over) and HIP maps streams
 *                          onto 4 hardware queues by default. The library sets 16 when it is loaded unless the variable is set.
 *   CMX_PIPELINE_STREAMS   2 or 1: throughput mode for several streams per GPU -- fewer hardware queues per engine (8 or 6;
 *                          roles take turns on shared streams, the per-stream period grows)
 *   CMX_MIXNET_SPEC=0      the one-workgroup mixing-network kernel (1 compute unit per stream instead of 27: many streams per GPU)
 *   CMX_MIXNET_XCD=k       the mixing network's 27 workgroups on XCD k, hand-off words through that XCD's L2 (bit-exact either way)
 *   CMX_MIXNET_JITTER=1..15  test hook: pseudo-random stalls in the mixing network's roles (bit-exact by construction; tests/test_gpu_mixnet.py)
 *   (tolerance mode is NOT an environment switch: cmx_mixnet_set_tolerance / cmx_lstm_set_tolerance / cmx_pipeline_set_tolerance --
 *    no program that writes files turns it on)
 *   CMX_P8CM_SERIAL        1: paq8's table families walk every instance serially (A/B timing); 2: the ContextMap family's narrowed walk takes
 *                          its whole-instance fall-back at every second visit (test switch)
 * Co-residency: a stream's stage kernels run for a whole chunk and wait for each other inside the launch, so their workgroups -- 94 per
 * stream (68 with CMX_MIXNET_SPEC=0), most of them a compute unit each -- must all be resident. cmx_pipeline_create / _enable_fxcm /
 * _enable_paq8 keep count per device and refuse (cmx_last_error says why) an engine that would exceed the device's compute units.
 *   CMX_FXCM_PROFILE, CMX_P8FAM_PROFILE, CMX_MIXNET_DBG     in-kernel phase timers / timing experiments (scripts/gpu_*prof*) */
#define CMX_PIPELINE_SLOTS 8   /* chunks in flight per stream (layer-0 matrices the caller cycles through) */
/* Construction ahead of time: start building the vocabulary-independent stages of an engine for `device` -- mixing
 * network, paq8 stage and (with_fxcm != 0; dictionary_path as for cmx_pipeline_enable_fxcm) the fxcm stage, ~16 GB of tables -- on a
 * thread of the library, and return at once. The caller goes on with what has to precede the predictor (runner.cpp:166-202:
 * preprocessing into the temp file, the vocabulary scan); the next cmx_pipeline_create / _enable_fxcm / _enable_paq8 on that device
 * (also behind cmx_stage_input) adopts whatever is ready and waits for what is not. Once per process; purely an optimisation: every
 * stage that was not prewarmed (or failed to) is built by the normal path. */
int cmx_prewarm(int device, const char* dictionary_path, int with_fxcm);

typedef struct cmx_pipeline cmx_pipeline_t;
cmx_pipeline_t* cmx_pipeline_create(const uint8_t vocab[256], int device, size_t max_chunk_bytes);
void cmx_pipeline_destroy(cmx_pipeline_t*);
/* Code the next n <= max_chunk_bytes bytes of the stream.
 *   bytes    HOST   [n]
 *   d_layer0 DEVICE [8n][2078] f32: columns 3..2024 (fxcm, paq8 -- no stage yet) must already hold the
 *                   Model::Predict values of every bit and be complete on the device; all other columns
 *                   are written by the engine
 *   d_p_out  DEVICE [8n] f32 OUT: the value Predictor::Predict() returns for each bit
 * Returns after enqueueing (it only waits for the chunk before the previous one); d_layer0 / d_p_out must
 * stay valid until cmx_pipeline_sync() or until two further submits have returned. */
int cmx_pipeline_submit(cmx_pipeline_t*, const uint8_t* bytes, size_t n, float* d_layer0, float* d_p_out);
/* The same chunk in steps, for look-ahead coding while some model families still run on the host:
 *   _begin   PPMd (this thread), upload, context stage and LSTM are enqueued; up to CMX_PIPELINE_SLOTS chunks may be begun and not
 *            finished;
 *   _hints   (optional) for the oldest begun chunk that has not handed them out: lstm_p[t], lstm_ex[t], t = 0 .. 8n,
 *            = the LSTM byte mixer's ByteModel::Predict value and `ex` for bit t of the chunk (t = 8n: the first bit
 *            after it). After coding bit t the reference's globals hold lstmpr = 1 + 4094 * lstm_p[t+1] (float
 *            arithmetic, truncated) and lstmex = lstm_ex[t+1] (predictor.cpp:180-182,462-465): what a host fxcm
 *            needs to run ahead. HOST arrays of 8n+1 entries; waits for that chunk's LSTM stage;
 *   _finish  the oldest begun chunk: its layer-0 columns 3..2024 from HOST rows cols[8n][2022] (NULL = already in
 *            d_layer0), then the mixing network -> d_p_out[8n]. Asynchronous.
 * cmx_pipeline_submit = _begin + _finish(NULL). */
int cmx_pipeline_begin(cmx_pipeline_t*, const uint8_t* bytes, size_t nbytes, float* d_layer0);
int cmx_pipeline_hints(cmx_pipeline_t*, float* lstm_p, int* lstm_ex);
int cmx_pipeline_finish(cmx_pipeline_t*, const float* cols, float* d_p_out);
/* The fxcm stage on the device (section 2f), opt-in, before the first chunk: cmx_pipeline_begin then also derives the
 * per-bit lstmpr / lstmex from the LSTM stage's output on the device and runs the fxcm stage (text parser on the calling
 * thread, kernel on a fourth HIP stream) into columns 3..433; cmx_pipeline_pretrain covers it; the caller hands in only
 * the remaining columns with cmx_pipeline_finish_cols (HOST rows of ncols floats = layer-0 columns first_col ..
 * first_col + ncols - 1; with the fxcm stage enabled first_col >= 434) -- cmx_pipeline_finish with rows then fails.
 * dictionary_path: as cmx_fxcm_create. */
int cmx_pipeline_enable_fxcm(cmx_pipeline_t*, const char* dictionary_path);
/* The paq8 stage on the device (section 2f at the end of this file), opt-in, before the first chunk: cmx_pipeline_begin then
 * also runs it (front end on the calling thread, kernels on the stage's own streams) into columns 434..2024 of d_layer0;
 * cmx_pipeline_pretrain feeds it the dictionary. With both vendored families on the device cmx_pipeline_finish takes no
 * columns (cmx_pipeline_submit does everything) and the Predictor shim needs no cmx_set_model_outputs. */
int cmx_pipeline_enable_paq8(cmx_pipeline_t*);
/* Wait until chunk number `index` (0 = the first submitted) has left the mixing network; only the last CMX_PIPELINE_SLOTS can be waited for. */
int cmx_pipeline_wait(cmx_pipeline_t*, uint64_t index);
/* cmx_pipeline_wait(index), then that chunk's probabilities (8 x its byte count floats, from the d_p_out it was submitted with) into
 * the HOST buffer p_host (page-locked for speed). Waits for this chunk only; the chunks behind it stay in flight. */
int cmx_pipeline_fetch(cmx_pipeline_t*, uint64_t index, float* p_host);
int cmx_pipeline_paq8_enabled(cmx_pipeline_t*);
int cmx_pipeline_paq8_total_ms(cmx_pipeline_t*, double* ms);
/* the paq8 stage's role-kernel times (cmx_p8stage_role_ms) since the last reset of the stage totals */
int cmx_pipeline_paq8_role_ms(cmx_pipeline_t*, double ms[7], uint64_t* chunks);
/* wall time of the calling thread inside begin / finish since the last reset of the stage totals, ms: [0] waiting for a
 * slot, [1] PPMd, [2] uploads + context stage + LSTM enqueue, [3] fxcm (parser + enqueue), [4] paq8 (front end + enqueue),
 * [5] mixing network enqueue */
int cmx_pipeline_host_ms(cmx_pipeline_t*, double ms[6]);
int cmx_pipeline_fxcm_enabled(cmx_pipeline_t*);
int cmx_pipeline_finish_cols(cmx_pipeline_t*, const float* cols, int first_col, int ncols, float* d_p_out);
int cmx_pipeline_fxcm_total_ms(cmx_pipeline_t*, double* ms);
/* the mixing network's tolerance mode for this stream (cmx_mixnet_set_tolerance), before the first chunk; _mixnet_mode: 0 strict, 1 tolerance */
/* diagnostics of a long run (all synchronise the device): rows allocated by each of the 47 final mixers (cap 10 000 + the shared
 * overflow row, mixer.cpp:16-36); speculation statistics of the mixing network (cmx_mixnet_spec_stats); the paq8 family kernel's
 * phase / path counters (CMX_P8FAM_PROFILE=1 at create time; cmx_p8stage_profile); PPMd's arena: {bytes reserved, untouched, in use} */
int cmx_mixnet_rows(cmx_mixnet_t*, uint32_t rows[47]);
int cmx_ppmd_arena(cmx_ppmd_t*, uint64_t out3[3]);
int cmx_pipeline_mixnet_rows(cmx_pipeline_t*, uint32_t rows[47]);
/* diagnosis: from now on every mixer's output (Mixer::Mix, mixer.cpp:38-55, all 47) of every bit of the stream's look-ahead chunks goes to
 * the DEVICE area d_mix[cap_bits][47] in stream order (bits past cap_bits are not recorded); NULL switches it off */
int cmx_pipeline_debug_mix_out(cmx_pipeline_t*, float* d_mix, uint64_t cap_bits);
/* diagnosis: the DEVICE selectors [8 n][47] and coded bits [8 n] of chunk number `index` (one of the last CMX_PIPELINE_SLOTS submitted), valid until its slot is reused */
int cmx_pipeline_debug_slot(cmx_pipeline_t*, uint64_t index, const uint32_t** d_sel, const uint8_t** d_bits, size_t* nbytes);
int cmx_pipeline_spec_stats(cmx_pipeline_t*, uint64_t out[5]);
int cmx_pipeline_paq8_profile(cmx_pipeline_t*, unsigned long long out128[128]);
int cmx_pipeline_ppmd_arena(cmx_pipeline_t*, uint64_t out3[3]);
int cmx_pipeline_set_tolerance(cmx_pipeline_t*, int on);
int cmx_pipeline_mixnet_mode(cmx_pipeline_t*);
/* Predictor::Pretrain over n dictionary bytes (HOST pointer), before the first submit: only the stages holding
 * `models_` learn (today: contexts + small models); mixers, SSE, LSTM and PPMd are not trained (predictor.cpp:471-487). */
int cmx_pipeline_pretrain(cmx_pipeline_t*, const uint8_t* bytes, size_t n);
int cmx_pipeline_sync(cmx_pipeline_t*);
/* HIP-event time (ms) the last submitted chunk spent in [0] contexts+small models, [1] LSTM, [2] mixing network. */
int cmx_pipeline_last_stage_ms(cmx_pipeline_t*, float ms[3]);
/* HIP-event time of each stage (contexts, LSTM, mixing network; ms) summed over every chunk that has finished since
 * the last reset, and the number of those chunks: call after cmx_pipeline_sync(). */
int cmx_pipeline_stage_totals(cmx_pipeline_t*, double ms[3], uint64_t* chunks, int reset);

/* ------------------------------------------------------------------------
 * Device libm probes (parity tests): evaluate the engine's expf / tanhf /
 * logistic on the device for n host floats. which: 0 expf, 1 tanhf, 2 logistic
 * ------------------------------------------------------------------------ */
int cmx_probe_libm(int device, int which, const float* x, float* y, size_t n);

/* ---- 2f. The paq8 stage: layer-0 columns 434..2024 = the 1591 values PAQ8::Predict() returns per bit (replaces
 *        src/models/paq8.{h,cpp} as wired at src/predictor.cpp:85-97: PAQ8::Predict / Perceive around
 *        paq8::Predictor::update, reference src/models/paq8.cpp:8248-8362, and contextModel2 :8101-8207).
 *        One handle per stream. cmx_p8stage_run takes the NEXT nbytes bytes of the stream in HOST memory: the front end
 *        (word / text / record / XML / x86 / match parsers, cmix_amd/csrc/p8front/) runs on the calling thread, the
 *        tables, the 1552 x 28 mixer and the APM chains run as kernels ordered behind `stream`.
 *        d_out: device matrix of f32, row t (ld floats apart, ld >= 1591) receives the 1591 values valid BEFORE bit t of
 *        this chunk is coded (pass layer0 + 434 with ld = 2078 to fill the predictor's columns in place).
 *        Scope: general data and text blocks; a stream that would switch on paq8's image / audio / JPEG sub-models makes
 *        the call fail (cmx_last_error names the detector) -- never a silently different number. ~9 GB of HBM. ---- */
typedef struct cmx_p8stage cmx_p8stage_t;
cmx_p8stage_t* cmx_p8stage_create(int device);
void cmx_p8stage_destroy(cmx_p8stage_t*);
int cmx_p8stage_run(cmx_p8stage_t*, const uint8_t* bytes_host, size_t nbytes, float* d_out, size_t ld, void* stream);
int cmx_p8stage_sync(cmx_p8stage_t*);
/* HIP-event time of the role kernels summed over the chunks collected so far (a chunk is collected when its staging
 * buffer comes round again, or by _sync): ms[0] family, [1] mixer + APM chains, [2..4] ContextMap2 x 3, [5] small learners (lanes), [6] DMC forest. */
int cmx_p8stage_role_ms(cmx_p8stage_t*, double ms[7], uint64_t* chunks, int reset);
int cmx_p8stage_set_upload_stream(cmx_p8stage_t*, void* stream);   /* see cmx_mixnet_set_upload_stream */
/* diagnostics (CMX_P8FAM_PROFILE=1 at create time): the family kernel's clocks by bit position and phase, and how often it left the
 * common path: out[8 bp + k] (k = 0 per-step values, 1 phase 1, 2 barrier, 3 run / rounds, 4 rest), out[64 + bp] steps in rounds,
 * out[72 + bp] instances walked, out[80 + bp] steps */
int cmx_p8stage_profile(cmx_p8stage_t*, unsigned long long out128[128]);


/* ------------------------------------------------------------------------
 * 4. The DECODER's form of the stages: the late-bit protocol (cmix_amd/csrc/cmx_late.h)
 *
 * Decoder::Decode (src/coder/decoder.cpp:20-39) learns bit t only after Predictor::Predict() has returned p(t): nothing can be
 * looked ahead. The stage kernels are the chunk kernels of section 3, but each waits where it reads a coded bit, a host record or
 * another stage's row: they are launched for a chunk of bytes that do not exist yet, and a BOX in host-coherent memory carries the
 * bits in (with each bit: the host records of the step after it) and p out. cmx_pipeline_late_* below drive one stream this way --
 * every model family on the device, no column from the caller, no reference object; cmx_predict()/cmx_perceive() of section 1 sit
 * on top of them when no input was staged. The per-stage entry points are what the pipeline (and the stage-level tests) call.
 * ------------------------------------------------------------------------ */
void* cmx_late_alloc(size_t bytes);   /* zeroed host-coherent pinned memory: boxes, rows, records */
void cmx_late_free(void* p);
void* cmx_late_alloc_dev(int device, size_t bytes);   /* zeroed UNCACHED device memory: rows / row counters kernels hand to each other while they run */
void cmx_late_free_dev(void* p);
size_t cmx_late_box_bytes(size_t nbits);   /* allocation size of a box for a chunk of nbits */
/* `box` of the per-stage entry points below: a `const CmxLate*` (cmx_late.h) = the host box, the chunk's row counters and their base */
int cmx_late_bump(int device, uint32_t* counter, uint32_t value, uint32_t* counter2, uint32_t value2, void* stream);
int cmx_ctxmodels_run_late(cmx_ctxmodels_t*, void* box, size_t nbytes, float* probs, size_t pstride, uint32_t* sel, float* brk_dist,
                           const float** brk_dist0_out, void* stream);
int cmx_bytemodel_late_run(int device, void* box, size_t nbytes, const float* brk0, const float* brk, const float* ppmd, const float* lstm0, const float* lstm,
                           const uint32_t* c0_brk, uint32_t c0_brk_want, const uint32_t* c0_lstm, uint32_t c0_lstm_want, float* layer0, size_t pstride,
                           int16_t* hint_pr, uint8_t* hint_ex, uint8_t* dbit0, const void* relay_dev, int nrelay, void* stream);
/* what a stage's records need from the stream's relay wave (an array of cmx_late_relay_t, cmx_late.h); returns the number of entries, -1 on error */
int cmx_p8stage_late_relay(cmx_p8stage_t*, int slot, void* out, int max);
int cmx_fxcm_late_relay(cmx_fxcm_t*, int slot, void* out, int max);
const float* cmx_lstm_byte_probs(cmx_lstm_t*);
/* _late_prepare: everything a stage allocates for chunks of that size, BEFORE the stream's first kernels are launched (an allocation
 * that maps memory into the device can wait for running kernels -- which, here, wait for the host) */
int cmx_fxcm_late_prepare(cmx_fxcm_t*, size_t nbytes);
int cmx_p8stage_late_prepare(cmx_p8stage_t*, size_t nbytes);
int cmx_mixnet_late_prepare(cmx_mixnet_t*, size_t nbits);
int cmx_fxcm_run_late(cmx_fxcm_t*, void* box, size_t nbytes, const int16_t* hint_pr, const uint8_t* hint_ex, float* probs, size_t pstride, int slot, void* stream);
int cmx_fxcm_late_byte(cmx_fxcm_t*, int slot, size_t b, uint8_t byte);
int cmx_p8stage_run_late(cmx_p8stage_t*, void* box, size_t nbytes, float* out, size_t ld, int slot);
int cmx_p8stage_late_emit(cmx_p8stage_t*, int slot, size_t step);
int cmx_p8stage_late_bit(cmx_p8stage_t*, int bit);
int cmx_p8stage_mixfail(cmx_p8stage_t*);
/* test hook: the counter of the ContextMap family's shared generator (paq8.cpp:152-165: `int i`, only i mod 64 matters), a multiple of 64, before the
 * first byte -- so that a test passes 2^31 / 2^32 draws (4 / 8 MB into a stream) within a few KB */
int cmx_p8stage_set_generator_counter(cmx_p8stage_t*, uint32_t counter);
/* test hook (state injection): the front end's byte position `pos` (paq8.cpp:167: the index into the 2^30-byte history ring at level 11), before the first
 * byte -- a test passes the ring's end, which a stream reaches after 1 GB, within a few KB */
int cmx_p8stage_debug_set_pos(cmx_p8stage_t*, int pos);
int cmx_mixnet_run_late(cmx_mixnet_t*, void* box, const float* probs, const uint32_t* sel, size_t nbits, void* stream);
/* the LSTM byte mixer in a decoder's chunk (round 6): one forward launch for the bytes b0 .. up to the end of the truncated-BPTT block or of the chunk -- its steps
 * wait for their bytes inside the launch and count the distributions they write (LC_LSTM); returns the number of bytes covered, -1 on error */
int cmx_lstm_run_late(cmx_lstm_t*, const void* late_box, const float* d_ppmd, const float* h_ppmd, const uint8_t* h_bytes, float* d_out, size_t b0, size_t nchunk, void* stream);
/* One stream, bit by bit. The handle is a pipeline of section 3 with the fxcm and paq8 stages enabled (and, if there is a
 * dictionary, pretrained): cmx_pipeline_late_start() launches the first chunks' kernels (last_bit: the bit coded before the first
 * one -- the last Pretrain bit, else 0); then strictly alternating cmx_pipeline_late_predict() -> p, cmx_pipeline_late_perceive(bit).
 * cmx_pipeline_late_stop() (also called by cmx_pipeline_destroy) unwinds the kernels of the chunk in progress. */
int cmx_pipeline_late_start(cmx_pipeline_t*, int last_bit);
float cmx_pipeline_late_predict(cmx_pipeline_t*);
int cmx_pipeline_late_perceive(cmx_pipeline_t*, int bit);
int cmx_pipeline_late_stop(cmx_pipeline_t*);
/* predict / perceive for n known bits in one call (a decoder minus the arithmetic decoder): p_out[i] = p before bits[i] */
int cmx_pipeline_late_replay(cmx_pipeline_t*, const uint8_t* bits, size_t n, float* p_out);
/* diagnostics: the calling thread's wall time since start, in ms: [0] waiting for p, [1] PPMd, [2] paq8 front end, [3] fxcm parser, [4] LSTM launches, [5] chunk launches */
int cmx_pipeline_late_host_ms(cmx_pipeline_t*, double ms[6], uint64_t* bits);
/* diagnostics: every row counter of the chunk being decoded and the device time (100 MHz) it was last moved at: out[2 i], out[2 i + 1] for
 * counter i of cmx_late.h (14 = the relay's); out[30..39] the mixing network's stamps of the current bit. Call between predict() and perceive(). */
int cmx_pipeline_late_debug_times(cmx_pipeline_t*, uint32_t out[48]);
/* test hook: the layer-0 row (2078 f32, host memory) and 47 selectors of the bit predicted last; valid until the matching perceive */
const float* cmx_pipeline_late_debug_row(cmx_pipeline_t*, const uint32_t** sel);
/* test hooks (CMX_LATE_DEBUG=1 in the environment when the handle is built): the 47 Mixer::Mix values of the bit BEFORE the one predicted last */
int cmx_pipeline_late_debug_mix(cmx_pipeline_t*, float out47[47]);
const float* cmx_mixnet_late_debug_mix(cmx_mixnet_t*, uint64_t chunk, size_t bit);
uint64_t cmx_mixnet_runs(cmx_mixnet_t*);

#ifdef __cplusplus
}
#endif
#endif /* CMIX_AMD_H */
