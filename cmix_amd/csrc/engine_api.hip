// engine_api.hip -- the whole-predictor surface (cmx_create .. cmx_destroy): `class Predictor`'s four members
// (src/predictor.h:17-22) in their strict per-bit protocol, the form a Decoder needs (src/coder/decoder.cpp:20-39:
// bit t is unknown until Predict() of bit t has returned).
//
// Per Predict() (predictor.cpp:361-419):
//   device  contexts + 54 small models, "dry" pass over the partially known byte -> columns 0-2, 2025-2075 and the
//           47 selectors of this bit (ctxmodels_kernels.hip, dry mode: no state survives the pass)
//   device  ByteModel::Predict of PPMd and of the LSTM byte mixer over the same partial byte -> columns 2076, 2077
//   caller  columns 3..2024 (fxcm, paq8) -- their device stages walk whole chunks of known bytes (cmx_pipeline_*), there is
//           no bit-synchronous form of them yet; on this surface the shim that owns the reference's two vendored models
//           hands their outputs in through cmx_set_model_outputs()
//   device  mixing network + SSE, bit-synchronous (cmx_mixnet_predict)
// Per Perceive(bit) (predictor.cpp:421-469): the mixing network learns; on the 8th bit the byte is committed to the
// context stage (the same kernel, now for real), to PPMd (host stage) and to the LSTM; then lstmpr / lstmex
// (predictor.cpp:462-466), which fxcm reads in its own Perceive, are refreshed for the caller.
// Pretrain(bit) (predictor.cpp:471-487) collects bytes and trains the context stage in batches.
//
// Everything of a handle goes through one HIP stream in order, host buffers are page-locked, and the host waits
// exactly twice per bit: for p at the end of cmx_predict(), and for lstmpr/lstmex in cmx_get_lstm_hint() -- not in
// cmx_perceive(), so the caller's own work (the shim runs paq8's Perceive there) overlaps the device's. Still
// launch-latency bound by construction (about ten small launches/copies per bit): this is the decode path and the
// parity anchor; compression throughput comes from the chunk pipeline (pipeline_api.hip). No CPU fallback.
//
// LOOK-AHEAD MODE ("optional cmx_stage_input"): a compressor knows every bit in advance. When the caller
// stages the bytes it is about to code (cmx_stage_input, before the first cmx_predict), the handle builds the chunk
// pipeline instead of the per-bit stages -- every model family on the device, fxcm and paq8 included, so no column is
// taken from the caller -- and keeps CMX_PIPELINE_SLOTS chunks of 4 KB in flight: cmx_predict() pops the next
// probability from the chunk that has left the mixing network, cmx_perceive(bit) checks the bit against the staged one
// (a mismatch is an error: the device has already learnt the staged bit) and, when a chunk is used up, submits the next.
// The reference's unmodified coder (encoder.cpp:14-30) and runner (runner.cpp:101-119) drive exactly this protocol.
// The handle is built lazily for that reason: cmx_create() checks the device and records the vocabulary; the first call
// that needs device state decides which set of stages exists (never both).
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <string>
#include <vector>

#include "../../include/cmix_amd.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip

namespace {
struct Pinned {               // one page-locked block: everything the host hands to / takes from the stream
  float staged[2022];         // fxcm / paq8 columns of the pending bit
  float ppmd[256];            // PPMd's distribution after the byte just committed
  float p;                    // Predict() value
  float hint_p[8];            // LSTM bit predictions of the current byte position (refresh_hint)
  int hint_ex[8];
  float row[CMX_N_INPUTS];    // cmx_debug_last_row
  unsigned lstm_fail;         // the LSTM stage's sticky hand-off flag, copied back behind every byte's LSTM step
  uint8_t byte[16];           // ring: partial / committed bytes on their way to the device
};
}  // namespace

namespace {
constexpr size_t kLaChunk = 4096;             // bytes per look-ahead chunk (the size bench.py and cmix_engine use)
constexpr size_t kLaSlots = CMX_PIPELINE_SLOTS;
struct LookAhead {                            // the handle in look-ahead mode (cmx_stage_input)
  cmx_pipeline_t* pipe = nullptr;
  std::vector<uint8_t> data;                  // staged bytes from stream offset `base` on
  size_t base = 0, staged = 0, submitted = 0; // stream offsets: data[0], end of staged input, end of submitted chunks
  bool ended = false;                         // the caller has marked the end of its input: the ragged tail may go
  struct Chunk { size_t off = 0, n = 0; } ring[kLaSlots];
  uint64_t n_sub = 0, n_done = 0;             // chunks submitted / used up
  float* d_layer0[kLaSlots] = {};
  float* d_p[kLaSlots] = {};
  float* h_p = nullptr;                       // page-locked [8 * kLaChunk]: probabilities of chunk n_done once loaded
  bool loaded = false;
  size_t bit = 0;                             // next bit of chunk n_done
};
}  // namespace

struct cmx_engine {
  int device = 0;
  uint8_t vocab[256];
  std::string dict;               // the hidden global `dictionary_path` (runner.cpp:17), read by the fxcm stage
  bool has_dict = false;
  int mode = 0;                   // 0 undecided, 1 per-bit stages built (columns from the caller), 2 look-ahead pipeline built, 3 the decoder's pipeline (late-bit protocol)
  LookAhead* la = nullptr;
  cmx_pipeline_t* late = nullptr; // mode 3: every model family on the device, bits arriving one at a time (cmx_pipeline_late_*)
  cmx_ctxmodels_t* ctx = nullptr;
  cmx_lstm_t* lstm = nullptr;
  cmx_mixnet_t* mix = nullptr;
  cmx_ppmd_t* ppmd = nullptr;
  hipStream_t st = nullptr;       // every launch and copy of this handle, in order
  Pinned* pin = nullptr;
  uint8_t* d_byte = nullptr;      // [16] ring, mirrors pin->byte
  float* d_rows = nullptr;        // [8][2078]: the dry pass writes all 8 rows, row j is the valid one
  uint32_t* d_sel = nullptr;      // [8][47]
  float* d_ppmd = nullptr;        // [256]: PPMd's distribution while the current byte is coded
  float* d_lstm = nullptr;        // [256]: the LSTM byte mixer's
  float* d_lstm_next = nullptr;   // [256]
  float* d_hint = nullptr;        // [8] f32 bit predictions + [8] i32 arg-max symbols
  float* d_p = nullptr;
  float* d_scratch = nullptr;     // [8][2078] outputs of the committing pass (unused)
  uint32_t* d_scratch_sel = nullptr;
  unsigned byte_slot = 0;
  bool have_staged = false, predicted = false, started = false, hint_pending = false;
  int j = 0;                      // bits of the current byte already coded
  int row_j = 0;                  // row of d_rows the last predict used
  unsigned partial = 0;           // those bits
  int lstmpr = 0, lstmex = 0;
  bool failed = false;            // sticky: a step failed after the handle's state had begun to change (every later call is refused)
  double t_predict = 0, t_perceive = 0, t_outside = 0, t_last = 0; unsigned long long n_calls = 0;   // CMX_TIMING=1: wall time inside cmx_predict / cmx_perceive and between them
  std::vector<uint8_t> pre;       // Pretrain bytes not yet trained
  int pre_j = 0;
  unsigned pre_partial = 0;
};

#define E_HIP(x)                                                                          \
  do {                                                                                    \
    hipError_t e_ = (x);                                                                  \
    if (e_ != hipSuccess) {                                                               \
      cmx_set_err(std::string(where) + ": " #x ": " + hipGetErrorString(e_));             \
      return fail;                                                                        \
    }                                                                                     \
  } while (0)

namespace {

// A call that fails after it has begun to change the handle's state (mixers trained, a stage half-enqueued) leaves the
// stream in no state the reference ever is in: the handle is poisoned instead of silently diverging on a retry.
struct Txn {
  cmx_engine* h; bool ok = false;
  explicit Txn(cmx_engine* e) : h(e) {}
  ~Txn() { if (!ok) h->failed = true; }
};
bool refused(cmx_engine* h, const char* where) {
  if (!h->failed) return false;
  cmx_set_err(std::string(where) + ": an earlier call on this handle failed part-way; its state is void (destroy it)");
  return true;
}

// A byte value on its way to the device: a ring of 16 page-locked slots, each mirrored by its own device byte. The
// caller synchronises at least once per coded bit, far more often than the ring wraps.
const uint8_t* send_byte(cmx_engine* h, uint8_t v) {
  const unsigned s = h->byte_slot++ & 15;
  h->pin->byte[s] = v;
  if (hipMemcpyAsync(h->d_byte + s, &h->pin->byte[s], 1, hipMemcpyHostToDevice, h->st) != hipSuccess) return nullptr;
  return h->d_byte + s;
}

// lstmpr / lstmex for the next bit: Discretize(byte_mixer->Predict()[0]) and byte_mixer->ex (predictor.cpp:180-182,
// 462-465) = ByteModel::Predict of the LSTM's distribution over the bits coded so far in this byte. Enqueued
// here, collected by finish_hint() -- the host is free in between (the shim runs paq8's Perceive there).
int enqueue_hint(cmx_engine* h) {
  const char* where = "cmx_perceive";
  const int fail = 1;
  const uint8_t* db = send_byte(h, (uint8_t)(h->partial << (8 - h->j)));
  if (!db) { cmx_set_err("cmx_perceive: upload failed"); return 1; }
  // the value is also the LSTM column (2077) of the row the next Predict() assembles
  if (cmx_bytemodel_bit_run(h->device, h->d_lstm, db, h->j, h->d_hint, 1, (int*)(h->d_hint + 8),
                            h->d_rows + (size_t)h->j * CMX_N_INPUTS + 2077, h->st)) return 1;
  E_HIP(hipMemcpyAsync(h->pin->hint_p, h->d_hint, 16 * 4, hipMemcpyDeviceToHost, h->st));  // hint_p[8] + hint_ex[8]
  h->hint_pending = true;
  return 0;
}

// The multi-workgroup LSTM kernels bound every in-launch wait; one that ran out leaves garbage behind, not a hang, and
// sets a sticky flag. It is copied back behind every byte's LSTM step (perceive) and looked at here, after the stream
// has been waited for anyway: a decoder must never go on with a void distribution.
bool lstm_timed_out(cmx_engine* h, const char* where) {
  if (!h->pin->lstm_fail) return false;
  cmx_set_err(std::string(where) + ": an in-launch hand-off of the LSTM kernels timed out (workgroups not co-resident?): "
              "the handle's LSTM state is void");
  h->failed = true;
  return true;
}

int finish_hint(cmx_engine* h) {
  const char* where = "cmx_get_lstm_hint";
  const int fail = 1;
  if (!h->hint_pending) return 0;
  E_HIP(hipStreamSynchronize(h->st));
  if (lstm_timed_out(h, where)) return 1;
  volatile float prod = 4094.0f * h->pin->hint_p[h->j];
  const float s = 1.0f + prod;
  h->lstmpr = (int)(unsigned)s;
  h->lstmex = h->pin->hint_ex[h->j];
  h->hint_pending = false;
  return 0;
}

void free_perbit(cmx_engine* h) {
  for (void* p : {(void*)h->d_byte, (void*)h->d_rows, (void*)h->d_sel, (void*)h->d_ppmd, (void*)h->d_lstm,
                  (void*)h->d_lstm_next, (void*)h->d_hint, (void*)h->d_p, (void*)h->d_scratch, (void*)h->d_scratch_sel})
    if (p) (void)hipFree(p);
  h->d_byte = nullptr; h->d_rows = nullptr; h->d_sel = nullptr; h->d_ppmd = nullptr; h->d_lstm = nullptr;
  h->d_lstm_next = nullptr; h->d_hint = nullptr; h->d_p = nullptr; h->d_scratch = nullptr; h->d_scratch_sel = nullptr;
  if (h->pin) (void)hipHostFree(h->pin);
  h->pin = nullptr;
  if (h->st) (void)hipStreamDestroy(h->st);
  h->st = nullptr;
  cmx_mixnet_destroy(h->mix); h->mix = nullptr;
  cmx_lstm_destroy(h->lstm); h->lstm = nullptr;
  cmx_ctxmodels_destroy(h->ctx); h->ctx = nullptr;
  cmx_ppmd_destroy(h->ppmd); h->ppmd = nullptr;
}

// The per-bit stages (mode 1): built by the first call that needs them when no input has been staged.
int ensure_perbit(cmx_engine* h, const char* where) {
  if (h->mode == 1) return 0;
  if (h->mode == 2) { cmx_set_err(std::string(where) + ": the handle is in look-ahead mode (cmx_stage_input was called)"); return 1; }
  if (h->mode == 3) { cmx_set_err(std::string(where) + ": the handle is decoding with every model family on the device (no columns are taken from the caller)"); return 1; }
  h->ctx = cmx_ctxmodels_create(h->vocab, h->device);
  h->lstm = h->ctx ? cmx_lstm_create(h->vocab, 31, h->device) : nullptr;  // 31 rand() draws precede the LSTM (indirect.cpp:10)
  h->mix = h->lstm ? cmx_mixnet_create(h->device) : nullptr;
  h->ppmd = h->mix ? cmx_ppmd_create(h->vocab) : nullptr;
  if (!h->ppmd) { free_perbit(h); return 1; }  // the failing stage has set the error
  bool ok = hipSetDevice(h->device) == hipSuccess;
  ok = ok && hipStreamCreateWithFlags(&h->st, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&h->pin, sizeof(Pinned), hipHostMallocDefault) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_byte, 16) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_rows, 8 * CMX_N_INPUTS * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_sel, 8 * CMX_N_MIXERS * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_ppmd, 256 * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_lstm, 256 * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_lstm_next, 256 * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_hint, 16 * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_p, 16) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_scratch, 8 * CMX_N_INPUTS * 4) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_scratch_sel, 8 * CMX_N_MIXERS * 4) == hipSuccess;
  if (!ok) { cmx_set_err(std::string(where) + ": stream / buffer allocation failed"); free_perbit(h); return 1; }
  memset(h->pin, 0, sizeof(Pinned));
  float u[256];
  for (int i = 0; i < 256; ++i) u[i] = (float)(1.0 / 256);  // ByteModel constructor (byte-model.cpp:5-6)
  if (hipMemcpy(h->d_ppmd, u, sizeof u, hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(h->d_lstm, u, sizeof u, hipMemcpyHostToDevice) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    cmx_set_err(std::string(where) + ": upload failed");
    free_perbit(h);
    return 1;
  }
  if (enqueue_hint(h) || finish_hint(h)) { free_perbit(h); return 1; }  // lstmpr/lstmex + column 2077 of bit 0
  h->mode = 1;
  return 0;
}

// Pretrain bytes collected so far -> the context stage, 64 KB at a time (per-bit mode)
int flush_pretrain(cmx_engine* h) {
  const char* where = "cmx_pretrain";
  const int fail = 1;
  if (h->pre.empty()) return 0;
  uint8_t* d = nullptr;
  E_HIP(hipMalloc((void**)&d, h->pre.size()));
  bool ok = hipMemcpy(d, h->pre.data(), h->pre.size(), hipMemcpyHostToDevice) == hipSuccess;
  ok = ok && cmx_ctxmodels_pretrain(h->ctx, d, h->pre.size(), h->st) == 0;
  ok = hipStreamSynchronize(h->st) == hipSuccess && ok;
  (void)hipFree(d);
  h->pre.clear();
  h->pre.shrink_to_fit();
  if (!ok) { cmx_set_err("cmx_pretrain: device error"); return 1; }
  return 0;
}

// ---- the decoder's mode (3): the chunk pipeline under the late-bit protocol (cmx_late.h) -------------------------------
// What a Decoder gets (decoder.cpp:20-39) when it has staged nothing and hands in no columns: the same stage kernels as the
// look-ahead compressor -- fxcm and paq8 included --, launched for chunks of bytes that do not exist yet; cmx_perceive(bit)
// publishes the bit (and the host stages' records of the step after it), cmx_predict() waits for the mixing network's p.
double cmx_timing_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec + 1e-9 * ts.tv_nsec; }
bool cmx_timing_on() { static const bool on = getenv("CMX_TIMING") != nullptr; return on; }
#define CMX_TIMING(what, t0) do { if (cmx_timing_on()) fprintf(stderr, "[cmx timing] %-28s %.3f s\n", what, cmx_timing_now() - (t0)); } while (0)
int ensure_late(cmx_engine* h, const char* where) {
  if (h->mode == 3) return 0;
  if (h->mode != 0) { cmx_set_err(std::string(where) + ": the handle is already in another mode"); return 1; }
  if (h->pre_j) { cmx_set_err(std::string(where) + ": Pretrain() stopped inside a byte"); return 1; }
  double t0 = cmx_timing_now();
  if (cmx_timing_on()) fprintf(stderr, "[cmx timing] first predict (decoder) at    %.3f s (process clock)\n", t0);
  h->late = cmx_pipeline_create(h->vocab, h->device, kLaChunk);
  CMX_TIMING("pipeline_create", t0); t0 = cmx_timing_now();
  bool ok = h->late && cmx_pipeline_enable_fxcm(h->late, h->has_dict ? h->dict.c_str() : nullptr) == 0;
  CMX_TIMING("enable_fxcm", t0); t0 = cmx_timing_now();
  ok = ok && cmx_pipeline_enable_paq8(h->late) == 0;
  CMX_TIMING("enable_paq8", t0); t0 = cmx_timing_now();
  // Predictor::Pretrain's bytes (predictor.cpp:471-487), collected since cmx_create, in one batch through the stages
  ok = ok && (h->pre.empty() || cmx_pipeline_pretrain(h->late, h->pre.data(), h->pre.size()) == 0);
  CMX_TIMING("pretrain", t0); t0 = cmx_timing_now();
  ok = ok && cmx_pipeline_late_start(h->late, h->pre.empty() ? 0 : (int)(h->pre.back() & 1)) == 0;
  CMX_TIMING("late_start", t0);
  if (!ok) { cmx_pipeline_destroy(h->late); h->late = nullptr; return 1; }   // the failing stage has set the error
  h->pre.clear();
  h->pre.shrink_to_fit();
  h->mode = 3;
  h->started = true;
  return 0;
}

// ---- look-ahead mode --------------------------------------------------------------------------------------------
void free_lookahead(cmx_engine* h) {
  LookAhead* la = h->la;
  if (!la) return;
  cmx_pipeline_destroy(la->pipe);   // synchronises the device first
  for (size_t i = 0; i < kLaSlots; ++i) {
    if (la->d_layer0[i]) cmx_device_free(h->device, la->d_layer0[i]);
    if (la->d_p[i]) cmx_device_free(h->device, la->d_p[i]);
  }
  if (la->h_p) cmx_host_free(la->h_p);
  delete la;
  h->la = nullptr;
}

int ensure_lookahead(cmx_engine* h) {
  if (h->mode == 2) return 0;
  if (h->mode == 1 || h->mode == 3) {
    cmx_set_err("cmx_stage_input: the per-bit stages of this handle already exist (a predict / pretrain flush came first): "
                "stage the input before the first cmx_predict()");
    return 1;
  }
  if (h->pre_j) { cmx_set_err("cmx_stage_input: Pretrain() stopped inside a byte"); return 1; }
  LookAhead* la = new LookAhead();
  h->la = la;
  la->pipe = cmx_pipeline_create(h->vocab, h->device, kLaChunk);
  bool ok = la->pipe && cmx_pipeline_enable_fxcm(la->pipe, h->has_dict ? h->dict.c_str() : nullptr) == 0 &&
            cmx_pipeline_enable_paq8(la->pipe) == 0;
  if (!ok) { free_lookahead(h); return 1; }   // the failing stage has set the error
  const size_t T = 8 * kLaChunk;
  for (size_t i = 0; ok && i < kLaSlots; ++i) {
    la->d_layer0[i] = (float*)cmx_device_alloc(h->device, T * CMX_N_INPUTS * sizeof(float));
    la->d_p[i] = (float*)cmx_device_alloc(h->device, T * sizeof(float));
    ok = la->d_layer0[i] && la->d_p[i];
  }
  la->h_p = ok ? (float*)cmx_host_alloc(T * sizeof(float)) : nullptr;
  if (!ok || !la->h_p) { cmx_set_err("cmx_stage_input: buffer allocation failed"); free_lookahead(h); return 1; }
  // Predictor::Pretrain's bytes (predictor.cpp:471-487), collected since cmx_create, in one batch through the stages
  if (!h->pre.empty() && cmx_pipeline_pretrain(la->pipe, h->pre.data(), h->pre.size())) { free_lookahead(h); return 1; }
  h->pre.clear();
  h->pre.shrink_to_fit();
  h->mode = 2;
  h->started = true;
  return 0;
}

// Submit staged chunks while a slot is free: whole chunks as they become available, the ragged tail once the caller
// has marked the end of its input (or a predict needs it, see la_predict).
int la_top_up(cmx_engine* h, bool force_tail) {
  LookAhead* la = h->la;
  while (la->n_sub - la->n_done < kLaSlots && la->submitted < la->staged) {
    size_t n = la->staged - la->submitted;
    if (n > kLaChunk) n = kLaChunk;
    else if (n < kLaChunk && !la->ended && !force_tail) break;
    const size_t k = la->n_sub % kLaSlots;
    if (cmx_pipeline_submit(la->pipe, la->data.data() + (la->submitted - la->base), n, la->d_layer0[k], la->d_p[k])) return 1;
    la->ring[k].off = la->submitted;
    la->ring[k].n = n;
    la->submitted += n;
    la->n_sub++;
    force_tail = false;
  }
  return 0;
}

float la_predict(cmx_engine* h) {
  LookAhead* la = h->la;
  if (la->n_done == la->n_sub) {   // nothing in flight: a staged tail shorter than a chunk goes now
    if (la_top_up(h, true)) return -1.0f;
    if (la->n_done == la->n_sub) {
      cmx_set_err("cmx_predict: look-ahead mode, but no staged input is left (every bit to be coded must be staged first: "
                  "cmx_stage_input)");
      return -1.0f;
    }
  }
  if (!la->loaded) {
    if (cmx_pipeline_fetch(la->pipe, la->n_done, la->h_p)) return -1.0f;   // this chunk only: the seven behind it stay in flight
    la->loaded = true;
    la->bit = 0;
  }
  return la->h_p[la->bit];
}

int la_perceive(cmx_engine* h, int bit) {
  LookAhead* la = h->la;
  const LookAhead::Chunk& c = la->ring[la->n_done % kLaSlots];
  const uint8_t B = la->data[c.off + (la->bit >> 3) - la->base];
  if ((int)((B >> (7 - (la->bit & 7))) & 1) != (bit ? 1 : 0)) {
    cmx_set_err("cmx_perceive: the coded bit differs from the staged input (look-ahead mode: the device stages have already "
                "learnt the staged bit); the handle's state is void");
    h->failed = true;
    return 1;
  }
  if (++la->bit == 8 * c.n) {   // chunk used up: its slot and its bytes are free
    la->n_done++;
    la->loaded = false;
    la->bit = 0;
    const size_t keep_from = la->n_done < la->n_sub ? la->ring[la->n_done % kLaSlots].off : la->submitted;
    if (keep_from - la->base >= (1u << 20)) {
      la->data.erase(la->data.begin(), la->data.begin() + (keep_from - la->base));
      la->base = keep_from;
    }
    if (la_top_up(h, false)) return 1;
  }
  return 0;
}

}  // namespace

extern "C" {

void cmx_destroy(cmx_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->late) (void)cmx_pipeline_late_stop(h->late);   // FIRST: a decoder's stage kernels wait for bits that will not come -- without the abort word a
                                                        // device-wide synchronisation waits for their wall-clock time-outs (30 s per chunk in flight)
  (void)hipDeviceSynchronize();
  free_lookahead(h);
  free_perbit(h);
  const double t0 = cmx_timing_now();
  if (cmx_timing_on()) fprintf(stderr, "[cmx timing] cmx_destroy at                %.3f s (process clock); %llu bits: %.3f s inside cmx_predict, %.3f s inside cmx_perceive, %.3f s in the caller between them\n", t0,
                               h->n_calls, h->t_predict, h->t_perceive, h->t_outside);
  if (h->late) cmx_pipeline_destroy(h->late);   // unwinds the kernels of the chunk in progress first
  CMX_TIMING("pipeline_destroy", t0);
  delete h;
}

cmx_t* cmx_create(const uint8_t vocab[256], const char* dict_path, int device) {
  if (!vocab) { cmx_set_err("cmx_create: null vocab"); return nullptr; }
  int ndev = 0;   // no CPU fallback: without a HIP device there is no handle
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev || hipSetDevice(device) != hipSuccess) {
    cmx_set_err("cmx_create: no HIP device " + std::to_string(device) + " (the engine runs on an MI355X; there is no CPU path)");
    return nullptr;
  }
  cmx_engine* h = new cmx_engine();
  h->device = device;
  memcpy(h->vocab, vocab, 256);
  if (dict_path) { h->dict = dict_path; h->has_dict = true; }
  return h;
}

int cmx_set_model_outputs(cmx_t* h, const float* cols) {
  if (!h || !cols) { cmx_set_err("cmx_set_model_outputs: bad argument"); return 1; }
  if (h->mode == 2 || h->mode == 3) { cmx_set_err("cmx_set_model_outputs: this handle takes no columns from the caller (fxcm and paq8 are device stages)"); return 1; }
  if (h->predicted) { cmx_set_err("cmx_set_model_outputs: between predict() and perceive()"); return 1; }
  if (refused(h, "cmx_set_model_outputs") || ensure_perbit(h, "cmx_set_model_outputs")) return 1;
  memcpy(h->pin->staged, cols, sizeof h->pin->staged);  // the previous bit's upload finished before its p came back
  h->have_staged = true;
  return 0;
}

float cmx_predict(cmx_t* h) {
  const char* where = "cmx_predict";
  const float fail = -1.0f;
  if (!h) { cmx_set_err("cmx_predict: null handle"); return fail; }
  if (refused(h, where)) return fail;
  if (h->predicted) { cmx_set_err("cmx_predict: called twice without perceive()"); return fail; }
  if (h->mode == 2) {
    Txn txn(h);
    const float p = la_predict(h);
    if (p < 0) return fail;
    h->predicted = true;
    txn.ok = true;
    return p;
  }
  if (h->mode == 3 || (h->mode == 0 && !h->have_staged)) {   // a decoder: nothing staged, no columns handed in -- the whole engine, bit by bit
    const double tq = cmx_timing_on() ? cmx_timing_now() : 0;
    if (h->mode != 3) E_HIP(hipSetDevice(h->device));
    Txn txn(h);
    if (ensure_late(h, where)) return fail;
    const float p = cmx_pipeline_late_predict(h->late);
    if (p < 0) return fail;
    h->predicted = true;
    txn.ok = true;
    if (cmx_timing_on()) { h->t_predict += cmx_timing_now() - tq; h->n_calls++; if (h->t_last) h->t_outside += tq - h->t_last; }
    return p;
  }
  if (!h->have_staged) {
    cmx_set_err("cmx_predict: this handle takes the fxcm / paq8 columns from the caller (cmx_set_model_outputs was used): hand in the "
                "columns of this bit as well");
    return fail;
  }
  E_HIP(hipSetDevice(h->device));
  if (h->pre_j) { cmx_set_err("cmx_predict: Pretrain() stopped inside a byte"); return fail; }
  Txn txn(h);
  if (flush_pretrain(h)) return fail;
  h->started = true;
  const int j = h->j;
  float* const row = h->d_rows + (size_t)j * CMX_N_INPUTS;
  const uint8_t* db = send_byte(h, (uint8_t)(h->partial << (8 - j)));  // coded bits on top, zeros below
  if (!db) { cmx_set_err("cmx_predict: upload failed"); return fail; }
  if (cmx_ctxmodels_peek(h->ctx, db, j, h->d_rows, CMX_N_INPUTS, h->d_sel, h->st)) return fail;
  if (cmx_bytemodel_bit_run(h->device, h->d_ppmd, db, j, h->d_rows + 2076, CMX_N_INPUTS, nullptr, nullptr, h->st))
    return fail;
  // column 2077 of this row was written when the LSTM hint of this bit position was formed (enqueue_hint)
  E_HIP(hipMemcpyAsync(row + 3, h->pin->staged, sizeof h->pin->staged, hipMemcpyHostToDevice, h->st));
  if (cmx_mixnet_predict_async(h->mix, row, h->d_sel + (size_t)j * CMX_N_MIXERS, h->d_p, h->st)) return fail;
  E_HIP(hipMemcpyAsync(&h->pin->p, h->d_p, 4, hipMemcpyDeviceToHost, h->st));
  E_HIP(hipStreamSynchronize(h->st));
  if (lstm_timed_out(h, where)) return fail;
  h->row_j = j;
  h->predicted = true;
  h->have_staged = false;
  txn.ok = true;
  return h->pin->p;
}

int cmx_perceive(cmx_t* h, int bit) {
  const char* where = "cmx_perceive";
  const int fail = 1;
  if (!h) { cmx_set_err("cmx_perceive: null handle"); return 1; }
  if (refused(h, where)) return 1;
  if (!h->predicted) { cmx_set_err("cmx_perceive: no pending predict()"); return 1; }
  if (h->mode == 2) {
    Txn txn(h);
    if (la_perceive(h, bit)) return 1;
    h->predicted = false;
    txn.ok = true;
    return 0;
  }
  if (h->mode == 3) {
    const double tq = cmx_timing_on() ? cmx_timing_now() : 0;
    Txn txn(h);
    if (cmx_pipeline_late_perceive(h->late, bit)) return 1;
    h->predicted = false;
    txn.ok = true;
    if (cmx_timing_on()) { h->t_last = cmx_timing_now(); h->t_perceive += h->t_last - tq; }
    return 0;
  }
  E_HIP(hipSetDevice(h->device));
  Txn txn(h);
  if (cmx_mixnet_perceive_async(h->mix, bit, h->st)) return 1;
  h->predicted = false;
  h->partial = (h->partial << 1) | (bit ? 1u : 0u);
  if (++h->j == 8) {  // byte boundary (predictor.cpp:439-461)
    const uint8_t B = (uint8_t)h->partial;
    const uint8_t* db = send_byte(h, B);
    if (!db) { cmx_set_err("cmx_perceive: upload failed"); return 1; }
    if (cmx_ctxmodels_run(h->ctx, db, 1, h->d_scratch, CMX_N_INPUTS, h->d_scratch_sel, h->st)) return 1;
    if (cmx_ppmd_run(h->ppmd, &B, 1, h->pin->ppmd)) return 1;  // host stage, while the device commits the byte
    E_HIP(hipMemcpyAsync(h->d_ppmd, h->pin->ppmd, sizeof h->pin->ppmd, hipMemcpyHostToDevice, h->st));
    if (cmx_lstm_run(h->lstm, h->d_ppmd, db, 1, h->d_lstm_next, nullptr, 0, nullptr, h->st)) return 1;
    E_HIP(hipMemcpyAsync(&h->pin->lstm_fail, cmx_lstm_fail_flag(h->lstm), 4, hipMemcpyDeviceToHost, h->st));
    float* t = h->d_lstm; h->d_lstm = h->d_lstm_next; h->d_lstm_next = t;
    h->j = 0;
    h->partial = 0;
  }
  const int rc = enqueue_hint(h);  // nothing waits here: cmx_get_lstm_hint() / the next cmx_predict() synchronise
  txn.ok = rc == 0;
  return rc;
}

int cmx_get_lstm_hint(cmx_t* h, int* lstmpr, int* lstmex) {
  if (!h || !lstmpr || !lstmex) { cmx_set_err("cmx_get_lstm_hint: bad argument"); return 1; }
  if (h->mode == 2 || h->mode == 3) { cmx_set_err("cmx_get_lstm_hint: this handle keeps the LSTM hints on the device (the fxcm stage reads them there)"); return 1; }
  if (refused(h, "cmx_get_lstm_hint") || ensure_perbit(h, "cmx_get_lstm_hint")) return 1;
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  if (finish_hint(h)) return 1;
  *lstmpr = h->lstmpr;
  *lstmex = h->lstmex;
  return 0;
}

const float* cmx_debug_last_row(cmx_t* h) {
  if (!h || h->mode != 1 || hipSetDevice(h->device) != hipSuccess) return nullptr;
  if (hipMemcpyAsync(h->pin->row, h->d_rows + (size_t)h->row_j * CMX_N_INPUTS, sizeof h->pin->row, hipMemcpyDeviceToHost,
                     h->st) != hipSuccess || hipStreamSynchronize(h->st) != hipSuccess) return nullptr;
  return h->pin->row;
}

// Bits are collected into bytes; which stages train on them is decided with the handle's mode: look-ahead mode hands the
// whole batch to cmx_pipeline_pretrain (cmx_stage_input), the per-bit mode trains the context stage 64 KB at a time.
int cmx_pretrain(cmx_t* h, int bit) {
  if (!h) { cmx_set_err("cmx_pretrain: null handle"); return 1; }
  if (h->started) { cmx_set_err("cmx_pretrain: only before the first predict() (preprocessor.cpp:37-69)"); return 1; }
  h->pre_partial = (h->pre_partial << 1) | (bit ? 1u : 0u);
  if (++h->pre_j == 8) {
    h->pre.push_back((uint8_t)h->pre_partial);
    h->pre_j = 0;
    h->pre_partial = 0;
    if (h->mode == 1 && h->pre.size() >= (1u << 16)) {   // (modes 2 and 3 take the whole batch when they are built)
      if (refused(h, "cmx_pretrain")) return 1;
      if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
      Txn txn(h);
      const int rc = flush_pretrain(h);
      txn.ok = rc == 0;
      return rc;
    }
  }
  return 0;
}

// Look-ahead: the next n bytes the caller is going to code, in order. May be called repeatedly (the
// bytes are appended); n == 0 marks the end of the input, so that the last, ragged chunk is submitted without waiting
// for more. The first call decides the handle's mode and must come before the first cmx_predict().
int cmx_stage_input(cmx_t* h, const uint8_t* bytes, size_t n) {
  if (!h || (n && !bytes)) { cmx_set_err("cmx_stage_input: bad argument"); return 1; }
  if (refused(h, "cmx_stage_input")) return 1;
  if (h->predicted) { cmx_set_err("cmx_stage_input: between predict() and perceive()"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  if (ensure_lookahead(h)) return 1;
  LookAhead* la = h->la;
  if (la->ended && n) { cmx_set_err("cmx_stage_input: input after the end-of-input mark"); return 1; }
  Txn txn(h);
  if (n) {
    la->data.insert(la->data.end(), bytes, bytes + n);
    la->staged += n;
  } else {
    la->ended = true;
  }
  const int rc = la_top_up(h, false);
  txn.ok = rc == 0;
  return rc;
}

// Decoder::Decode (decoder.cpp:20-39) over a whole stream inside the library: `code` = the arithmetic code (what follows the container header, runner.cpp:34-84),
// nbytes = the stream's length from that header; out receives the bytes the predictor saw (the preprocessed stream). The handle decodes from its first bit
// (late-bit mode): every model family is a device stage, the host stages (paq8's front end per step, PPMd / fxcm's parser per byte) run on this
// thread between two bits -- which is also why the arithmetic decoder's dozen integer operations stay here: the front end needs every bit on the host before it
// can emit the next step's records, so a decoder on the device would not take the host off the per-bit path (DESIGN.md 4.10 has the measured split).
// The same loop as the reference's Decompress + decoder.cpp over cmx_predict / cmx_perceive, without an ABI crossing per bit.
int cmx_decode_stream(cmx_t* h, const uint8_t* code, size_t code_len, uint8_t* out, size_t nbytes) {
  if (!h || (!code && code_len) || (!out && nbytes)) { cmx_set_err("cmx_decode_stream: bad argument"); return 1; }
  cmx_decoder_t* d = cmx_decoder_create(code, code_len);
  if (!d) { cmx_set_err("cmx_decode_stream: decoder construction failed"); return 1; }
  int rc = 0;
  for (size_t i = 0; i < nbytes && !rc; ++i) {
    unsigned byte = 0;
    for (int j = 0; j < 8; ++j) {
      const float p = cmx_predict(h);
      if (p < 0.0f) { rc = 1; break; }
      const int bit = cmx_decoder_decode(d, p);
      if (bit < 0 || cmx_perceive(h, bit)) { rc = 1; break; }
      byte = (byte << 1) | (unsigned)bit;
    }
    out[i] = (uint8_t)byte;
  }
  cmx_decoder_destroy(d);
  return rc;
}

// 0 undecided, 1 per-bit stages, 2 look-ahead pipeline; in mode 2 *chunks_in_flight (may be NULL) = submitted - used up
int cmx_mode(cmx_t* h, int* chunks_in_flight) {
  if (!h) return -1;
  if (chunks_in_flight) *chunks_in_flight = h->la ? (int)(h->la->n_sub - h->la->n_done) : 0;
  return h->mode;
}

}  // extern "C"
