// pipeline_api.hip -- one input stream through every stage the engine has: the native host-side
// orchestration behind the Predictor surface (C ABI: cmx_pipeline_* in include/cmix_amd.h).
//
// What Predictor::Predict/Perceive do bit by bit (src/predictor.cpp:361-469) is done here a chunk of
// already-known bytes at a time (compression look-ahead):
//
//   host:   PPMd byte model (ppmd_host.cpp) over the chunk            predictor.cpp:447-449
//   s_lstm: H2D of its distributions; LSTM byte mixer -> column 2077   predictor.cpp:378-387,450-467
//   s_ctx:  contexts + 54 small models -> columns 0-2, 2025-2075, 47 selectors; PPMd bits -> column 2076
//                                                                       predictor.cpp:362-377,421-446
//   s_mix:  (after both) final mixing network + SSE -> p per bit       predictor.cpp:388-418,432-437
//
// The three device stages of a chunk run on their own HIP streams and the stages of consecutive chunks
// overlap (up to four chunks in flight). Columns 3..2024 (fxcm, paq8) have no stage yet: the caller supplies
// them inside d_layer0. No CPU fallback exists for any device stage.
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include <string>
#include <thread>

#include "../../include/cmix_amd.h"
#include "cmx_late.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip
extern "C" int cmx_make_stream(hipStream_t* st, int which);   // cmx_api.hip: a stage's kernel stream (compute-unit mask when the mixing network owns an XCD)
static double late_now();   // ms on the steady clock (defined with the late-bit pipeline below)

// ---- construction ahead of time ------------------------------------------------------------------------------
// Most of an engine's construction does not depend on the input: the mixing network's 2.8 GB of weight rows and SSE tables, the
// paq8 stage's 9 GB and the fxcm stage's 4.4 GB of tables are allocated and initialised the same way for every stream (only the
// context stage, the LSTM and PPMd are sized by the vocabulary, which the preprocessed stream defines). cmx_prewarm() builds those
// three stages on a thread of its own while the caller still preprocesses its input (runner.cpp:166-186: type detection, dictionary
// transform, temp file) and scans the vocabulary; cmx_pipeline_create / _enable_fxcm / _enable_paq8 then adopt them.
namespace {
struct Prewarm {
  std::thread th;
  int device = 0;
  bool want_fxcm = false, has_dict = false;
  std::string dict;
  cmx_mixnet_t* mix = nullptr;
  cmx_fxcm_t* fx = nullptr;
  cmx_p8stage_t* p8 = nullptr;
  ~Prewarm() {
    // Never adopted (or only in part), and the process is ending (the only caller is the static destructor below): wait for the builder
    // thread -- it must not be inside the HIP runtime while the process tears down -- and leave what it built to the operating system.
    // Destroying the stages here (hipFree, hipStreamDestroy from a static destructor of a shared library, next to the HIP runtime's own
    // teardown) corrupted the heap of `cmix_dropin -d` on an empty file, the one program run that never builds an engine.
    if (th.joinable()) th.join();
  }
};
std::mutex g_pw_mu;
Prewarm* g_pw = nullptr;
struct PrewarmAtExit { ~PrewarmAtExit() { std::lock_guard<std::mutex> l(g_pw_mu); delete g_pw; g_pw = nullptr; } } g_pw_at_exit;
// exit(): handlers run in reverse order of registration, and the HIP runtime registered its own when cmx_prewarm() made the first HIP call --
// this one is registered right after it, so it runs BEFORE the runtime goes away and waits for the builder thread (a thread inside
// hipMalloc while the runtime is torn down under it crashed `cmix_dropin -d` on an empty file, which exits 0.1 s after it started)
void prewarm_join_at_exit() {
  std::lock_guard<std::mutex> l(g_pw_mu);
  if (getenv("CMX_TIMING")) fprintf(stderr, "[cmx timing] exit handler at               %.3f s\n", std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count());
  if (g_pw && g_pw->th.joinable()) g_pw->th.join();
}
// the prewarmed set for `device`, its thread joined; nullptr if there is none
Prewarm* prewarmed(int device) {
  std::lock_guard<std::mutex> l(g_pw_mu);
  if (!g_pw || g_pw->device != device) return nullptr;
  if (g_pw->th.joinable()) g_pw->th.join();
  return g_pw;
}
}  // namespace

extern "C" int cmx_prewarm(int device, const char* dictionary_path, int with_fxcm) {
  std::lock_guard<std::mutex> l(g_pw_mu);
  if (g_pw) { cmx_set_err("cmx_prewarm: already called in this process"); return 1; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) { cmx_set_err("cmx_prewarm: no HIP device " + std::to_string(device)); return 1; }
  Prewarm* p = new Prewarm();
  p->device = device; p->want_fxcm = with_fxcm != 0;
  if (dictionary_path) { p->dict = dictionary_path; p->has_dict = true; }
  const bool timing = getenv("CMX_TIMING") != nullptr;
  if (timing) fprintf(stderr, "[cmx timing] cmx_prewarm called at        %.3f s (process clock)\n", late_now() / 1e3);
  p->th = std::thread([p, timing]() {   // a stage that fails here is simply built (and its error reported) by the normal path later
    (void)hipSetDevice(p->device);
    p->mix = cmx_mixnet_create(p->device);
    if (timing) fprintf(stderr, "[cmx timing] prewarm: mixnet built at       %.3f s\n", late_now() / 1e3);
    p->p8 = cmx_p8stage_create(p->device);
    if (timing) fprintf(stderr, "[cmx timing] prewarm: paq8 stage built at   %.3f s\n", late_now() / 1e3);
    if (p->want_fxcm) p->fx = cmx_fxcm_create(p->has_dict ? p->dict.c_str() : nullptr, p->device);
  });
  g_pw = p;
  static const int registered = atexit(prewarm_join_at_exit);
  (void)registered;
  return 0;
}

namespace {
constexpr int kSlots = CMX_PIPELINE_SLOTS;  // chunks in flight: a chunk's latency (paq8 maps -> family -> paq8 mixer -> mixing network) is ~3 chunk periods
struct Slot {
  uint8_t* d_bytes = nullptr;
  uint8_t* d_bits = nullptr;
  uint32_t* d_sel = nullptr;
  float* d_ppmd = nullptr;      // [max+1][256]: row 0 = distribution going into the chunk
  float* d_lstm_out = nullptr;  // [max][256]
  uint8_t* h_bytes = nullptr;   // pinned: bytes, then the 8n unpacked bits
  float* h_ppmd = nullptr;      // pinned [max+1][256]
  float* d_hint = nullptr;      // [8 max + 1] f32 LSTM bit predictions, then [8 max + 1] i32 `ex` (look-ahead hybrid)
  float* h_hint = nullptr;      // pinned mirror
  unsigned* h_fail = nullptr;   // pinned [4]: the LSTM's / fxcm's / mixing network's sticky hand-off flags as they stood behind this chunk's kernels
  size_t n = 0;                 // bytes of the chunk in this slot
  float* d_layer0 = nullptr;    // the caller's layer-0 rows of that chunk
  float* d_p = nullptr;         // the caller's p[] buffer of that chunk (cmx_pipeline_fetch)
  hipEvent_t ev_in = nullptr, ev_ctx0 = nullptr, ev_ctx1 = nullptr, ev_lstm0 = nullptr, ev_lstm1 = nullptr,
             ev_mix0 = nullptr, ev_mix1 = nullptr, ev_cols = nullptr;
  int16_t* d_fx_pr = nullptr;   // [8 max] fxcm stage (opt-in): lstmpr per update
  uint8_t* d_fx_ex = nullptr;   // [8 max] lstmex per update
  hipEvent_t ev_fxin = nullptr, ev_fx0 = nullptr, ev_fx1 = nullptr;
  hipEvent_t ev_p80 = nullptr, ev_p81 = nullptr;   // paq8 stage (opt-in)
  bool used = false;
  bool untimed = false;  // the chunk in this slot has not been added to the stage totals yet
};
}  // namespace

namespace {
// ---- the decoder's form (cmx_late.h): chunks of kLateChunk bytes, three sets of host-coherent buffers --------------------------
// A chunk's kernels are launched while its predecessor is still being decoded (they queue behind it on every stage's stream), and a
// chunk's first byte still reads its predecessor's last distributions: three sets, so that the one being launched is never one that
// is still read.
constexpr size_t kLateChunk = 512;
struct LateSet {
  // host-coherent pinned memory: what the decoder thread writes / reads while the kernels run
  CmxLateBox* box = nullptr;
  float* ppmd = nullptr;         // [n + 1][256]  PPMd's distributions; row 0 = going into the chunk
  uint8_t* bytes = nullptr;      // [n]
  float* d_ppmd = nullptr;       // device mirror of ppmd (the relay wave copies row b over with byte b's first step)
  uint8_t* d_bits = nullptr;     // device mirror of the bits: [8] prefix (byte 7 = the bit before the chunk), then [8 n]
  cmx_late_relay_t* d_relay = nullptr; int nrelay = 0;   // what the relay wave copies, per step / per byte (device copy of the table)
  cmx_late_relay_t relay[CMX_LATE_RELAY_MAX];            // the same table on the host: in push mode this thread does the copies
  // uncached device memory: what one stage kernel writes and another reads while both run
  uint32_t* cnt = nullptr;       // [LC_N][CMX_LATE_CNT_STRIDE] row counters (base | rows, never cleared)
  float* layer0 = nullptr;       // [8 n][2078]
  uint32_t* sel = nullptr;       // [8 n][47]
  float* brk = nullptr;          // [n][256]   Bracket model's distribution after each byte (context stage)
  float* lstm = nullptr;         // [n][256]   the LSTM byte mixer's
  int16_t* hint_pr = nullptr; uint8_t* hint_ex = nullptr;   // [8 n] fxcm's LSTM hints per update
  CmxLate lt;                    // what the chunk's kernels are launched with
};
struct Late {
  LateSet set[3];
  bool active = false, failed = false, predicted = false;
  uint64_t launched = 0;   // chunks whose kernels are enqueued
  bool touched = false;    // a launch was attempted (some of a chunk's kernels may be running even if the attempt failed half-way)
  uint64_t cur = 0;        // chunk being decoded
  size_t t = 0;            // its next bit
  unsigned partial = 0;
  hipStream_t s_bm = nullptr;
  const float* lstm0 = nullptr;
  double ms[6] = {0, 0, 0, 0, 0, 0};
  uint64_t bits = 0;
  uint32_t timeout_s = 0;  // CMX_LATE_TIMEOUT_S: seconds an in-launch wait of a decoder's kernels may last without progress (0: the 30 s default)
  bool push = false;       // the decoder thread stores steps into device memory itself (large BAR; CMX_LATE_PULL=1: rounds 4 / 5's relay wave instead)
  int lstm_covered = 0;    // bytes of the chunk in progress that the LSTM's last forward launch still covers (cmx_lstm_run_late)
  bool lstm_per_byte = false;   // CMX_LATE_LSTM_PER_BYTE=1: rounds 4 / 5's one launch per byte (A/B)
  float* dbg_row = nullptr; uint32_t* dbg_sel = nullptr;   // pinned: cmx_pipeline_late_debug_row
  uint64_t mix_chunk0 = 0;                                  // the mixing-network handle's launch count when the stream started (debug hook)
};
}  // namespace

struct cmx_pipeline {
  int device = 0;
  Late* late = nullptr;
  bool lstm_tolerance = false;   // cmx_pipeline_set_tolerance switched the LSTM's weight update to its MFMA form
  size_t max_chunk = 0;
  cmx_ppmd_t* ppmd = nullptr;
  cmx_ctxmodels_t* ctx = nullptr;
  cmx_lstm_t* lstm = nullptr;
  cmx_mixnet_t* mix = nullptr;
  hipStream_t s_ctx = nullptr, s_lstm = nullptr, s_mix = nullptr;
  hipStream_t s_up = nullptr;   // every host-to-device copy of the stream's chunks (inputs, fxcm records, paq8 records, decay schedule): never a kernel in front of a copy
  cmx_fxcm_t* fxcm = nullptr;    // device fxcm stage (cmx_pipeline_enable_fxcm); NULL: the caller supplies columns 3..433
  hipStream_t s_fx = nullptr;
  float* d_fx_scratch = nullptr; // pretraining writes its (discarded) rows here
  double fx_ms = 0;
  cmx_p8stage_t* p8 = nullptr;   // device paq8 stage (cmx_pipeline_enable_paq8); NULL: the caller supplies columns 434..2024
  hipStream_t s_p8 = nullptr;
  float* d_p8_scratch = nullptr; // pretraining writes its (discarded) rows here
  double p8_ms = 0;
  int compact = 0;       // CMX_PIPELINE_STREAMS: 0 = one stream per stage / role, 2 and 1 = throughput modes with fewer hardware queues per engine
  bool failed = false;   // sticky: a chunk failed after its stages had begun to be enqueued (the stream's state is void)
  int wgs = 0;           // workgroups this engine's persistent kernels keep resident (wg_claim)
  double host_ms[6] = {0, 0, 0, 0, 0, 0};   // calling thread, since the last reset: slot wait, PPMd, ctx + LSTM enqueue, fxcm (parser + enqueue), paq8 (front end + enqueue), mixing network enqueue
  Slot slot[kSlots];
  uint64_t chunks = 0;    // chunks begun
  uint64_t hinted = 0;    // chunks whose LSTM hints were handed out (<= chunks)
  uint64_t finished = 0;  // chunks whose mixing network was enqueued (<= chunks)
  float last_dist[256];
  float stage_ms[3] = {0, 0, 0};
  int last_slot = -1;
  double tot_ms[3] = {0, 0, 0};  // HIP-event time of every finished chunk's stages since the last reset
  uint64_t tot_chunks = 0;
  float* dbg_mix = nullptr;      // diagnosis (cmx_pipeline_debug_mix_out): the 47 mixer outputs of every bit, [bit][47], as far as dbg_mix_cap bits
  uint64_t dbg_mix_cap = 0, dbg_mix_bits = 0;
};

namespace {
void collect(cmx_pipeline* h, Slot& s) {  // the slot's chunk has finished (its ev_mix1 was waited for)
  if (!s.untimed) return;
  float ms[3] = {0, 0, 0};
  (void)hipEventElapsedTime(&ms[0], s.ev_ctx0, s.ev_ctx1);
  (void)hipEventElapsedTime(&ms[1], s.ev_lstm0, s.ev_lstm1);
  (void)hipEventElapsedTime(&ms[2], s.ev_mix0, s.ev_mix1);
  for (int i = 0; i < 3; ++i) h->tot_ms[i] += ms[i];
  if (h->fxcm) { float f = 0; (void)hipEventElapsedTime(&f, s.ev_fx0, s.ev_fx1); h->fx_ms += f; }
  if (h->p8) { float f = 0; (void)hipEventElapsedTime(&f, s.ev_p80, s.ev_p81); h->p8_ms += f; }
  h->tot_chunks++;
  s.untimed = false;
}
}  // namespace

__global__ void cmx_fxcm_hints_kernel(const float* layer0, long stride, const float* p_after, const int* ex, int T, int16_t* lstmpr, uint8_t* lstmex) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= T) return;
  const float p = t + 1 < T ? layer0[(long)(t + 1) * stride + 2077] : *p_after;
  const float prod = 4094.0f * p;                 // Discretize (predictor.cpp:180-182): the product is rounded to float, then 1 is added
  lstmpr[t] = (int16_t)(unsigned)(1.0f + prod);
  lstmex[t] = (uint8_t)ex[t + 1];
}

extern "C" {

// Co-residency: the stage kernels of a stream run for a whole chunk and hand values to each other inside the launch (bounded waits), so
// all of their workgroups must be resident at once -- mixing network 27 (1 with CMX_MIXNET_SPEC=0), LSTM 52, contexts 1, fxcm 4, paq8 10 --
// and most of them own a compute unit (130-160 KB of LDS). An engine that would take the device past its compute-unit count is refused at
// construction (with the reason) instead of timing out in the middle of a stream. Process-wide per device.
static std::mutex g_wg_mu;
static int g_wg_used[64];
static bool wg_claim(cmx_pipeline* h, int n, const char* what) {
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device) != hipSuccess || cus <= 0) return true;   // unknown: do not guess
  std::lock_guard<std::mutex> g(g_wg_mu);
  int& used = g_wg_used[h->device & 63];
  if (used + n > cus) {
    cmx_set_err(std::string(what) + ": " + std::to_string(used) + " workgroups of persistent stage kernels are already resident on device " + std::to_string(h->device) + " and this stage needs " +
                std::to_string(n) + " more, the device has " + std::to_string(cus) + " compute units (fewer streams per GPU, or CMX_MIXNET_SPEC=0 for the one-workgroup mixing network)");
    return false;
  }
  used += n; h->wgs += n;
  return true;
}
static void wg_release(cmx_pipeline* h) {
  std::lock_guard<std::mutex> g(g_wg_mu);
  g_wg_used[h->device & 63] -= h->wgs;
  h->wgs = 0;
}

void cmx_pipeline_destroy(cmx_pipeline_t* h) {
  if (!h) return;
  if (h->late) {
    (void)cmx_pipeline_late_stop(h);
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    for (LateSet& q : h->late->set) {
      cmx_late_free(q.box); cmx_late_free(q.ppmd); cmx_late_free(q.bytes);
      cmx_late_free_dev(q.d_ppmd); cmx_late_free_dev(q.d_bits); cmx_late_free_dev(q.d_relay);
      cmx_late_free_dev(q.cnt); cmx_late_free_dev(q.layer0); cmx_late_free_dev(q.sel); cmx_late_free_dev(q.brk); cmx_late_free_dev(q.lstm);
      cmx_late_free_dev(q.hint_pr); cmx_late_free_dev(q.hint_ex);
    }
    if (h->late->s_bm) (void)hipStreamDestroy(h->late->s_bm);
    if (h->late->dbg_row) (void)hipHostFree(h->late->dbg_row);
    if (h->late->dbg_sel) (void)hipHostFree(h->late->dbg_sel);
    delete h->late;
    h->late = nullptr;
  }
  wg_release(h);
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (Slot& s : h->slot) {
    if (s.d_bytes) (void)hipFree(s.d_bytes);
    if (s.d_bits) (void)hipFree(s.d_bits);
    if (s.d_sel) (void)hipFree(s.d_sel);
    if (s.d_ppmd) (void)hipFree(s.d_ppmd);
    if (s.d_lstm_out) (void)hipFree(s.d_lstm_out);
    if (s.h_bytes) (void)hipHostFree(s.h_bytes);
    if (s.h_ppmd) (void)hipHostFree(s.h_ppmd);
    if (s.d_hint) (void)hipFree(s.d_hint);
    if (s.h_hint) (void)hipHostFree(s.h_hint);
    if (s.h_fail) (void)hipHostFree(s.h_fail);
    if (s.d_fx_pr) (void)hipFree(s.d_fx_pr);
    if (s.d_fx_ex) (void)hipFree(s.d_fx_ex);
    for (hipEvent_t e : {s.ev_fxin, s.ev_fx0, s.ev_fx1, s.ev_p80, s.ev_p81})
      if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {s.ev_in, s.ev_ctx0, s.ev_ctx1, s.ev_lstm0, s.ev_lstm1, s.ev_mix0, s.ev_mix1, s.ev_cols})
      if (e) (void)hipEventDestroy(e);
  }
  if (h->s_ctx && h->s_ctx != h->s_lstm) (void)hipStreamDestroy(h->s_ctx);
  if (h->s_lstm) (void)hipStreamDestroy(h->s_lstm);
  if (h->s_mix) (void)hipStreamDestroy(h->s_mix);
  if (h->s_up) (void)hipStreamDestroy(h->s_up);
  if (h->s_fx && h->s_fx != h->s_lstm) (void)hipStreamDestroy(h->s_fx);
  if (h->d_fx_scratch) (void)hipFree(h->d_fx_scratch);
  if (h->s_p8 && h->s_p8 != h->s_mix) (void)hipStreamDestroy(h->s_p8);
  if (h->d_p8_scratch) (void)hipFree(h->d_p8_scratch);
  cmx_p8stage_destroy(h->p8);
  cmx_fxcm_destroy(h->fxcm);
  cmx_mixnet_destroy(h->mix);
  cmx_lstm_destroy(h->lstm);
  cmx_ctxmodels_destroy(h->ctx);
  cmx_ppmd_destroy(h->ppmd);
  delete h;
}

cmx_pipeline_t* cmx_pipeline_create(const uint8_t vocab[256], int device, size_t max_chunk_bytes) {
  if (max_chunk_bytes == 0 || max_chunk_bytes > (1u << 24)) { cmx_set_err("cmx_pipeline_create: bad max_chunk_bytes"); return nullptr; }
  cmx_pipeline_t* h = new cmx_pipeline();
  h->device = device;
  h->max_chunk = max_chunk_bytes;
  {
    const char* sp = getenv("CMX_MIXNET_SPEC");
    if (cmx_device_count() > 0 && !wg_claim(h, ((sp && sp[0] == '0') ? 1 : 27) + 52 + 1, "cmx_pipeline_create")) { delete h; return nullptr; }
  }
  // every stage reports its own failure (no device, out of memory) through cmx_last_error()
  h->ctx = cmx_ctxmodels_create(vocab, device);
  h->lstm = h->ctx ? cmx_lstm_create(vocab, 31, device) : nullptr;  // 31 rand() draws precede the LSTM (indirect.cpp:10)
  h->ppmd = h->lstm ? cmx_ppmd_create(vocab) : nullptr;
  if (h->ppmd) {   // the vocabulary-independent mixing network may have been built ahead of time (cmx_prewarm), beside the three above
    if (Prewarm* pw = prewarmed(device)) { h->mix = pw->mix; pw->mix = nullptr; }
    if (!h->mix) h->mix = cmx_mixnet_create(device);
  }
  if (!h->mix) { cmx_pipeline_destroy(h); return nullptr; }
  bool ok = hipSetDevice(device) == hipSuccess;
  // CMX_PIPELINE_STREAMS=2 (throughput mode, many streams per GPU): the context stage shares the LSTM's HIP stream.
  // Every HIP stream is a hardware queue; past ~24 active queues the hardware scheduler starts time-slicing them
  // (profiles/r01_multiproc.txt: microsecond kernels then show 13 ms durations), so 8+ streams per GPU want 2 each.
  const char* ns = getenv("CMX_PIPELINE_STREAMS");
  const bool two = ns && (ns[0] == '2' || ns[0] == '1');   // 2: contexts on the LSTM's stream, paq8 on 4 streams; 1: fxcm there too, paq8 on 3
  h->compact = ns && ns[0] == '1' ? 1 : two ? 2 : 0;
  ok = ok && cmx_make_stream(&h->s_lstm, 0) == 0;
  if (two) h->s_ctx = h->s_lstm;
  else ok = ok && cmx_make_stream(&h->s_ctx, 0) == 0;
  ok = ok && cmx_make_stream(&h->s_mix, 1) == 0;
  ok = ok && hipStreamCreateWithFlags(&h->s_up, hipStreamNonBlocking) == hipSuccess;
  ok = ok && cmx_mixnet_set_upload_stream(h->mix, h->s_up) == 0;
  const size_t n = max_chunk_bytes;
  for (Slot& s : h->slot) {
    ok = ok && hipMalloc((void**)&s.d_bytes, n) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.d_bits, 8 * n) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.d_sel, 8 * n * CMX_N_MIXERS * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.d_ppmd, (n + 1) * 256 * 4) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.d_lstm_out, n * 256 * 4) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&s.h_bytes, 9 * n, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&s.h_ppmd, (n + 1) * 256 * 4, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.d_hint, 2 * (8 * n + 1) * 4) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&s.h_hint, 2 * (8 * n + 1) * 4, hipHostMallocDefault) == hipSuccess;
    ok = ok && hipHostMalloc((void**)&s.h_fail, 16, hipHostMallocDefault) == hipSuccess;
    if (ok) s.h_fail[0] = s.h_fail[1] = s.h_fail[2] = s.h_fail[3] = 0;
    for (hipEvent_t* e : {&s.ev_in, &s.ev_ctx0, &s.ev_ctx1, &s.ev_lstm0, &s.ev_lstm1, &s.ev_mix0, &s.ev_mix1, &s.ev_cols})
      ok = ok && hipEventCreate(e) == hipSuccess;
  }
  if (!ok) { cmx_set_err("cmx_pipeline_create: stream / buffer allocation failed"); cmx_pipeline_destroy(h); return nullptr; }
  for (int i = 0; i < 256; ++i) h->last_dist[i] = (float)(1.0 / 256);  // ByteModel constructor (byte-model.cpp:5-6)
  return h;
}

// ---- the fxcm stage on the device (opt-in) ----------------------------------------------------------------------
// After coding bit t the reference holds lstmpr = Discretize(p of bit t + 1) = 1 + 4094 * p in float arithmetic,
// truncated (predictor.cpp:180-182,463), and lstmex = the LSTM byte mixer's `ex` at that moment (:464); fxcm reads both
// in the Perceive of bit t (:465). p of bit t + 1 is column 2077 of row t + 1; the entry after the chunk's last bit
// comes from the distribution after the chunk's last byte (hint arrays, entry 8n).
int cmx_pipeline_enable_fxcm(cmx_pipeline_t* h, const char* dictionary_path) {
  if (!h) { cmx_set_err("cmx_pipeline_enable_fxcm: null handle"); return 1; }
  if (h->fxcm) return 0;
  if (h->chunks) { cmx_set_err("cmx_pipeline_enable_fxcm: only before the first chunk"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  if (!wg_claim(h, 4, "cmx_pipeline_enable_fxcm")) return 1;
  cmx_fxcm_t* fx = nullptr;
  if (Prewarm* pw = prewarmed(h->device))   // built ahead of time for the same dictionary?
    if (pw->fx && pw->has_dict == (dictionary_path != nullptr) && (!dictionary_path || pw->dict == dictionary_path)) { fx = pw->fx; pw->fx = nullptr; }
  if (!fx) fx = cmx_fxcm_create(dictionary_path, h->device);
  if (!fx) return 1;
  const size_t n = h->max_chunk;
  bool ok = true;
  if (h->compact == 1) h->s_fx = h->s_lstm;   // throughput mode: LSTM, contexts and fxcm take turns on one stream
  else ok = cmx_make_stream(&h->s_fx, 0) == 0;
  ok = ok && hipMalloc((void**)&h->d_fx_scratch, 8 * n * 434 * sizeof(float)) == hipSuccess;
  for (Slot& s : h->slot) {
    ok = ok && hipMalloc((void**)&s.d_fx_pr, 8 * n * 2) == hipSuccess;
    ok = ok && hipMalloc((void**)&s.d_fx_ex, 8 * n) == hipSuccess;
    for (hipEvent_t* e : {&s.ev_fxin, &s.ev_fx0, &s.ev_fx1}) ok = ok && hipEventCreate(e) == hipSuccess;
  }
  if (!ok) {   // leave the handle as it was: no half-enabled stage (a retry starts from scratch)
    for (Slot& s : h->slot) {
      if (s.d_fx_pr) (void)hipFree(s.d_fx_pr);
      if (s.d_fx_ex) (void)hipFree(s.d_fx_ex);
      s.d_fx_pr = nullptr; s.d_fx_ex = nullptr;
      for (hipEvent_t* e : {&s.ev_fxin, &s.ev_fx0, &s.ev_fx1}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    }
    if (h->s_fx && h->s_fx != h->s_lstm) (void)hipStreamDestroy(h->s_fx);
    if (h->d_fx_scratch) (void)hipFree(h->d_fx_scratch);
    h->s_fx = nullptr; h->d_fx_scratch = nullptr;
    cmx_fxcm_destroy(fx);
    cmx_set_err("cmx_pipeline_enable_fxcm: stream / buffer allocation failed");
    return 1;
  }
  (void)cmx_fxcm_set_upload_stream(fx, h->s_up);
  h->fxcm = fx;
  return 0;
}

// ---- the paq8 stage on the device (opt-in): layer-0 columns 434..2024 (include/cmix_amd.h section 2f) ------------
int cmx_pipeline_enable_paq8(cmx_pipeline_t* h) {
  if (!h) { cmx_set_err("cmx_pipeline_enable_paq8: null handle"); return 1; }
  if (h->p8) return 0;
  if (h->chunks) { cmx_set_err("cmx_pipeline_enable_paq8: only before the first chunk"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  if (!wg_claim(h, 10, "cmx_pipeline_enable_paq8")) return 1;
  cmx_p8stage_t* p8 = nullptr;
  if (Prewarm* pw = prewarmed(h->device)) { p8 = pw->p8; pw->p8 = nullptr; }
  if (!p8) p8 = cmx_p8stage_create(h->device);
  if (!p8) return 1;
  bool ok = true;
  if (h->compact) h->s_p8 = h->s_mix;   // throughput mode: the mixing network's stream itself waits for the stage's mixer
  else ok = cmx_make_stream(&h->s_p8, 0) == 0;
  ok = ok && hipMalloc((void**)&h->d_p8_scratch, 8 * h->max_chunk * 1591 * sizeof(float)) == hipSuccess;
  for (Slot& s : h->slot)
    for (hipEvent_t* e : {&s.ev_p80, &s.ev_p81}) ok = ok && hipEventCreate(e) == hipSuccess;
  if (!ok) {
    for (Slot& s : h->slot)
      for (hipEvent_t* e : {&s.ev_p80, &s.ev_p81}) { if (*e) (void)hipEventDestroy(*e); *e = nullptr; }
    if (h->s_p8 && h->s_p8 != h->s_mix) (void)hipStreamDestroy(h->s_p8);
    if (h->d_p8_scratch) (void)hipFree(h->d_p8_scratch);
    h->s_p8 = nullptr; h->d_p8_scratch = nullptr;
    cmx_p8stage_destroy(p8);
    cmx_set_err("cmx_pipeline_enable_paq8: stream / buffer allocation failed");
    return 1;
  }
  (void)cmx_p8stage_set_upload_stream(p8, h->s_up);
  h->p8 = p8;
  return 0;
}
// the mixing network's tolerance mode (cmx_mixnet_set_tolerance: NOT bit-exact, bench / measurement only); before the first chunk
int cmx_pipeline_set_tolerance(cmx_pipeline_t* h, int on) {
  if (!h) { cmx_set_err("cmx_pipeline_set_tolerance: null handle"); return 1; }
  if (h->chunks || h->late) { cmx_set_err("cmx_pipeline_set_tolerance: only before the first chunk"); return 1; }
  // the mixing network first (it can refuse: CMX_MIXNET_SPEC=0 / CMX_MIXNET_V1 have no tolerance form); the LSTM's switch (its weight-update contraction
  // on the matrix cores) follows and is rolled back with the network's if it fails, so that the two never disagree about the mode
  const int was = cmx_mixnet_mode(h->mix);
  if (cmx_mixnet_set_tolerance(h->mix, on)) return 1;
  if (cmx_lstm_set_tolerance(h->lstm, on)) { (void)cmx_mixnet_set_tolerance(h->mix, was == 1); return 1; }
  h->lstm_tolerance = on != 0;
  return 0;
}
// 0 strict, 1 tolerance: not strict as soon as EITHER the mixing network or the LSTM computes in its tolerance form
int cmx_pipeline_mixnet_mode(cmx_pipeline_t* h) { return !h ? -1 : (h->lstm_tolerance ? 1 : cmx_mixnet_mode(h->mix)); }
// ---- diagnostics of a long run (scripts/gpu_long_run.py): all of them synchronise the device ----
int cmx_pipeline_mixnet_rows(cmx_pipeline_t* h, uint32_t rows[47]) { return h ? cmx_mixnet_rows(h->mix, rows) : 1; }
int cmx_pipeline_spec_stats(cmx_pipeline_t* h, uint64_t out[5]) { return h ? cmx_mixnet_spec_stats(h->mix, out) : 1; }
int cmx_pipeline_paq8_profile(cmx_pipeline_t* h, unsigned long long out128[128]) { return h && h->p8 ? cmx_p8stage_profile(h->p8, out128) : 1; }
int cmx_pipeline_ppmd_arena(cmx_pipeline_t* h, uint64_t out3[3]) { return h ? cmx_ppmd_arena(h->ppmd, out3) : 1; }
int cmx_pipeline_paq8_enabled(cmx_pipeline_t* h) { return h && h->p8 ? 1 : 0; }
int cmx_pipeline_paq8_total_ms(cmx_pipeline_t* h, double* ms) { if (!h || !ms) return 1; *ms = h->p8_ms; return 0; }
// wall time of the CALLING thread inside cmx_pipeline_begin / _finish since the last reset of the stage totals, in ms:
// [0] waiting for a slot (the device is behind), [1] PPMd, [2] uploads + context stage + LSTM enqueue, [3] fxcm (text parser,
// staging wait, enqueue), [4] paq8 (front end, staging wait, enqueue), [5] mixing network enqueue
int cmx_pipeline_host_ms(cmx_pipeline_t* h, double ms[6]) { if (!h || !ms) return 1; for (int i = 0; i < 6; i++) ms[i] = h->host_ms[i]; return 0; }
int cmx_pipeline_paq8_role_ms(cmx_pipeline_t* h, double ms[7], uint64_t* chunks) { return h && h->p8 ? cmx_p8stage_role_ms(h->p8, ms, chunks, 0) : 1; }
int cmx_pipeline_fxcm_enabled(cmx_pipeline_t* h) { return h && h->fxcm ? 1 : 0; }
// HIP-event time of the fxcm kernel over the chunks counted by cmx_pipeline_stage_totals (same reset)
int cmx_pipeline_fxcm_total_ms(cmx_pipeline_t* h, double* ms) { if (!h || !ms) return 1; *ms = h->fx_ms; return 0; }

// ---- a chunk in two steps -------------------------------------------------------------------------------------
// begin: everything that does not need the fxcm/paq8 columns -- PPMd on this thread, upload, context stage, LSTM.
int cmx_pipeline_begin(cmx_pipeline_t* h, const uint8_t* bytes, size_t n, float* d_layer0) {
  if (!h) { cmx_set_err("cmx_pipeline_begin: null handle"); return 1; }
  if (!bytes || !d_layer0 || n == 0 || n > h->max_chunk) { cmx_set_err("cmx_pipeline_begin: bad argument"); return 1; }
  if (h->chunks - h->finished >= (uint64_t)kSlots) { cmx_set_err("cmx_pipeline_begin: too many chunks begun and not finished"); return 1; }
  if (h->failed) { cmx_set_err("cmx_pipeline_begin: an earlier chunk of this handle failed part-way; its state is void (destroy it)"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  struct Txn { cmx_pipeline* h; bool ok = false; ~Txn() { if (!ok) h->failed = true; } } txn{h};   // from here on a failure leaves stages half-enqueued
  Slot& s = h->slot[h->chunks % kSlots];
  auto now = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
  double t_mark = now();
  auto lap = [&](int k) { const double t = now(); h->host_ms[k] += t - t_mark; t_mark = t; };
  if (s.used && hipEventSynchronize(s.ev_mix1) != hipSuccess) {  // the chunk that used these buffers kSlots submits ago
    cmx_set_err("cmx_pipeline_begin: device error in an earlier chunk");
    return 1;
  }
  collect(h, s);
  lap(0);
  s.n = n;
  s.d_layer0 = d_layer0;
  // ---- host stage: PPMd runs ahead of the device on this thread ----
  memcpy(s.h_ppmd, h->last_dist, 256 * 4);
  if (cmx_ppmd_run(h->ppmd, bytes, n, s.h_ppmd + 256)) return 1;
  memcpy(h->last_dist, s.h_ppmd + n * 256, 256 * 4);
  lap(1);
  memcpy(s.h_bytes, bytes, n);
  uint8_t* hb = s.h_bytes + n;
  for (size_t i = 0; i < n; ++i)
    for (int j = 0; j < 8; ++j) hb[8 * i + j] = (bytes[i] >> (7 - j)) & 1;  // MSB first (runner.cpp:106-108)
  // ---- inputs ----
  // on the upload stream: behind the previous chunk's kernels in a stage's own stream the copies would wait in stream order, and
  // a waiting host-to-device copy holds up every later copy of the process (kernels of chunks further ahead start late)
  bool ok = hipMemcpyAsync(s.d_ppmd, s.h_ppmd, (n + 1) * 256 * 4, hipMemcpyHostToDevice, h->s_up) == hipSuccess;
  ok = ok && hipMemcpyAsync(s.d_bytes, s.h_bytes, n, hipMemcpyHostToDevice, h->s_up) == hipSuccess;
  ok = ok && hipMemcpyAsync(s.d_bits, hb, 8 * n, hipMemcpyHostToDevice, h->s_up) == hipSuccess;
  ok = ok && hipEventRecord(s.ev_in, h->s_up) == hipSuccess;
  ok = ok && hipStreamWaitEvent(h->s_lstm, s.ev_in, 0) == hipSuccess;
  if (h->s_ctx != h->s_lstm) ok = ok && hipStreamWaitEvent(h->s_ctx, s.ev_in, 0) == hipSuccess;
  if (!ok) { cmx_set_err("cmx_pipeline_begin: input upload failed"); return 1; }
  // ---- context / small-model stage + PPMd's bit predictions ----
  (void)hipEventRecord(s.ev_ctx0, h->s_ctx);
  if (cmx_ctxmodels_run(h->ctx, s.d_bytes, n, d_layer0, CMX_N_INPUTS, s.d_sel, h->s_ctx)) return 1;
  if (cmx_bytemodel_bits_run(h->device, s.d_ppmd, s.d_ppmd + 256, s.d_bytes, n, d_layer0 + 2076, CMX_N_INPUTS, nullptr,
                             h->s_ctx)) return 1;
  (void)hipEventRecord(s.ev_ctx1, h->s_ctx);
  // ---- LSTM byte mixer ----
  (void)hipEventRecord(s.ev_lstm0, h->s_lstm);
  if (cmx_lstm_run(h->lstm, s.d_ppmd + 256, s.d_bytes, n, s.d_lstm_out, d_layer0 + 2077, CMX_N_INPUTS,
                   (int*)(s.d_hint + (8 * h->max_chunk + 1)), h->s_lstm))
    return 1;
  // the stage's sticky time-out flag, in stream order behind its kernels: cmx_pipeline_wait looks at it per chunk
  (void)hipMemcpyAsync(&s.h_fail[0], cmx_lstm_fail_flag(h->lstm), 4, hipMemcpyDeviceToHost, h->s_lstm);
  (void)hipEventRecord(s.ev_lstm1, h->s_lstm);
  lap(2);
  if (h->fxcm) {  // ---- fxcm stage: hints from the LSTM's columns, then the parser (this thread) and the kernel on its own stream ----
    const size_t T = 8 * n;
    float* dp = s.d_hint;
    int* dx = (int*)(s.d_hint + (8 * h->max_chunk + 1));
    if (cmx_bytemodel_bit_run(h->device, s.d_lstm_out + (n - 1) * 256, s.d_bytes, 0, dp + T, 1, dx + T, nullptr, h->s_lstm)) return 1;
    hipLaunchKernelGGL(cmx_fxcm_hints_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), 0, h->s_lstm, d_layer0, (long)CMX_N_INPUTS, dp + T, dx, (int)T,
                       s.d_fx_pr, s.d_fx_ex);
    (void)hipEventRecord(s.ev_fxin, h->s_lstm);
    (void)hipStreamWaitEvent(h->s_fx, s.ev_fxin, 0);
    (void)hipEventRecord(s.ev_fx0, h->s_fx);
    if (cmx_fxcm_run(h->fxcm, bytes, s.d_bytes, n, s.d_fx_pr, s.d_fx_ex, d_layer0, CMX_N_INPUTS, h->s_fx)) return 1;
    (void)hipMemcpyAsync(&s.h_fail[1], cmx_fxcm_fail_flag(h->fxcm), 4, hipMemcpyDeviceToHost, h->s_fx);
    (void)hipEventRecord(s.ev_fx1, h->s_fx);
  }
  lap(3);
  if (h->p8) {  // ---- paq8 stage: front end on this thread, role kernels on its own streams; needs nothing from the other stages ----
    (void)hipEventRecord(s.ev_p80, h->s_p8);
    if (cmx_p8stage_run(h->p8, bytes, n, d_layer0 + 434, CMX_N_INPUTS, h->s_p8)) return 1;
    (void)hipEventRecord(s.ev_p81, h->s_p8);
  }
  lap(4);
  s.used = true;
  s.untimed = false;  // becomes true once its mixing network is enqueued
  h->chunks++;
  txn.ok = true;
  return 0;
}

// hints: what Predictor::Perceive leaves in `lstmpr` / `lstmex` (predictor.cpp:462-465) along the oldest begun chunk
// that has not handed them out: lstm_p[t] / lstm_ex[t] = ByteModel::Predict value and `ex` of the LSTM byte mixer for
// bit t of the chunk, t = 0 .. 8n (entry 8n = the first bit after the chunk, known from the last distribution alone).
// After coding bit t the reference holds lstmpr = 1 + 4094 * lstm_p[t + 1] (float arithmetic), lstmex = lstm_ex[t + 1].
// Waits for that chunk's LSTM stage.
int cmx_pipeline_hints(cmx_pipeline_t* h, float* lstm_p, int* lstm_ex) {
  if (!h || !lstm_p || !lstm_ex) { cmx_set_err("cmx_pipeline_hints: bad argument"); return 1; }
  if (h->hinted >= h->chunks) { cmx_set_err("cmx_pipeline_hints: no begun chunk is waiting for them"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  Slot& s = h->slot[h->hinted % kSlots];
  const size_t n = s.n, T = 8 * n;
  float* dp = s.d_hint;
  int* dx = (int*)(s.d_hint + (8 * h->max_chunk + 1));  // bits 0 .. 8n-1 were written by the LSTM stage (begin)
  float* hp = s.h_hint;
  int* hx = (int*)(s.h_hint + (8 * h->max_chunk + 1));
  // entry 8n: bit 0 of whatever byte follows, from the distribution after the chunk's last byte
  if (cmx_bytemodel_bit_run(h->device, s.d_lstm_out + (n - 1) * 256, s.d_bytes, 0, dp + T, 1, dx + T, nullptr, h->s_lstm))
    return 1;
  // bits 0 .. 8n-1: column 2077 of the chunk's layer-0 rows, gathered
  bool ok = hipMemcpy2DAsync(hp, 4, s.d_layer0 + 2077, CMX_N_INPUTS * 4, 4, T, hipMemcpyDeviceToHost, h->s_lstm) == hipSuccess;
  ok = ok && hipMemcpyAsync(hp + T, dp + T, 4, hipMemcpyDeviceToHost, h->s_lstm) == hipSuccess;
  ok = ok && hipMemcpyAsync(hx, dx, (T + 1) * 4, hipMemcpyDeviceToHost, h->s_lstm) == hipSuccess;
  ok = ok && hipStreamSynchronize(h->s_lstm) == hipSuccess;
  if (!ok) { cmx_set_err("cmx_pipeline_hints: device error"); return 1; }
  memcpy(lstm_p, hp, (T + 1) * 4);
  memcpy(lstm_ex, hx, (T + 1) * 4);
  h->hinted++;
  return 0;
}

// finish: the oldest begun chunk gets its fxcm/paq8 columns (HOST rows of 2022 floats = layer-0 columns 3..2024; NULL
// = the caller has already written them into d_layer0) and its mixing network. Asynchronous.
static int finish_impl(cmx_pipeline_t* h, const float* cols, int first_col, int ncols, float* d_p_out);
int cmx_pipeline_finish(cmx_pipeline_t* h, const float* cols, float* d_p_out) {
  if (h && h->fxcm && h->p8 && cols) { cmx_set_err("cmx_pipeline_finish: the fxcm and paq8 stages are on the device: there are no columns to hand in"); return 1; }
  if (h && h->fxcm && cols) { cmx_set_err("cmx_pipeline_finish: the fxcm stage is on the device; hand in columns 434..2024 with cmx_pipeline_finish_cols"); return 1; }
  return finish_impl(h, cols, 3, 2022, d_p_out);
}
// the same with the caller's rows covering layer-0 columns first_col .. first_col + ncols - 1 only (HOST rows of ncols floats)
int cmx_pipeline_finish_cols(cmx_pipeline_t* h, const float* cols, int first_col, int ncols, float* d_p_out) {
  if (!cols || first_col < 3 || ncols <= 0 || first_col + ncols > 2025 || (h && h->fxcm && first_col < 434) || (h && h->p8 && first_col + ncols > 434)) { cmx_set_err("cmx_pipeline_finish_cols: bad column range"); return 1; }
  return finish_impl(h, cols, first_col, ncols, d_p_out);
}
static int finish_impl(cmx_pipeline_t* h, const float* cols, int first_col, int ncols, float* d_p_out) {
  if (!h || !d_p_out) { cmx_set_err("cmx_pipeline_finish: bad argument"); return 1; }
  if (h->finished >= h->chunks) { cmx_set_err("cmx_pipeline_finish: no begun chunk"); return 1; }
  if (h->failed) { cmx_set_err("cmx_pipeline_finish: an earlier chunk of this handle failed part-way; its state is void (destroy it)"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  struct Txn { cmx_pipeline* h; bool ok = false; ~Txn() { if (!ok) h->failed = true; } } txn{h};
  Slot& s = h->slot[h->finished % kSlots];
  const size_t n = s.n;
  if (cols) {  // returns once the rows have been read: the caller may reuse `cols` right away
    bool ok = hipMemcpy2DAsync(s.d_layer0 + first_col, CMX_N_INPUTS * 4, cols, (size_t)ncols * 4, (size_t)ncols * 4, 8 * n, hipMemcpyHostToDevice,
                               h->s_mix) == hipSuccess;
    ok = ok && hipEventRecord(s.ev_cols, h->s_mix) == hipSuccess && hipEventSynchronize(s.ev_cols) == hipSuccess;
    if (!ok) { cmx_set_err("cmx_pipeline_finish: column upload failed"); return 1; }
  }
  // ---- final mixing network, once both producers have written their columns ----
  const auto t_fin = std::chrono::steady_clock::now();
  (void)hipStreamWaitEvent(h->s_mix, s.ev_ctx1, 0);
  (void)hipStreamWaitEvent(h->s_mix, s.ev_lstm1, 0);
  if (h->fxcm) (void)hipStreamWaitEvent(h->s_mix, s.ev_fx1, 0);
  if (h->p8) (void)hipStreamWaitEvent(h->s_mix, s.ev_p81, 0);
  (void)hipEventRecord(s.ev_mix0, h->s_mix);
  float* dmix = nullptr;
  if (h->dbg_mix && h->dbg_mix_bits + 8 * n <= h->dbg_mix_cap) dmix = h->dbg_mix + h->dbg_mix_bits * 47;   // (CMX_MIXERS, mixnet_state.h)
  h->dbg_mix_bits += 8 * n;
  if (cmx_mixnet_run(h->mix, s.d_layer0, s.d_sel, s.d_bits, 8 * n, d_p_out, dmix, h->s_mix)) return 1;
  // the kernel's sticky time-out word, in stream order behind it: cmx_pipeline_wait looks at it per chunk (a look-ahead coder never
  // calls cmx_pipeline_sync, the only other place where it is read)
  (void)hipMemcpyAsync(&s.h_fail[2], cmx_mixnet_error_flag(h->mix), 4, hipMemcpyDeviceToHost, h->s_mix);
  (void)hipEventRecord(s.ev_mix1, h->s_mix);
  s.d_p = d_p_out;
  h->host_ms[5] += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_fin).count();
  s.untimed = true;
  h->last_slot = (int)(h->finished % kSlots);
  h->finished++;
  if (h->hinted < h->finished) h->hinted = h->finished;  // hints nobody asked for are skipped
  txn.ok = true;
  return 0;
}

// diagnosis (scripts/gpu_foreign_load.py): a DEVICE area of cap_bits x 47 floats that receives every mixer's output (Mixer::Mix, before the clamp) of
// every bit of the stream from now on, in stream order; NULL switches it off. Look-ahead chunks only.
int cmx_pipeline_debug_mix_out(cmx_pipeline_t* h, float* d_mix, uint64_t cap_bits) {
  if (!h) { cmx_set_err("cmx_pipeline_debug_mix_out: null handle"); return 1; }
  h->dbg_mix = d_mix; h->dbg_mix_cap = d_mix ? cap_bits : 0; h->dbg_mix_bits = 0;
  return 0;
}

// diagnosis (scripts/gpu_mixnet_vote.py): the DEVICE buffers of chunk number `index` (0 = the first submitted; one of the last CMX_PIPELINE_SLOTS) that the mixing
// network reads beside the layer-0 rows -- its 47 selectors per bit and the coded bits -- valid until the slot is reused CMX_PIPELINE_SLOTS submits later
int cmx_pipeline_debug_slot(cmx_pipeline_t* h, uint64_t index, const uint32_t** d_sel, const uint8_t** d_bits, size_t* nbytes) {
  if (!h || index >= h->finished || index + kSlots < h->finished) { cmx_set_err("cmx_pipeline_debug_slot: that chunk is not in flight"); return 1; }
  const Slot& s = h->slot[index % kSlots];
  if (d_sel) *d_sel = s.d_sel;
  if (d_bits) *d_bits = s.d_bits;
  if (nbytes) *nbytes = s.n;
  return 0;
}

int cmx_pipeline_submit(cmx_pipeline_t* h, const uint8_t* bytes, size_t n, float* d_layer0, float* d_p_out) {
  if (!h) { cmx_set_err("cmx_pipeline_submit: null handle"); return 1; }
  if (n == 0) return 0;
  if (!bytes || !d_layer0 || !d_p_out || n > h->max_chunk) { cmx_set_err("cmx_pipeline_submit: bad argument"); return 1; }
  if (h->finished != h->chunks) { cmx_set_err("cmx_pipeline_submit: a begun chunk is waiting for cmx_pipeline_finish"); return 1; }
  if (cmx_pipeline_begin(h, bytes, n, d_layer0)) return 1;
  return cmx_pipeline_finish(h, nullptr, d_p_out);
}

// Predictor::Pretrain (predictor.cpp:471-487) over dictionary bytes: only `models_` learn -- here the context /
// small-model stage; PPMd, the LSTM, the mixers and the SSE are untouched, exactly as in the reference.
int cmx_pipeline_pretrain(cmx_pipeline_t* h, const uint8_t* bytes, size_t n) {
  if (!h) { cmx_set_err("cmx_pipeline_pretrain: null handle"); return 1; }
  if (n == 0) return 0;
  if (!bytes) { cmx_set_err("cmx_pipeline_pretrain: bad argument"); return 1; }
  if (h->chunks) { cmx_set_err("cmx_pipeline_pretrain: only before the first submit (preprocessor.cpp:37-69)"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  uint8_t* d = nullptr;
  if (hipMalloc((void**)&d, n) != hipSuccess) { cmx_set_err("cmx_pipeline_pretrain: hipMalloc failed"); return 1; }
  bool ok = hipMemcpyAsync(d, bytes, n, hipMemcpyHostToDevice, h->s_ctx) == hipSuccess;
  hipEvent_t ev_d = nullptr;   // the dictionary bytes are on the device before the fxcm kernels read them
  ok = ok && hipEventCreateWithFlags(&ev_d, hipEventDisableTiming) == hipSuccess && hipEventRecord(ev_d, h->s_ctx) == hipSuccess;
  ok = ok && cmx_ctxmodels_pretrain(h->ctx, d, n, h->s_ctx) == 0;
  // fxcm and paq8 are two of models_: Predict + Perceive per dictionary bit (predictor.cpp:471-476), fxcm with the hints at their start-up
  // value 0 (:359); the rows are discarded. The three stages learn independently, so their chunks are enqueued side by side (context
  // stage on its stream, fxcm and paq8 chunk by chunk on theirs) and waited for once: the dictionary costs the slowest stage, not the sum.
  const size_t C = h->max_chunk;
  int16_t* zpr = nullptr; uint8_t* zex = nullptr;
  if (ok && h->fxcm) {
    ok = hipMalloc((void**)&zpr, 8 * C * 2) == hipSuccess && hipMalloc((void**)&zex, 8 * C) == hipSuccess;
    ok = ok && hipMemsetAsync(zpr, 0, 8 * C * 2, h->s_fx) == hipSuccess && hipMemsetAsync(zex, 0, 8 * C, h->s_fx) == hipSuccess;
    ok = ok && hipStreamWaitEvent(h->s_fx, ev_d, 0) == hipSuccess;
  }
  for (size_t off = 0; ok && off < n; off += C) {
    const size_t m = n - off < C ? n - off : C;
    if (h->fxcm) ok = cmx_fxcm_run(h->fxcm, bytes + off, d + off, m, zpr, zex, h->d_fx_scratch, 434, h->s_fx) == 0;
    if (ok && h->p8) ok = cmx_p8stage_run(h->p8, bytes + off, m, h->d_p8_scratch, 1591, h->s_p8) == 0;
  }
  ok = hipStreamSynchronize(h->s_ctx) == hipSuccess && ok;
  if (h->fxcm) ok = hipStreamSynchronize(h->s_fx) == hipSuccess && ok;
  if (h->p8) ok = hipStreamSynchronize(h->s_p8) == hipSuccess && ok;
  if (zpr) (void)hipFree(zpr);
  if (zex) (void)hipFree(zex);
  if (ev_d) (void)hipEventDestroy(ev_d);
  (void)hipFree(d);
  if (!ok) { cmx_set_err("cmx_pipeline_pretrain: device error"); return 1; }
  return 0;
}

// Wait until chunk number `index` (0 = the first chunk submitted) has left the mixing network: its p[] is complete.
// Only the last CMX_PIPELINE_SLOTS chunks can be waited for (their slots still hold the events). Also reports a timed-out
// in-launch hand-off of the multi-workgroup LSTM / fxcm kernels as of this chunk (their sticky flags were copied back in
// stream order behind the chunk's kernels, which the mixing network waited for): a caller that codes chunk by chunk
// learns of a void stream at the chunk where it happened, not at the end of the file.
int cmx_pipeline_wait(cmx_pipeline_t* h, uint64_t index) {
  if (!h) { cmx_set_err("cmx_pipeline_wait: null handle"); return 1; }
  if (index >= h->finished || index + kSlots < h->finished) { cmx_set_err("cmx_pipeline_wait: that chunk is not in flight"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  Slot& s = h->slot[index % kSlots];
  if (hipEventSynchronize(s.ev_mix1) != hipSuccess) { cmx_set_err("cmx_pipeline_wait: device error"); return 1; }
  const bool p8fail = h->p8 && cmx_p8stage_mixfail(h->p8);   // (host-mapped: the paq8 mixer's workgroup 0 gave up waiting for another workgroup)
  if (s.h_fail[0] || s.h_fail[1] || s.h_fail[2] || p8fail) {
    cmx_set_err(std::string("cmx_pipeline_wait: an in-launch hand-off of the ") + (s.h_fail[0] ? "LSTM" : s.h_fail[1] ? "fxcm" : s.h_fail[2] ? "mixing network" : "paq8 mixer") +
                " kernels timed out (workgroups not co-resident?): the stream's output is void from chunk " + std::to_string(index) + " on");
    h->failed = true;
    return 1;
  }
  return 0;
}

// cmx_pipeline_wait(index), then the chunk's probabilities (8 x its byte count floats) into p_host. Only this chunk is waited for:
// the chunks behind it stay in flight (a device-wide synchronisation here -- what cmx_copy_to_host does -- would drain the whole
// pipeline once per chunk and cost two thirds of the throughput: a chunk's latency is three chunk periods).
int cmx_pipeline_fetch(cmx_pipeline_t* h, uint64_t index, float* p_host) {
  if (!p_host) { cmx_set_err("cmx_pipeline_fetch: bad argument"); return 1; }
  if (cmx_pipeline_wait(h, index)) return 1;
  const Slot& s = h->slot[index % kSlots];
  // the upload stream is idle between submits (its copies were waited for by the stages long ago) and non-blocking: the copy runs at once
  if (hipMemcpyAsync(p_host, s.d_p, 8 * s.n * sizeof(float), hipMemcpyDeviceToHost, h->s_up) != hipSuccess || hipStreamSynchronize(h->s_up) != hipSuccess) {
    cmx_set_err("cmx_pipeline_fetch: device-to-host copy failed");
    return 1;
  }
  return 0;
}

int cmx_pipeline_sync(cmx_pipeline_t* h) {
  if (!h) return 1;
  (void)hipSetDevice(h->device);
  bool ok = hipStreamSynchronize(h->s_ctx) == hipSuccess;
  ok = hipStreamSynchronize(h->s_lstm) == hipSuccess && ok;
  ok = hipStreamSynchronize(h->s_mix) == hipSuccess && ok;
  if (h->s_fx) ok = hipStreamSynchronize(h->s_fx) == hipSuccess && ok;
  if (h->p8) ok = cmx_p8stage_sync(h->p8) == 0 && ok;
  if (!ok) { cmx_set_err("cmx_pipeline_sync: device error"); h->failed = true; return 1; }
  if (cmx_ctxmodels_sync(h->ctx) || cmx_mixnet_sync(h->mix)) { h->failed = true; return 1; }
  // the multi-workgroup kernels bound every in-launch wait: one that ran out left garbage behind, not a hang
  if (cmx_lstm_failed(h->lstm)) { cmx_set_err("cmx_pipeline_sync: an in-launch hand-off of the LSTM kernels timed out (workgroups not co-resident?): the stream's output is void"); h->failed = true; return 1; }
  if (h->fxcm && cmx_fxcm_failed(h->fxcm)) { cmx_set_err("cmx_pipeline_sync: an in-launch hand-off of the fxcm kernel timed out (workgroups not co-resident?): the stream's output is void"); h->failed = true; return 1; }
  for (Slot& s : h->slot) collect(h, s);
  if (h->last_slot >= 0) {
    Slot& s = h->slot[h->last_slot];
    (void)hipEventElapsedTime(&h->stage_ms[0], s.ev_ctx0, s.ev_ctx1);
    (void)hipEventElapsedTime(&h->stage_ms[1], s.ev_lstm0, s.ev_lstm1);
    (void)hipEventElapsedTime(&h->stage_ms[2], s.ev_mix0, s.ev_mix1);
  }
  return 0;
}

int cmx_pipeline_stage_totals(cmx_pipeline_t* h, double ms[3], uint64_t* chunks, int reset) {
  if (!h || !ms || !chunks) return 1;
  for (int i = 0; i < 3; ++i) ms[i] = h->tot_ms[i];
  *chunks = h->tot_chunks;
  if (reset) { h->tot_ms[0] = h->tot_ms[1] = h->tot_ms[2] = 0; h->tot_chunks = 0; h->fx_ms = 0; h->p8_ms = 0; for (double& v : h->host_ms) v = 0; if (h->p8) { double t[7]; uint64_t c; (void)cmx_p8stage_role_ms(h->p8, t, &c, 1); } }
  return 0;
}

int cmx_pipeline_last_stage_ms(cmx_pipeline_t* h, float ms[3]) {
  if (!h || !ms) return 1;
  memcpy(ms, h->stage_ms, sizeof h->stage_ms);
  return 0;
}

// ================================================================================================================================
// The decoder's form of the pipeline: the late-bit protocol (cmx_late.h; include/cmix_amd.h section 4)
// ================================================================================================================================
size_t cmx_late_box_bytes(size_t nbits) { return sizeof(CmxLateBox) + nbits + 64; }

// HOST PUSH (cmx_late.h, CmxLate::pad): what late_relay() does on the device, done by the decoder thread with stores into device memory -- bit s - 1 of the chunk
// (s == 0: the bit before it), the host stages' records of step s into their device mirrors, then the step count. The stores go through the PCIe BAR in program
// order (the mirrors and counters are uncached device memory); the fence drains the write-combining buffers before the counter, and again behind it.
static void late_push_step(LateSet& q, size_t s, size_t nbits, int ybit) {
  q.d_bits[8 + (ptrdiff_t)s - 1] = (uint8_t)(ybit & 1);
  for (int e = 0; e < q.nrelay; ++e) {
    const cmx_late_relay_t& r = q.relay[e];
    size_t row;
    if (r.kind == 0) { if (s >= nbits) continue; row = s; }
    else if (r.kind == 1) { if ((s & 7) || s >= nbits) continue; row = s >> 3; }
    else { if ((s & 7) || s < 8) continue; row = (s >> 3) - 1; }
    memcpy((char*)r.dst + row * r.stride, (const char*)r.src + row * r.stride, r.stride);
  }
  __builtin_ia32_sfence();
  *(volatile uint32_t*)(q.cnt + LC_KNOWN * CMX_LATE_CNT_STRIDE) = q.lt.base | (uint32_t)(s + 1);
  __builtin_ia32_sfence();
}

static double late_now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// the previous chunk's counter value that says "all n of its bytes' distributions are in place"
static uint32_t pq_want(uint64_t c, size_t n) { return c ? ((uint32_t)((c - 1) & 0xFFFFu) << 16) | (uint32_t)n : 0u; }
// enqueue every stage kernel of chunk number c (its bits arrive later)
static int late_launch(cmx_pipeline* h, uint64_t c) {
  Late* L = h->late;
  L->touched = true;
  const int s = (int)(c % 3);
  LateSet& q = L->set[s];
  const size_t n = kLateChunk;
  CmxLateBox* B = q.box;
  // the box of the chunk three back: nothing reads it any more (its kernels ended before those of chunk c - 2 began)
  memset(B, 0, sizeof(CmxLateBox));
  B->nbits = (uint32_t)(8 * n);
  B->timeout_s = L->timeout_s;
  __sync_synchronize();
  q.lt.box = B; q.lt.cnt = q.cnt; q.lt.base = (uint32_t)(c & 0xFFFFu) << 16; q.lt.pad = L->push ? 1u : 0u; q.lt.dbit0 = q.d_bits + 8;   // the counters are never cleared: a value of the chunk three back is below this base
  void* const LT = &q.lt;
  const float* brk0 = nullptr;
  if (cmx_ctxmodels_run_late(h->ctx, LT, n, q.layer0, CMX_N_INPUTS, q.sel, q.brk, &brk0, h->s_ctx)) return 1;
  const uint32_t *c0_brk = nullptr, *c0_lstm = nullptr;
  const float* lstm0 = L->lstm0;
  if (c > 0) {   // the distributions going into the chunk are the previous chunk's last rows, valid when ITS stages have counted them
    LateSet& pq = L->set[(int)((c - 1) % 3)];
    brk0 = pq.brk + (n - 1) * 256; c0_brk = pq.cnt + LC_BRK * CMX_LATE_CNT_STRIDE;
    lstm0 = pq.lstm + (n - 1) * 256; c0_lstm = pq.cnt + LC_LSTM * CMX_LATE_CNT_STRIDE;
  }
  const uint32_t pwant = pq_want(c, n);
  if (cmx_bytemodel_late_run(h->device, LT, n, brk0, q.brk, q.d_ppmd, lstm0, q.lstm, c0_brk, pwant, c0_lstm, pwant, q.layer0, CMX_N_INPUTS, q.hint_pr, q.hint_ex,
                             q.d_bits + 8, q.d_relay, q.nrelay, L->s_bm)) return 1;
  if (cmx_fxcm_run_late(h->fxcm, LT, n, q.hint_pr, q.hint_ex, q.layer0, CMX_N_INPUTS, s, h->s_fx)) return 1;
  if (cmx_p8stage_run_late(h->p8, LT, n, q.layer0 + 434, CMX_N_INPUTS, s)) return 1;
  if (cmx_mixnet_run_late(h->mix, LT, q.layer0, q.sel, 8 * n, h->s_mix)) return 1;
  L->launched = c + 1;
  return 0;
}

int cmx_pipeline_late_start(cmx_pipeline_t* h, int last_bit) {
  if (!h) { cmx_set_err("cmx_pipeline_late_start: null handle"); return 1; }
  if (h->late) { cmx_set_err("cmx_pipeline_late_start: already started"); return 1; }
  if (!h->fxcm || !h->p8) { cmx_set_err("cmx_pipeline_late_start: enable the fxcm and paq8 stages first (a decoder takes no columns from the caller)"); return 1; }
  if (h->chunks) { cmx_set_err("cmx_pipeline_late_start: the handle has coded chunks of known bytes (a stream is decoded from its first bit)"); return 1; }
  if (h->compact) { cmx_set_err("cmx_pipeline_late_start: the stages share HIP streams (CMX_PIPELINE_STREAMS); a decoder needs every stage kernel running at once"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  if (hipDeviceSynchronize() != hipSuccess) { cmx_set_err("cmx_pipeline_late_start: device error"); return 1; }   // pretraining is complete
  Late* L = new Late();
  h->late = L;
  const size_t n = kLateChunk, T = 8 * n;
  bool ok = hipStreamCreateWithFlags(&L->s_bm, hipStreamNonBlocking) == hipSuccess;
  ok = ok && hipHostMalloc((void**)&L->dbg_row, CMX_N_INPUTS * 4, hipHostMallocDefault) == hipSuccess && hipHostMalloc((void**)&L->dbg_sel, CMX_N_MIXERS * 4, hipHostMallocDefault) == hipSuccess;
  for (LateSet& q : L->set) {
    q.box = (CmxLateBox*)cmx_late_alloc(cmx_late_box_bytes(T));
    q.ppmd = (float*)cmx_late_alloc((n + 1) * 256 * 4);
    q.bytes = (uint8_t*)cmx_late_alloc(n);
    q.cnt = (uint32_t*)cmx_late_alloc_dev(h->device, LC_N * CMX_LATE_CNT_STRIDE * 4);
    q.d_ppmd = (float*)cmx_late_alloc_dev(h->device, (n + 1) * 256 * 4);
    q.d_bits = (uint8_t*)cmx_late_alloc_dev(h->device, T + 16);
    q.d_relay = (cmx_late_relay_t*)cmx_late_alloc_dev(h->device, CMX_LATE_RELAY_MAX * sizeof(cmx_late_relay_t));
    ok = ok && q.d_ppmd && q.d_bits && q.d_relay;
    q.layer0 = (float*)cmx_late_alloc_dev(h->device, T * CMX_N_INPUTS * 4);
    q.sel = (uint32_t*)cmx_late_alloc_dev(h->device, T * CMX_N_MIXERS * 4);
    q.brk = (float*)cmx_late_alloc_dev(h->device, n * 256 * 4);
    q.lstm = (float*)cmx_late_alloc_dev(h->device, n * 256 * 4);
    q.hint_pr = (int16_t*)cmx_late_alloc_dev(h->device, T * 2);
    q.hint_ex = (uint8_t*)cmx_late_alloc_dev(h->device, T);
    ok = ok && q.box && q.cnt && q.layer0 && q.sel && q.brk && q.lstm && q.ppmd && q.bytes && q.hint_pr && q.hint_ex;
  }
  if (!ok) { cmx_set_err("cmx_pipeline_late_start: buffer allocation failed"); L->failed = true; return 1; }
  // every stage allocates what it needs NOW: once the first chunk's kernels run they wait for this thread, and an allocation that maps
  // memory into the device may wait for them
  if (cmx_fxcm_late_prepare(h->fxcm, n) || cmx_p8stage_late_prepare(h->p8, n) || cmx_mixnet_late_prepare(h->mix, T)) { L->failed = true; return 1; }
  for (int s = 0; s < 3; ++s) {   // the relay wave's table of each buffer set: the stages' record arrays + PPMd's distributions
    LateSet& q = L->set[s];
    cmx_late_relay_t tab[CMX_LATE_RELAY_MAX];
    int k = cmx_p8stage_late_relay(h->p8, s, tab, CMX_LATE_RELAY_MAX - 2);
    const int kf = k < 0 ? -1 : cmx_fxcm_late_relay(h->fxcm, s, tab + k, 1);
    if (k < 0 || kf < 0) { L->failed = true; return 1; }
    k += kf;
    tab[k].src = q.ppmd; tab[k].dst = q.d_ppmd; tab[k].stride = 1024; tab[k].kind = 1; ++k;
    q.nrelay = k;
    memcpy(q.relay, tab, (size_t)k * sizeof(cmx_late_relay_t));
    if (hipMemcpy(q.d_relay, tab, (size_t)k * sizeof(cmx_late_relay_t), hipMemcpyHostToDevice) != hipSuccess) { cmx_set_err("cmx_pipeline_late_start: table upload failed"); L->failed = true; return 1; }
  }
  if (hipDeviceSynchronize() != hipSuccess) { cmx_set_err("cmx_pipeline_late_start: device error"); L->failed = true; return 1; }
  L->lstm0 = cmx_lstm_byte_probs(h->lstm);
  { const char* v = getenv("CMX_LATE_LSTM_PER_BYTE"); L->lstm_per_byte = v && v[0] == '1'; }
  { const char* v = getenv("CMX_LATE_TIMEOUT_S"); const long t = v ? atol(v) : 0; L->timeout_s = t > 0 ? (uint32_t)(t < 86400 ? t : 86400) : 0u; }   // (the host's own bound in late_predict follows: twice this, at least 60 s)
  {   // host push needs the device's memory in this process's address space (large BAR); CMX_LATE_PULL=1 keeps the relay wave (A/B, and the fall-back)
    int bar = 0;
    const char* v = getenv("CMX_LATE_PULL");
    L->push = !(v && v[0] == '1') && hipDeviceGetAttribute(&bar, hipDeviceAttributeIsLargeBar, h->device) == hipSuccess && bar == 1;
  }
  L->mix_chunk0 = cmx_mixnet_runs(h->mix);
  // chunk 0 and, queued behind it, chunk 1
  if (late_launch(h, 0)) { L->failed = true; return 1; }
  LateSet& q0 = L->set[0];
  memcpy(q0.ppmd, h->last_dist, 256 * 4);                       // PPMd's distribution going into the stream
  if (cmx_p8stage_late_emit(h->p8, 0, 0)) { L->failed = true; return 1; }   // the records of step 0
  q0.box->last_y = last_bit ? 1u : 0u;
  __sync_synchronize();
  q0.box->start = 1;
  if (L->push) late_push_step(q0, 0, T, last_bit ? 1 : 0);
  if (late_launch(h, 1)) { L->failed = true; return 1; }
  L->active = true;
  return 0;
}

int cmx_pipeline_late_stop(cmx_pipeline_t* h) {
  if (!h || !h->late) return 0;
  Late* L = h->late;
  if (!L->active && !L->touched) return 0;   // (a start that failed AFTER its first launch leaves kernels waiting: they are unwound here too)
  for (LateSet& q : L->set) if (q.box) { q.box->abort = 1; }
  __sync_synchronize();
  (void)hipSetDevice(h->device);
  const double t0 = late_now();
  const bool ok = hipDeviceSynchronize() == hipSuccess;
  if (getenv("CMX_TIMING")) fprintf(stderr, "[cmx timing] %-28s %.3f s\n", "late_stop: kernels unwound", (late_now() - t0) / 1e3);
  L->active = false; L->touched = false;
  if (!ok) { cmx_set_err("cmx_pipeline_late_stop: device error"); return 1; }
  return 0;
}

float cmx_pipeline_late_predict(cmx_pipeline_t* h) {
  if (!h || !h->late || !h->late->active) { cmx_set_err("cmx_pipeline_late_predict: not started"); return -1.0f; }
  Late* L = h->late;
  if (L->failed) { cmx_set_err("cmx_pipeline_late_predict: the stream failed earlier"); return -1.0f; }
  if (L->predicted) { cmx_set_err("cmx_pipeline_late_predict: called twice without perceive()"); return -1.0f; }
  LateSet& q = L->set[(int)(L->cur % 3)];
  volatile unsigned long long* w = (volatile unsigned long long*)&q.box->p_word[L->t % CMX_LATE_P_RING];
  const double t0 = late_now();
  unsigned long long v;
  unsigned spins = 0;
  while ((unsigned)((v = *w) >> 32) != (unsigned)(L->t + 1)) {
    if ((++spins & 0xfffu) == 0) {
      const double el = late_now() - t0;
      const char* why = nullptr;
      static int slow_reports = 0;
      if (el > 1000.0 && slow_reports < 4 && getenv("CMX_TIMING")) {   // diagnostics: a wait of more than a second -- which stage is the stream waiting for?
        ++slow_reports;
        uint32_t cv[LC_N * CMX_LATE_CNT_STRIDE] = {0};
        (void)hipMemcpyAsync(cv, q.cnt, sizeof cv, hipMemcpyDeviceToHost, h->s_up);
        (void)hipStreamSynchronize(h->s_up);
        fprintf(stderr, "\n[cmx timing] waiting > 1 s for p of bit %llu (bit %zu of chunk %llu); rows:", (unsigned long long)L->bits, L->t, (unsigned long long)L->cur);
        static const char* const nm2[] = {"ctx", "bm0", "bm1", "bm2", "fx", "p8", "cm2a", "cm2b", "cm2c", "fam", "lanes", "dmc", "brk", "lstm", "known"};
        for (int i = 0; i < 15; ++i) fprintf(stderr, " %s=%u", nm2[i], cv[i * CMX_LATE_CNT_STRIDE] & 0xFFFFu);
        fprintf(stderr, " | box nknown %u start %u fail %u\n", q.box->nknown, q.box->start, q.box->fail);
      }
      if (*(volatile uint32_t*)&q.box->fail) why = "a stage kernel's wait ran out of time (are all stage kernels co-resident?)";
      else if (cmx_p8stage_mixfail(h->p8)) why = "the paq8 mixer's workgroup 0 timed out waiting for another workgroup";
      else if (el > (L->timeout_s > 30 ? 2000.0 * L->timeout_s : 60000.0)) why = "no prediction from the device for 60 s (or twice CMX_LATE_TIMEOUT_S)";
      if (why) {
        std::string st = " [box: nknown " + std::to_string(q.box->nknown) + " start " + std::to_string(q.box->start) + " rows";
        static const char* const nm[] = {"ctx", "bm0", "bm1", "bm2", "fx", "p8", "cm2a", "cm2b", "cm2c", "fam", "lanes", "dmc", "brk", "lstm", "known"};
        uint32_t cv[LC_N * CMX_LATE_CNT_STRIDE] = {0};
        for (LateSet& z : L->set) if (z.box) z.box->abort = 1;   // the kernels leave, the copy below can run
        if (hipMemcpy(cv, q.cnt, sizeof cv, hipMemcpyDeviceToHost) == hipSuccess)
          for (int i = 0; i < 15; ++i) st += std::string(" ") + nm[i] + "=" + std::to_string(cv[i * CMX_LATE_CNT_STRIDE] & 0xFFFFu);
        st += "]";
        cmx_set_err(std::string("cmx_pipeline_late_predict: ") + why + " at bit " + std::to_string(L->bits) + " (bit " + std::to_string(L->t) + " of its chunk)" + st);
        L->failed = true;
        for (LateSet& z : L->set) if (z.box) z.box->abort = 1;
        return -1.0f;
      }
    }
  }
  L->ms[0] += late_now() - t0;
  L->predicted = true;
  float p;
  const unsigned pb = (unsigned)v;
  memcpy(&p, &pb, 4);
  return p;
}

int cmx_pipeline_late_perceive(cmx_pipeline_t* h, int bit) {
  if (!h || !h->late || !h->late->active) { cmx_set_err("cmx_pipeline_late_perceive: not started"); return 1; }
  Late* L = h->late;
  if (L->failed) { cmx_set_err("cmx_pipeline_late_perceive: the stream failed earlier"); return 1; }
  if (!L->predicted) { cmx_set_err("cmx_pipeline_late_perceive: no pending predict()"); return 1; }
  struct Txn { Late* L; bool ok = false; ~Txn() { if (!ok) L->failed = true; } } txn{L};
  const int s = (int)(L->cur % 3);
  LateSet& q = L->set[s];
  const size_t n = kLateChunk, T = 8 * n, t = L->t;
  bit = bit ? 1 : 0;
  q.box->bit[t] = (uint8_t)bit;
  L->partial = (L->partial << 1) | (unsigned)bit;
  double t0 = late_now();
  auto lap = [&](int k) { const double now = late_now(); L->ms[k] += now - t0; t0 = now; };
  // ---- host stages for the step after this bit: their records are in place before the bit is published ----
  if (cmx_p8stage_late_bit(h->p8, bit)) return 1;
  const bool byte_done = (t & 7) == 7, chunk_done = t + 1 == T;
  const size_t b = t >> 3;
  uint8_t byte = 0;
  if (byte_done) {   // predictor.cpp:439-461
    byte = (uint8_t)L->partial;
    L->partial = 0;
    q.bytes[b] = byte;
    if (cmx_ppmd_run(h->ppmd, &byte, 1, q.ppmd + (b + 1) * 256)) return 1;
    memcpy(h->last_dist, q.ppmd + (b + 1) * 256, 256 * 4);
    lap(1);
    if (cmx_fxcm_late_byte(h->fxcm, s, b, byte)) return 1;
    lap(3);
  }
  LateSet* nq = nullptr;
  if (!chunk_done) { if (cmx_p8stage_late_emit(h->p8, s, t + 1)) return 1; }
  else {
    nq = &L->set[(int)((L->cur + 1) % 3)];
    if (cmx_p8stage_late_emit(h->p8, (int)((L->cur + 1) % 3), 0)) return 1;
    memcpy(nq->ppmd, q.ppmd + n * 256, 256 * 4);
    nq->box->last_y = (uint32_t)bit;
  }
  lap(2);
  __sync_synchronize();   // the records are written before the counter moves
  *(volatile uint32_t*)&q.box->nknown = (uint32_t)(t + 1);
  *(volatile unsigned long long*)&q.box->kb = ((unsigned long long)(unsigned)bit << 32) | (unsigned long long)(t + 1);
  if (nq) *(volatile uint32_t*)&nq->box->start = 1;
  if (L->push) {   // this thread is the relay: the step's bit and records into the device mirrors, then the count (the next chunk's step 0 with this chunk's last)
    late_push_step(q, t + 1, T, bit);
    if (nq) late_push_step(*nq, 0, T, bit);
  }
  // ---- the LSTM byte mixer's step for the completed byte (predictor.cpp:450-461), then "its distribution is there" ----
  if (byte_done) {
    if (L->lstm_per_byte) {
      if (cmx_lstm_run(h->lstm, q.ppmd + (b + 1) * 256, q.bytes + b, 1, q.lstm + b * 256, nullptr, 0, nullptr, h->s_lstm)) return 1;
      if (cmx_late_bump(h->device, q.cnt + LC_LSTM * CMX_LATE_CNT_STRIDE, q.lt.base | (uint32_t)(b + 1), nullptr, 0, h->s_lstm)) { cmx_set_err("cmx_pipeline_late_perceive: launch failed"); return 1; }
    } else if (L->lstm_covered == 0) {   // a new truncated-BPTT block or a new chunk begins with this byte: one launch for all of its bytes, which wait for theirs inside it
      const int c = cmx_lstm_run_late(h->lstm, &q.lt, q.d_ppmd, q.ppmd, q.bytes, q.lstm, b, n, h->s_lstm);
      if (c < 1) return 1;
      L->lstm_covered = c - 1;
    } else --L->lstm_covered;
    lap(4);
  }
  L->predicted = false;
  L->bits++;
  if (chunk_done) {
    L->cur++;
    L->t = 0;
    if (late_launch(h, L->cur + 1)) return 1;   // the chunk after the next, queued behind it
    lap(5);
  } else L->t = t + 1;
  txn.ok = true;
  return 0;
}

// test hook: the layer-0 row (2078 f32, host-coherent memory) and the 47 selectors of the bit whose p was predicted last -- valid
// between cmx_pipeline_late_predict() and cmx_pipeline_late_perceive()
const float* cmx_pipeline_late_debug_row(cmx_pipeline_t* h, const uint32_t** sel) {
  if (!h || !h->late || !h->late->active || !h->late->predicted) return nullptr;
  Late* L = h->late;
  const LateSet& q = L->set[(int)(L->cur % 3)];
  // the rows live in (uncached) device memory: copied out on the upload stream, which is idle while a stream is decoded
  if (hipMemcpyAsync(L->dbg_row, q.layer0 + L->t * CMX_N_INPUTS, CMX_N_INPUTS * 4, hipMemcpyDeviceToHost, h->s_up) != hipSuccess ||
      hipMemcpyAsync(L->dbg_sel, q.sel + L->t * CMX_N_MIXERS, CMX_N_MIXERS * 4, hipMemcpyDeviceToHost, h->s_up) != hipSuccess ||
      hipStreamSynchronize(h->s_up) != hipSuccess) return nullptr;
  if (sel) *sel = L->dbg_sel;
  return L->dbg_row;
}
// test hook (CMX_LATE_DEBUG=1 when the handle was built): the 47 mixer outputs (Mixer::Mix values: 26 layer-0, 20 layer-1, the final one) of
// the bit BEFORE the one predicted last (the waves write them when they learn, i.e. after the bit has arrived), copied to `out47`; 1 if the
// hook is off or no bit has been coded yet
int cmx_pipeline_late_debug_mix(cmx_pipeline_t* h, float out47[47]) {
  if (!h || !h->late || !h->late->active || !h->late->predicted || !out47 || !h->late->bits) return 1;
  Late* L = h->late;
  const uint64_t chunk = L->t ? L->cur : L->cur - 1;
  const size_t bit = L->t ? L->t - 1 : 8 * kLateChunk - 1;
  const float* d = cmx_mixnet_late_debug_mix(h->mix, L->mix_chunk0 + chunk, bit);
  if (!d) return 1;
  if (hipMemcpyAsync(L->dbg_row, d, CMX_N_MIXERS * 4, hipMemcpyDeviceToHost, h->s_up) != hipSuccess || hipStreamSynchronize(h->s_up) != hipSuccess) return 1;
  memcpy(out47, L->dbg_row, CMX_N_MIXERS * 4);
  return 0;
}

// Replay: predict / perceive for n known bits in one call (what a decoder does, minus the arithmetic decoder): p_out[i] = p before bits[i].
// For tests against reference traces and for timing the protocol without a caller's per-call overhead.
int cmx_pipeline_late_replay(cmx_pipeline_t* h, const uint8_t* bits, size_t n, float* p_out) {
  if (!h || !bits || !p_out) { cmx_set_err("cmx_pipeline_late_replay: bad argument"); return 1; }
  for (size_t i = 0; i < n; ++i) {
    const float p = cmx_pipeline_late_predict(h);
    if (p < 0) return 1;
    p_out[i] = p;
    if (cmx_pipeline_late_perceive(h, bits[i])) return 1;
  }
  return 0;
}

// diagnostics: the row counters of the chunk being decoded and when each was last moved (device clock, 100 MHz): out[2 i] = rows of
// counter i (cmx_late.h: LC_*; 14 = steps the relay has brought over), out[2 i + 1] = its time stamp; out[30..39]: the mixing network's
// stamps of the current bit (p out, row complete, sums there, layer 1 done, helpers fed). Taken between predict() and perceive() every
// counter stands at the same bit. Copies 1 KB from the device on the upload stream.
int cmx_pipeline_late_debug_times(cmx_pipeline_t* h, uint32_t out[48]) {
  if (!h || !h->late || !h->late->active || !out) return 1;
  Late* L = h->late;
  const LateSet& q = L->set[(int)(L->cur % 3)];
  uint32_t cv[LC_N * CMX_LATE_CNT_STRIDE];
  if (hipMemcpyAsync(cv, q.cnt, sizeof cv, hipMemcpyDeviceToHost, h->s_up) != hipSuccess || hipStreamSynchronize(h->s_up) != hipSuccess) return 1;
  for (int i = 0; i < 15; ++i) { out[2 * i] = cv[i * CMX_LATE_CNT_STRIDE] & 0xFFFFu; out[2 * i + 1] = cv[i * CMX_LATE_CNT_STRIDE + 1]; }
  for (int k = 0; k < 10; ++k) out[30 + k] = cv[15 * CMX_LATE_CNT_STRIDE + k];
  return 0;
}

int cmx_pipeline_late_host_ms(cmx_pipeline_t* h, double ms[6], uint64_t* bits) {
  if (!h || !h->late || !ms || !bits) return 1;
  for (int i = 0; i < 6; ++i) ms[i] = h->late->ms[i];
  *bits = h->late->bits;
  return 0;
}

}  // extern "C"
