// cmx_api.hip -- host side of the C ABI declared in include/cmix_amd.h.
//
// Owns device memory, builds the read-only tables with the HOST libm (so they
// are bit-identical to what the reference process computes at start-up:
// logit LUT ref src/mixer/sigmoid.cpp:5-10, SSE stretch/squash tables ref
// src/mixer/sse.cpp:80-135, mixer decay schedule ref src/mixer/mixer.cpp:58),
// and launches the persistent kernels.  No CPU implementation of the per-bit
// path exists in this library: every entry point needs a gfx950 device.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stddef.h>
#include <string>
#include <vector>

#include "../../include/cmix_amd.h"
#include "mixnet_state.h"
#include "cmx_late.h"

extern "C" __global__ void cmx_mixnet_kernel(MixState*, const float*, const uint32_t*,
                                             const uint8_t*, const float*, int, float*, float*, int);
extern "C" __global__ void cmx_mixnet_chunk_kernel(MixState*, const float*, const uint32_t*,
                                                   const uint8_t*, const float*, int, float*, float*, int);
extern "C" __global__ void cmx_mixnet_spec_kernel(MixState*, SpecXfer*, const float*, const uint32_t*, const uint8_t*, const float*, int, float*,
                                                  float*, int);
extern "C" __global__ void cmx_mixnet_spec_jitter_kernel(MixState*, SpecXfer*, const float*, const uint32_t*, const uint8_t*, const float*, int, float*, float*, int);
extern "C" __global__ void cmx_mixnet_spec_late_kernel(MixState*, SpecXfer*, const float*, const uint32_t*, const float*, int, float*, float*, int, CmxLate);
extern "C" __global__ void cmx_sse_init_kernel(MixState*);
extern "C" __global__ void cmx_probe_libm_kernel(int, const float*, float*, size_t);

namespace {

thread_local std::string g_err;
void set_err(const std::string& s) { g_err = s; }

#define HIP_OK(call)                                                                  \
  do {                                                                                \
    hipError_t e_ = (call);                                                           \
    if (e_ != hipSuccess) {                                                           \
      set_err(std::string(#call) + ": " + hipGetErrorString(e_));                     \
      return fail_value;                                                              \
    }                                                                                 \
  } while (0)

// learning rates in construction order, ref src/predictor.cpp:199-356
const float kLr[CMX_MIXERS] = {
    0.005f, 0.0005f, 0.005f, 0.0005f, 0.005f, 0.002f, 0.002f, 0.005f, 0.00005f, 0.0007f, 0.0005f,
    0.002f, 0.0005f, 0.001f, 0.001f, 0.001f, 0.005f, 0.001f, 0.001f, 0.005f, 0.001f, 0.001f,
    0.005f, 0.005f, 0.005f, 0.003f,
    0.005f, 0.0005f, 0.005f, 0.0005f, 0.00001f, 0.005f, 0.005f, 0.005f, 0.0005f, 0.002f, 0.001f,
    0.001f, 0.001f, 0.001f, 0.001f, 0.001f, 0.001f, 0.001f, 0.001f, 0.001f,
    0.0003f};

#define CMX_MIXNET_THREADS 512
constexpr size_t kLdsBytes = (size_t)(17 * CMX_MIX0 * 68 + 2112 + 32 + 32 + 64 + 48 + 48 + 48) * 4;

// ---- host-side table construction (host libm on purpose) -------------------
void build_logit(std::vector<float>& t) {  // ref sigmoid.cpp:5-10,23-25
  t.resize(100001);
  for (int i = 0; i < 100001; ++i) {
    float p = (i + 0.5f) / 100001;
    t[i] = logf(p / (1 - p));
  }
}

void build_sse_tables(std::vector<uint16_t>& t_st, std::vector<uint16_t>& t_sq) {
  // ref sse.cpp:80-135. Note sse.cpp defines its own log2/exp2 via log/exp.
  const double LOG2E = 1.44269504088896340736;
  auto l2 = [&](double a) { return LOG2E * log(a); };
  auto e2 = [&](double a) { return exp(a / LOG2E); };
  auto st = [&](double p) { return l2((1 - p) / p); };
  auto sq = [&](double p) { return 1.0 / (1.0 + e2(p)); };
  const int SCALE = 32768, hSCALE = 16384;
  const double st_coef = (hSCALE - 1) / l2(SCALE - 1);
  const double sq_coef = 1.0 / st_coef;
  t_st.assign(SCALE, 0);
  t_sq.assign(SCALE, 0);
  for (int i = 1; i < SCALE; i++)
    t_sq[i] = (uint16_t)(unsigned)(sq((double)(i - hSCALE) * sq_coef) * SCALE);
  unsigned x = 0;
  for (unsigned i = 1; i < (unsigned)SCALE; i++) {
    unsigned s = (unsigned)(st((double)i / SCALE) * st_coef + hSCALE);
    t_st[i] = (uint16_t)s;
    if ((uint16_t)s != t_st[x]) {
      unsigned y = i - 1;
      t_sq[t_st[x]] = (uint16_t)((x + y + 1) / 2);
      x = i;
    }
  }
}

}  // namespace

void cmx_set_err(const std::string& s) { set_err(s); }  // shared with the other stage files

// HIP maps a process's streams onto 4 hardware queues unless told otherwise, which serialises the stage kernels of one input
// stream that are meant to overlap (3.3 -> 6.2 KB/s when it was found). The runtime reads the variable at its
// first API call, so setting it when the library is loaded covers every host program (the reference's own main() included);
// a value the user has set is left alone.
__attribute__((constructor)) static void cmx_default_hw_queues() { setenv("GPU_MAX_HW_QUEUES", "16", 0); }

struct cmx_mixnet {
  int device = 0;
  MixState* d_state = nullptr;
  MixState h_state;  // host copy of the pointer block
  std::vector<void*> allocs;
  float* d_decay = nullptr;
  hipStream_t run_stream = nullptr; bool run_stream_set = false;
  hipStream_t s_up = nullptr; bool own_up = false;                   // uploads go on a stream that never has a kernel in front of them (cmx_mixnet_set_upload_stream)
  hipEvent_t ev_kdone[CMX_PIPELINE_SLOTS] = {};                       // the kernel that read decay slot i has run   // the stream cmx_mixnet_run was first called with
  size_t decay_cap = 0;
  float* h_decay = nullptr;  // pinned, DECAY_SLOTS x decay_cap: a slot is rewritten only after its copy ran
  hipEvent_t ev_decay[CMX_PIPELINE_SLOTS] = {};
  bool decay_used[CMX_PIPELINE_SLOTS] = {};
  uint64_t runs = 0;
  uint64_t bits_done = 0;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  bool timed = false;
  // bit-synchronous staging
  float* d_sync_probs = nullptr;
  uint32_t* d_sync_sel = nullptr;
  uint8_t* d_sync_bit = nullptr;
  float* d_sync_p = nullptr;
  float h_sync_decay = 0;
  bool predicted = false;
  // asynchronous bit-synchronous mode (device operands): the pending predict's operands, pinned (bit, decay) ring
  const float* pend_probs = nullptr;
  const uint32_t* pend_sel = nullptr;
  unsigned char* h_sync_pin = nullptr;  // 16 slots x 8 bytes
  unsigned sync_slot = 0;
  int profile = 0;
  int dbg = 0;          // CMX_MIXNET_DBG: timing experiments (results invalid when nonzero)
  int rotate = 0;       // CMX_MIXNET_ROTATE=1 (test hook): the roles' workgroups move to another XCD with every launch (see spec_kernel_body)
  int jitter = 0;       // CMX_MIXNET_JITTER=1..15 (test hook, tests/test_gpu_mixnet.py): pseudo-random stalls in every role but the gather wave -- same results, every lead / lag between the roles
  int xcd = -1;         // CMX_MIXNET_XCD=k: place the persistent kernel on XCD k (speed only; -1 = wherever block 0 lands)
  bool use_v1 = false;  // CMX_MIXNET_V1=1: run chunks through the bit-synchronous kernel
  bool tolerance = false;   // cmx_mixnet_set_tolerance (opt-in through the API, NOT bit-exact): layer-0 dot products as f64 tree sums rounded once (cmx_mixnet_spec_kernel only)
  bool use_spec = true; // cmx_mixnet_spec_kernel (26 helper workgroups, speculative segment-parallel chains); CMX_MIXNET_SPEC=0: the one-workgroup kernel
  SpecXfer* d_xfer = nullptr;
  float* d_late_p = nullptr; size_t late_p_cap = 0;   // the decoder's form: the kernel's p[] array (the host reads p from the box)
  float* late_mix_cur[3] = {nullptr, nullptr, nullptr};
  float* d_late_mix = nullptr;                         // CMX_LATE_DEBUG=1: the 47 mixer outputs per bit of the chunk launched last (test hook)
};

extern "C" {

const char* cmx_last_error(void) { return g_err.c_str(); }
const char* cmx_version(void) { return "cmix_amd 0.1 (gfx950; cmix v21 predictor path)"; }

int cmx_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// whole-predictor surface (cmx_create .. cmx_destroy): engine_api.hip

// Plain-C memory helpers so that a host program written against this header needs no HIP headers of its own.
void* cmx_device_alloc(int device, size_t bytes) {
  void* p = nullptr;
  if (hipSetDevice(device) != hipSuccess || hipMalloc(&p, bytes ? bytes : 1) != hipSuccess) {
    set_err("cmx_device_alloc: hipMalloc failed");
    return nullptr;
  }
  return p;
}
void cmx_device_free(int device, void* p) {
  if (p && hipSetDevice(device) == hipSuccess) (void)hipFree(p);
}
void* cmx_host_alloc(size_t bytes) {  // page-locked
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { set_err("cmx_host_alloc: hipHostMalloc failed"); return nullptr; }
  return p;
}
void cmx_host_free(void* p) { if (p) (void)hipHostFree(p); }
// Memory that kernels of DIFFERENT launches read and write while all of them run, and that the host writes while they run (the
// decoder's late-bit protocol, cmx_late.h): page-locked host memory mapped coherent (uncached on the device), zeroed.
void* cmx_late_alloc(size_t bytes) {
  void* p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess) { set_err("cmx_late_alloc: hipHostMalloc failed"); return nullptr; }
  memset(p, 0, bytes);
  return p;
}
void cmx_late_free(void* p) { if (p) (void)hipHostFree(p); }
// Memory that kernels of different launches hand rows to each other through WHILE THEY RUN (and the row counters): device memory with
// the uncached attribute -- a store followed by s_waitcnt vmcnt(0) has reached the device's memory, a load never hits a stale line of
// another XCD's L2 --, zeroed.
void* cmx_late_alloc_dev(int device, size_t bytes) {
  void* p = nullptr;
  if (hipSetDevice(device) != hipSuccess) { set_err("cmx_late_alloc_dev: hipSetDevice failed"); return nullptr; }
  if (hipExtMallocWithFlags(&p, bytes ? bytes : 1, hipDeviceMallocUncached) != hipSuccess) {
    (void)hipGetLastError();
    if (hipExtMallocWithFlags(&p, bytes ? bytes : 1, hipDeviceMallocFinegrained) != hipSuccess) { set_err("cmx_late_alloc_dev: hipExtMallocWithFlags failed"); return nullptr; }
  }
  if (hipMemset(p, 0, bytes) != hipSuccess) { (void)hipFree(p); set_err("cmx_late_alloc_dev: hipMemset failed"); return nullptr; }
  return p;
}
void cmx_late_free_dev(void* p) { if (p) (void)hipFree(p); }
int cmx_copy_to_host(int device, void* dst, const void* d_src, size_t bytes) {  // synchronous, after all prior work of the device
  if (hipSetDevice(device) != hipSuccess || hipDeviceSynchronize() != hipSuccess ||
      hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost) != hipSuccess) {
    set_err("cmx_copy_to_host: device error");
    return 1;
  }
  return 0;
}

// --------------------------------------------------------------------------
// mixing-network stage
// --------------------------------------------------------------------------
void cmx_mixnet_destroy(cmx_mixnet_t* h) {
  if (!h) return;
  hipSetDevice(h->device);
  hipDeviceSynchronize();
  for (void* p : h->allocs) hipFree(p);
  if (h->h_decay) hipHostFree(h->h_decay);
  if (h->h_sync_pin) hipHostFree(h->h_sync_pin);
  if (h->ev0) hipEventDestroy(h->ev0);
  if (h->ev1) hipEventDestroy(h->ev1);
  for (int i = 0; i < CMX_PIPELINE_SLOTS; ++i) { if (h->ev_decay[i]) hipEventDestroy(h->ev_decay[i]); if (h->ev_kdone[i]) hipEventDestroy(h->ev_kdone[i]); }
  if (h->own_up && h->s_up) hipStreamDestroy(h->s_up);
  delete h;
}

cmx_mixnet_t* cmx_mixnet_create(int device) {
  cmx_mixnet_t* const fail_value = nullptr;
  int n = cmx_device_count();
  if (n <= 0) { set_err("cmx_mixnet_create: no HIP device visible (a gfx950 GPU is required)"); return nullptr; }
  if (device < 0 || device >= n) { set_err("cmx_mixnet_create: bad device index"); return nullptr; }
  HIP_OK(hipSetDevice(device));
  hipDeviceProp_t prop;
  HIP_OK(hipGetDeviceProperties(&prop, device));
  if (!strstr(prop.gcnArchName, "gfx950")) {
    set_err(std::string("cmx_mixnet_create: device is ") + prop.gcnArchName + ", this library is built for gfx950 only");
    return nullptr;
  }
  cmx_mixnet_t* h = new cmx_mixnet();
  h->device = device;
  MixState& S = h->h_state;
  memset(&S, 0, sizeof S);
  auto dalloc = [&](size_t bytes, bool zero) -> void* {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) return nullptr;
    h->allocs.push_back(p);
    if (zero) hipMemset(p, 0, bytes);
    return p;
  };
#define ALLOC(field, type, count, zero)                                       \
  do {                                                                        \
    S.field = (type*)dalloc((size_t)(count) * sizeof(type), zero);            \
    if (!S.field) { set_err("hipMalloc failed for " #field); cmx_mixnet_destroy(h); return nullptr; } \
  } while (0)
  std::vector<float> lut;
  std::vector<uint16_t> t_st, t_sq;
  build_logit(lut);
  build_sse_tables(t_st, t_sq);
  float* d_lut = (float*)dalloc(lut.size() * 4, false);
  uint16_t* d_st = (uint16_t*)dalloc(32768 * 2, false);
  uint16_t* d_sq = (uint16_t*)dalloc(32768 * 2, false);
  if (!d_lut || !d_st || !d_sq) { set_err("hipMalloc failed for tables"); cmx_mixnet_destroy(h); return nullptr; }
  hipMemcpy(d_lut, lut.data(), lut.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(d_st, t_st.data(), 32768 * 2, hipMemcpyHostToDevice);
  hipMemcpy(d_sq, t_sq.data(), 32768 * 2, hipMemcpyHostToDevice);
  S.logit_lut = d_lut;
  S.t_st = d_st;
  S.t_sq = d_sq;
  S.stretch_min = lut[0];        // Logit(0), ref mixer-input.cpp:5
  S.stretch_max = lut[100000];   // Logit(1)
  memcpy(S.lr, kLr, sizeof kLr);
  ALLOC(rows0, float, (size_t)CMX_MIX0 * CMX_ROWS_PER_MIXER * CMX_ROW0_STRIDE, true);
  ALLOC(rows1, float, (size_t)CMX_MIX1 * CMX_ROWS_PER_MIXER * CMX_ROW1_STRIDE, true);
  ALLOC(rows2, float, (size_t)CMX_ROWS_PER_MIXER * CMX_ROW2_STRIDE, true);
  ALLOC(row_steps, uint64_t, (size_t)CMX_MIXERS * CMX_ROWS_PER_MIXER, true);
  ALLOC(map_keys, uint32_t, (size_t)CMX_MIXERS * CMX_MAP_SLOTS, true);
  ALLOC(map_vals, uint32_t, (size_t)CMX_MIXERS * CMX_MAP_SLOTS, true);
  ALLOC(s6, uint16_t, (size_t)CMX_SM6_VOL * 8, false);
  ALLOC(s7, uint16_t, (size_t)CMX_SM7_VOL * 8, false);
  ALLOC(x1, int, CMX_MIX1_VOL, false);
  ALLOC(x2, int, CMX_MIX2_VOL, false);
#undef ALLOC
  for (int i = 0; i <= CMX_MIXERS; ++i) S.max_steps[i] = 1;  // ref mixer.cpp:13
  S.steps = 0;
  S.sse_j = 1;  // ref sse.cpp:226
  S.sse_pc = 0;
  S.sse_ffl = 0;
  for (int i = 0; i <= CMX_MIXERS; ++i) S.fwd_p[i] = 0.5f;
  h->d_state = (MixState*)dalloc(sizeof(MixState), false);
  if (!h->d_state) { set_err("hipMalloc failed for state"); cmx_mixnet_destroy(h); return nullptr; }
  hipMemcpy(h->d_state, &S, sizeof S, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(cmx_sse_init_kernel, dim3(2048), dim3(256), 0, 0, h->d_state);
  h->d_sync_probs = (float*)dalloc(CMX_IN0 * 4, true);
  h->d_sync_sel = (uint32_t*)dalloc(CMX_MIXERS * 4, true);
  h->d_sync_bit = (uint8_t*)dalloc(16, true);
  h->d_sync_p = (float*)dalloc(16, true);
  if (hipFuncSetAttribute((const void*)cmx_mixnet_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)kLdsBytes) != hipSuccess) {
    set_err("hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed");
    cmx_mixnet_destroy(h);
    return nullptr;
  }
  if (hipFuncSetAttribute((const void*)cmx_mixnet_chunk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                          CMX_CHUNK_LDS_BYTES) != hipSuccess) {
    set_err("hipFuncSetAttribute(MaxDynamicSharedMemorySize, chunk kernel) failed");
    cmx_mixnet_destroy(h);
    return nullptr;
  }
  { const char* v = getenv("CMX_MIXNET_V1"); h->use_v1 = v && v[0] == '1'; }
  { const char* v = getenv("CMX_MIXNET_SPEC"); h->use_spec = !(v && v[0] == '0'); }
  h->d_xfer = (SpecXfer*)dalloc(sizeof(SpecXfer), true);   // incl. the zero padding of the input ring
  bool attr_ok = true;
  for (const void* k : {(const void*)cmx_mixnet_spec_kernel, (const void*)cmx_mixnet_spec_jitter_kernel, (const void*)cmx_mixnet_spec_late_kernel})
    attr_ok = attr_ok && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, CMX_SPEC_LDS_BYTES) == hipSuccess;
  if (!h->d_xfer || !attr_ok) {
    set_err("cmx_mixnet_create: hand-off area / kernel attribute (spec kernel) failed");
    cmx_mixnet_destroy(h);
    return nullptr;
  }
  { const char* v = getenv("CMX_MIXNET_DBG"); h->dbg = v ? atoi(v) : 0; }
  { const char* v = getenv("CMX_MIXNET_JITTER"); h->jitter = v ? atoi(v) & 15 : 0; }
  { const char* v = getenv("CMX_MIXNET_ROTATE"); h->rotate = v && v[0] == '1'; }
  { const char* v = getenv("CMX_MIXNET_XCD"); h->xcd = v ? atoi(v) : CMX_MIXNET_XCD_DEFAULT; if (h->xcd > 7) h->xcd = -1; }
  hipEventCreate(&h->ev0);
  hipEventCreate(&h->ev1);
  for (int i = 0; i < CMX_PIPELINE_SLOTS; ++i) { hipEventCreateWithFlags(&h->ev_decay[i], hipEventDisableTiming); hipEventCreateWithFlags(&h->ev_kdone[i], hipEventDisableTiming); }
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { set_err(std::string("init: ") + hipGetErrorString(e)); cmx_mixnet_destroy(h); return nullptr; }
  return h;
}

// Mixer::Perceive's global schedule, ref mixer.cpp:58: a pure function of the
// mixer's step counter, evaluated with the host's double pow().
static inline float decay_of(uint64_t steps) {
  return (float)(0.9 / pow(0.0000001 * steps + 0.8, 0.8));
}

static int ensure_decay(cmx_mixnet_t* h, size_t nbits) {
  const int fail_value = 1;
  if (nbits <= h->decay_cap) return 0;
  size_t cap = nbits < 4096 ? 4096 : nbits;
  if (h->h_decay) hipHostFree(h->h_decay);
  h->h_decay = nullptr;
  for (int i = 0; i < CMX_PIPELINE_SLOTS; ++i) {  // nothing of the old buffers may still be waiting to be copied or read
    if (h->decay_used[i]) { (void)hipEventSynchronize(h->ev_decay[i]); (void)hipEventSynchronize(h->ev_kdone[i]); }
    h->decay_used[i] = false;
  }
  HIP_OK(hipHostMalloc((void**)&h->h_decay, (size_t)CMX_PIPELINE_SLOTS * cap * 4, hipHostMallocDefault));
  void* p = nullptr;
  HIP_OK(hipMalloc(&p, (size_t)CMX_PIPELINE_SLOTS * cap * 4));   // one device slot per host slot (slot 0 doubles as the bit-synchronous scalar)
  h->allocs.push_back(p);
  h->d_decay = (float*)p;
  h->decay_cap = cap;
  return 0;
}

// Speculation statistics of cmx_mixnet_spec_kernel since the handle was created: out[0] speculative segments run, [1] resolved from a
// candidate lane, [2..4] re-runs (misses) of segment 1, 2, 3. Synchronises the device.
int cmx_mixnet_spec_stats(cmx_mixnet_t* h, uint64_t out[5]) {
  const int fail_value = 1;
  if (!h || !out) { set_err("cmx_mixnet_spec_stats: bad argument"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipDeviceSynchronize());
  unsigned long long st[8];
  HIP_OK(hipMemcpy(st, (char*)h->d_xfer + offsetof(SpecXfer, stat), sizeof st, hipMemcpyDeviceToHost));
  for (int i = 0; i < 5; ++i) out[i] = st[i];
  return 0;
}

// A HIP stream for a stage's kernels. With the mixing network placed on one XCD through a compute-unit mask (CMX_MIXNET_XCD=k CMX_CUMASK=1) the network's
// stream may use ONLY that XCD's compute units and every other stage's stream everything BUT them: a stage whose workgroups land on the crowded XCD is held
// up there (profiles/r05_xcd_placement.txt: the LSTM's block kernels 4.7 -> 6.5 us/bit next to 27 spinning workgroups, whichever XCD). Bit i of the mask is
// compute unit i / 8 of XCC i % 8 (the KFD spreads a queue's mask over the XCCs bit by bit). which: 0 = any other stage, 1 = the mixing network.
static bool g_cumask_applied = false;   // at least one masked stream was really created (the kernels' placement flag follows THIS, not the environment: advisor, round 5)
static bool cumask_wanted() {
  static const char* const xe = getenv("CMX_MIXNET_XCD");
  // (ignored in the compact stream modes, CMX_PIPELINE_STREAMS: there roles share the mixing network's stream, and a one-XCD mask would confine them to the
  // compute units its 27 spinning workgroups already occupy)
  static const bool w = getenv("CMX_CUMASK") != nullptr && xe && atoi(xe) >= 0 && atoi(xe) < 8 && getenv("CMX_PIPELINE_STREAMS") == nullptr;
  return w;
}
int cmx_make_stream(hipStream_t* st, int which) {
  static const char* const xe = getenv("CMX_MIXNET_XCD");
  const bool masked = cumask_wanted();
  if (masked) {
    const int x = atoi(xe);
    uint32_t m[8];
    for (int w = 0; w < 8; ++w) {
      m[w] = 0;
      for (int b = 0; b < 32; ++b) { const int cu = 32 * w + b; if (((cu & 7) == x) == (which == 1)) m[w] |= 1u << b; }
    }
    if (hipExtStreamCreateWithCUMask(st, 8, m) == hipSuccess) { if (which == 1) g_cumask_applied = true; return 0; }
    (void)hipGetLastError();
  }
  return hipStreamCreateWithFlags(st, hipStreamNonBlocking) == hipSuccess ? 0 : 1;
}
int cmx_cumask_on(void) { return cumask_wanted() && g_cumask_applied; }   // the mixing network's stream really carries its one-XCD mask

// The stream the handle's host-to-device copies go on (the pipeline gives all its stages ONE upload stream that never has
// a kernel in front of a copy); without it the handle creates its own on first use.
// Tolerance mode (NOT bit-exact; north_star's "per-bit probabilities within a tolerance"): the layer-0 dot products as f64 tree sums
// rounded once. An explicit switch of the handle, never an environment variable: a file coded with it has the reference's header
// but cannot be decoded by the reference or by this library's decoder (which is strict), so no file-writing program sets it.
int cmx_mixnet_set_tolerance(cmx_mixnet_t* h, int on) {
  if (!h) { set_err("cmx_mixnet_set_tolerance: null handle"); return 1; }
  if (on && (!h->use_spec || h->use_v1)) { set_err("cmx_mixnet_set_tolerance: only the 27-workgroup kernel has the mode (CMX_MIXNET_SPEC=0 / CMX_MIXNET_V1 are set)"); return 1; }
  if (h->runs || h->bits_done) { set_err("cmx_mixnet_set_tolerance: only before the first bit of the stream"); return 1; }
  h->tolerance = on != 0;
  return 0;
}
// DEVICE address of MixState::error (set by a chunk kernel whose bounded in-launch wait ran out), for callers that copy it back in
// stream order behind the chunk's kernel (cmx_pipeline_finish) instead of synchronising the device
const int* cmx_mixnet_error_flag(cmx_mixnet_t* h) { return h ? &h->d_state->error : nullptr; }
// diagnostics: rows allocated so far by each of the 47 mixers (Mixer::GetContextData's context_map_.size(); the cap is 10 000 + the
// shared overflow row, mixer.cpp:16-36). Synchronises the device.
int cmx_mixnet_rows(cmx_mixnet_t* h, uint32_t rows[47]) {
  if (!h || !rows) { set_err("cmx_mixnet_rows: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { set_err("cmx_mixnet_rows: device error"); return 1; }
  return hipMemcpy(rows, (const char*)h->d_state + offsetof(MixState, n_rows), 47 * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
uint64_t cmx_mixnet_runs(cmx_mixnet_t* h) { return h ? h->runs : 0; }   // chunks launched so far
// test hook (CMX_LATE_DEBUG=1): DEVICE address of the 47 mixer outputs of bit `bit` of the late chunk launched as number `chunk` of this handle
const float* cmx_mixnet_late_debug_mix(cmx_mixnet_t* h, uint64_t chunk, size_t bit) {
  if (!h || !h->d_late_mix) return nullptr;
  return h->d_late_mix + (size_t)(chunk % 3) * h->late_p_cap * CMX_MIXERS + bit * CMX_MIXERS;
}
// 0 strict (bit-exact, the default), 1 tolerance
int cmx_mixnet_mode(cmx_mixnet_t* h) { return h && h->tolerance ? 1 : 0; }

int cmx_mixnet_set_upload_stream(cmx_mixnet_t* h, void* stream) {
  if (!h) { set_err("cmx_mixnet_set_upload_stream: null handle"); return 1; }
  if (h->own_up && h->s_up) hipStreamDestroy(h->s_up);
  h->s_up = (hipStream_t)stream; h->own_up = false;
  return 0;
}

static int mixnet_run_impl(cmx_mixnet_t* h, const float* d_probs, const uint32_t* d_sel, const uint8_t* d_bits, size_t nbits, float* d_p_out, float* d_mix_out,
                           void* stream, const CmxLate* box);
int cmx_mixnet_run(cmx_mixnet_t* h, const float* d_probs, const uint32_t* d_sel,
                   const uint8_t* d_bits, size_t nbits, float* d_p_out, float* d_mix_out,
                   void* stream) {
  return mixnet_run_impl(h, d_probs, d_sel, d_bits, nbits, d_p_out, d_mix_out, stream, nullptr);
}
// The decoder's form of a chunk (cmx_late.h): rows and selectors (memory the producing kernels and this one see coherently) are
// consumed as their stages count them in `box`; p(t) goes to the box, bit t comes back through it. Only the 27-workgroup kernel.
int cmx_mixnet_run_late(cmx_mixnet_t* h, void* box, const float* probs, const uint32_t* sel, size_t nbits, void* stream) {
  if (!h || !box) { set_err("cmx_mixnet_run_late: bad argument"); return 1; }
  if (!h->use_spec || h->use_v1 || h->tolerance) { set_err("cmx_mixnet_run_late: a decoder needs the strict 27-workgroup kernel (CMX_MIXNET_SPEC=0, CMX_MIXNET_V1 or the tolerance switch is set)"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { set_err("hipSetDevice failed"); return 1; }
  if (h->late_p_cap < nbits || h->decay_cap < nbits) { set_err("cmx_mixnet_run_late: call cmx_mixnet_late_prepare first (nothing may be allocated while the stream's kernels run)"); return 1; }
  // (debug: three chunk-sized areas of mixer outputs in rotation, like the pipeline's buffer sets)
  float* mix = h->d_late_mix ? h->d_late_mix + (size_t)(h->runs % 3) * h->late_p_cap * CMX_MIXERS : nullptr;
  h->late_mix_cur[h->runs % 3] = mix;
  return mixnet_run_impl(h, probs, sel, nullptr, nbits, h->d_late_p, mix, stream, (const CmxLate*)box);
}
// everything the decoder's form allocates, for chunks of up to nbits bits: before the first chunk's kernels are launched
int cmx_mixnet_late_prepare(cmx_mixnet_t* h, size_t nbits) {
  if (!h || !nbits) { set_err("cmx_mixnet_late_prepare: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { set_err("hipSetDevice failed"); return 1; }
  if (h->late_p_cap < nbits) {
    void* p = nullptr;
    if (hipMalloc(&p, nbits * 4) != hipSuccess) { set_err("cmx_mixnet_late_prepare: hipMalloc failed"); return 1; }
    h->allocs.push_back(p);
    h->d_late_p = (float*)p; h->late_p_cap = nbits;
    const char* dbg = getenv("CMX_LATE_DEBUG");
    if (dbg && dbg[0] == '1') {
      if (hipMalloc(&p, 3 * nbits * CMX_MIXERS * 4) != hipSuccess) { set_err("cmx_mixnet_late_prepare: hipMalloc failed"); return 1; }
      h->allocs.push_back(p);
      h->d_late_mix = (float*)p;
    }
  }
  if (!h->s_up) { if (hipStreamCreateWithFlags(&h->s_up, hipStreamNonBlocking) != hipSuccess) { set_err("cmx_mixnet_late_prepare: stream creation failed"); return 1; } h->own_up = true; }
  return ensure_decay(h, nbits);
}
static int mixnet_run_impl(cmx_mixnet_t* h, const float* d_probs, const uint32_t* d_sel, const uint8_t* d_bits, size_t nbits, float* d_p_out, float* d_mix_out,
                           void* stream, const CmxLate* box) {
  const int fail_value = 1;
  if (!h) { set_err("cmx_mixnet_run: null handle"); return 1; }
  if (h->predicted) { set_err("cmx_mixnet_run: a bit-synchronous predict() is pending"); return 1; }
  if (nbits == 0) return 0;
  if (nbits > 0x7fffffff) { set_err("cmx_mixnet_run: chunk too large"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  hipStream_t st = (hipStream_t)stream;
  // One in-order stream per handle: the decay schedule goes through ONE device buffer and the kernel trains the handle's
  // state, so chunks enqueued on different streams would race on both.
  if (h->run_stream_set && st != h->run_stream) { set_err("cmx_mixnet_run: every chunk of a handle must be enqueued on the same stream"); return 1; }
  h->run_stream = st; h->run_stream_set = true;
  // The decay schedule is staged through one of CMX_PIPELINE_SLOTS pinned host slots, each with its own device slot, and
  // copied on the UPLOAD stream: a copy enqueued on the kernel's stream sits behind the previous chunk's persistent kernel,
  // and on this hardware a host-to-device copy that waits in stream order holds up every later copy of the process (the
  // other stages' uploads of chunks further ahead: their kernels then start a mixing-network period late).
  if (ensure_decay(h, nbits)) return 1;
  if (!h->s_up) { HIP_OK(hipStreamCreateWithFlags(&h->s_up, hipStreamNonBlocking)); h->own_up = true; }
  const int slot = (int)(h->runs++ % CMX_PIPELINE_SLOTS);
  if (h->decay_used[slot]) HIP_OK(hipEventSynchronize(h->ev_kdone[slot]));   // the kernel that read this slot (implies its copy)
  float* hd = h->h_decay + (size_t)slot * h->decay_cap;
  float* dd = h->d_decay + (size_t)slot * h->decay_cap;
  for (size_t t = 0; t < nbits; ++t) hd[t] = decay_of(h->bits_done + t);
  HIP_OK(hipMemcpyAsync(dd, hd, nbits * 4, hipMemcpyHostToDevice, h->s_up));
  HIP_OK(hipEventRecord(h->ev_decay[slot], h->s_up));
  HIP_OK(hipStreamWaitEvent(st, h->ev_decay[slot], 0));
  h->decay_used[slot] = true;
  HIP_OK(hipEventRecord(h->ev0, st));
  if (h->use_v1)
    hipLaunchKernelGGL(cmx_mixnet_kernel, dim3(1), dim3(CMX_MIXNET_THREADS), kLdsBytes, st, h->d_state,
                       d_probs, d_sel, d_bits, dd, (int)nbits, d_p_out, d_mix_out,
                       3 | (h->profile ? 4 : 0));
  else if (h->use_spec) {
    // epochs and value|tag words restart at 0 with every launch; 1 main + 26 helper workgroups, co-resident (27 of 256 CUs)
    HIP_OK(hipMemsetAsync(h->d_xfer, 0, CMX_SPEC_HEADER_BYTES, st));
    const bool cumask = cmx_cumask_on() != 0;   // the stream's compute-unit mask does the placement: 27 workgroups, all of them work, the XCC census still decides the hand-off's form
    const int rot = h->rotate && h->xcd < 0 ? (int)((h->runs * 3) & 7) : 0;   // (h->runs was advanced above: the launch's number + 1; x 3: not the neighbouring XCD every time)
    const int kmode = 3 | (rot << 8) | (h->profile ? 4 : 0) | ((h->dbg & 15) << 4) | (h->tolerance ? 0x1000 : 0) | (h->xcd >= 0 ? 0x20000 | ((h->xcd & 7) << 20) : 0) |
                      (getenv("CMX_MIXNET_XCD_NOLOCAL") ? 0x800000 : 0) | (cumask ? 0x1000000 : 0) | (h->jitter ? 0x2000000 | ((h->jitter & 15) << 26) : 0);
    static const bool padgrid = getenv("CMX_MIXNET_PADGRID") != nullptr;   // diagnostic: the 8 x 27 grid of the one-XCD placement without the placement (blocks 27.. leave at once)
    const unsigned grid = (1 + CMX_SPEC_HELPERS) * ((h->xcd >= 0 && !cumask) || padgrid ? 8 : 1) + (unsigned)rot;   // placement: 8 x 27 workgroups, those with blockIdx % 8 == xcd work
    if (box && box->box)   // a decoder's chunk: the patient instantiation of the same roles
      hipLaunchKernelGGL(cmx_mixnet_spec_late_kernel, dim3(grid), dim3(CMX_SPEC_THREADS), CMX_SPEC_LDS_BYTES, st,
                         h->d_state, h->d_xfer, d_probs, d_sel, dd, (int)nbits, d_p_out, d_mix_out, kmode, *box);
    else {
      hipLaunchKernelGGL(h->jitter ? cmx_mixnet_spec_jitter_kernel : cmx_mixnet_spec_kernel, dim3(grid), dim3(CMX_SPEC_THREADS), CMX_SPEC_LDS_BYTES, st,
                         h->d_state, h->d_xfer, d_probs, d_sel, d_bits, dd, (int)nbits, d_p_out, d_mix_out, kmode);
    }
  } else
    // XCD placement (observed: block b runs on XCD b % 8): 8 blocks, all but block `xcd` leave at once
    hipLaunchKernelGGL(cmx_mixnet_chunk_kernel, dim3(h->xcd >= 0 ? 8 : 1), dim3(CMX_CHUNK_THREADS), CMX_CHUNK_LDS_BYTES, st,
                       h->d_state, d_probs, d_sel, d_bits, dd, (int)nbits, d_p_out, d_mix_out,
                       3 | (h->profile ? 4 : 0) | ((h->dbg & 15) << 4) | (h->xcd >= 0 ? ((h->xcd & 7) + 1) << 8 : 0));
  HIP_OK(hipGetLastError());
  HIP_OK(hipEventRecord(h->ev_kdone[slot], st));
  HIP_OK(hipEventRecord(h->ev1, st));
  h->timed = true;
  h->bits_done += nbits;
  return 0;
}

float cmx_mixnet_predict(cmx_mixnet_t* h, const float* probs, const uint32_t* sel) {
  const float fail_value = -1.0f;
  if (!h) { set_err("cmx_mixnet_predict: null handle"); return -1.0f; }
  if (h->predicted) { set_err("cmx_mixnet_predict: called twice without perceive()"); return -1.0f; }
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipMemcpy(h->d_sync_probs, probs, CMX_IN0 * 4, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h->d_sync_sel, sel, CMX_MIXERS * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(cmx_mixnet_kernel, dim3(1), dim3(CMX_MIXNET_THREADS), kLdsBytes, 0, h->d_state,
                     h->d_sync_probs, h->d_sync_sel, h->d_sync_bit, h->d_decay, 1, h->d_sync_p,
                     (float*)nullptr, 1);
  HIP_OK(hipGetLastError());
  float p = -1.0f;
  HIP_OK(hipMemcpy(&p, h->d_sync_p, 4, hipMemcpyDeviceToHost));
  h->predicted = true;
  return p;
}

int cmx_mixnet_perceive(cmx_mixnet_t* h, int bit) {
  const int fail_value = 1;
  if (!h) { set_err("cmx_mixnet_perceive: null handle"); return 1; }
  if (!h->predicted) { set_err("cmx_mixnet_perceive: no pending predict()"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (ensure_decay(h, 1)) return 1;
  uint8_t b = bit ? 1 : 0;
  float d = decay_of(h->bits_done);
  HIP_OK(hipMemcpy(h->d_sync_bit, &b, 1, hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(h->d_decay, &d, 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(cmx_mixnet_kernel, dim3(1), dim3(CMX_MIXNET_THREADS), kLdsBytes, 0, h->d_state,
                     h->d_sync_probs, h->d_sync_sel, h->d_sync_bit, h->d_decay, 1, h->d_sync_p,
                     (float*)nullptr, 2);
  HIP_OK(hipGetLastError());
  HIP_OK(hipDeviceSynchronize());
  h->predicted = false;
  h->bits_done += 1;
  return 0;
}

// Bit-synchronous mode with DEVICE operands, asynchronous on `stream` (engine_api.hip drives these): same kernel,
// same protocol; the operands must stay untouched until the matching perceive has run.
int cmx_mixnet_predict_async(cmx_mixnet_t* h, const float* d_probs, const uint32_t* d_sel, float* d_p, void* stream) {
  const int fail_value = 1;
  if (!h || !d_probs || !d_sel || !d_p) { set_err("cmx_mixnet_predict_async: bad argument"); return 1; }
  if (h->predicted) { set_err("cmx_mixnet_predict_async: called twice without perceive()"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  hipLaunchKernelGGL(cmx_mixnet_kernel, dim3(1), dim3(CMX_MIXNET_THREADS), kLdsBytes, (hipStream_t)stream, h->d_state,
                     d_probs, d_sel, h->d_sync_bit, (const float*)(h->d_sync_bit + 4), 1, d_p, (float*)nullptr, 1);
  HIP_OK(hipGetLastError());
  h->pend_probs = d_probs;
  h->pend_sel = d_sel;
  h->predicted = true;
  return 0;
}

int cmx_mixnet_perceive_async(cmx_mixnet_t* h, int bit, void* stream) {
  const int fail_value = 1;
  if (!h) { set_err("cmx_mixnet_perceive_async: null handle"); return 1; }
  if (!h->predicted || !h->pend_probs) { set_err("cmx_mixnet_perceive_async: no pending predict_async()"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  if (!h->h_sync_pin) HIP_OK(hipHostMalloc((void**)&h->h_sync_pin, 16 * 8, hipHostMallocDefault));
  // the caller synchronises at least once per bit (it needs p), so 16 slots can never wrap onto a pending copy
  unsigned char* slot = h->h_sync_pin + 8 * (h->sync_slot++ & 15);
  slot[0] = bit ? 1 : 0;
  const float d = decay_of(h->bits_done);
  memcpy(slot + 4, &d, 4);
  hipStream_t st = (hipStream_t)stream;
  HIP_OK(hipMemcpyAsync(h->d_sync_bit, slot, 8, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(cmx_mixnet_kernel, dim3(1), dim3(CMX_MIXNET_THREADS), kLdsBytes, st, h->d_state, h->pend_probs,
                     h->pend_sel, h->d_sync_bit, (const float*)(h->d_sync_bit + 4), 1, h->d_sync_p, (float*)nullptr, 2);
  HIP_OK(hipGetLastError());
  h->predicted = false;
  h->pend_probs = nullptr;
  h->bits_done += 1;
  return 0;
}

int cmx_mixnet_profile(cmx_mixnet_t* h, int enable, uint64_t* out16) {
  const int fail_value = 1;
  if (!h) { set_err("cmx_mixnet_profile: null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipDeviceSynchronize());
  MixState tmp;
  HIP_OK(hipMemcpy(&tmp, h->d_state, sizeof tmp, hipMemcpyDeviceToHost));
  if (out16) memcpy(out16, tmp.prof, sizeof tmp.prof);
  memset(tmp.prof, 0, sizeof tmp.prof);
  HIP_OK(hipMemcpy((char*)h->d_state + offsetof(MixState, prof), tmp.prof, sizeof tmp.prof, hipMemcpyHostToDevice));
  h->profile = enable;
  return 0;
}

// Test hook (state injection; tests/golden/make_wrap_traces.py, the twin of the reference harness's ref_debug_set_mixer_steps): the network as after `steps`
// bits -- Mixer::steps_ of all 47 mixers (mixer.cpp:58,61: the argument of the decay schedule). Between chunks only.
int cmx_mixnet_debug_set_steps(cmx_mixnet_t* h, uint64_t steps) {
  const int fail_value = 1;
  if (!h) { set_err("cmx_mixnet_debug_set_steps: null handle"); return 1; }
  if (h->predicted) { set_err("cmx_mixnet_debug_set_steps: a bit-synchronous predict() is pending"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipDeviceSynchronize());
  HIP_OK(hipMemcpy((char*)h->d_state + offsetof(MixState, steps), &steps, 8, hipMemcpyHostToDevice));
  h->bits_done = steps;
  return 0;
}

int cmx_mixnet_bits_done(const cmx_mixnet_t* h, uint64_t* out) {
  if (!h || !out) { set_err("cmx_mixnet_bits_done: null argument"); return 1; }
  *out = h->bits_done;
  return 0;
}

int cmx_mixnet_sync(cmx_mixnet_t* h) {
  const int fail_value = 1;
  if (!h) { set_err("cmx_mixnet_sync: null handle"); return 1; }
  HIP_OK(hipSetDevice(h->device));
  HIP_OK(hipDeviceSynchronize());
  int err = 0;
  HIP_OK(hipMemcpy(&err, (char*)h->d_state + offsetof(MixState, error), 4, hipMemcpyDeviceToHost));
  if (err) { set_err("cmx_mixnet: a device-side wait timed out (kernel aborted); state is invalid"); return 1; }
  return 0;
}

int cmx_mixnet_last_kernel_ms(cmx_mixnet_t* h, float* ms) {
  const int fail_value = 1;
  if (!h || !ms) { set_err("cmx_mixnet_last_kernel_ms: null argument"); return 1; }
  if (!h->timed) { set_err("cmx_mixnet_last_kernel_ms: no chunk has been run"); return 1; }
  HIP_OK(hipEventSynchronize(h->ev1));
  HIP_OK(hipEventElapsedTime(ms, h->ev0, h->ev1));
  return 0;
}

int cmx_probe_libm(int device, int which, const float* x, float* y, size_t n) {
  const int fail_value = 1;
  if (cmx_device_count() <= 0) { set_err("cmx_probe_libm: no HIP device"); return 1; }
  HIP_OK(hipSetDevice(device));
  float *dx = nullptr, *dy = nullptr;
  HIP_OK(hipMalloc((void**)&dx, n * 4));
  HIP_OK(hipMalloc((void**)&dy, n * 4));
  HIP_OK(hipMemcpy(dx, x, n * 4, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(cmx_probe_libm_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, 0, which,
                     dx, dy, n);
  HIP_OK(hipGetLastError());
  HIP_OK(hipMemcpy(y, dy, n * 4, hipMemcpyDeviceToHost));
  hipFree(dx);
  hipFree(dy);
  return 0;
}

}  // extern "C"
