// lstm_api.hip -- host side of the LSTM byte-mixer stage (C ABI: cmx_lstm_* in include/cmix_amd.h).
//
// Initialisation follows the reference constructor chain exactly: glibc rand() after
// srand(0xDEADBEEF) (predictor.cpp:26) -- re-implemented here (TYPE_3 additive-feedback generator,
// glibc stdlib/random_r.c) so the library never touches the process-global rand() state -- with the
// 31 draws of the Indirect models constructed before the LSTM skipped (indirect.cpp:10), then
// lstm-layer.cpp:52-59. Adam's per-step scalars (lstm-layer.cpp:14-30: sqrtf / powf / double pow)
// come from the HOST libm as a 3001-entry table.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stddef.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/cmix_amd.h"
#include "mixnet_state.h"   // CMX_MIXNET_XCD_DEFAULT: the XCD the block kernels leave to the mixing network
#include "lstm_state.h"
#include "cmx_late.h"
#include "cmx_glibc_rand.h"

extern "C" __global__ void cmx_lstm_prep(const LstmState, const float*, const uint8_t*, size_t, int, int);
extern "C" __global__ void cmx_lstm_bptt_acc(const LstmState, int, int, int, int);
extern "C" __global__ void cmx_lstm_bptt_gb(const LstmState, int, int);
extern "C" __global__ void cmx_lstm_bptt_acc_mfma(const LstmState, int, int);
extern "C" __global__ void cmx_lstm_fwdblk(const LstmState, const uint8_t*, const float*, float*, size_t, int, int, int);
extern "C" __global__ void cmx_lstm_fwdblk_late(const LstmState, CmxLate, const float*, const float*, float*, int, int, int, int, int);
extern "C" __global__ void cmx_lstm_bpttblk(const LstmState);
extern "C" __global__ void cmx_bytemodel_bits(const float*, const float*, const uint8_t*, size_t, float*, int*, size_t, int,
                                              float*);
extern "C" __global__ void cmx_bytemodel_late_kernel(CmxLate, size_t, const float*, const float*, const float*, const float*, const float*, const uint32_t*, uint32_t,
                                                     const uint32_t*, uint32_t, float*, size_t, int16_t*, uint8_t*, uint8_t*, const cmx_late_relay_t*, int);
extern "C" __global__ void cmx_late_bump_kernel(uint32_t*, uint32_t, uint32_t*, uint32_t);

void cmx_set_err(const std::string& s);  // cmx_api.hip


struct cmx_lstm {
  int device = 0;
  LstmState h_state;   // passed to every kernel by value
  std::vector<void*> allocs;
  float* d_prev_probs = nullptr;  // byte distribution at chunk start
  uint64_t bytes_done = 0;
  int hc = 0;                // which hid[] / stateb[] buffer holds the current hidden_ / state_
  size_t fb_lds = 0, bp_lds = 0;   // dynamic LDS of the block kernels (lstm_block.hip)
  uint64_t bptt_rounds = 0;  // LstmLayer::update_steps_ = min(rounds, 3000) (lstm-layer.cpp:131-133)
  bool tolerance = false;    // cmx_lstm_set_tolerance: the weight-update contraction on the matrix cores (NOT bit-exact)
};


extern "C" {

void cmx_lstm_destroy(cmx_lstm_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->allocs) (void)hipFree(p);
  delete h;
}

int cmx_glibc_rand_selftest(uint32_t seed, int n, int* out) {  // test hook: first n draws
  GlibcRand g(seed);
  for (int i = 0; i < n; ++i) out[i] = g.next();
  return 0;
}

cmx_lstm_t* cmx_lstm_create(const uint8_t vocab[256], int skip_rand, int device) {
  int ndev = cmx_device_count();
  if (ndev <= 0) { cmx_set_err("cmx_lstm_create: no HIP device visible (a gfx950 GPU is required)"); return nullptr; }
  if (device < 0 || device >= ndev) { cmx_set_err("cmx_lstm_create: bad device index"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return nullptr; }
  cmx_lstm_t* h = new cmx_lstm();
  h->device = device;
  LstmState& S = h->h_state;
  memset(&S, 0, sizeof S);
  int V = 0;
  for (int i = 0; i < 256; ++i) {
    S.vocab[i] = vocab[i] != 0;
    S.byte_map[i] = V;
    if (S.vocab[i]) ++V;
  }
  if (V == 0) { cmx_set_err("cmx_lstm_create: empty vocabulary"); delete h; return nullptr; }
  S.V = V;
  S.insz[0] = 1 + LSTM_C + V;
  S.insz[1] = V + 1 + 2 * LSTM_C;
  S.rowlen[0] = S.insz[0] + V;
  S.rowlen[1] = S.insz[1] + V;
  S.lr = 0.03f;
  { const char* v = getenv("CMX_LSTM_XCD"); S.xcd = v ? atoi(v) : -1; }
  { const char* v = getenv("CMX_LSTM_SLEEP"); S.poll_sleep = v && v[0] == '1'; }
  { const char* v = getenv("CMX_LSTM_AVOID_XCD"); if (!v) v = getenv("CMX_MIXNET_XCD"); S.avoid_xcd = v ? atoi(v) : CMX_MIXNET_XCD_DEFAULT; if (S.avoid_xcd > 7) S.avoid_xcd = -1; }   // the mixing network's XCD (mixnet_state.h)
  bool fail = false;
  auto dallocf = [&](size_t count, const float* init) -> float* {
    void* p = nullptr;
    if (hipMalloc(&p, count * 4) != hipSuccess) { fail = true; return nullptr; }
    h->allocs.push_back(p);
    if (init) (void)hipMemcpy(p, init, count * 4, hipMemcpyHostToDevice);
    else (void)hipMemset(p, 0, count * 4);
    return (float*)p;
  };
  // weights, reference draw order: per cell i, per column j: forget, input node, output (lstm-layer.cpp:52-59)
  GlibcRand rng(0xDEADBEEFu);
  for (int i = 0; i < skip_rand; ++i) rng.next();
  for (int l = 0; l < LSTM_L; ++l) {
    const int rl = S.rowlen[l];
    std::vector<float> w[3], wt[3];
    for (int g = 0; g < 3; ++g) { w[g].resize((size_t)LSTM_C * rl); wt[g].resize((size_t)LSTM_C * rl); }
    const float val = sqrtf(6.0f / (float)(V + V));
    const float low = -val, range = 2 * val;
    for (int i = 0; i < LSTM_C; ++i) {
      for (int j = 0; j < rl; ++j)
        for (int g = 0; g < 3; ++g) w[g][(size_t)i * rl + j] = low + ((float)rng.next() / (float)2147483647) * range;
      w[0][(size_t)i * rl + rl - 1] = 1;
    }
    for (int g = 0; g < 3; ++g) {
      for (int i = 0; i < LSTM_C; ++i)
        for (int j = 0; j < rl; ++j) wt[g][lstm_wt_index(V, S.insz[l], j, i)] = w[g][(size_t)i * rl + j];
      S.W[l][g] = dallocf((size_t)LSTM_C * rl, w[g].data());
      S.WT[l][g] = dallocf((size_t)LSTM_C * rl, wt[g].data());
      S.M[l][g] = dallocf((size_t)LSTM_C * rl, nullptr);
      S.Vv[l][g] = dallocf((size_t)LSTM_C * rl, nullptr);
      std::vector<float> gb(8 * LSTM_C, 0.0f);
      for (int i = 0; i < LSTM_C; ++i) gb[i] = 1.0f;  // gamma_ (lstm-layer.h:12)
      S.gb[l][g] = dallocf(8 * LSTM_C, gb.data());
      S.norm[l][g] = dallocf((size_t)LSTM_H * LSTM_C, nullptr);
      S.gstate[l][g] = dallocf((size_t)LSTM_H * LSTM_C, nullptr);
      S.ivar[l][g] = dallocf(LSTM_H, nullptr);
      S.raw[l][g] = dallocf(LSTM_C, nullptr);
      S.E[l][g] = dallocf((size_t)LSTM_H * LSTM_C, nullptr);
    }
    S.last_state[l] = dallocf((size_t)LSTM_H * LSTM_C, nullptr);
    S.tanh_state[l] = dallocf((size_t)LSTM_H * LSTM_C, nullptr);
    S.in_gate_state[l] = dallocf((size_t)LSTM_H * LSTM_C, nullptr);
    S.state[l] = dallocf(LSTM_C, nullptr);
    S.stateb[0][l] = S.state[l];
    S.stateb[1][l] = dallocf(LSTM_C, nullptr);
    std::vector<float> li((size_t)LSTM_H * S.insz[l], 0.0f);
    for (int e = 0; e < LSTM_H; ++e) li[(size_t)e * S.insz[l] + S.insz[l] - 1] = 1.0f;  // lstm.cpp:21-23
    S.layer_input[l] = dallocf(li.size(), li.data());
  }
  S.OL = dallocf((size_t)LSTM_H * V * LSTM_NH, nullptr);
  S.OLT = nullptr;   // (the transposed output-layer copy of the retired one-workgroup forward kernel)
  {
    std::vector<float> out((size_t)LSTM_H * LSTM_VP, 0.0f);
    for (int e = 0; e < LSTM_H; ++e)
      for (int i = 0; i < V; ++i) out[(size_t)e * LSTM_VP + i] = (float)(1.0 / V);  // lstm.cpp:16
    S.output = dallocf(out.size(), out.data());
  }
  {
    std::vector<float> hid(LSTM_NH, 0.0f);
    hid[LSTM_NH - 1] = 1.0f;  // lstm.cpp:18
    S.hid[0] = dallocf(LSTM_NH, hid.data());
    S.hid[1] = dallocf(LSTM_NH, hid.data());
  }
  S.input_history = (unsigned*)dallocf(LSTM_H, nullptr);
  S.bp_symbol = (unsigned*)dallocf(LSTM_H, nullptr);
  S.logits = dallocf(LSTM_VP, nullptr);
  {
    // Adam scalars per update step t (lstm-layer.cpp:14-30), host libm.
    std::vector<float> tab(4 * (LSTM_UPDATE_LIMIT + 1), 0.0f);
    const float beta1 = 0.025f, beta2 = 0.9999f, lr = S.lr;
    const unsigned long long limit = LSTM_UPDATE_LIMIT;
    for (unsigned long long s = 0; s <= limit; ++s) {
      float t = (float)s, alpha, b1, b2;
      if (t < limit) {
        alpha = lr * 0.1f / sqrtf(5e-5f * t + 1.0f);
        b1 = (float)(1.0f - powf(beta1, t));
        b2 = (float)(1.0f - powf(beta2, t));
      } else {
        alpha = lr * 0.1f / sqrtf(5e-5f * limit + 1.0f);
        b1 = (float)(1.0f - pow(beta1, limit));
        b2 = (float)(1.0f - pow(beta2, limit));
      }
      tab[4 * s] = alpha;
      tab[4 * s + 1] = b1;
      tab[4 * s + 2] = b2;
    }
    S.adam_tab = dallocf(tab.size(), tab.data());
  }
  {
    std::vector<float> bp(256, (float)(1.0 / 256));  // ByteModel ctor, byte-model.cpp:5-6
    S.byte_probs = dallocf(256, bp.data());
    h->d_prev_probs = dallocf(256, bp.data());
  }
  S.dyn = (int*)dallocf(4, nullptr);
  S.sync = (LstmSync*)dallocf((sizeof(LstmSync) + 3) / 4, nullptr);
  S.raw_ring = dallocf((size_t)LSTM_L * LSTM_H * 3 * LSTM_C, nullptr);
  S.h_ring = dallocf((size_t)LSTM_H * LSTM_NH, nullptr);
  S.logit_ring = dallocf((size_t)LSTM_H * LSTM_VP, nullptr);
  S.bp_pub = dallocf((size_t)2 * LSTM_H * 2 * LSTM_C, nullptr);
  {
    // dynamic LDS of the block kernels (the carve-up is in lstm_block.hip)
    const size_t nq1 = (size_t)(S.insz[1] + 3) / 4;
    const size_t gate = nq1 * LSTM_FB_R * 16 + (836 + 3 * LSTM_C + 4 + 256) * 4;
    const size_t ro = (size_t)(V + LSTM_FB_GO - 1) / LSTM_FB_GO, rp = ro | 1;
    const size_t outl = (size_t)((LSTM_NH + 3) / 4) * rp * 16 + (2 * 404 + 256 + 256 + 36 + 36 + 4) * 4;
    h->fb_lds = gate > outl ? gate : outl;
    h->bp_lds = (size_t)9 * 50 * LSTM_BP_J * 16 +
                ((size_t)3 * 256 * LSTM_BP_J + 3 * 256 + 3 * LSTM_C + 3 * LSTM_C + 4 + 6 * LSTM_BP_J) * 4;
    if (hipFuncSetAttribute((const void*)cmx_lstm_fwdblk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->fb_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)cmx_lstm_fwdblk_late, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->fb_lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)cmx_lstm_bpttblk, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bp_lds) != hipSuccess)
      fail = true;
  }
  S.blk = (LstmBlockArgs*)dallocf((sizeof(LstmBlockArgs) + 3) / 4 + 4, nullptr);
  if (fail) {
    cmx_set_err("cmx_lstm_create: hipMalloc failed");
    cmx_lstm_destroy(h);
    return nullptr;
  }
  if (hipDeviceSynchronize() != hipSuccess) { cmx_set_err("cmx_lstm_create: init failed"); cmx_lstm_destroy(h); return nullptr; }
  return h;
}

int cmx_lstm_vocab_size(const cmx_lstm_t* h) { return h ? h->h_state.V : -1; }

int cmx_lstm_run(cmx_lstm_t* h, const float* d_in_probs, const uint8_t* d_bytes, size_t nbytes,
                 float* d_out_probs, float* d_bit_p, size_t bit_p_stride, int* d_bit_ex, void* stream) {
  if (!h) { cmx_set_err("cmx_lstm_run: null handle"); return 1; }
  if (nbytes == 0) return 0;
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipStream_t st = (hipStream_t)stream;
  const LstmState& S = h->h_state;
  const int V = S.V;
  if (d_bit_p) (void)hipMemcpyAsync(h->d_prev_probs, S.byte_probs, 256 * 4, hipMemcpyDeviceToDevice, st);   // (the distribution going into the run: only ByteModel's bit predictions below read it)
  if (!d_in_probs) { cmx_set_err("cmx_lstm_run: d_in_probs is required"); return 1; }
  auto sync_reset = [&]() {
    return hipMemsetAsync((char*)S.sync + 16, 0, sizeof(LstmSync) - 16, st) == hipSuccess;
  };
  for (size_t n = 0; n < nbytes;) {
    const int e = (int)(h->bytes_done % LSTM_H);   // Lstm::epoch_
    const int hc = h->hc;                          // which hid[] buffer holds hidden_
    int us = 0;
    if (e == 0) {
      h->bptt_rounds += 1;
      us = (int)(h->bptt_rounds < LSTM_UPDATE_LIMIT ? h->bptt_rounds : LSTM_UPDATE_LIMIT);
    }
    // multi-workgroup kernels (lstm_block.hip): [bookkeeping, BPTT round] at an epoch-0 byte, then ONE launch for the
    // bytes up to the end of the block
    if (e == 0) {                                // lstm.cpp:93
      hipLaunchKernelGGL(cmx_lstm_prep, dim3(1), dim3(256), 0, st, S, d_in_probs + n * 256, d_bytes, n, e, -1);
      if (!sync_reset()) { cmx_set_err("cmx_lstm_run: hipMemsetAsync failed"); return 1; }
      hipLaunchKernelGGL(cmx_lstm_bpttblk, dim3(lstm_grid_for_roles(LSTM_BP_G, S.avoid_xcd)), dim3(LSTM_BP_THREADS), h->bp_lds, st, S);
      if (h->tolerance) hipLaunchKernelGGL(cmx_lstm_bptt_acc_mfma, dim3((S.rowlen[1] + 15) / 16, (LSTM_C + 15) / 16, 6), dim3(64), 0, st, S, us, -1);
      else { const int gx = (S.rowlen[1] + 63) / 64, gy = LSTM_C / 4; hipLaunchKernelGGL(cmx_lstm_bptt_acc, dim3(lstm_grid_for_roles(gx * gy * 6, S.avoid_xcd)), dim3(64, 4), 0, st, S, us, -1, gx, gy); }
      hipLaunchKernelGGL(cmx_lstm_bptt_gb, dim3(6), dim3(256), 0, st, S, us, -1);
    }
    const size_t left = nbytes - n;
    const int cnt = (int)(left < (size_t)(LSTM_H - e) ? left : (size_t)(LSTM_H - e));
    if (!sync_reset()) { cmx_set_err("cmx_lstm_run: hipMemsetAsync failed"); return 1; }
    hipLaunchKernelGGL(cmx_lstm_fwdblk, dim3(lstm_grid_for_roles(2 * LSTM_FB_GL + LSTM_FB_GO, S.avoid_xcd)), dim3(LSTM_FB_THREADS), h->fb_lds, st, S, d_bytes,
                       d_in_probs, d_out_probs, n, cnt, e, hc);
    h->hc ^= 1;
    h->bytes_done += cnt;
    n += cnt;
  }
  if (d_bit_p) {
    if (!d_out_probs) { cmx_set_err("cmx_lstm_run: bit predictions need d_out_probs"); return 1; }
    hipLaunchKernelGGL(cmx_bytemodel_bits, dim3((unsigned)nbytes), dim3(64), 0, st, h->d_prev_probs, d_out_probs,
                       d_bytes, nbytes, d_bit_p, d_bit_ex, bit_p_stride ? bit_p_stride : (size_t)1, -1, (float*)nullptr);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_lstm_run: ") + hipGetErrorString(e)); return 1; }
  return 0;
}

// The decoder's form of a run (cmx_late.h; round 6): bytes b0 .. of a decoder's chunk, of which only byte b0 is known (its value and the byte models' distribution
// after it: HOST pointers h_in256 / byte0, for the bookkeeping and the BPTT round an epoch-0 byte begins with, lstm.cpp:93). ONE forward launch covers the bytes up to
// the end of the truncated-BPTT block or of the chunk (nchunk bytes), whichever comes first: its steps wait for their bytes inside the launch
// (cmx_lstm_fwdblk_late), read PPMd's distributions from the relay's device mirror d_ppmd ([nchunk + 1][256], row b + 1 = after byte b; the chunk's last byte:
// from the host's own array h_ppmd, same layout, device-mapped) and count every distribution that goes out into d_out ([nchunk][256]) on LC_LSTM.
// Returns the number of bytes the launch covers (>= 1), or -1. The caller calls again when those bytes have been decoded.
int cmx_lstm_run_late(cmx_lstm_t* h, const void* late_box, const float* d_ppmd, const float* h_ppmd, const uint8_t* h_bytes, float* d_out, size_t b0, size_t nchunk, void* stream) {
  if (!h || !late_box || !d_ppmd || !h_ppmd || !h_bytes || !d_out || b0 >= nchunk) { cmx_set_err("cmx_lstm_run_late: bad argument"); return -1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return -1; }
  if (h->tolerance) { cmx_set_err("cmx_lstm_run_late: a decoder is strict"); return -1; }
  hipStream_t st = (hipStream_t)stream;
  const LstmState& S = h->h_state;
  const int e = (int)(h->bytes_done % LSTM_H);
  const int hc = h->hc;
  auto sync_reset = [&]() { return hipMemsetAsync((char*)S.sync + 16, 0, sizeof(LstmSync) - 16, st) == hipSuccess; };
  if (e == 0) {   // lstm.cpp:93: the BPTT round over the block that has just ended, with the bookkeeping of this byte in front of it (device-mapped host rows)
    h->bptt_rounds += 1;
    const int us = (int)(h->bptt_rounds < LSTM_UPDATE_LIMIT ? h->bptt_rounds : LSTM_UPDATE_LIMIT);
    hipLaunchKernelGGL(cmx_lstm_prep, dim3(1), dim3(256), 0, st, S, h_ppmd + (b0 + 1) * 256, h_bytes, b0, e, -1);
    if (!sync_reset()) { cmx_set_err("cmx_lstm_run_late: hipMemsetAsync failed"); return -1; }
    hipLaunchKernelGGL(cmx_lstm_bpttblk, dim3(lstm_grid_for_roles(LSTM_BP_G, S.avoid_xcd)), dim3(LSTM_BP_THREADS), h->bp_lds, st, S);
    { const int gx = (S.rowlen[1] + 63) / 64, gy = LSTM_C / 4; hipLaunchKernelGGL(cmx_lstm_bptt_acc, dim3(lstm_grid_for_roles(gx * gy * 6, S.avoid_xcd)), dim3(64, 4), 0, st, S, us, -1, gx, gy); }
    hipLaunchKernelGGL(cmx_lstm_bptt_gb, dim3(6), dim3(256), 0, st, S, us, -1);
  }
  const size_t left = nchunk - b0;
  const int cnt = (int)(left < (size_t)(LSTM_H - e) ? left : (size_t)(LSTM_H - e));
  if (!sync_reset()) { cmx_set_err("cmx_lstm_run_late: hipMemsetAsync failed"); return -1; }
  hipLaunchKernelGGL(cmx_lstm_fwdblk_late, dim3(lstm_grid_for_roles(2 * LSTM_FB_GL + LSTM_FB_GO, S.avoid_xcd)), dim3(LSTM_FB_THREADS), h->fb_lds, st, S, *(const CmxLate*)late_box,
                     d_ppmd, h_ppmd, d_out, (int)b0, cnt, e, hc, (int)nchunk);
  h->hc ^= 1;
  h->bytes_done += cnt;
  const hipError_t er = hipGetLastError();
  if (er != hipSuccess) { cmx_set_err(std::string("cmx_lstm_run_late: ") + hipGetErrorString(er)); return -1; }
  return cnt;
}

// 1 = a bounded in-launch wait of a block kernel ran out (the stream's LSTM results are void); synchronises the device
/* TOLERANCE mode of the stage (NOT bit-exact; measurement only -- no file-writing tool turns it on): the BPTT round's weight-update
 * contraction runs as v_mfma_f32_16x16x4_f32 tiles (cmx_lstm_bptt_acc_mfma) instead of the reference's ordered, separately rounded chain */
int cmx_lstm_set_tolerance(cmx_lstm_t* h, int on) {
  if (!h) { cmx_set_err("cmx_lstm_set_tolerance: null handle"); return 1; }
  if (h->bytes_done && h->tolerance != (on != 0)) { cmx_set_err("cmx_lstm_set_tolerance: only before the first byte (a stream is strict or tolerant from its start)"); return 1; }
  h->tolerance = on != 0;
  return 0;
}
int cmx_lstm_failed(cmx_lstm_t* h) {
  if (!h) return 1;
  (void)hipSetDevice(h->device);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  unsigned f = 0;
  if (hipMemcpy(&f, &h->h_state.sync->fail, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  return f ? 1 : 0;
}

// DEVICE address of the sticky flag above, for callers that copy it back in stream order behind the stage's kernels
// (4 bytes, asynchronously) instead of synchronising the device: the per-bit surface and cmx_pipeline_wait
const unsigned* cmx_lstm_fail_flag(cmx_lstm_t* h) { return h ? &h->h_state.sync->fail : nullptr; }

int cmx_bytemodel_bits_run(int device, const float* d_dist0, const float* d_dist_rest, const uint8_t* d_bytes,
                           size_t nbytes, float* d_bit_p, size_t bit_p_stride, int* d_bit_ex, void* stream) {
  if (nbytes == 0) return 0;
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipLaunchKernelGGL(cmx_bytemodel_bits, dim3((unsigned)nbytes), dim3(64), 0, (hipStream_t)stream, d_dist0,
                     d_dist_rest, d_bytes, nbytes, d_bit_p, d_bit_ex, bit_p_stride ? bit_p_stride : (size_t)1, -1,
                     (float*)nullptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_bytemodel_bits_run: ") + hipGetErrorString(e)); return 1; }
  return 0;
}

// The decoder's form (cmx_late.h): ByteModel::Predict / Perceive of the stream's three byte-level distributions bit by bit as the
// bits arrive through `box` -> layer-0 columns 0, 2076, 2077 and the fxcm stage's LSTM hints (see cmx_bytemodel_late_kernel).
int cmx_bytemodel_late_run(int device, void* box, size_t nbytes, const float* brk0, const float* brk, const float* ppmd, const float* lstm0, const float* lstm,
                           const uint32_t* c0_brk, uint32_t c0_brk_want, const uint32_t* c0_lstm, uint32_t c0_lstm_want, float* layer0, size_t pstride,
                           int16_t* hint_pr, uint8_t* hint_ex, uint8_t* dbit0, const void* relay_dev, int nrelay, void* stream) {
  if (!box || !nbytes || !brk0 || !brk || !ppmd || !lstm0 || !lstm || !layer0 || !hint_pr || !hint_ex || !dbit0 || nrelay < 0 || nrelay > CMX_LATE_RELAY_MAX || (nrelay && !relay_dev)) {
    cmx_set_err("cmx_bytemodel_late_run: bad argument");
    return 1;
  }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipLaunchKernelGGL(cmx_bytemodel_late_kernel, dim3(1), dim3(192 + 64 * 6), 0, (hipStream_t)stream, *(const CmxLate*)box, nbytes, brk0, brk, ppmd, lstm0, lstm, c0_brk, c0_brk_want,
                     c0_lstm, c0_lstm_want, layer0, pstride, hint_pr, hint_ex, dbit0, (const cmx_late_relay_t*)relay_dev, nrelay);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_bytemodel_late_run: ") + hipGetErrorString(e)); return 1; }
  return 0;
}
// one system-scope store behind everything already in `stream`: *counter = value (and *counter2 = value2 when given)
int cmx_late_bump(int device, uint32_t* counter, uint32_t value, uint32_t* counter2, uint32_t value2, void* stream) {
  if (!counter) { cmx_set_err("cmx_late_bump: bad argument"); return 1; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipLaunchKernelGGL(cmx_late_bump_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counter, value, counter2, value2);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
// the distribution the LSTM byte mixer holds between bytes (DEVICE memory, [256]; uniform until the first byte)
const float* cmx_lstm_byte_probs(cmx_lstm_t* h) { return h ? h->h_state.byte_probs : nullptr; }

// Bit-synchronous mode: ByteModel::Predict for bit `k` (0..7) of the byte whose top k bits are the coded ones:
// d_bit_p[k * stride] (and *d_p_copy, may be NULL) <- the value, d_bit_ex[k] <- `ex` when given.
int cmx_bytemodel_bit_run(int device, const float* d_dist, const uint8_t* d_byte, int k, float* d_bit_p,
                          size_t bit_p_stride, int* d_bit_ex, float* d_p_copy, void* stream) {
  if (!d_dist || !d_byte || !d_bit_p || k < 0 || k > 7) { cmx_set_err("cmx_bytemodel_bit_run: bad argument"); return 1; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipLaunchKernelGGL(cmx_bytemodel_bits, dim3(1), dim3(64), 0, (hipStream_t)stream, d_dist, d_dist, d_byte, (size_t)1,
                     d_bit_p, d_bit_ex, bit_p_stride ? bit_p_stride : (size_t)1, k, d_p_copy);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_bytemodel_bit_run: ") + hipGetErrorString(e)); return 1; }
  return 0;
}

// test hook: copy gate weights (reference layout [200][rowlen]) back to the host
int cmx_lstm_get_gate_weights(cmx_lstm_t* h, int layer, int gate, float* out) {
  if (!h) return 1;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  size_t n = (size_t)LSTM_C * h->h_state.rowlen[layer];
  return hipMemcpy(out, h->h_state.W[layer][gate], n * 4, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
int cmx_lstm_gate_rowlen(const cmx_lstm_t* h, int layer) { return h ? h->h_state.rowlen[layer] : -1; }

}  // extern "C"
