// lstm_block.hip -- the byte-level LSTM of one stream spread over many compute units (gfx950).
//
// Reference: Lstm::Perceive/Predict (src/mixer/lstm.cpp:87-150), LstmLayer::ForwardPass/BackwardPass
// (src/mixer/lstm-layer.cpp:62-99, :108-183), ByteMixer::ByteUpdate (src/mixer/byte-mixer.cpp:22-38).
//
// Strict mode as in lstm_kernels.hip: every dot product is an ordered f32 chain, so the parallelism is ACROSS
// chains. One workgroup streams ~50 GB/s, and the 2.8 MB of weights a byte touches made the one-workgroup forward
// kernel a 41 us/byte bandwidth problem. Here the weights stay where the chains run:
//
//  cmx_lstm_fwdblk  one launch per run of <= 100 bytes (one truncated-BPTT block). 12 workgroups per gate layer
//    hold 50 gate rows each in LDS (<= 132 KB), 8 workgroups hold the output layer's rows (and do its per-byte SGD
//    in LDS). The three groups form a pipeline over the bytes: layer 0 of byte t+1 overlaps layer 1 of byte t
//    overlaps output layer / softmax of byte t-1. Inside a group the only exchange is an all-gather of the raw
//    gate sums (600 floats) per byte; every workgroup then does the RMS norm and the cell update redundantly, so
//    a hand-off costs one store -> counter -> poll round trip and nothing else. The parts of a chain that do not
//    depend on the recurrence (one-hot column, the V symbol inputs, for layer 1 its own previous hidden) run
//    before the wait.
//  cmx_lstm_bpttblk the sequential part of BPTT (100 epochs x 2 layers) on 20 workgroups: each owns 10 cells,
//    keeps the recurrent weight columns of those cells in LDS (72 KB) and computes their W^T chains, hidden-error
//    chains and clipped errors; the per-cell results are all-gathered once per (epoch, layer) step, the
//    element-wise / normalisation phases are replicated in every workgroup.
//
// Hand-offs: data that crosses workgroups inside a launch is written and read with agent-scope atomics (they go
// to the coherence point, never a stale line), the producer waits for its stores (s_waitcnt vmcnt(0)) before it
// moves the counter. Every wait is bounded: a counter that does not arrive sets LstmSync::fail and the launch
// drains (cmx_lstm_failed reports it) instead of hanging the GPU.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cmx_libm.h"
#include "lstm_state.h"
#include "cmx_late.h"

namespace {

__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }
__device__ __forceinline__ float fdiv(float a, float b) { return __fdiv_rn(a, b); }
__device__ __forceinline__ float fsqrt(float a) { return __builtin_sqrtf(a); }

constexpr int C = LSTM_C, H = LSTM_H, NH = LSTM_NH, VP = LSTM_VP;
constexpr int GL = LSTM_FB_GL, R = LSTM_FB_R, GO = LSTM_FB_GO, FT = LSTM_FB_THREADS;
constexpr int GB = LSTM_BP_G, J = LSTM_BP_J, BT = LSTM_BP_THREADS;
constexpr unsigned SPIN_LIMIT = 1u << 23;

__device__ __forceinline__ unsigned ld_u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_f(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_f(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// a workgroup barrier that orders LDS traffic only: __syncthreads() also drains vmcnt(0), i.e. waits for every global load the
// wave has in flight -- the prefetches issued at the top of a step would be awaited at its first barrier instead of at their use
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// lane 0 polls the counter, the workgroup waits at the barrier
__device__ __forceinline__ void wg_wait(const unsigned* p, unsigned want, unsigned* fail, int sleepy) {
  if (threadIdx.x == 0) {
    unsigned it = 0;
    while (ld_u(p) < want) {
      if (sleepy) __builtin_amdgcn_s_sleep(1);   // A/B (CMX_LSTM_SLEEP): a poll every ~64 clocks instead of back to back
      if ((++it & 1023u) == 0 && (it > SPIN_LIMIT || ld_u(fail))) {
        __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        break;
      }
    }
  }
  lds_barrier();
}
// The decoder's form of a forward block (cmx_lstm_fwdblk_late): the launch covers bytes that do not exist yet, so every in-launch wait may last as long as the
// decoder (host) takes -- bounded by the box's abort / fail words and 30 s of wall-clock time (cmx_late.h), not by a spin count. A wait that ends that way sets
// LstmSync::fail: every later wait of the launch then falls through and the kernel drains.
template <bool LATE> __device__ __forceinline__ void wg_wait_t(const unsigned* p, unsigned want, unsigned* fail, int sleepy, CmxLateBox* B) {
  if (!LATE) { wg_wait(p, want, fail, sleepy); return; }
  if (threadIdx.x == 0 && ld_u(p) < want) {
    unsigned it = 0;
    unsigned long long t0 = 0;
    for (;;) {
      __builtin_amdgcn_s_sleep(1);
      if (ld_u(p) >= want) break;
      if ((++it & 255u) == 0) {
        bool out = ld_u(fail) || late_ld(&B->abort) || late_ld(&B->fail);
        const unsigned long long now = wall_clock64();
        if (!t0) t0 = now;
        else if (now - t0 > late_timeout_ticks(B)) { late_st(&B->fail, 1u); out = true; }
        if (out) { __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
      }
    }
  }
  lds_barrier();
}
// the decoder's byte b of the chunk (bit 8 b + 7 decoded, the host stages' records of the step behind it in place): thread 0 waits for the relay's count,
// the workgroup at the barrier; then every thread assembles the byte from the device mirror of the bits. -1: aborted / timed out (LstmSync::fail is set).
__device__ __forceinline__ int late_byte(const CmxLate& LT, int b, unsigned* fail, int* flag_s) {
  if (threadIdx.x == 0) {
    const bool ok = !ld_u(fail) && late_wait_cnt(LT, LC_KNOWN, (uint32_t)(8 * (b + 1) + 1));
    if (!ok) __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag_s = ok ? 1 : 0;
  }
  lds_barrier();
  if (!*flag_s) return -1;
  int v = 0;
#pragma unroll
  for (int j = 0; j < 8; ++j) v = (v << 1) | (int)(*(volatile const uint8_t*)(LT.dbit0 + 8 * b + j) & 1);
  return v;
}

// the calling wave's agent-scope stores are complete, then the counter moves (callers: lanes of wave 0 only)
__device__ __forceinline__ void wave_signal(unsigned* p) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (threadIdx.x == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// every wave's stores are complete (all threads call)
__device__ __forceinline__ void wg_signal(unsigned* p) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float quad(float f, const float4 x, const float4 w) {
  f = fadd(f, fmul(x.x, w.x));
  f = fadd(f, fmul(x.y, w.y));
  f = fadd(f, fmul(x.z, w.z));
  return fadd(f, fmul(x.w, w.w));
}
// Ordered chain f += x[d] * w[d], d = d0 .. d1-1. Weights: LDS quads Wq[(d>>2) * stride + r] (term d in component
// d & 3); x: LDS vector, 16-byte aligned at d = 0. The next four quads are fetched while the current four are added.
__device__ __forceinline__ float lds_chain(float f, const float4* Wq, int stride, int r, const float* xv, int d0, int d1) {
  int d = d0;
  for (; d < d1 && (d & 3); ++d) f = fadd(f, fmul(xv[d], reinterpret_cast<const float*>(Wq + (size_t)(d >> 2) * stride + r)[d & 3]));
  const int nq = (d1 - d) >> 2;
  const float4* wp = Wq + (size_t)(d >> 2) * stride + r;
  const float4* xp = reinterpret_cast<const float4*>(xv + d);
  int q = 0;
  if (nq >= 4) {
    float4 w0[4], x0[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { w0[k] = wp[(size_t)k * stride]; x0[k] = xp[k]; }
    for (; q + 8 <= nq; q += 4) {
      float4 w1[4], x1[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) { w1[k] = wp[(size_t)(q + 4 + k) * stride]; x1[k] = xp[q + 4 + k]; }
#pragma unroll
      for (int k = 0; k < 4; ++k) f = quad(f, x0[k], w0[k]);
#pragma unroll
      for (int k = 0; k < 4; ++k) { w0[k] = w1[k]; x0[k] = x1[k]; }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) f = quad(f, x0[k], w0[k]);
    q += 4;
  }
  for (; q < nq; ++q) f = quad(f, xp[q], wp[(size_t)q * stride]);
  d += nq << 2;
  for (; d < d1; ++d) f = fadd(f, fmul(xv[d], reinterpret_cast<const float*>(Wq + (size_t)(d >> 2) * stride + r)[d & 3]));
  return f;
}

// ---------------------------------------------------------------------------------------------------------------
// forward block: a gate-layer workgroup
// ---------------------------------------------------------------------------------------------------------------
struct FbArgs {
  const uint8_t* bytes; const float* in_probs; float* out_probs;
  size_t n0; int cnt; int e0; int hc;
  // the decoder's form: n0 = the first byte's index IN ITS CHUNK; bytes come from the device mirror of the decoded bits, PPMd's distribution after byte b from
  // row b + 1 of its device mirror (the relay brings a row with the first step of the byte behind it) -- the chunk's last byte: from the host's own row
  CmxLate late; const float* d_ppmd; const float* h_ppmd; int nchunk;
};

// RMS norm + activations + cell update of layer `layer` at epoch e from the gathered raw sums (lstm-layer.cpp:62-83,
// :93-98), redundantly in every workgroup of the layer's group; the lead workgroup keeps the per-epoch caches BPTT
// reads and publishes the new hidden vector. xh <- the 200 new hidden values.
template <bool LATE> __device__ __forceinline__ void fb_finish_layer(const LstmState* S, int layer, int e, bool lead, float* rawl, float* ivar_s,
                                                float* xh, float& st, CmxLateBox* LB, const void* late_ptr = nullptr) {
  const int tid = threadIdx.x;
  LstmSync* Y = S->sync;
  wg_wait_t<LATE>(&Y->raw_cnt[layer][e], GL, &Y->fail, S->poll_sleep, LB);
  const float* rr = S->raw_ring + ((size_t)layer * H + e) * (3 * C);
  for (int idx = tid; idx < 3 * C; idx += FT) rawl[idx] = ld_f(rr + idx);
  lds_barrier();
  if (tid < 3) {  // (norm_*norm_).sum(): expression-template sum runs backward from the last element
    const float* rw = rawl + tid * C;
    float s = fmul(rw[C - 1], rw[C - 1]);
#pragma unroll 16
    for (int i = C - 2; i >= 0; --i) s = fadd(s, fmul(rw[i], rw[i]));
    const float iv = fdiv(1.0f, fsqrt(fadd(fdiv(s, (float)C), 1e-5f)));
    ivar_s[tid] = iv;
    if (lead) S->ivar[layer][tid][e] = iv;
  }
  lds_barrier();
  if (tid < C) {
    float stg[3];
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const float* gb = S->gb[layer][g];
      const float nrm = fmul(rawl[g * C + tid], ivar_s[g]);
      if (lead) S->norm[layer][g][(size_t)e * C + tid] = nrm;
      stg[g] = fadd(fmul(nrm, gb[tid]), gb[C + tid]);  // norm*gamma + beta
    }
    const float fg = cmx_logistic(stg[0]);
    const float inn = cmx_tanhf(stg[1]);
    const float og = cmx_logistic(stg[2]);
    const float igs = fsub(1.0f, fg);
    const float last = st;
    st = fadd(fmul(st, fg), fmul(inn, igs));
    const float th = cmx_tanhf(st);
    const float h = fmul(og, th);
    xh[tid] = h;
    if (lead) {
      const size_t ec = (size_t)e * C + tid;
      S->gstate[layer][0][ec] = fg;
      S->gstate[layer][1][ec] = inn;
      S->gstate[layer][2][ec] = og;
      S->last_state[layer][ec] = last;
      S->in_gate_state[layer][ec] = igs;
      S->tanh_state[layer][ec] = th;
      st_f(&S->h_ring[(size_t)e * NH + layer * C + tid], h);
    }
  }
  if (lead) wg_signal(&Y->h_flag[layer][e]);
  else lds_barrier();
  if (LATE && lead && tid == 0) late_stamp(*reinterpret_cast<const CmxLate*>(late_ptr), 7 + layer);   // diagnostics: this layer's hidden vector of the step is out
}

template <bool LATE> __device__ void fb_gate_wg(const LstmState* S, const FbArgs& A, int layer, int w, float* lds) {
  const int tid = threadIdx.x, V = S->V, insz = S->insz[layer];
  const bool lead = w == 0;
  LstmSync* Y = S->sync;
  const int nq = (insz + 3) >> 2, full = insz & ~3, nqf = full >> 2;
  float4* Wl = reinterpret_cast<float4*>(lds);          // [nq][R]: the workgroup's 50 gate rows, dense columns
  float* xv = reinterpret_cast<float*>(Wl + (size_t)nq * R);   // [836] the layer's input vector (lstm.cpp:120-131)
  float* rawl = xv + 836;                               // [600]
  float* ivar_s = rawl + 3 * C;                         // [4]
  int* sym2byte = reinterpret_cast<int*>(ivar_s + 4);   // [256]
  __shared__ int late_flag;
  CmxLateBox* const LB = LATE ? A.late.box : nullptr;
  const int row0 = w * R, g = row0 / C, i0 = row0 - g * C;   // 50 divides 200: one gate per workgroup
  const float* wt = S->WT[layer][g];
  for (int idx = tid; idx < nq * R; idx += FT) {
    const int q = idx / R, r = idx - q * R, i = i0 + r;
    float4 v;
    if (q < nqf) {
      v = *reinterpret_cast<const float4*>(wt + (size_t)V * C + ((size_t)q * C + i) * 4);
    } else {  // the insz % 4 trailing columns are stored [d][i] (lstm_wt_index)
      const float* t = wt + (size_t)V * C + (size_t)full * C + i;
      const int rem = insz - full;
      v.x = t[0]; v.y = rem > 1 ? t[C] : 0.0f; v.z = rem > 2 ? t[2 * C] : 0.0f; v.w = 0.0f;
    }
    Wl[idx] = v;
  }
  for (int b = tid; b < 256; b += FT)
    if (S->vocab[b]) sym2byte[S->byte_map[b]] = b;
  if (tid < C) xv[V + tid] = S->hid[A.hc][layer * C + tid];   // own previous hidden (lstm.cpp:122-124)
  if (tid == 0) xv[insz - 1] = 1.0f;                           // bias
  float st = tid < C ? S->stateb[A.hc][layer][tid] : 0.0f;     // LstmLayer::state_
  const int r = tid < R ? tid : R - 1;
  for (int k = 0; k < A.cnt; ++k) {
    const int e = A.e0 + k;
    const size_t n = A.n0 + k;
    // A decoder: byte k only exists once the distribution of byte k - 1 has gone out, which needs this layer's hidden vector of that step -- the previous
    // step is finished BEFORE the wait (a compressor finishes it under the first part of this step's chains, below: the same operations on the same operands
    // in another order of independent parts)
    if (LATE && k > 0) fb_finish_layer<LATE>(S, layer, e - 1, lead, rawl, ivar_s, xv + V, st, LB, &A.late);
    int byte_k;
    if (LATE) { byte_k = late_byte(A.late, (int)n, &Y->fail, &late_flag); if (byte_k < 0) byte_k = 0; }
    else byte_k = A.bytes[n];
    if (LATE && lead && layer == 0 && tid == 0) late_stamp(A.late, 6);   // diagnostics (scripts/gpu_late_time.py): the byte has arrived
    const int cur_sym = S->byte_map[byte_k];
    float* li = S->layer_input[layer] + (size_t)e * insz;
    lds_barrier();
    if (tid < V) {  // ByteMixer::SetInput / Lstm::SetInput (byte-mixer.cpp:15-20): inputs_ *= 2 / num_models_
      float pin;
      if (LATE) {
        const float* src = (int)n + 1 < A.nchunk ? A.d_ppmd + (n + 1) * 256 : A.h_ppmd + (n + 1) * 256;
        pin = __hip_atomic_load(src + sym2byte[tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else pin = A.in_probs[n * 256 + sym2byte[tid]];
      const float v = fmul(pin, 2.0f);
      xv[tid] = v;
      if (lead && e != 0) li[tid] = v;   // epoch 0: written by cmx_lstm_prep ahead of the BPTT round
    }
    if (lead && layer == 0 && tid == 0 && e != 0) {  // Lstm::Perceive bookkeeping (lstm.cpp:88-92)
      S->dyn[0] = (int)S->input_history[e - 1];
      S->input_history[e - 1] = (unsigned)cur_sym;
    }
    lds_barrier();
    float f = 0.0f;
    if (tid < 64) {  // LstmLayer::ForwardPass dot products (lstm-layer.cpp:85-92): one-hot column, then the symbol inputs
      f = wt[(size_t)cur_sym * C + i0 + r];
      f = lds_chain(f, Wl, R, r, xv, 0, V);
    }
    if (!LATE && k > 0) fb_finish_layer<LATE>(S, layer, e - 1, lead, rawl, ivar_s, xv + V, st, LB, &A.late);
    if (lead && tid < C) li[V + tid] = xv[V + tid];   // keep the assembled vector for BPTT
    if (tid < 64) f = lds_chain(f, Wl, R, r, xv, V, V + C);
    if (layer == 1) {  // layer 0's new hidden (lstm.cpp:127-131)
      wg_wait_t<LATE>(&Y->h_flag[0][e], 1, &Y->fail, S->poll_sleep, LB);
      if (tid < C) {
        const float h = ld_f(&S->h_ring[(size_t)e * NH + tid]);
        xv[V + C + tid] = h;
        if (lead) li[V + C + tid] = h;
      }
      lds_barrier();
    }
    if (tid < 64) {
      f = lds_chain(f, Wl, R, r, xv, V + C, insz);   // layer 1: layer 0's new hidden; then the bias
      if (tid < R) st_f(S->raw_ring + ((size_t)layer * H + e) * (3 * C) + row0 + tid, f);
      wave_signal(&Y->raw_cnt[layer][e]);
    }
  }
  lds_barrier();
  const int el = A.e0 + A.cnt - 1;
  fb_finish_layer<LATE>(S, layer, el, lead, rawl, ivar_s, xv + V, st, LB, &A.late);
  if (lead && tid < C) {
    S->stateb[A.hc ^ 1][layer][tid] = st;
    S->hid[A.hc ^ 1][layer * C + tid] = xv[V + tid];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward block: an output-layer workgroup. Rows i0 .. i0+nr-1 of output_layer_ live in LDS; per byte: the SGD step
// of lstm.cpp:112-116 on them (and the copy BPTT reads, output_layer_[epoch]), the 401-term chains (lstm.cpp:132-140),
// all-gather of the logits, softmax (lstm.cpp:141-149) redundantly, ByteMixer::ByteUpdate tail (byte-mixer.cpp:27-37).
// ---------------------------------------------------------------------------------------------------------------
template <bool LATE> __device__ void fb_out_wg(const LstmState* S, const FbArgs& A, int w, float* lds) {
  const int tid = threadIdx.x, V = S->V;
  const bool lead = w == 0;
  LstmSync* Y = S->sync;
  const int RO = (V + GO - 1) / GO, i0 = w * RO;
  const int nr = V - i0 < 0 ? 0 : (V - i0 < RO ? V - i0 : RO);
  const int RP = RO | 1;                     // odd row pitch: the element-wise pass walks q across lanes
  constexpr int NQ = (NH + 3) / 4;           // 101 quads; the last holds the bias column only
  float4* OLs = reinterpret_cast<float4*>(lds);                 // [NQ][RP]
  float* hbuf = reinterpret_cast<float*>(OLs + (size_t)NQ * RP);   // [2][404] hidden_ of the previous / current byte
  float* lgs = hbuf + 2 * 404;               // [256]
  float* red = lgs + 256;                    // [256]
  float* le_s = red + 256;                   // [36]
  float* outp = le_s + 36;                   // [36]  output_[last] of the own rows
  float* tot_s = outp + 36;                  // [4]
  __shared__ int late_flag;
  CmxLateBox* const LB = LATE ? A.late.box : nullptr;
  const int last0 = A.e0 == 0 ? H - 1 : A.e0 - 1;
  for (int idx = tid; idx < nr * NQ; idx += FT) {
    const int rr = idx / NQ, q = idx - rr * NQ, j = q * 4;
    const float* src = S->OL + ((size_t)last0 * V + i0 + rr) * NH + j;
    float4 v;
    v.x = src[0];
    v.y = j + 1 < NH ? src[1] : 0.0f; v.z = j + 2 < NH ? src[2] : 0.0f; v.w = j + 3 < NH ? src[3] : 0.0f;
    OLs[(size_t)q * RP + rr] = v;
  }
  for (int j = tid; j < NH; j += FT) hbuf[j] = S->hid[A.hc][j];
  if (tid < nr) outp[tid] = S->output[(size_t)last0 * VP + i0 + tid];
  int pb = 0;
  const int r = tid < nr ? tid : (nr > 0 ? nr - 1 : 0);
  for (int k = 0; k < A.cnt; ++k) {
    const int e = A.e0 + k;
    const size_t n = A.n0 + k;
    int byte_k;
    if (LATE) { byte_k = late_byte(A.late, (int)n, &Y->fail, &late_flag); if (byte_k < 0) byte_k = 0; }
    else byte_k = A.bytes[n];
    const int cur_sym = S->byte_map[byte_k];
    float* hprev = hbuf + pb * 404;
    float* hcur = hbuf + (pb ^ 1) * 404;
    lds_barrier();
    if (tid < nr) {
      const float o = outp[tid];
      const float err = (i0 + tid == cur_sym) ? fsub(o, 1.0f) : o;
      le_s[tid] = fmul(S->lr, err);
    }
    lds_barrier();
    for (int idx = tid; idx < nr * NQ; idx += FT) {   // slot[e] = slot[last] - (lr*err_i) * hidden_
      const int rr = idx / NQ, q = idx - rr * NQ, j = q * 4;
      const float le = le_s[rr];
      float4 v = OLs[(size_t)q * RP + rr];
      float* dst = S->OL + ((size_t)e * V + i0 + rr) * NH + j;
      v.x = fsub(v.x, fmul(le, hprev[j]));
      dst[0] = v.x;
      if (j + 3 < NH) {
        v.y = fsub(v.y, fmul(le, hprev[j + 1]));
        v.z = fsub(v.z, fmul(le, hprev[j + 2]));
        v.w = fsub(v.w, fmul(le, hprev[j + 3]));
        dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
      }
      OLs[(size_t)q * RP + rr] = v;
    }
    wg_wait_t<LATE>(&Y->h_flag[0][e], 1, &Y->fail, S->poll_sleep, LB);
    if (tid < C) hcur[tid] = ld_f(&S->h_ring[(size_t)e * NH + tid]);
    lds_barrier();
    float sum = 0.0f;
    if (tid < 64) sum = lds_chain(sum, OLs, RP, r, hcur, 0, C);
    wg_wait_t<LATE>(&Y->h_flag[1][e], 1, &Y->fail, S->poll_sleep, LB);
    if (tid < C) hcur[C + tid] = ld_f(&S->h_ring[(size_t)e * NH + C + tid]);
    if (tid == 0) hcur[2 * C] = 1.0f;   // bias element of hidden_ (lstm.cpp:18)
    lds_barrier();
    if (tid < 64) {
      sum = lds_chain(sum, OLs, RP, r, hcur, C, NH);
      if (tid < nr) st_f(&S->logit_ring[(size_t)e * VP + i0 + tid], sum);
      wave_signal(&Y->logit_cnt[e]);
    }
    wg_wait_t<LATE>(&Y->logit_cnt[e], GO, &Y->fail, S->poll_sleep, LB);
    float lg = 0.0f;
    if (tid < V) lg = ld_f(&S->logit_ring[(size_t)e * VP + tid]);
    red[tid] = tid < V ? lg : 0.0f;   // max_out starts at 0 (lstm.cpp:132)
    lds_barrier();
    for (int s = 128; s > 0; s >>= 1) {
      if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
      lds_barrier();
    }
    const float mx = red[0];
    if (tid < V) lgs[tid] = cmx_expf(fsub(lg, mx));
    lds_barrier();
    if (tid == 0) {  // valarray::sum(): forward from 0
      float t = 0.0f;
#pragma unroll 16
      for (int i = 0; i < V; ++i) t = fadd(t, lgs[i]);
      tot_s[0] = t;
    }
    lds_barrier();
    float p = 0.0f;
    if (tid < V) p = fdiv(lgs[tid], tot_s[0]);
    lds_barrier();
    if (tid < V) {
      lgs[tid] = p;
      if (lead) S->output[(size_t)e * VP + tid] = p;
    }
    lds_barrier();
    if (tid < nr) outp[tid] = lgs[i0 + tid];
    if (lead) {
      const float pbv = S->vocab[tid] ? lgs[S->byte_map[tid]] : 0.0f;
      S->byte_probs[tid] = pbv;
      if (A.out_probs) A.out_probs[n * 256 + tid] = pbv;
    }
    if (LATE && lead) {   // the distribution after byte n of the chunk is in place (uncached device memory): count it, as the one-thread kernel behind a per-byte launch did
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0 && !ld_u(&Y->fail)) { late_publish(A.late, LC_LSTM, (uint32_t)(n + 1)); late_stamp(A.late, 9); }
    }
    pb ^= 1;
  }
}

}  // namespace

extern "C" __global__ __launch_bounds__(LSTM_FB_THREADS) void cmx_lstm_fwdblk(const LstmState P, const uint8_t* bytes,
                                                                             const float* in_probs, float* out_probs,
                                                                             size_t n0, int cnt, int e0, int hc) {
  extern __shared__ float4 fb_lds[];
  FbArgs A;
  A.bytes = bytes; A.in_probs = in_probs; A.out_probs = out_probs; A.n0 = n0; A.cnt = cnt; A.e0 = e0; A.hc = hc;
  const int b = lstm_role_of_block((int)blockIdx.x, P.avoid_xcd);
  if (b < 0 || b >= 2 * GL + GO) return;
  A.late = CmxLate(); A.d_ppmd = nullptr; A.h_ppmd = nullptr; A.nchunk = 0;
  if (b < GL) fb_gate_wg<false>(&P, A, 0, b, reinterpret_cast<float*>(fb_lds));
  else if (b < 2 * GL) fb_gate_wg<false>(&P, A, 1, b - GL, reinterpret_cast<float*>(fb_lds));
  else fb_out_wg<false>(&P, A, b - 2 * GL, reinterpret_cast<float*>(fb_lds));
}

// The decoder's form (round 6): ONE launch for the bytes b0 .. b0 + cnt - 1 of a decoder's chunk (to the end of the truncated-BPTT block or of the chunk),
// none of which but the first exists when it starts -- every step waits for its byte (the relay's step count, cmx_late.h), the output group counts the
// distributions as they go out (LC_LSTM). Round 4 / 5 launched cmx_lstm_fwdblk once per byte: 32 workgroups reloading 4 MB of weights into LDS, ~120 us of
// every byte's first bit.
extern "C" __global__ __launch_bounds__(LSTM_FB_THREADS) void cmx_lstm_fwdblk_late(const LstmState P, CmxLate late, const float* d_ppmd, const float* h_ppmd,
                                                                                  float* out_probs, int b0, int cnt, int e0, int hc, int nchunk) {
  extern __shared__ float4 fb_lds[];
  FbArgs A;
  A.bytes = nullptr; A.in_probs = nullptr; A.out_probs = out_probs; A.n0 = (size_t)b0; A.cnt = cnt; A.e0 = e0; A.hc = hc;
  A.late = late; A.d_ppmd = d_ppmd; A.h_ppmd = h_ppmd; A.nchunk = nchunk;
  const int b = lstm_role_of_block((int)blockIdx.x, P.avoid_xcd);
  if (b < 0 || b >= 2 * GL + GO) return;
  if (b < GL) fb_gate_wg<true>(&P, A, 0, b, reinterpret_cast<float*>(fb_lds));
  else if (b < 2 * GL) fb_gate_wg<true>(&P, A, 1, b - GL, reinterpret_cast<float*>(fb_lds));
  else fb_out_wg<true>(&P, A, b - 2 * GL, reinterpret_cast<float*>(fb_lds));
}

// ---------------------------------------------------------------------------------------------------------------
// BPTT, sequential part (lstm.cpp:93-110; lstm-layer.cpp:108-183) on LSTM_BP_G workgroups. Step s = 0..199 is
// (epoch 99 - s/2, layer 1 - (s & 1)). Workgroup b owns cells b*10 .. b*10+9: the hidden-error chain of those cells
// (lstm.cpp:98-103) and the W^T chains that end in them (lstm-layer.cpp:164-181, weights in LDS); everything
// element-wise and the per-gate normalisation backward are replicated in every workgroup on all 200 cells. Exchange
// per step: [hidden error after the output-layer chain | clipped stored error] of every cell.
// ---------------------------------------------------------------------------------------------------------------
namespace {
// one (gate, cell) item of the normalisation-backward phase: 600 items on 512 threads, so threads 0..87 carry two.
// _c = the step's layer, _o = the other layer; the pairs swap at the end of every step.
struct BpItem {
  int g, c; bool on;
  float gam_c, gam_o, gu_c, gu_o, bu_c, bu_o;
  float norm, iv, err;
};
}  // namespace

extern "C" __global__ __launch_bounds__(LSTM_BP_THREADS) void cmx_lstm_bpttblk(const LstmState P) {
  extern __shared__ float4 bp_lds[];
  const LstmState* S = &P;
  LstmSync* Y = S->sync;
  const int tid = threadIdx.x, V = S->V, b = lstm_role_of_block((int)blockIdx.x, P.avoid_xcd), j0 = b * J;
  if (b < 0 || b >= GB) return;
  const bool lead = b == 0;
  // LDS
  float4* Wb = bp_lds;                                   // [9 slots][50 quads of j][J] : W[l][g][j][2V + kind*C + j0 + cl]
  float* olb = reinterpret_cast<float*>(Wb + 9 * 50 * J);   // [3][256][J] output-layer slices of steps s, s+1, s+2
  float* errs = olb + 3 * 256 * J;                       // [3][256]    softmax-CE error of those steps' epochs
  float* gerr = errs + 3 * 256;                          // [3][C]
  float* gprod = gerr + 3 * C;                           // [3][C]
  float* gsm = gprod + 3 * C;                            // [4]
  float* fres = gsm + 4;                                 // [2][3][J]
  // recurrent weight columns of the own cells: slot 0..2 = layer 0 kind 0 (gate), 3..5 = layer 1 kind 0, 6..8 = layer 1 kind 1
  for (int idx = tid; idx < 9 * 50 * J; idx += BT) {
    const int slot = idx / (50 * J), rem = idx - slot * 50 * J, q = rem / J, cl = rem - q * J;
    const int layer = slot < 3 ? 0 : 1, gg = slot % 3, kind = slot >= 6 ? 1 : 0;
    const int rl = S->rowlen[layer];
    const float* wsrc = S->W[layer][gg] + 2 * V + kind * C + j0 + cl;
    float4 v;
    v.x = wsrc[(size_t)(4 * q) * rl]; v.y = wsrc[(size_t)(4 * q + 1) * rl];
    v.z = wsrc[(size_t)(4 * q + 2) * rl]; v.w = wsrc[(size_t)(4 * q + 3) * rl];
    Wb[idx] = v;
  }
  // thread roles: cell j = tid (< 200); items (gate, cell) = tid and 512 + tid (< 600)
  float stored_c = 0.0f, stored_o = 0.0f, serr_c = 0.0f, serr_o = 0.0f;   // step 0 is layer 1
  BpItem ia, ib;
  auto item_init = [&](BpItem& it, int idx) {
    it.on = idx < 3 * C;
    const int id = it.on ? idx : 0;
    it.g = id / C; it.c = id - it.g * C;
    it.gam_o = S->gb[0][it.g][it.c]; it.gam_c = S->gb[1][it.g][it.c];
    it.gu_c = it.gu_o = it.bu_c = it.bu_o = 0.0f;
    it.norm = it.iv = it.err = 0.0f;
  };
  item_init(ia, tid);
  item_init(ib, BT + tid);
  // staging registers: output-layer slice and error vector of a step ahead
  constexpr int NSL = (256 * J + BT - 1) / BT;   // 5
  float olr[NSL];
  float errr = 0.0f;
  auto slice_load = [&](int s) {   // -> registers
    const int epoch = H - 1 - (s >> 1), layer = 1 - (s & 1);
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const int idx = tid + u * BT, i = idx / J, cl = idx - i * J;
      olr[u] = i < V ? S->OL[((size_t)epoch * V + i) * NH + layer * C + j0 + cl] : 0.0f;
    }
    if (tid < V) {
      const float o = S->output[(size_t)epoch * VP + tid];
      errr = ((unsigned)tid == S->input_history[epoch]) ? fsub(o, 1.0f) : o;
    }
  };
  auto slice_store = [&](int s) {  // registers -> LDS buffer s % 3
    float* ob = olb + (s % 3) * 256 * J;
#pragma unroll
    for (int u = 0; u < NSL; ++u) {
      const int idx = tid + u * BT;
      if (idx < 256 * J) ob[idx] = olr[u];
    }
    if (tid < V) errs[(s % 3) * 256 + tid] = errr;
  };
  // hidden_error_[j] += output_layer_[epoch][i][j+offset] * error_i, i ascending (lstm.cpp:98-103), own cells
  auto p1_chain = [&](int s, float h) -> float {
    const float* ob = olb + (s % 3) * 256 * J + (tid < J ? tid : 0);
    const float* ev = errs + (s % 3) * 256;
    int i = 0;
    for (; i + 8 <= V; i += 8) {
      float a[8], x[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) { a[k] = ob[(i + k) * J]; x[k] = ev[i + k]; }
#pragma unroll
      for (int k = 0; k < 8; ++k) h = fadd(h, fmul(a[k], x[k]));
    }
    for (; i < V; ++i) h = fadd(h, fmul(ob[i * J], ev[i]));
    return h;
  };
  slice_load(0); slice_store(0);
  slice_load(1); slice_store(1);
  slice_load(2);
  lds_barrier();
  float* pub = S->bp_pub;   // [200 steps][2][C]
  if (tid < 64) {  // step 0's hidden error: hidden_error_ is zero on entry
    const float h = p1_chain(0, 0.0f);
    if (tid < J) { st_f(pub + j0 + tid, h); st_f(pub + C + j0 + tid, 0.0f); }
    wave_signal(&Y->bp_cnt[0]);
  }
  float hpre = 0.0f;
#pragma unroll 1
  for (int s = 0; s < 2 * H; ++s) {
    const int epoch = H - 1 - (s >> 1), layer = 1 - (s & 1);
    // staging: slice s+2 -> LDS, slice s+3 -> registers
    if (s + 2 < 2 * H) slice_store(s + 2);
    if (s + 3 < 2 * H) slice_load(s + 3);
    // this step's caches, in flight during the wait
    float th = 0, os = 0, fs = 0, is = 0, igs = 0, ls = 0;
    if (tid < C) {
      const size_t ec = (size_t)epoch * C + tid;
      th = S->tanh_state[layer][ec]; os = S->gstate[layer][2][ec]; fs = S->gstate[layer][0][ec];
      is = S->gstate[layer][1][ec]; igs = S->in_gate_state[layer][ec]; ls = S->last_state[layer][ec];
    }
    ia.norm = S->norm[layer][ia.g][(size_t)epoch * C + ia.c]; ia.iv = S->ivar[layer][ia.g][epoch];
    if (ib.on) { ib.norm = S->norm[layer][ib.g][(size_t)epoch * C + ib.c]; ib.iv = S->ivar[layer][ib.g][epoch]; }
    if (lead && tid == 0 && layer == 1) S->bp_symbol[epoch] = epoch == 0 ? (unsigned)S->dyn[0] : S->input_history[epoch - 1];
    lds_barrier();   // slice s+2 is in LDS
    // the next step's chain does not depend on this step when it starts from zero (layer 0 leaves hidden_error_ = 0)
    if (layer == 0 && s + 1 < 2 * H && tid < 64) hpre = p1_chain(s + 1, 0.0f);
    wg_wait(&Y->bp_cnt[s], GB, &Y->fail, S->poll_sleep);
    if (tid < C) {
      const float h = ld_f(pub + (size_t)s * 2 * C + tid);
      if (s > 0) stored_o = ld_f(pub + (size_t)s * 2 * C + C + tid);   // the previous step's layer
      // LstmLayer::BackwardPass head (lstm-layer.cpp:110-132)
      if (epoch == H - 1) { stored_c = h; serr_c = 0.0f; }
      else stored_c = fadd(stored_c, h);
      const float og_e = fmul(fmul(fmul(th, stored_c), os), fsub(1.0f, os));
      serr_c = fadd(serr_c, fmul(fmul(stored_c, os), fsub(1.0f, fmul(th, th))));
      const float in_e = fmul(fmul(serr_c, igs), fsub(1.0f, fmul(is, is)));
      const float fg_e = fmul(fmul(fmul(fsub(ls, is), serr_c), fs), igs);
      gerr[tid] = fg_e;
      gerr[C + tid] = in_e;
      gerr[2 * C + tid] = og_e;
      if (epoch > 0) { serr_c = fmul(serr_c, fs); stored_c = 0.0f; }
    }
    lds_barrier();
    // per-gate normalisation backward (lstm-layer.cpp:158-163); item = (gate, cell)
    auto norm_a = [&](BpItem& it) {
      if (epoch == H - 1) { it.gu_c = 0.0f; it.bu_c = 0.0f; }
      float e = gerr[it.g * C + it.c];
      it.bu_c = fadd(it.bu_c, e);
      it.gu_c = fadd(it.gu_c, fmul(e, it.norm));
      e = fmul(e, fmul(it.gam_c, it.iv));
      it.err = e;
      gprod[it.g * C + it.c] = fmul(e, it.norm);
    };
    norm_a(ia);
    if (ib.on) norm_a(ib);
    lds_barrier();
    if (tid < 3) {  // (error_*norm_).sum(): backward
      const float* gp = gprod + tid * C;
      float sm = gp[C - 1];
#pragma unroll 16
      for (int i = C - 2; i >= 0; --i) sm = fadd(sm, gp[i]);
      gsm[tid] = fdiv(sm, (float)C);
    }
    lds_barrier();
    auto norm_b = [&](BpItem& it) {
      const float e = fsub(it.err, fmul(gsm[it.g], it.norm));
      gerr[it.g * C + it.c] = e;
      if (lead) S->E[layer][it.g][(size_t)epoch * C + it.c] = e;
    };
    norm_b(ia);
    if (ib.on) norm_b(ib);
    lds_barrier();
    // W^T chains into the own cells (lstm-layer.cpp:164-181): f = sum_j error_[j] * W[j][col + c], j ascending
    if (tid < 64) {
      const int nl = layer == 0 ? 3 * J : 6 * J;
      const int L = tid < nl ? tid : nl - 1;
      const int kind = L / (3 * J), gg = (L - kind * 3 * J) / J, cl = L - kind * 3 * J - gg * J;
      const int slot = layer == 0 ? gg : 3 + kind * 3 + gg;
      const bool need = kind == 0 ? epoch > 0 : layer > 0;
      if (need) {
        int nterm = C;
        asm volatile("" : "+s"(nterm));   // opaque trip count: a fully unrolled chain hoists 200 products into registers
        const float f = lds_chain(0.0f, Wb + (size_t)slot * 50 * J, J, cl, gerr + gg * C, 0, nterm);
        if (tid < nl) fres[(kind * 3 + gg) * J + cl] = f;
      }
    }
    lds_barrier();
    if (tid < C) serr_c = fminf(fmaxf(serr_c, -10.0f), 10.0f);   // ClipGradients (lstm-layer.cpp:140-142), replicated part
    if (tid < 64) {
      float h = 0.0f, sn = 0.0f;
      if (tid < J) {
        // *hidden_error = 0, then += f per gate in order forget, input node, output (lstm-layer.cpp:126,137-139)
        float he = 0.0f;
        if (layer > 0) { he = fadd(he, fres[3 * J + tid]); he = fadd(he, fres[4 * J + tid]); he = fadd(he, fres[5 * J + tid]); }
        float se = 0.0f;   // stored error of the own cell: zeroed above when epoch > 0
        if (epoch > 0) { se = fadd(se, fres[tid]); se = fadd(se, fres[J + tid]); se = fadd(se, fres[2 * J + tid]); }
        sn = fminf(fmaxf(se, -10.0f), 10.0f);
        h = fminf(fmaxf(he, -10.0f), 10.0f);
      }
      if (s + 1 < 2 * H) {
        h = layer == 1 ? p1_chain(s + 1, h) : hpre;
        if (tid < J) {
          st_f(pub + (size_t)(s + 1) * 2 * C + j0 + tid, h);
          st_f(pub + (size_t)(s + 1) * 2 * C + C + j0 + tid, sn);
        }
        wave_signal(&Y->bp_cnt[s + 1]);
      }
    }
    float t;
    t = stored_c; stored_c = stored_o; stored_o = t;
    t = serr_c; serr_c = serr_o; serr_o = t;
    t = ia.gam_c; ia.gam_c = ia.gam_o; ia.gam_o = t;
    t = ia.gu_c; ia.gu_c = ia.gu_o; ia.gu_o = t;
    t = ia.bu_c; ia.bu_c = ia.bu_o; ia.bu_o = t;
    t = ib.gam_c; ib.gam_c = ib.gam_o; ib.gam_o = t;
    t = ib.gu_c; ib.gu_c = ib.gu_o; ib.gu_o = t;
    t = ib.bu_c; ib.bu_c = ib.bu_o; ib.bu_o = t;
  }
  if (lead) {  // gamma / beta updates of the round (lstm-layer.cpp:158-160) for cmx_lstm_bptt_gb
    auto fin = [&](const BpItem& it) {   // after an even number of swaps _c is layer 1 again
      float* gb1 = S->gb[1][it.g];
      gb1[6 * C + it.c] = it.gu_c;
      gb1[7 * C + it.c] = it.bu_c;
      float* gb0 = S->gb[0][it.g];
      gb0[6 * C + it.c] = it.gu_o;
      gb0[7 * C + it.c] = it.bu_o;
    };
    fin(ia);
    if (ib.on) fin(ib);
  }
}
