// lstm_kernels.hip -- gfx950 kernels of the byte-level LSTM byte mixer stage.
//
// Reference: ByteMixer::ByteUpdate (src/mixer/byte-mixer.cpp:22-38), Lstm::Perceive/Predict
// (src/mixer/lstm.cpp:87-150), LstmLayer::ForwardPass/BackwardPass and Adam
// (src/mixer/lstm-layer.cpp:11-197), ByteModel::Predict (src/models/byte-model.cpp:8-24).
//
// Strict mode: every dot product is an ordered f32 chain (product rounded, then added, in the
// reference's index order; `(expr).sum()` of libstdc++ runs BACKWARD, `valarray::sum()` forward),
// libm through cmx_libm.h, no FMA contraction. The parallelism is ACROSS chains: 600 gate rows
// per layer, V output rows, 200 hidden-error lanes, and every weight element in the BPTT
// accumulation/Adam sweep. MFMA is not used: an MFMA step is a fused multiply-add with one
// rounding and a blocked K order, which cannot reproduce the reference stream.
//
// This file: the per-block bookkeeping (prep), the element-parallel half of BPTT (update accumulation, Adam: one lane per
// weight) and ByteModel's bit predictions. The sequential halves -- the forward pass of a block of up to 100 bytes and the
// BPTT walk -- are the multi-workgroup kernels of lstm_block.hip (cmx_lstm_fwdblk, cmx_lstm_bpttblk).
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
// NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS that maps to the approximate native sqrt.
__device__ __forceinline__ float fsqrt(float a) { return __builtin_sqrtf(a); }  // IEEE-correct (hipcc default)

constexpr int C = LSTM_C, H = LSTM_H, NH = LSTM_NH, VP = LSTM_VP;

}  // namespace

// ---- ByteMixer::SetInput/ByteUpdate head + Lstm::Perceive bookkeeping (byte-mixer.cpp:15-26,
//      lstm.cpp:80-92). in256 = the byte model's distribution, byte = the byte just coded.
__device__ __forceinline__ void lstm_prep(const LstmState* S, const float* in256, const uint8_t* bytes, size_t n, int e) {
  const int tid = threadIdx.x, V = S->V;
  if (S->vocab[tid]) {
    float v = fmul(in256[tid], 2.0f);  // inputs_ (0 + val) *= 2 / num_models_  (unsigned division = 2)
    int k = S->byte_map[tid];
    S->layer_input[0][(size_t)e * S->insz[0] + k] = v;
    S->layer_input[1][(size_t)e * S->insz[1] + k] = v;
  }
  if (tid == 0) {
    int sym = S->byte_map[bytes[n]];
    int last = e == 0 ? H - 1 : e - 1;
    S->dyn[0] = (int)S->input_history[last];  // old_input (lstm.cpp:90)
    S->input_history[last] = (unsigned)sym;
  }
  (void)V;
}

// on BPTT bytes (epoch 0) the bookkeeping must precede the backward pass (lstm.cpp:88-93). The trailing `k` of these kernels
// is always -1 (explicit arguments); k >= 0 read them from the device-resident LstmBlockArgs of a replayed block.
extern "C" __global__ void cmx_lstm_prep(const LstmState P, const float* in256, const uint8_t* bytes, size_t n, int e, int k) {
  if (k >= 0) { const LstmBlockArgs a = *P.blk; in256 = a.in_probs + a.n0 * 256; bytes = a.bytes; n = a.n0; e = 0; }
  lstm_prep(&P, in256, bytes, n, e);
}

// ---- BPTT sweep: update_[i][c] accumulated over epochs 99..0 (lstm-layer.cpp:182-186) held in a
//      register, then Adam (lstm-layer.cpp:11-32) and both weight layouts rewritten.
//      roles (ceil(rowlen/64), 50, 6): z = layer*3 + gate; block (64, 4) = 64 columns x 4 rows.
extern "C" __global__ void cmx_lstm_bptt_acc(const LstmState P, int update_steps, int k, int gx, int gy) {
  const LstmState* S = &P;
  if (k >= 0) update_steps = P.blk->us;
  __shared__ float es[H][4];
  __shared__ float ins[H][64];
  __shared__ unsigned sym[H];
  // launched as a 1-D grid of gx * gy * 6 roles (+ padding when one XCD is left to the mixing network, LstmState::avoid_xcd: a tile on that XCD would
  // share its compute units with 27 spinning workgroups and hold the whole sweep up -- 0.10 -> 0.89 ms, profiles/r05_kernel_stats_xcd7.csv)
  const int role = lstm_role_of_block((int)blockIdx.x, P.avoid_xcd);
  if (role < 0 || role >= gx * gy * 6) return;
  const int bx = role % gx, by = (role / gx) % gy, bz = role / (gx * gy);
  const int layer = bz / 3, g = bz % 3;
  const int V = S->V, rl = S->rowlen[layer], insz = S->insz[layer];
  const int c = bx * 64 + threadIdx.x, i = by * 4 + threadIdx.y;
  const int t = threadIdx.y * 64 + threadIdx.x;
  for (int k = t; k < H * 4; k += 256) es[k >> 2][k & 3] = S->E[layer][g][(size_t)(k >> 2) * C + by * 4 + (k & 3)];
  for (int k = t; k < H * 64; k += 256) {
    int ee = k >> 6, cc = bx * 64 + (k & 63);
    ins[ee][k & 63] = (cc >= V && cc < rl) ? S->layer_input[layer][(size_t)ee * insz + (cc - V)] : 0.0f;
  }
  if (t < H) sym[t] = S->bp_symbol[t];
  __syncthreads();
  if (c >= rl) return;
  float acc = 0.0f;
  if (c < V) {
    for (int e = H - 1; e >= 0; --e)
      if (sym[e] == (unsigned)c) acc = fadd(acc, es[e][threadIdx.y]);
  } else {
    for (int e = H - 1; e >= 0; --e) acc = fadd(acc, fmul(es[e][threadIdx.y], ins[e][threadIdx.x]));
  }
  const float* tab = S->adam_tab + 4 * update_steps;
  const float alpha = tab[0], b1 = tab[1], b2 = tab[2];
  const float beta1 = 0.025f, beta2 = 0.9999f, eps = 1e-6f;
  const size_t ix = (size_t)i * rl + c;
  float m = S->M[layer][g][ix], v = S->Vv[layer][g][ix], w = S->W[layer][g][ix];
  m = fmul(m, beta1);
  m = fadd(m, fmul(fsub(1.0f, beta1), acc));
  v = fmul(v, beta2);
  v = fadd(v, fmul(fmul(fsub(1.0f, beta2), acc), acc));
  w = fsub(w, fmul(alpha, fdiv(fdiv(m, b1), fsqrt(fadd(fdiv(v, b2), eps)))));
  S->M[layer][g][ix] = m;
  S->Vv[layer][g][ix] = v;
  S->W[layer][g][ix] = w;
  S->WT[layer][g][lstm_wt_index(V, insz, c, i)] = w;
}

// ---- the same sweep in TOLERANCE mode (cmx_lstm_set_tolerance; NOT bit-exact): the dense part of update_[i][c] = sum over the 100 epochs of
//      error_[e][i] * layer_input[e][c] is a 200 x rowlen x 100 matrix product per gate -- here on the matrix cores, v_mfma_f32_16x16x4_f32
//      (f32 in, f32 accumulate: the products are fused into the running sum and the epochs are summed upwards, four at a time, instead of the
//      reference's separately rounded fmul + fadd from epoch 99 down; lstm-layer.cpp:182-186). One wavefront per 16 x 16 tile of the weight matrix:
//      lane l feeds A[i0 + (l & 15)][e0 + (l >> 4)] = error_ and B[e0 + (l >> 4)][c0 + (l & 15)] = the input; its four accumulators are rows
//      i0 + 4 (l >> 4) + r of column c0 + (l & 15). The one-hot part (c < V) keeps the reference's order; Adam is the strict kernel's.
//      grid (ceil(rowlen / 16), ceil(200 / 16), 6), block 64.
extern "C" __global__ __launch_bounds__(64) void cmx_lstm_bptt_acc_mfma(const LstmState P, int update_steps, int k) {
  const LstmState* S = &P;
  if (k >= 0) update_steps = P.blk->us;
  typedef float f32x4 __attribute__((ext_vector_type(4)));
  const int layer = blockIdx.z / 3, g = blockIdx.z % 3;
  const int V = S->V, rl = S->rowlen[layer], insz = S->insz[layer];
  const int lane = threadIdx.x, c0 = blockIdx.x * 16, i0 = blockIdx.y * 16;
  const int am = i0 + (lane & 15), bn = c0 + (lane & 15), kk = lane >> 4;
  const float* E = S->E[layer][g];
  const float* LI = S->layer_input[layer];
  const bool aval = am < C, bval = bn >= V && bn < rl;
  f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
  for (int e0 = 0; e0 < H; e0 += 4) {
    const int e = e0 + kk;
    const float a = (aval && e < H) ? E[(size_t)e * C + am] : 0.0f;
    const float b = (bval && e < H) ? LI[(size_t)e * insz + (bn - V)] : 0.0f;
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  }
  const int c = bn;
  if (c >= rl) return;
  const float* tab = S->adam_tab + 4 * update_steps;
  const float alpha = tab[0], b1 = tab[1], b2 = tab[2];
  const float beta1 = 0.025f, beta2 = 0.9999f, eps = 1e-6f;
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + 4 * kk + r;
    if (i >= C) continue;
    float a_ = acc[r];
    if (c < V) {
      a_ = 0.0f;
      for (int e = H - 1; e >= 0; --e) if (S->bp_symbol[e] == (unsigned)c) a_ = fadd(a_, E[(size_t)e * C + i]);
    }
    const size_t ix = (size_t)i * rl + c;
    float m = S->M[layer][g][ix], v = S->Vv[layer][g][ix], w = S->W[layer][g][ix];
    m = fmul(m, beta1);
    m = fadd(m, fmul(fsub(1.0f, beta1), a_));
    v = fmul(v, beta2);
    v = fadd(v, fmul(fmul(fsub(1.0f, beta2), a_), a_));
    w = fsub(w, fmul(alpha, fdiv(fdiv(m, b1), fsqrt(fadd(fdiv(v, b2), eps)))));
    S->M[layer][g][ix] = m;
    S->Vv[layer][g][ix] = v;
    S->W[layer][g][ix] = w;
    S->WT[layer][g][lstm_wt_index(V, insz, c, i)] = w;
  }
}

// ---- Adam for gamma / beta (lstm-layer.cpp:191-195); grid 6 blocks of 256
extern "C" __global__ void cmx_lstm_bptt_gb(const LstmState P, int update_steps, int k) {
  const LstmState* S = &P;
  if (k >= 0) update_steps = P.blk->us;
  const int layer = blockIdx.x / 3, g = blockIdx.x % 3, c = threadIdx.x;
  if (c >= C) return;
  float* gb = S->gb[layer][g];
  const float* tab = S->adam_tab + 4 * update_steps;
  const float alpha = tab[0], b1 = tab[1], b2 = tab[2];
  const float beta1 = 0.025f, beta2 = 0.9999f, eps = 1e-6f;
  for (int which = 0; which < 2; ++which) {  // 0: gamma (w 0, m 2, v 3, u 6)   1: beta (w 1, m 4, v 5, u 7)
    float* w = gb + (which ? 1 : 0) * C;
    float* m = gb + (which ? 4 : 2) * C;
    float* v = gb + (which ? 5 : 3) * C;
    float gr = gb[(which ? 7 : 6) * C + c];
    float mm = fmul(m[c], beta1);
    mm = fadd(mm, fmul(fsub(1.0f, beta1), gr));
    float vv = fmul(v[c], beta2);
    vv = fadd(vv, fmul(fmul(fsub(1.0f, beta2), gr), gr));
    w[c] = fsub(w[c], fmul(alpha, fdiv(fdiv(mm, b1), fsqrt(fadd(fdiv(vv, b2), eps)))));
    m[c] = mm;
    v[c] = vv;
  }
}

// ---- ByteModel::Predict / Perceive along the known bits of a byte (byte-model.cpp:8-37).
//      One block per byte: dist(n) is the distribution the byte model held while byte n was coded.
//      Outputs p[n][8] (Model::Predict value per bit) and ex[n][8] (arg-max symbol, `ex`).
extern "C" __global__ void cmx_bytemodel_bits(const float* dist0, const float* dist_rest, const uint8_t* bytes,
                                              size_t nbytes, float* p_out, int* ex_out, size_t pstride, int only_k,
                                              float* p_out2) {
  // only_k >= 0 (bit-synchronous mode): just bit only_k of the byte -- the interval after only_k halvings needs no
  // sums -- and p_out2 (may be NULL) receives a second copy of that one value. The arg-max scan runs only when
  // ex_out is given.
  __shared__ float pr[256];
  const size_t n = blockIdx.x;
  const float* d = n == 0 ? dist0 : dist_rest + (n - 1) * 256;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) pr[i] = d[i];
  __syncthreads();
  if (threadIdx.x != 0) return;
  int top = 255, bot = 0;
  const int byte = bytes[n];
  for (int k = 0; k < 8; ++k) {
    int mid = bot + ((top - bot) / 2);
    if (only_k < 0 || k == only_k) {
      float num = 0.0f;
      for (int i = mid + 1; i <= top; ++i) num = fadd(num, pr[i]);
      float denom = num;
      for (int i = bot; i <= mid; ++i) denom = fadd(denom, pr[i]);
      const float p = denom == 0.0f ? 0.5f : fdiv(num, denom);
      p_out[(n * 8 + k) * pstride] = p;  // pstride 1, or a layer-0 row stride
      if (p_out2) *p_out2 = p;
      if (ex_out) {
        int ex = bot;
        float mx = pr[bot];
        for (int i = bot + 1; i <= top; ++i)
          if (pr[i] > mx) { mx = pr[i]; ex = i; }
        ex_out[n * 8 + k] = ex;
      }
      if (k == only_k) return;
    }
    if ((byte >> (7 - k)) & 1) bot = mid + 1;
    else top = mid;
  }
}

// ---- ByteModel::Predict / Perceive for a DECODER (cmx_late.h): the three byte-level distributions of the stream -- Bracket's
//      (layer-0 column 0), PPMd's (2076), the LSTM byte mixer's (2077) -- one wavefront each, bit by bit as the bits arrive.
//      Wave 2 also forms what Predictor::Perceive leaves in lstmpr / lstmex (predictor.cpp:180-182,462-465) for the fxcm stage:
//      hint[q] = Discretize(p of bit q + 1), `ex` at that moment, q = chunk-local update; entry 8 nbytes - 1 comes from the
//      distribution after the chunk's last byte. Counters: LC_BM0/1/2 = rows whose column is written (LC_BM2 = r also means
//      hint[r - 2] is there). A distribution after byte n - 1 is awaited on its producer's counter (Bracket: LC_BRK of the context
//      kernel; LSTM: LC_LSTM, bumped behind the byte's LSTM launch; PPMd: a host record, in place once the byte's last bit is
//      published); the one going into the chunk on the previous chunk's counters (c0_*: may be null).
//      Waves 3.. are the stream's RELAY (cmx_late.h): the one wavefront that talks to the host -- it brings every published bit and the host
//      stages' records of the step over into device memory and counts the step (LC_KNOWN), which is what every kernel waits on.
#define CMX_RELAY_WAVES 6
extern "C" __global__ void __launch_bounds__(192 + 64 * CMX_RELAY_WAVES)
cmx_bytemodel_late_kernel(CmxLate B, size_t nbytes, const float* brk0, const float* brk, const float* ppmd, const float* lstm0,
                          const float* lstm, const uint32_t* c0_brk, uint32_t c0_brk_want, const uint32_t* c0_lstm, uint32_t c0_lstm_want,
                          float* layer0, size_t pstride, int16_t* hint_pr, uint8_t* hint_ex, uint8_t* dbit0, const cmx_late_relay_t* relay, int nrelay) {
  __shared__ float prs[3][256];
  __shared__ unsigned relay_done, relay_seen;
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (threadIdx.x == 0) { relay_done = 0; relay_seen = 0; }
  __syncthreads();
  if (w >= 3) {   // the relay -- unless the HOST pushes the steps into device memory itself (CmxLate::pad = 1: round 6, cmx_late.h)
    if (!B.pad) late_relay(B, dbit0, relay, nrelay, (int)(8 * nbytes), lane, w - 3, CMX_RELAY_WAVES, &relay_done, &relay_seen);
    return;
  }
  float* const pr = prs[w];
  const int col = w == 0 ? 0 : w == 1 ? 2076 : 2077;
  const int which = w == 0 ? LC_BM0 : w == 1 ? LC_BM1 : LC_BM2;
  const size_t T = 8 * nbytes;
  for (size_t n = 0; n <= nbytes; ++n) {
    if (n == nbytes && w != 2) break;   // the entry after the chunk's last bit: the LSTM hint only
    const float* d;
    bool ok = true;
    if (n == 0) {
      ok = late_wait_step(B, 0);
      if (ok && w == 0 && c0_brk) ok = late_wait_ge(B.box, c0_brk, c0_brk_want);
      if (ok && w == 2 && c0_lstm) ok = late_wait_ge(B.box, c0_lstm, c0_lstm_want);
      d = w == 0 ? brk0 : w == 1 ? ppmd : lstm0;
    } else {
      if (w == 0) { ok = late_wait_cnt(B, LC_BRK, (uint32_t)n); d = brk + (n - 1) * 256; }
      else if (w == 1) { ok = late_wait_step(B, (int)(8 * n)); d = ppmd + n * 256; }   // (the relay has copied PPMd's row n with the byte's first step)
      else { ok = late_wait_cnt(B, LC_LSTM, (uint32_t)n); d = lstm + (n - 1) * 256; }
    }
    if (!ok) return;
    asm volatile("" ::: "memory");
    for (int i = lane; i < 256; i += 64) pr[i] = *(volatile const float*)(d + i);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    int top = 255, bot = 0;
    for (int k = 0; k < 8; ++k) {
      const size_t t = 8 * n + k;
      const int mid = bot + ((top - bot) / 2);
      {   // byte-model.cpp:8-24. Round 6: the two ordered sums (up to 128 + 128 dependent adds at a byte's first bit) run on every lane from broadcast LDS reads,
          // eight in flight ahead of the adds -- one lane reading and adding a term at a time put ~25 us between the LSTM's distribution and its column of the
          // byte's first bit (profiles/r06_decode_time.txt); the likeliest byte (`ex`: the first index of the range's maximum, byte-model.cpp:19-23) is a wave reduction
          // with that tie rule instead of a 256-step scan.
        auto ordered = [&](float acc, int a, int b) {   // acc + pr[a] + pr[a + 1] + ... + pr[b], in that order
          int i = a;
          for (; i + 7 <= b; i += 8) {
            const float v0 = pr[i], v1 = pr[i + 1], v2 = pr[i + 2], v3 = pr[i + 3], v4 = pr[i + 4], v5 = pr[i + 5], v6 = pr[i + 6], v7 = pr[i + 7];
            acc = fadd(acc, v0); acc = fadd(acc, v1); acc = fadd(acc, v2); acc = fadd(acc, v3);
            acc = fadd(acc, v4); acc = fadd(acc, v5); acc = fadd(acc, v6); acc = fadd(acc, v7);
          }
          for (; i <= b; ++i) acc = fadd(acc, pr[i]);
          return acc;
        };
        const float num = ordered(0.0f, mid + 1, top);
        const float denom = ordered(num, bot, mid);
        const float p = denom == 0.0f ? 0.5f : fdiv(num, denom);
        int ex = 0;
        if (w == 2 && t > 0) {
          float bv = -1.0f;   // (probabilities: every value of the range is >= 0)
          int bi = 0x7fffffff;
          for (int i = bot + lane; i <= top; i += 64) { const float v = pr[i]; if (v > bv) { bv = v; bi = i; } }   // ascending i: the lane's first maximum
          for (int sft = 32; sft > 0; sft >>= 1) {
            const float ov = __shfl_xor(bv, sft);
            const int oi = __shfl_xor(bi, sft);
            if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
          }
          ex = bi;
        }
        if (lane == 0) {
          if (n < nbytes) layer0[t * pstride + col] = p;
          if (w == 2 && t > 0) {
            const float prod = fmul(4094.0f, p);     // Discretize (predictor.cpp:180-182): the product is rounded to float, then 1 is added
            hint_pr[t - 1] = (int16_t)(unsigned)fadd(1.0f, prod);
            hint_ex[t - 1] = (uint8_t)ex;
          }
          late_publish(B, which, (uint32_t)(t + 1));
        }
      }
      if (n == nbytes) break;
      const int bit = late_y(B, (int)t + 1);
      if (bit < 0) return;
      if (bit) bot = mid + 1; else top = mid;   // ByteModel::Perceive (byte-model.cpp:30-37)
    }
  }
  (void)T;
}

// one store behind whatever is in front of it in the stream: the late pipeline's "the LSTM distribution of byte n is there"
extern "C" __global__ void cmx_late_bump_kernel(uint32_t* counter, uint32_t value, uint32_t* counter2, uint32_t value2) {
  if (threadIdx.x == 0) {
    __hip_atomic_store(counter, value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (counter2) __hip_atomic_store(counter2, value2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}
