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
// Per byte (stream order): prep -> [bptt_seq -> bptt_acc -> bptt_gb every 100 bytes] -> sgd -> fwd
// (both layers, output layer and softmax fused in one workgroup).  Bit-level predictions for a whole
// chunk are produced by bytemodel_bits at the end (the coded bytes are known in compression).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cmx_libm.h"
#include "lstm_state.h"

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

// ---- output layer SGD (lstm.cpp:112-116): slot[e] = slot[last] - (lr*err_i)*hidden_, both layouts
// stand-alone form: on BPTT bytes (epoch 0) the bookkeeping must precede the backward pass (lstm.cpp:88-93)
// Every per-byte kernel takes a trailing `k`: k < 0 = direct launch, the explicit arguments hold; k >= 0 = node
// k of a captured 100-byte block, the arguments come from the device-resident LstmBlockArgs (epoch = k).
extern "C" __global__ void cmx_lstm_setblk(LstmBlockArgs* dst, const LstmBlockArgs v) {
  if (threadIdx.x == 0) *dst = v;
}

extern "C" __global__ void cmx_lstm_prep(const LstmState P, const float* in256, const uint8_t* bytes, size_t n, int e, int k) {
  if (k >= 0) { const LstmBlockArgs a = *P.blk; in256 = a.in_probs + a.n0 * 256; bytes = a.bytes; n = a.n0; e = 0; }
  lstm_prep(&P, in256, bytes, n, e);
}

extern "C" __global__ void cmx_lstm_sgd(const LstmState P, const float* in256, const uint8_t* bytes, size_t n, int e, int hid_cur,
                                        int k) {
  const LstmState* S = &P;
  if (k >= 0) {
    const LstmBlockArgs a = *P.blk;
    n = a.n0 + k; e = k; hid_cur = (a.hc0 + k) & 1; bytes = a.bytes;
    in256 = k == 0 ? nullptr : a.in_probs + n * 256;  // node 0 follows the stand-alone bookkeeping kernel
  }
  if (blockIdx.x == gridDim.x - 1) {  // the extra block does ByteMixer::SetInput / Lstm::Perceive bookkeeping
    if (in256) lstm_prep(S, in256, bytes, n, e);   // NULL: already done by cmx_lstm_prep
    return;
  }
  const int V = S->V, last = e == 0 ? H - 1 : e - 1;
  const int cur_sym = S->byte_map[bytes[n]];
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= V * NH) return;
  const int i = idx / NH, j = idx - i * NH;
  float o = S->output[(size_t)last * VP + i];
  float err = (i == cur_sym) ? fsub(o, 1.0f) : o;
  float le = fmul(S->lr, err);
  const float* hid = S->hid[hid_cur];
  float v = fsub(S->OL[((size_t)last * V + i) * NH + j], fmul(le, hid[j]));
  S->OL[((size_t)e * V + i) * NH + j] = v;
  S->OLT[((size_t)e * NH + j) * VP + i] = v;
}

// ---- Fused forward pass of one byte: both layers' gate dot products, RMS norm + activations + cell
//      update, the output-layer matvec and the softmax (lstm.cpp:120-150; lstm-layer.cpp:62-99;
//      byte-mixer.cpp:27-37) in ONE workgroup of 640 lanes. The byte-rate recurrence is a chain of
//      six dependent steps; as six launches it was bound by launch latency (4-10 us each), not by
//      work. Here the steps are separated by workgroup barriers. The 600 gate rows of a layer are
//      600 ordered chains, one per lane; the transposed weight layout makes a wave's loads of one term
//      a contiguous 256-byte run, so the CU streams the layer's weights from L2 at its L1 fill rate.
extern "C" __global__ __launch_bounds__(640) void cmx_lstm_fwd(const LstmState P, const uint8_t* bytes, size_t n, int e,
                                                               int hid_cur, float* out_probs256, int k) {
  if (P.xcd >= 0 && (int)blockIdx.x != (P.xcd & 7)) return;
  const LstmState* S = &P;
  if (k >= 0) {
    const LstmBlockArgs a = *P.blk;
    n = a.n0 + k; e = k; hid_cur = (a.hc0 + k) & 1; bytes = a.bytes;
    out_probs256 = a.out_probs ? a.out_probs + n * 256 : nullptr;
  }
  __shared__ float in[832];
  __shared__ float raw[3][C];
  __shared__ float ivar_s[3];
  __shared__ float hid[NH];
  __shared__ float ex[VP];
  __shared__ float red[256];
  __shared__ float tot_s;
  const int tid = threadIdx.x, V = S->V;
  const int cur_sym = S->byte_map[bytes[n]];
  const float* hold = S->hid[hid_cur];
  float* hnew = S->hid[hid_cur ^ 1];
  if (tid == 0) hid[NH - 1] = 1.0f;  // bias element of hidden_ (lstm.cpp:18)
#pragma unroll 1
  for (int layer = 0; layer < LSTM_L; ++layer) {
    const int insz = S->insz[layer];
    float* li = S->layer_input[layer] + (size_t)e * insz;
    __syncthreads();
    for (int j = tid; j < insz; j += 640) {
      float v;
      if (j < V) v = li[j];                                  // Lstm::SetInput (written by cmx_lstm_prep)
      else if (j < V + C) v = hold[layer * C + (j - V)];     // own previous hidden (lstm.cpp:122-124)
      else if (j < insz - 1) v = hid[j - V - C];             // layer 1: layer 0's new hidden (lstm.cpp:127-131)
      else v = 1.0f;                                         // bias
      in[j] = v;
      if (j >= V) li[j] = v;                                 // keep the assembled vector for BPTT
    }
    __syncthreads();
    if (tid < 3 * C) {  // LstmLayer::ForwardPass(NeuronLayer&) dot products (lstm-layer.cpp:85-92)
      const int g = tid / C, i = tid - g * C;
      const float* wt = S->WT[layer][g];
      float f = wt[(size_t)cur_sym * C + i];
      // dense part: 16-byte loads of four consecutive terms (layout: lstm_wt_index), 8 loads = 32 terms
      // in flight per lane, then the ordered chain over them
      const float4* w4 = reinterpret_cast<const float4*>(wt + (size_t)V * C) + i;
      const int full = insz & ~3, nq = full >> 2;
      // (A single CU sustains ~50 GB/s from L2/HBM whatever the load width or depth -- measured with dword,
      // 16-byte and 16-deep ring variants -- so this workgroup is bandwidth-bound at ~45 us per byte; the
      // next step is to spread the rows over several workgroups with in-launch hand-offs.)
      int q0 = 0;
      for (; q0 + 8 <= nq; q0 += 8) {  // 8 x 16-byte loads (32 terms) in flight per lane, then the ordered chain
        float4 w[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = w4[(size_t)(q0 + k) * C];
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float* x = in + 4 * (q0 + k);
          f = fadd(f, fmul(x[0], w[k].x));
          f = fadd(f, fmul(x[1], w[k].y));
          f = fadd(f, fmul(x[2], w[k].z));
          f = fadd(f, fmul(x[3], w[k].w));
        }
      }
      for (; q0 < nq; ++q0) {
        const float4 v = w4[(size_t)q0 * C];
        const float* x = in + 4 * q0;
        f = fadd(f, fmul(x[0], v.x));
        f = fadd(f, fmul(x[1], v.y));
        f = fadd(f, fmul(x[2], v.z));
        f = fadd(f, fmul(x[3], v.w));
      }
      int j = full;
      const float* wr = wt + (size_t)V * C + (size_t)full * C + i;
      for (; j < insz; ++j) f = fadd(f, fmul(in[j], wr[(size_t)(j - full) * C]));
      raw[g][i] = f;
    }
    __syncthreads();
    // RMS norm, activations, cell update (lstm-layer.cpp:62-83, 93-98)
    if (tid < 3) {  // (norm_*norm_).sum(): expression-template sum runs backward from the last element
      float s = fmul(raw[tid][C - 1], raw[tid][C - 1]);
      for (int i = C - 2; i >= 0; --i) s = fadd(s, fmul(raw[tid][i], raw[tid][i]));
      float iv = fdiv(1.0f, fsqrt(fadd(fdiv(s, (float)C), 1e-5f)));
      ivar_s[tid] = iv;
      S->ivar[layer][tid][e] = iv;
    }
    __syncthreads();
    if (tid < C) {
      float st[3];
      for (int g = 0; g < 3; ++g) {
        const float* gb = S->gb[layer][g];
        float nrm = fmul(raw[g][tid], ivar_s[g]);
        S->norm[layer][g][(size_t)e * C + tid] = nrm;
        st[g] = fadd(fmul(nrm, gb[tid]), gb[C + tid]);  // norm*gamma + beta
      }
      float fg = cmx_logistic(st[0]);
      float inn = cmx_tanhf(st[1]);
      float og = cmx_logistic(st[2]);
      S->gstate[layer][0][(size_t)e * C + tid] = fg;
      S->gstate[layer][1][(size_t)e * C + tid] = inn;
      S->gstate[layer][2][(size_t)e * C + tid] = og;
      float state = S->state[layer][tid];
      S->last_state[layer][(size_t)e * C + tid] = state;
      float igs = fsub(1.0f, fg);
      S->in_gate_state[layer][(size_t)e * C + tid] = igs;
      state = fmul(state, fg);
      state = fadd(state, fmul(inn, igs));
      S->state[layer][tid] = state;
      float th = cmx_tanhf(state);
      S->tanh_state[layer][(size_t)e * C + tid] = th;
      const float h = fmul(og, th);
      hid[layer * C + tid] = h;
      hnew[layer * C + tid] = h;
    }
  }
  __syncthreads();
  // output layer matvec (lstm.cpp:132-140): one ordered 401-term chain per vocabulary symbol
  float lg = 0.0f;
  if (tid < V) {
    const float* ot = S->OLT + (size_t)e * NH * VP + tid;
    float sum = 0.0f;
    int j = 0;
    for (; j + 32 <= NH; j += 32) {
      float w[32];
#pragma unroll
      for (int k = 0; k < 32; ++k) w[k] = ot[(size_t)(j + k) * VP];
#pragma unroll
      for (int k = 0; k < 32; ++k) sum = fadd(sum, fmul(hid[j + k], w[k]));
    }
    for (; j < NH; ++j) sum = fadd(sum, fmul(hid[j], ot[(size_t)j * VP]));
    lg = sum;
  }
  // softmax (lstm.cpp:141-149), ByteMixer::ByteUpdate tail (byte-mixer.cpp:27-37)
  if (tid < 256) red[tid] = tid < V ? lg : 0.0f;  // max_out starts at 0 (lstm.cpp:132)
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]);
    __syncthreads();
  }
  const float mx = red[0];
  if (tid < V) ex[tid] = cmx_expf(fsub(lg, mx));
  __syncthreads();
  if (tid == 0) {  // valarray::sum(): forward from 0
    float t = 0.0f;
    for (int i = 0; i < V; ++i) t = fadd(t, ex[i]);
    tot_s = t;
  }
  __syncthreads();
  if (tid < V) {
    float p = fdiv(ex[tid], tot_s);
    S->output[(size_t)e * VP + tid] = p;
    ex[tid] = p;
  }
  __syncthreads();
  if (tid < 256) {
    float pb = S->vocab[tid] ? ex[S->byte_map[tid]] : 0.0f;
    S->byte_probs[tid] = pb;
    if (out_probs256) out_probs256[tid] = pb;
  }
}

// ---- BPTT, sequential part (lstm.cpp:93-110; lstm-layer.cpp:108-183): one block of 1024 walks
//      epoch 99..0 x layer 1..0, leaving the final gate errors E[l][g][epoch][.] for the sweep.
extern "C" __global__ __launch_bounds__(1024) void cmx_lstm_bptt_seq(const LstmState P) {
  if (P.xcd >= 0 && (int)blockIdx.x != (P.xcd & 7)) return;
  const LstmState* S = &P;
  __shared__ float errv[VP];        // softmax-CE error of the epoch
  __shared__ float herr[C];         // Lstm::hidden_error_
  __shared__ float gerr[3][C];      // NeuronLayer::error_
  __shared__ float gprod[3][C];
  __shared__ float gsm[3];
  __shared__ float fres[2][3][C];   // matvec results: [stored|hidden][gate][i]
  const int tid = threadIdx.x, V = S->V;
  const int j = tid;                // cell index for the element-wise phases
  float stored[LSTM_L] = {0, 0}, state_err[LSTM_L] = {0, 0};
  if (tid < C) herr[tid] = 0.0f;    // hidden_error_ is zero on entry (cleared by the last BackwardPass)
  __syncthreads();
  for (int epoch = H - 1; epoch >= 0; --epoch) {
    if (tid < V) {
      float o = S->output[(size_t)epoch * VP + tid];
      errv[tid] = ((unsigned)tid == S->input_history[epoch]) ? fsub(o, 1.0f) : o;
    }
    if (tid == 0) S->bp_symbol[epoch] = epoch == 0 ? (unsigned)S->dyn[0] : S->input_history[epoch - 1];
    __syncthreads();
#pragma unroll
    for (int layer = LSTM_L - 1; layer >= 0; --layer) {
      // hidden_error_[j] += output_layer_[epoch][i][j+offset] * error_i, i ascending (lstm.cpp:98-103)
      if (j < C) {
        float h = herr[j];
        const float* ol = S->OL + (size_t)epoch * V * NH + layer * C + j;
        int i = 0;
        for (; i + 16 <= V; i += 16) {
          float a[16];
#pragma unroll
          for (int k = 0; k < 16; ++k) a[k] = ol[(size_t)(i + k) * NH];
#pragma unroll
          for (int k = 0; k < 16; ++k) h = fadd(h, fmul(a[k], errv[i + k]));
        }
        for (; i < V; ++i) h = fadd(h, fmul(ol[(size_t)i * NH], errv[i]));
        // LstmLayer::BackwardPass head (lstm-layer.cpp:110-132)
        const size_t ec = (size_t)epoch * C + j;
        if (epoch == H - 1) { stored[layer] = h; state_err[layer] = 0.0f; }
        else stored[layer] = fadd(stored[layer], h);
        const float th = S->tanh_state[layer][ec], os = S->gstate[layer][2][ec], fs = S->gstate[layer][0][ec],
                    is = S->gstate[layer][1][ec], igs = S->in_gate_state[layer][ec], ls = S->last_state[layer][ec];
        float og_e = fmul(fmul(fmul(th, stored[layer]), os), fsub(1.0f, os));
        state_err[layer] = fadd(state_err[layer], fmul(fmul(stored[layer], os), fsub(1.0f, fmul(th, th))));
        float in_e = fmul(fmul(state_err[layer], igs), fsub(1.0f, fmul(is, is)));
        float fg_e = fmul(fmul(fmul(fsub(ls, is), state_err[layer]), fs), igs);
        gerr[0][j] = fg_e;
        gerr[1][j] = in_e;
        gerr[2][j] = og_e;
        if (epoch > 0) { state_err[layer] = fmul(state_err[layer], fs); stored[layer] = 0.0f; }
      }
      __syncthreads();
      // per-gate normalisation backward (lstm-layer.cpp:158-163); thread = (gate, cell)
      const int g = tid >> 8, c = tid & 255;
      float myerr = 0.0f, mynorm = 0.0f;
      if (g < 3 && c < C) {
        float* gb = S->gb[layer][g];
        float* gamma_u = gb + 6 * C;
        float* beta_u = gb + 7 * C;
        if (epoch == H - 1) { gamma_u[c] = 0.0f; beta_u[c] = 0.0f; }
        myerr = gerr[g][c];
        mynorm = S->norm[layer][g][(size_t)epoch * C + c];
        beta_u[c] = fadd(beta_u[c], myerr);
        gamma_u[c] = fadd(gamma_u[c], fmul(myerr, mynorm));
        myerr = fmul(myerr, fmul(gb[c], S->ivar[layer][g][epoch]));
        gprod[g][c] = fmul(myerr, mynorm);
      }
      __syncthreads();
      if (g < 3 && c == 0) {  // (error_*norm_).sum(): backward
        float s = gprod[g][C - 1];
        for (int i = C - 2; i >= 0; --i) s = fadd(s, gprod[g][i]);
        gsm[g] = fdiv(s, (float)C);
      }
      __syncthreads();
      if (g < 3 && c < C) {
        myerr = fsub(myerr, fmul(gsm[g], mynorm));
        gerr[g][c] = myerr;
        S->E[layer][g][(size_t)epoch * C + c] = myerr;
      }
      __syncthreads();
      // W^T matvecs (lstm-layer.cpp:164-181): f_i = sum_j error_[j] * W[j][col+i], j ascending
      for (int kind = 0; kind < 2; ++kind) {
        const bool need = kind == 0 ? epoch > 0 : layer > 0;
        if (need && g < 3 && c < C) {
          const int rl = S->rowlen[layer];
          const float* w = S->W[layer][g] + 2 * V + (kind == 1 ? C : 0) + c;
          float f = 0.0f;
          for (int jj = 0; jj < C; jj += 20) {   // C = 200 = 10 x 20
            float wv[20];
#pragma unroll
            for (int k = 0; k < 20; ++k) wv[k] = w[(size_t)(jj + k) * rl];
#pragma unroll
            for (int k = 0; k < 20; ++k) f = fadd(f, fmul(gerr[g][jj + k], wv[k]));
          }
          fres[kind][g][c] = f;
        }
      }
      __syncthreads();
      if (j < C) {
        // *hidden_error = 0, then += f per gate in order forget, input node, output (lstm-layer.cpp:126,137-139)
        float he = 0.0f;
        if (layer > 0) { he = fadd(he, fres[1][0][j]); he = fadd(he, fres[1][1][j]); he = fadd(he, fres[1][2][j]); }
        if (epoch > 0) {
          float se = stored[layer];
          se = fadd(se, fres[0][0][j]); se = fadd(se, fres[0][1][j]); se = fadd(se, fres[0][2][j]);
          stored[layer] = se;
        }
        // ClipGradients (lstm-layer.cpp:140-142)
        state_err[layer] = fminf(fmaxf(state_err[layer], -10.0f), 10.0f);
        stored[layer] = fminf(fmaxf(stored[layer], -10.0f), 10.0f);
        herr[j] = fminf(fmaxf(he, -10.0f), 10.0f);
      }
      __syncthreads();
    }
  }
}

// ---- BPTT sweep: update_[i][c] accumulated over epochs 99..0 (lstm-layer.cpp:182-186) held in a
//      register, then Adam (lstm-layer.cpp:11-32) and both weight layouts rewritten.
//      grid (ceil(rowlen/64), 50, 6): blockIdx.z = layer*3 + gate; block (64, 4) = 64 columns x 4 rows.
extern "C" __global__ void cmx_lstm_bptt_acc(const LstmState P, int update_steps, int k) {
  const LstmState* S = &P;
  if (k >= 0) update_steps = P.blk->us;
  __shared__ float es[H][4];
  __shared__ float ins[H][64];
  __shared__ unsigned sym[H];
  const int layer = blockIdx.z / 3, g = blockIdx.z % 3;
  const int V = S->V, rl = S->rowlen[layer], insz = S->insz[layer];
  const int c = blockIdx.x * 64 + threadIdx.x, i = blockIdx.y * 4 + threadIdx.y;
  const int t = threadIdx.y * 64 + threadIdx.x;
  for (int k = t; k < H * 4; k += 256) es[k >> 2][k & 3] = S->E[layer][g][(size_t)(k >> 2) * C + blockIdx.y * 4 + (k & 3)];
  for (int k = t; k < H * 64; k += 256) {
    int ee = k >> 6, cc = blockIdx.x * 64 + (k & 63);
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
