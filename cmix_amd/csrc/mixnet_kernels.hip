// mixnet_kernels.hip -- persistent gfx950 kernel for cmix's final mixing network.
//
// One workgroup (8 wavefronts) owns one input stream and walks its bits
// strictly in order; there is no launch per bit.  Per bit it evaluates
//   MixerInput::SetInput (clamp + logit LUT)      ref src/mixer/mixer-input.cpp:11-15
//   auxiliary_context_                            ref src/predictor.cpp:388-393
//   26 + 20 + 1 Mixer::Mix / Mixer::Perceive      ref src/mixer/mixer.cpp:38-72
//   Mixer::GetContextData (row select / create)   ref src/mixer/mixer.cpp:16-36
//   SSE::Predict / Perceive                       ref src/mixer/sse.cpp:243-328
// bit-exactly as the reference built `g++ -O3` evaluates them: every product is
// rounded to f32 and added to the running f32 sum in ascending index order
// (no FMA, no tree reduction), libm calls go through cmx_libm.h.
//
// Work split inside the workgroup (wave64):
//   wave 0        "chain" wave: lane m owns mixer m. It walks the 2078-term
//                 ordered add chain of all 26 layer-0 mixers at once (one
//                 dependent v_add_f32 per term), then the intra-layer extra-input
//                 chain, layers 1 and 2, the squash and the SSE stage.
//   waves 1..7    "producer" waves: own the layer-0 weight rows element-wise in
//                 registers (132 slots per lane), stream them from HBM with
//                 coalesced 256-byte row segments, form the rounded products
//                 x[i]*w[m][i] and stage them in LDS in a bank-conflict-free
//                 [chunk][mixer][64+4] layout for the chain wave's ds_read_b128;
//                 after the error is known they apply w -= u*x in registers and
//                 write the row back.
// LDS: 120 KB product stage (half of the 26x2078 products at a time) + 8 KB
// stretched inputs + small per-mixer arrays.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mixnet_dev.h"

namespace {

constexpr int NTHREADS = 512;
constexpr int NPROD_WAVES = 7;
constexpr int NCHUNKS = 33;            // 64-element chunks of a 2078-wide row (last one: 30)
constexpr int H0_CHUNKS = 17;          // chunks 0..16 are staged first
constexpr int H1_CHUNKS = 16;          // chunks 17..32
constexpr int QPW = 4;                 // producer wave pw owns mixers pw, pw+7, pw+14, pw+21
constexpr int NSLOTS = NCHUNKS * QPW;  // register slot s = chunk*4 + q  (compile-time chunk)
constexpr int PROW = 68;               // padded product row: 64 + 4 floats
constexpr int PROD_FLOATS = H0_CHUNKS * CMX_MIX0 * PROW;

// Ordered add chain over `nchunks` staged chunks for mixer lane m (ref mixer.cpp:40-43).
__device__ __forceinline__ float chain_half(const float* prod, int m, int nchunks, int last_valid,
                                            float p) {
  for (int cl = 0; cl < nchunks; ++cl) {
    const float4* row = reinterpret_cast<const float4*>(prod + (cl * CMX_MIX0 + m) * PROW);
    int n4 = (cl == nchunks - 1) ? last_valid / 4 : 16;
#pragma unroll 4
    for (int q = 0; q < n4; ++q) {
      float4 v = row[q];
      p = fadd(p, v.x);
      p = fadd(p, v.y);
      p = fadd(p, v.z);
      p = fadd(p, v.w);
    }
    if (cl == nchunks - 1) {
      const float* r = prod + (cl * CMX_MIX0 + m) * PROW;
      for (int k = n4 * 4; k < last_valid; ++k) p = fadd(p, r[k]);
    }
  }
  return p;
}

// mode bits: 1 = forward (Predict), 2 = update (Perceive). Chunk mode = 3.
// Both roles execute exactly the same sequence of workgroup barriers per bit.
struct Smem {
  float* prod; float* xs; float* out0; float* in1; float* in2; float* upd;
  uint32_t* rowidx; uint32_t* dflag;
};

__device__ __forceinline__ void stretch_inputs(const MixState* S, const float* pr, float* xs,
                                               int tid) {
  // ref mixer-input.cpp:11-15, sigmoid.cpp:12-17
  const float* __restrict__ lut = S->logit_lut;
  for (int i = tid; i < CMX_IN0; i += NTHREADS) {
    float p = pr[i];
    if (p < 1.0e-4f) p = 1.0e-4f;
    else if (p > 1 - 1.0e-4f) p = 1 - 1.0e-4f;
    int idx = (int)(p * 100001.0f);
    if (idx >= 100001) idx = 100000;
    else if (idx < 0) idx = 0;
    xs[i] = lut[idx];
  }
}

// ---------------------------------------------------------------- producers
// Producer wave pw owns, for each of its (up to) 4 mixers m = pw + 7q and each of
// the 33 chunks c, the 64 row elements [64c, 64c+64): one register per lane.
__device__ void producer_role(MixState* S, const Smem& sm, const float* probs, int nbits,
                              bool fwd, bool upd_on, int tid) {
  int pw = __builtin_amdgcn_readfirstlane(tid >> 6) - 1;
  const int lane = tid & 63;
  float* __restrict__ rows0 = S->rows0;
  float w[NSLOTS];
  for (int t = 0; t < nbits; ++t) {
    asm volatile("" : "+s"(pw));  // keep per-slot address math inside the loop (no 132-wide hoist)
    stretch_inputs(S, probs + (size_t)t * CMX_IN0, sm.xs, tid);
    __syncthreads();  // B1: xs ready
    __syncthreads();  // B2: rows selected
    // element offset of the selected row of mixer `lane` (lanes 0..25)
    uint32_t mybase = 0;
    if (lane < CMX_MIX0) mybase = (lane * CMX_ROWS_PER_MIXER + sm.rowidx[lane]) * CMX_ROW0_STRIDE;
    uint32_t base[QPW];
    bool okq[QPW];
#pragma unroll
    for (int q = 0; q < QPW; ++q) {
      int m = pw + NPROD_WAVES * q;
      okq[q] = m < CMX_MIX0;
      base[q] = __builtin_amdgcn_readlane(mybase, okq[q] ? m : 0);
    }
    // C. stream the selected rows (coalesced 256-byte segments) into registers
#pragma unroll
    for (int s = 0; s < NSLOTS; ++s) {
      const int c = s / QPW, q = s % QPW;
      w[s] = 0.0f;
      if (okq[q] && (c < NCHUNKS - 1 || lane < CMX_IN0 - 64 * (NCHUNKS - 1)))
        w[s] = rows0[base[q] + c * 64 + lane];
    }
    if (fwd) {
#pragma unroll
      for (int s = 0; s < H0_CHUNKS * QPW; ++s) {
        const int c = s / QPW, q = s % QPW;
        if (okq[q])
          sm.prod[(c * CMX_MIX0 + pw + NPROD_WAVES * q) * PROW + lane] =
              fmul(sm.xs[c * 64 + lane], w[s]);
      }
      __syncthreads();  // B3: half 0 staged
      __syncthreads();  // B4: half 0 consumed
#pragma unroll
      for (int s = H0_CHUNKS * QPW; s < NSLOTS; ++s) {
        const int c = s / QPW, q = s % QPW;
        if (okq[q])
          sm.prod[((c - H0_CHUNKS) * CMX_MIX0 + pw + NPROD_WAVES * q) * PROW + lane] =
              fmul(sm.xs[c * 64 + lane], w[s]);  // xs[2078..2111] is zero padding
      }
    }
    __syncthreads();  // B5: half 1 staged (or rows loaded, update-only mode)
    __syncthreads();  // B6: update scalars published
    // G. w -= u*x (+ periodic decay), write rows back (ref mixer.cpp:66-71)
    if (upd_on) {
      const float cdec = 1.0f - 3.0e-6f;
      float uq[QPW];
      bool dq[QPW];
#pragma unroll
      for (int q = 0; q < QPW; ++q) {
        int m = okq[q] ? pw + NPROD_WAVES * q : 0;
        uq[q] = sm.upd[m];
        dq[q] = sm.dflag[m] != 0;
      }
#pragma unroll
      for (int s = 0; s < NSLOTS; ++s) {
        const int c = s / QPW, q = s % QPW;
        if (okq[q] && (c < NCHUNKS - 1 || lane < CMX_IN0 - 64 * (NCHUNKS - 1))) {
          float v = fsub(w[s], fmul(uq[q], sm.xs[c * 64 + lane]));
          if (dq[q]) v = fmul(v, cdec);
          rows0[base[q] + c * 64 + lane] = v;
        }
      }
    }
    __syncthreads();  // B7: xs / rowidx may be overwritten
  }
}

// Mixer::Perceive scalar part (ref mixer.cpp:56-64) for mixer `mix`, one lane.
__device__ __forceinline__ float perceive_scalar(MixState* S, const Smem& sm, int mix, float pval,
                                                 double d1, int bit) {
  uint32_t r = sm.rowidx[mix];
  uint64_t* rs = S->row_steps + (size_t)mix * CMX_ROWS_PER_MIXER + r;
  uint64_t rsteps = *rs;
  uint64_t mx = S->max_steps[mix];
  float decay = (float)(d1 * (1.5 - ((1.0 * (double)rsteps) / (double)mx)));
  float u = fmul(fmul(decay, S->lr[mix]), fsub(cmx_logistic(pval), (float)bit));
  ++rsteps;
  *rs = rsteps;
  if (rsteps > mx) S->max_steps[mix] = rsteps;
  sm.dflag[mix] = (rsteps & 1023) == 0;
  sm.upd[mix] = u;
  return u;
}

// Optional phase timers (mode bit 4): shader-clock ticks accumulated per phase by lane 0 of the
// chain wave into S->prof[]; read back through cmx_mixnet_profile().
#define PROF(k)                                                        \
  do {                                                                 \
    if (prof_on) {                                                     \
      uint64_t now_ = __builtin_readcyclecounter();                    \
      if (tid == 0) S->prof[k] += now_ - tprev;                        \
      tprev = now_;                                                    \
    }                                                                  \
  } while (0)

// ---------------------------------------------------------------- chain wave
__device__ void chain_role(MixState* S, const Smem& sm, const float* probs, const uint32_t* sel,
                           const uint8_t* bits, const float* decay1, int nbits, float* p_out,
                           float* mix_out, bool fwd, bool upd_on, bool prof_on, int tid) {
  const int m = tid;
  uint64_t tprev = __builtin_readcyclecounter();  // lane = mixer index within its layer
  const float smin = S->stretch_min, smax = S->stretch_max;
  const float cdec = 1.0f - 3.0e-6f;
  for (int t = 0; t < nbits; ++t) {
    const float* pr = probs + (size_t)t * CMX_IN0;
    stretch_inputs(S, pr, sm.xs, tid);
    __syncthreads();  // B1
    PROF(0);
    // B. selectors -> weight rows
    if (m < CMX_MIXERS) {
      uint32_t key = sel[(size_t)t * CMX_MIXERS + m];
      if (m == CMX_AUX) {  // ref predictor.cpp:388-393
        float avg = 0;
        avg = fadd(avg, cmx_logistic(sm.xs[433]));
        avg = fadd(avg, cmx_logistic(sm.xs[2024]));
        avg = fadd(avg, cmx_logistic(sm.xs[2077]));
        avg = avg / 3.0f;
        key = (uint32_t)(unsigned long long)(avg * 15);
      }
      sm.rowidx[m] = select_row(S, m, key);
    }
    PROF(1);
    __syncthreads();  // B2
    const int bit = bits[t];
    float* row0 = nullptr;
    float ew[CMX_MIX0];
    if (m < CMX_MIX0) {
      row0 = S->rows0 + ((size_t)m * CMX_ROWS_PER_MIXER + sm.rowidx[m]) * CMX_ROW0_STRIDE;
#pragma unroll
      for (int j = 0; j < CMX_MIX0; ++j) ew[j] = row0[CMX_ROW0_EXTRA + j];
    }
    float p_ = 0.0f, p1_ = 0.0f, p2_ = 0.0f;
    float* row1 = S->rows1 + ((size_t)(m < CMX_MIX1 ? m : 0) * CMX_ROWS_PER_MIXER +
                              sm.rowidx[CMX_MIX0 + (m < CMX_MIX1 ? m : 0)]) * CMX_ROW1_STRIDE;
    float* row2 = S->rows2 + (size_t)sm.rowidx[CMX_MIXERS - 1] * CMX_ROW2_STRIDE;
    PROF(2);
    if (fwd) {
      __syncthreads();  // B3
      PROF(3);
      float p_main = 0.0f;
      if (m < CMX_MIX0) p_main = chain_half(sm.prod, m, H0_CHUNKS, 64, 0.0f);
      PROF(4);
      __syncthreads();  // B4
      __syncthreads();  // B5
      PROF(5);
      if (m < CMX_MIX0) p_main = chain_half(sm.prod, m, H1_CHUNKS, CMX_IN0 - 32 * 64, p_main);
      PROF(6);
      // intra-layer chain: mixer k also sees the clamped outputs of mixers 0..k-1
      // (ref predictor.cpp:395-400, mixer.cpp:45-53)
      float e = 0.0f;
#pragma unroll
      for (int j = 0; j < CMX_MIX0; ++j) {
        float mine = fadd(p_main, e);
        float oj = __shfl(mine, j);
        if (oj > smax) oj = smax;
        else if (oj < smin) oj = smin;
        if (m == j) { p_ = mine; sm.out0[j] = oj; }
        if (m > j && m < CMX_MIX0) e = fadd(e, fmul(oj, ew[j]));
      }
      if (m < CMX_MIX0) { S->fwd_p[m] = p_; S->fwd_out0[m] = sm.out0[m]; }
      PROF(7);
    } else {
      __syncthreads();  // B5
      if (m < CMX_MIX0) { p_ = S->fwd_p[m]; sm.out0[m] = S->fwd_out0[m]; }
    }
    // layer-1 / layer-2 inputs (ref predictor.cpp:397-406)
    if (m < CMX_MIX0) { sm.in1[m] = sm.out0[m]; sm.in2[m] = sm.out0[m]; }
    if (m < 3) {
      float v = sm.xs[m == 0 ? 433 : m == 1 ? 2024 : 2077];
      if (v > smax) v = smax;
      else if (v < smin) v = smin;
      sm.in1[CMX_MIX0 + m] = v;
      sm.in2[CMX_MIX0 + CMX_MIX1 + m] = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    if (fwd) {
      // layer 1: lane k (0..19) = mixer 26+k, 29 inputs + k extras
      float pm = 0.0f;
      for (int i = 0; i < CMX_IN1; ++i) pm = fadd(pm, fmul(sm.in1[i], row1[i]));
      float e = 0.0f;
#pragma unroll
      for (int j = 0; j < CMX_MIX1; ++j) {
        float mine = fadd(pm, e);
        float oj = __shfl(mine, j);
        if (oj > smax) oj = smax;
        else if (oj < smin) oj = smin;
        if (m == j) { p1_ = mine; sm.in2[CMX_MIX0 + j] = oj; }
        if (m > j && m < CMX_MIX1) e = fadd(e, fmul(oj, row1[CMX_ROW1_EXTRA + j]));
      }
      if (m < CMX_MIX1) { S->fwd_p[CMX_MIX0 + m] = p1_; S->fwd_in2[CMX_MIX0 + m] = sm.in2[CMX_MIX0 + m]; }
    } else if (m < CMX_MIX1) {
      p1_ = S->fwd_p[CMX_MIX0 + m];
      sm.in2[CMX_MIX0 + m] = S->fwd_in2[CMX_MIX0 + m];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    PROF(8);
    // layer 2 (lane 0): 49 inputs; squash; SSE; LSTM override (ref predictor.cpp:413-418)
    if (m == 0) {
      if (fwd) {
        float acc = 0.0f;
        for (int i = 0; i < CMX_IN2; ++i) acc = fadd(acc, fmul(sm.in2[i], row2[i]));
        p2_ = acc;
        S->fwd_p[CMX_MIXERS - 1] = p2_;
        float pf = sse_step(S, cmx_logistic(p2_), bit, upd_on);
        float lp = pr[CMX_IN0 - 1];
        if (lp == 0.0f || lp == 1.0f) pf = lp;
        p_out[t] = pf;
      } else {
        p2_ = S->fwd_p[CMX_MIXERS - 1];
        sse_step(S, cmx_logistic(p2_), bit, true);
      }
    }
    if (fwd && mix_out) {
      float* mo = mix_out + (size_t)t * CMX_MIXERS;
      if (m < CMX_MIX0) mo[m] = p_;
      if (m < CMX_MIX1) mo[CMX_MIX0 + m] = p1_;
      if (m == 0) mo[CMX_MIXERS - 1] = p2_;
    }
    PROF(9);
    // Mixer::Perceive for the rows this wave owns (ref mixer.cpp:56-72)
    if (upd_on) {
      const double d1 = (double)decay1[t];  // (float)(0.9/pow(1e-7*steps_+0.8,0.8)), host libm
      if (m < CMX_MIX0) {
        float u = perceive_scalar(S, sm, m, p_, d1, bit);
        bool df = sm.dflag[m];
#pragma unroll
        for (int j = 0; j < CMX_MIX0; ++j) {
          if (j < m) {
            float v = fsub(ew[j], fmul(u, sm.out0[j]));
            if (df) v = fmul(v, cdec);
            row0[CMX_ROW0_EXTRA + j] = v;
          }
        }
      }
      if (m < CMX_MIX1) {
        float u = perceive_scalar(S, sm, CMX_MIX0 + m, p1_, d1, bit);
        bool df = sm.dflag[CMX_MIX0 + m];
        for (int i = 0; i < CMX_IN1; ++i) {
          float v = fsub(row1[i], fmul(u, sm.in1[i]));
          if (df) v = fmul(v, cdec);
          row1[i] = v;
        }
        for (int j = 0; j < m; ++j) {
          float v = fsub(row1[CMX_ROW1_EXTRA + j], fmul(u, sm.in2[CMX_MIX0 + j]));
          if (df) v = fmul(v, cdec);
          row1[CMX_ROW1_EXTRA + j] = v;
        }
      }
      if (m == 0) {
        float u = perceive_scalar(S, sm, CMX_MIXERS - 1, p2_, d1, bit);
        bool df = sm.dflag[CMX_MIXERS - 1];
        for (int i = 0; i < CMX_IN2; ++i) {
          float v = fsub(row2[i], fmul(u, sm.in2[i]));
          if (df) v = fmul(v, cdec);
          row2[i] = v;
        }
        S->steps = S->steps + 1;
      }
    }
    PROF(10);
    __syncthreads();  // B6
    __syncthreads();  // B7
    PROF(11);
  }
}

}  // namespace

extern "C" __global__ __launch_bounds__(NTHREADS) void cmx_mixnet_kernel(
    MixState* __restrict__ S, const float* __restrict__ probs, const uint32_t* __restrict__ sel,
    const uint8_t* __restrict__ bits, const float* __restrict__ decay1, int nbits,
    float* __restrict__ p_out, float* __restrict__ mix_out, int mode) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Smem sm;
  sm.prod = smem;                      // PROD_FLOATS
  sm.xs = sm.prod + PROD_FLOATS;       // 2112 stretched layer-0 inputs
  sm.out0 = sm.xs + 2112;              // 32 clamped layer-0 outputs
  sm.in1 = sm.out0 + 32;               // 32
  sm.in2 = sm.in1 + 32;                // 64
  sm.upd = sm.in2 + 64;                // 48 per-mixer update scalar
  sm.rowidx = reinterpret_cast<uint32_t*>(sm.upd + 48);  // 48
  sm.dflag = sm.rowidx + 48;           // 48
  const int tid = threadIdx.x;
  const bool fwd = mode & 1, upd_on = mode & 2;
  if (tid < 2112 - CMX_IN0) sm.xs[CMX_IN0 + tid] = 0.0f;  // padding read by the last chunk
  if (tid < 64) chain_role(S, sm, probs, sel, bits, decay1, nbits, p_out, mix_out, fwd, upd_on, (mode & 4) != 0, tid);
  else producer_role(S, sm, probs, nbits, fwd, upd_on, tid);
}

// Fills the SSE tables with their initial interpolation nodes (ref sse.cpp:27-33,216-228).
extern "C" __global__ void cmx_sse_init_kernel(MixState* S) {
  size_t gid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = gid; i < (size_t)CMX_SM6_VOL; i += stride)
    for (int k = 0; k < 7; ++k) S->s6[i * 8 + k] = (uint16_t)(8192 + k * 5461);
  for (size_t i = gid; i < (size_t)CMX_SM7_VOL; i += stride)
    for (int k = 0; k < 7; ++k) S->s7[i * 8 + k] = (uint16_t)(12288 + k * 4096);
  for (size_t i = gid; i < (size_t)CMX_MIX1_VOL; i += stride) S->x1[i] = 7649 + 16384;
  for (size_t i = gid; i < (size_t)CMX_MIX2_VOL; i += stride) S->x2[i] = 2561 + 16384;
}

// libm probe for the parity tests.
extern "C" __global__ void cmx_probe_libm_kernel(int which, const float* x, float* y, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = x[i];
  y[i] = which == 0 ? cmx_expf(v) : which == 1 ? cmx_tanhf(v) : cmx_logistic(v);
}
