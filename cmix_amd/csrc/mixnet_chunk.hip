// mixnet_chunk.hip -- look-ahead (chunk mode) kernel of the final mixing network.
//
// Same arithmetic, same HBM state and same results as the bit-synchronous kernel in
// mixnet_kernels.hip, restructured as a 12-wave software pipeline inside ONE persistent
// workgroup, because in compression every bit of the chunk is already known:
//
//   wave 0  chain    lane m = layer-0 mixer m: the 26 ordered 2078-term add chains
//                    (mixer.cpp:40-43) in four LDS-staged segments, the intra-layer
//                    extra-input chain (predictor.cpp:395-400), Mixer::Perceive scalars.
//   wave 1  tail     one bit behind: layer 1, layer 2, squash, SSE, override, output
//                    (predictor.cpp:402-418) and the updates of those rows.
//   wave 2  scout    up to two bits ahead: MixerInput stretch of the 2078 inputs, aux
//                    context, Mixer::GetContextData row selection for all 47 mixers.
//   waves 3..11      producers p=0..8: own the selected layer-0 rows of mixers p, p+9,
//                    p+18 in registers as float4 (9 chunks x 3 mixers), apply the previous
//                    bit's update lazily (w -= u*x), swap rows whose selector changed
//                    (16-byte global stores / loads, issued a segment ahead), and stage
//                    the rounded products for the chain wave.
//
// No workgroup barrier in the bit loop: the roles are decoupled by monotonic LDS
// counters (staged / consumed / u_epoch / scout_epoch / tail_in / tail_done), so the
// critical recurrence per bit is  chain -> extras -> u -> first product segment.
// Every spin is bounded; a timeout sets S->error and unwinds all roles.
#include "mixnet_dev.h"
#include "cmx_late.h"

namespace {

constexpr int NTHREADS = 768;
constexpr int NPROD = 9;
constexpr int MPW = 3;             // mixers per producer wave
constexpr int NCH = 9;             // 256-float chunks per row (chunk 8 holds floats 2048..2079)
constexpr int SEG = 548;           // floats per (mixer, segment) product row: 544 + 4 pad (conflict-free b128)
constexpr int PBUF = CMX_MIX0 * SEG;
constexpr int XS = 2112;
constexpr unsigned SPIN_LIMIT = 1u << 27;

struct Ctl {
  int staged[2];
  int consumed;
  int u_epoch;
  int scout_epoch;
  int tail_in;
  int tail_done;
  int abort;
  int b_in, b_done;   // cmx_mixnet_spec_kernel: bit + 1 handed from the layer-1 wave to the layer-2 / SSE wave, and finished by it
  int bit_epoch;      // late mode (a decoder, cmx_late.h): bit + 1 whose value the output wave has received from the host and put into Lds::bitring
  unsigned late_lo, late_hi;   // late mode: the box's address (0: a compressor's chunk) -- waits that depend on the decoder are bounded by wall-clock time, not by a spin count
};

struct BitRec {            // written by the scout for bit t (slot t % 3)
  uint32_t rowidx[48];
  uint32_t changed[32];    // layer-0 row differs from the previous bit's
  float aux3[4];           // clamped stretch of the three auxiliary inputs
  float lstm_p;            // raw probs[t][2077] (override test)
  int bit;
  uint32_t auxkey;         // cmx_mixnet_spec_kernel: auxiliary_context_ of the bit (predictor.cpp:388-393), from the stretch wave
  int pad;
};

struct TailRec {           // written by the chain wave for bit t (slot t & 1)
  float out0[32];          // clamped layer-0 outputs
  float aux3[4];
  uint32_t rowidx[24];     // rows of mixers 26..46
  float lstm_p;
  int bit;
  int pad[2];
};

struct Lds {
  float* prod;     // [2][PBUF]
  float* xs;       // [3][XS]
  BitRec* rec;     // [rr]
  int rr;          // depth of the rec ring: 3 (one-workgroup kernel), 8 (cmx_mixnet_spec_kernel)
  int lead;        // bits the scout may run ahead of the chain / gather wave: 2 resp. 4
  TailRec* trec;   // [2]
  float* upd;      // [32]
  uint32_t* dflag; // [32]
  float* in2;      // [64] (tail wave scratch)
  Ctl* ctl;
  float* w2;       // [64] layer-2 weight row
  float* w1;       // [20][68] layer-1 weight rows
  unsigned pfdump; // LDS byte offset of a 256-byte dump area for the row-prefetch LDS-DMA loads
  uint64_t* exptab; // [32] expf's table (cmx_libm.h): the chain wave's error needs it on the serial path
  const uint16_t* lst;  // cmx_mixnet_spec_kernel: LDS copies of the SSE's t_st / t_sq (64 KB each: five dependent look-ups per bit
  const uint16_t* lsq;  //   sit on the tail wave's path); nullptr: read them from global memory
  int* sdone;       // [rr] cmx_mixnet_spec_kernel: bit + 1 whose stretched inputs a stretch wave has published
  float* h2;        // [2][64] cmx_mixnet_spec_kernel: the layer-2 inputs of a bit (49) + bit, lstm_p, layer-2 row, from tail_a_role to tail_b_role
  int* bitring;     // [8] late mode: the decoded bits, slot bit % 8 (Ctl::bit_epoch)
  CmxLate late;     // late mode: the decoder's box + row counters; late.box == nullptr: every bit of the chunk is known (compression)
  int jit;          // CMX_MIXNET_JITTER (test hook): 0 = off, else the seed of the roles' pseudo-random stalls (jitter_stall)
};

// Test hook (CMX_MIXNET_JITTER=seed, tests/test_gpu_mixnet.py): a role stalls at pseudo-random bits for 0 .. ~100 us. The roles of this kernel are
// coupled only through counters and value|tag words, so every result must be independent of how far any role runs ahead of or behind any other --
// in a clean run the gather wave is the bottleneck and the leads sit at their maxima, the short-lead paths are the least exercised code of the
// kernel. Wave-uniform (t and who are).
template <bool JIT> __device__ __forceinline__ void jitter_stall(int jit, int t, int who) {
  if constexpr (!JIT) return;
  unsigned h = (unsigned)t * 2654435761u + (unsigned)who * 0x9E3779B9u + (unsigned)jit * 0x85EBCA6Bu;
  h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12;
  if ((h & 7u) == 0) {
    const int n = (int)((h >> 3) & 31u);
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }
}

// All inter-wave traffic of this kernel goes through LDS, so its synchronisation only has to
// order LDS operations (lgkmcnt). The C++ workgroup-scope acquire/release atomics also drain
// vmcnt -- i.e. every poll would wait for the wave's outstanding HBM row loads/stores -- so the
// flags are accessed with explicit DS instructions instead.
typedef __attribute__((address_space(3))) int lds_int;
__device__ __forceinline__ int lds_poll(const int* p) {
  int v;
  asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"((const lds_int*)p) : "memory");
  return v;
}
// publish: every earlier LDS write of this wave is complete before the flag moves
__device__ __forceinline__ void lds_publish_store(int* p, int v) {
  asm volatile("s_waitcnt lgkmcnt(0)\n\tds_write_b32 %0, %1" :: "v"((lds_int*)p), "v"(v) : "memory");
}
__device__ __forceinline__ void lds_publish_add1(int* p) {
  int one = 1;
  asm volatile("s_waitcnt lgkmcnt(0)\n\tds_add_u32 %0, %1" :: "v"((lds_int*)p), "v"(one) : "memory");
}
// Touch one cache line per lane without a register destination: LDS-DMA load of one dword per lane
// into LDS[lds_dst + 4*lane] (M0 = LDS destination base, saved and restored in the same statement).
__device__ __forceinline__ void touch_line(gptr<const float> g, unsigned lds_dst) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(g), "s"(dst) : "memory");
}
__device__ __forceinline__ int ld_acq(const int* p) { return lds_poll(p); }
__device__ __forceinline__ void st_rel(int* p, int v) { lds_publish_store(p, v); }
// Wave-uniform bounded spin until *p >= target. Returns false on abort/timeout.
// late mode: what the wave waits for may depend on the decoder (host); then only the box's abort / fail words and 30 s of wall-clock time end the wait
__device__ __forceinline__ CmxLateBox* ctl_box(Ctl* ctl) {
  const unsigned lo = (unsigned)lds_poll((const int*)&ctl->late_lo), hi = (unsigned)lds_poll((const int*)&ctl->late_hi);
  return reinterpret_cast<CmxLateBox*>(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ bool late_expired(CmxLateBox* B, unsigned long long& t0) {
  if (late_ld(&B->abort) || late_ld(&B->fail)) return true;
  const unsigned long long now = wall_clock64();
  if (!t0) { t0 = now; return false; }
  if (now - t0 > late_timeout_ticks(B)) { late_st(&B->fail, 1u); return true; }
  return false;
}
template <bool LATE = false> __device__ __forceinline__ bool wait_ge(Ctl* ctl, const int* p, int target, bool sleepy) {
  unsigned spins = 0;
  unsigned long long t0 = 0;
  while (lds_poll(p) < target) {
    if (sleepy) __builtin_amdgcn_s_sleep(2);
    if ((++spins & 1023u) == 0) {
      bool out;
      if (LATE) { CmxLateBox* const B = ctl_box(ctl); out = B ? late_expired(B, t0) : spins > SPIN_LIMIT; }
      else out = spins > SPIN_LIMIT;
      if (lds_poll(&ctl->abort) || out) {
        lds_publish_store(&ctl->abort, 1);
        return false;
      }
    }
  }
  return true;
}

// Broadcast lane j (a compile-time constant in the unrolled extra-input chains) through an SGPR:
// v_readlane_b32 costs a few clocks, ds_bpermute_b32 a full LDS round trip per step of the chain.
__device__ __forceinline__ float bcast_lane(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}

// MixerInput::SetStretchedInput / SetExtraInput clamp (mixer-input.cpp:17-27): if (x > max) x = max;
// else if (x < min) x = min;  -- as two selects, no branch.
// The same clamp as ONE instruction on the serial extra-input chain: v_med3_f32(x, min, max) is the median of the three, i.e.
// the reference's two-sided clamp for every non-NaN x (min < max always; a NaN never reaches a mixer output: its weights
// would have been NaN for every later bit in the reference as well). 1.9 k -> 1.2 k clocks per bit for the 26 steps.
__device__ __forceinline__ float clamp_med3(float x, float mn, float mx) { return __builtin_amdgcn_fmed3f(x, mn, mx); }
__device__ __forceinline__ float clamp_out(float x, float mn, float mx) {
  const float lo = x < mn ? mn : x;
  return x > mx ? mx : lo;
}

__device__ __forceinline__ float4 f4sub_mul(float4 w, float u, float4 x) {
  w.x = fsub(w.x, fmul(u, x.x));
  w.y = fsub(w.y, fmul(u, x.y));
  w.z = fsub(w.z, fmul(u, x.z));
  w.w = fsub(w.w, fmul(u, x.w));
  return w;
}
__device__ __forceinline__ float4 f4scale(float4 w, float c) {
  w.x = fmul(w.x, c); w.y = fmul(w.y, c); w.z = fmul(w.z, c); w.w = fmul(w.w, c);
  return w;
}
__device__ __forceinline__ float4 f4mul(float4 a, float4 b) {
  return make_float4(fmul(a.x, b.x), fmul(a.y, b.y), fmul(a.z, b.z), fmul(a.w, b.w));
}
__device__ __forceinline__ float add4(float p, float4 v) {
  p = fadd(p, v.x); p = fadd(p, v.y); p = fadd(p, v.z); p = fadd(p, v.w);
  return p;
}

// Ordered add chain over one staged segment (n = 512, or 542 for the last one): software-
// pipelined so the LDS reads of the next 32 terms are in flight while the current 32 are added.
// A (mixer, segment) row is SEG = 548 floats, so reading float4 128..135 is always in bounds.
// NB: whole batches of 32 floats in the segment (16: 512 / 542 terms; 8: 256 / 286 terms, the helpers' eight-wave split)
template <int NB> __device__ __forceinline__ float chain_seg_n(const float* rowp, int n, float p) {
  const float4* row = reinterpret_cast<const float4*>(__builtin_assume_aligned(rowp, 16));
  float4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = row[i];
#pragma unroll 1
  for (int bi = 0; bi < NB; bi += 2) {
    // sched_barrier(0): nothing may be scheduled across, so the next batch's LDS reads are
    // issued BEFORE the current batch's 32 dependent adds and retire underneath them.
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = row[(bi + 1) * 8 + i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) p = add4(p, a[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = row[(bi + 2) * 8 + i];   // bi == NB - 2: the remainder batch
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) p = add4(p, b[i]);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int rem4 = (n >> 2) - NB * 8;  // 0 or 7
#pragma unroll
  for (int i = 0; i < 7; ++i)
    if (i < rem4) p = add4(p, a[i]);
  if (n & 2) {                      // 542 = 135 * 4 + 2, 286 = 71 * 4 + 2
    p = fadd(p, a[7].x);
    p = fadd(p, a[7].y);
  }
  return p;
}
__device__ __forceinline__ float chain_seg(const float* rowp, int n, float p) { return chain_seg_n<16>(rowp, n, p); }

// ------------------------------------------------------------------ scout (wave 2)
// X != nullptr (cmx_mixnet_spec_kernel): the stretched inputs and the layer-0 rows of the bit are also published to the helper
// workgroups through the global ring (agent-scope stores, then the epoch).
__device__ void scout_role(MixState* S, const Lds& L, const float* probs, const uint32_t* sel,
                           const uint8_t* bits, int nbits, int lane, bool prof_on, SpecXfer* X = nullptr) {
  uint64_t tprev = __builtin_readcyclecounter();
#define SPROF(k)                                                       \
  do {                                                                 \
    if (prof_on) {                                                     \
      uint64_t now_ = __builtin_readcyclecounter();                    \
      pacc[k - 6] += now_ - tprev;                                     \
      tprev = now_;                                                    \
    }                                                                  \
  } while (0)
  uint64_t pacc[6] = {0, 0, 0, 0, 0, 0};
  const gptr<const float> lut = as_global(S->logit_lut);
  const gptr<const float> gprobs = as_global(probs);
  const gptr<const uint32_t> gsel = as_global(sel);
  const float smin = S->stretch_min, smax = S->stretch_max;
  for (int t = 0; t < nbits; ++t) {
    SPROF(11);
    if (!X) { if (t >= 2 && !wait_ge(L.ctl, &L.ctl->consumed, 4 * t - 4, true)) return; }
    else if (t >= L.lead && !wait_ge(L.ctl, &L.ctl->consumed, 4 * (t - L.lead) + 1, true)) return;   // the gather wave has begun bit t - lead
    // rec slot t % rr still holds bit t-rr, whose layer-1/2 row indices the tail wave reads at the start of
    // its bit t-rr: wait until it has finished that bit
    if (t >= L.rr && !wait_ge(L.ctl, &L.ctl->tail_done, t - L.rr + 1, true)) return;
    SPROF(6);
    float* xs = L.xs + (t % 3) * XS;
    BitRec* rec = L.rec + (t % L.rr);
    const BitRec* prev = L.rec + ((t + L.rr - 1) % L.rr);
    const gptr<const float> pr = gprobs + (size_t)t * CMX_IN0;
    // MixerInput::SetInput (mixer-input.cpp:11-15) + Sigmoid::Logit (sigmoid.cpp:12-17)
    float pv[33];
#pragma unroll
    for (int r = 0; r < 33; ++r) {
      int i = r * 64 + lane;
      pv[r] = i < CMX_IN0 ? pr[i] : 0.5f;
    }
    uint32_t key = lane < CMX_MIXERS ? gsel[(size_t)t * CMX_MIXERS + lane] : 0;
    int bitv = bits[t];
#pragma unroll
    for (int r = 0; r < 33; ++r) {
      float p = pv[r];
      if (p < 1.0e-4f) p = 1.0e-4f;
      else if (p > 1 - 1.0e-4f) p = 1 - 1.0e-4f;
      int idx = (int)(p * 100001.0f);
      if (idx >= 100001) idx = 100000;
      else if (idx < 0) idx = 0;
      pv[r] = lut[idx];
    }
    // the three auxiliary inputs (columns 433, 2024, 2077) sit in lanes 49, 40, 29 of rows 6, 31, 32
    const float ax0 = bcast_lane(pv[6], 49), ax1 = bcast_lane(pv[31], 40), ax2 = bcast_lane(pv[32], 29);
    if (!X) {   // the producers of the one-workgroup kernel read the inputs from LDS; the helpers get them through the global ring
#pragma unroll
      for (int r = 0; r < 33; ++r) {
        int i = r * 64 + lane;
        if (i < CMX_IN0) xs[i] = pv[r];
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
    }
    SPROF(7);
    if (lane == CMX_AUX) {  // predictor.cpp:388-393
      float avg = 0;
      avg = fadd(avg, cmx_logistic(ax0));
      avg = fadd(avg, cmx_logistic(ax1));
      avg = fadd(avg, cmx_logistic(ax2));
      avg = avg / 3.0f;
      key = (uint32_t)(unsigned long long)(avg * 15);
    }
    if (lane < CMX_MIXERS) {
      uint32_t r = select_row(S, lane, key);
      rec->rowidx[lane] = r;
      const uint32_t chg = (t == 0) || (r != prev->rowidx[lane]);
      if (lane < CMX_MIX0) rec->changed[lane] = chg;
      if (X && lane < CMX_MIX0) {
        __hip_atomic_store(&X->rowidx[t % CMX_SPEC_RING][lane], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&X->changed[t % CMX_SPEC_RING][lane], chg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (X) {
      float* gx = X->xs[t % CMX_SPEC_RING];
#pragma unroll
      for (int r = 0; r < 33; ++r) {
        int i = r * 64 + lane;
        if (i < CMX_IN0) __hip_atomic_store(gx + i, pv[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the ring slot is complete before the epoch moves
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_store(&X->scout_epoch, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    SPROF(8);
    SPROF(9);
    if (lane < 3) {
      float v = lane == 0 ? ax0 : lane == 1 ? ax1 : ax2;
      if (v > smax) v = smax;
      else if (v < smin) v = smin;
      rec->aux3[lane] = v;
    }
    if (lane == 0) {
      rec->lstm_p = pr[CMX_IN0 - 1];
      rec->bit = bitv;
    }
    st_rel(&L.ctl->scout_epoch, t + 1);
    SPROF(10);
  }
  if (prof_on && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) S->prof[6 + i] += pacc[i];
  }
#undef SPROF
}

// ------------------------------------------------------------------ producers (waves 3..11)
__device__ void producer_role(MixState* S, const Lds& L, int nbits, int p, int lane, bool prof_on, int dbg) {
  const gptr<float> rows0 = as_global(S->rows0);
  float4 W[NCH][MPW];
  int mj[MPW];
  bool ok[MPW];
#pragma unroll
  for (int j = 0; j < MPW; ++j) {
    mj[j] = p + NPROD * j;
    ok[j] = mj[j] < CMX_MIX0;
    if (!ok[j]) mj[j] = 0;
  }
  const float cdec = 1.0f - 3.0e-6f;
  const bool lane_ok8 = lane < 8;  // chunk 8 = floats 2048..2079: 8 lanes
  uint64_t pacc[6] = {0, 0, 0, 0, 0, 0};
  uint64_t tprev = __builtin_readcyclecounter();
#define PPROF(k)                                                       \
  do {                                                                 \
    if (prof_on && p == 0 && !(dbg & 6)) {                             \
      uint64_t now_ = __builtin_readcyclecounter();                    \
      pacc[k - 6] += now_ - tprev;                                     \
      tprev = now_;                                                    \
    }                                                                  \
  } while (0)
  for (int t = 0; t <= nbits; ++t) {
    PPROF(11);
    const bool live = t < nbits;   // t == nbits: flush the last update and store every row
    if (live && !wait_ge(L.ctl, &L.ctl->scout_epoch, t + 1, true)) return;
    PPROF(6);
    const float* xs = L.xs + (t % 3) * XS;
    const float* xsp = L.xs + ((t + 2) % 3) * XS;
    const BitRec* rec = L.rec + (t % L.rr);
    const BitRec* prev = L.rec + ((t + L.rr - 1) % L.rr);
    // Everything that does not depend on bit t-1's error is fetched BEFORE waiting for it, so the
    // window between "u published" and "segment 0 staged" (the serial part of the bit) is as short
    // as possible: row indices, the previous and current inputs of chunks 0 and 1.
    bool chg[MPW];
    uint32_t bold[MPW], bnew[MPW];
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
      chg[j] = ok[j] && (!live || (rec->changed[mj[j]] != 0 && !((dbg & 1) && t > 0)));  // dbg&1: timing experiment only
      bold[j] = (mj[j] * CMX_ROWS_PER_MIXER + prev->rowidx[mj[j]]) * CMX_ROW0_STRIDE;
      bnew[j] = (mj[j] * CMX_ROWS_PER_MIXER + rec->rowidx[mj[j]]) * CMX_ROW0_STRIDE;
    }
    const float4 xp0 = *reinterpret_cast<const float4*>(xsp + 4 * lane);
    const float4 xp1 = *reinterpret_cast<const float4*>(xsp + 256 + 4 * lane);
    const float4 xc0 = *reinterpret_cast<const float4*>(xs + 4 * lane);
    const float4 xc1 = *reinterpret_cast<const float4*>(xs + 256 + 4 * lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (t > 0 && !wait_ge(L.ctl, &L.ctl->u_epoch, t, false)) return;
    PPROF(7);
    float u[MPW];
    bool anydf = false;
    bool df[MPW];
#pragma unroll
    for (int j = 0; j < MPW; ++j) {
      u[j] = L.upd[mj[j]];
      df[j] = L.dflag[mj[j]] != 0;
      anydf |= df[j];
    }
    // Every row load of the previous bit has long landed; telling the compiler so here (a wait it can
    // see, on every path) keeps it from draining vmcnt(0) in front of each conditional store/load below.
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0), expcnt/lgkmcnt untouched (gfx9 encoding)
    // upd: apply bit t-1's update (mixer.cpp:66-71). Pure VALU; the rare 1024-step weight decay is a
    // separate pass so the common path has no per-element branch.
    auto upd = [&](int k, float4 x) {
#pragma unroll
      for (int j = 0; j < MPW; ++j) W[k][j] = f4sub_mul(W[k][j], u[j], x);
      if (anydf) {
#pragma unroll
        for (int j = 0; j < MPW; ++j)
          if (df[j]) W[k][j] = f4scale(W[k][j], cdec);
      }
    };
    // swp: 16-byte store of the outgoing row / load of the incoming one, for mixers whose selector changed
    auto swp = [&](int j, int k0, int k1) {
      if (chg[j]) {
#pragma unroll
        for (int k = 0; k < NCH; ++k) {
          if (k >= k0 && k < k1) {
            const int i = 256 * k + 4 * lane;
            if (k < 8 || lane_ok8) {
              if (t > 0) gstore4_async(rows0 + bold[j] + i, W[k][j]);
              if (live) W[k][j] = gload4(rows0 + bnew[j] + i);
            }
          }
        }
      }
    };
    // stage the rounded products of chunk k into segment buffer (mixer.cpp:41: in[i]*w[i])
    auto stageX = [&](int k, int q, float* buf, float4 x) {
      const int i = 256 * k + 4 * lane;
#pragma unroll
      for (int j = 0; j < MPW; ++j)
        if (ok[j]) *reinterpret_cast<float4*>(buf + mj[j] * SEG + (i - 512 * q)) = f4mul(x, W[k][j]);
    };
    auto stageK = [&](int k, int q, float* buf) {
      if (k < 8 || lane_ok8) stageX(k, q, buf, *reinterpret_cast<const float4*>(xs + 256 * k + 4 * lane));
    };
    auto publish = [&](int g) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) lds_publish_add1(&L.ctl->staged[g & 1]);
    };
    // ---- serial window: chunks 0,1 -> segment 0 ----
    if (t > 0) { upd(0, xp0); upd(1, xp1); }
#pragma unroll
    for (int j = 0; j < MPW; ++j) swp(j, 0, 2);
    if (live) {
      // buffer (4t)&1 was last read for segment 2 of bit t-1, which the chain finished before it
      // published u: no need to poll `consumed` here (nor for segment 1)
      float* buf = L.prod + ((4 * t) & 1) * PBUF;
      stageX(0, 0, buf, xc0);
      stageX(1, 0, buf, xc1);
      publish(4 * t);
    }
    PPROF(8);
    // ---- the rest runs underneath the chain wave's segment 0 ----
    if (t > 0) {
#pragma unroll
      for (int k = 2; k < NCH; ++k)
        if (k < 8 || lane_ok8) upd(k, *reinterpret_cast<const float4*>(xsp + 256 * k + 4 * lane));
    }
#pragma unroll
    for (int j = 0; j < MPW; ++j) swp(j, 2, NCH);
    PPROF(9);
    if (live) {
#pragma unroll
      for (int q = 1; q < 4; ++q) {
        const int g = 4 * t + q;
        if (q >= 2 && !wait_ge(L.ctl, &L.ctl->consumed, g - 1, true)) return;
        float* buf = L.prod + (g & 1) * PBUF;
        stageK(2 * q, q, buf);
        stageK(2 * q + 1, q, buf);
        if (q == 3) stageK(8, q, buf);
        publish(g);
      }
    }
    // Pull the rows this wave will swap in at the NEXT bit towards L2 while it would otherwise idle
    // waiting for the error of this bit: one dword per 128-byte line, 66 lines per row. The values are
    // discarded: the loads are LDS-DMA (global_load_lds_dword) into a 256-byte dump area, so they have
    // NO register destination -- an ordinary load issued from asm and never waited for may land in its
    // VGPR after the compiler has re-assigned that register. They are drained by the vmcnt(0) at the
    // top of the next serial window, ~10k clocks later. Skipped when the scout has not published the
    // next bit yet.
    if (t + 1 < nbits && lds_poll(&L.ctl->scout_epoch) >= t + 2) {
      const BitRec* nxt = L.rec + ((t + 1) % L.rr);
#pragma unroll
      for (int j = 0; j < MPW; ++j) {
        if (ok[j] && nxt->changed[mj[j]]) {
          const uint32_t base = (mj[j] * CMX_ROWS_PER_MIXER + nxt->rowidx[mj[j]]) * CMX_ROW0_STRIDE;
          touch_line(rows0 + base + 32 * lane, L.pfdump);
          if (lane < 2) touch_line(rows0 + base + 2048 + 32 * lane, L.pfdump);
        }
      }
    }
    PPROF(10);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (prof_on && p == 0 && !(dbg & 6) && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) S->prof[6 + i] += pacc[i];
  }
#undef PPROF
}

// ------------------------------------------------------------------ chain (wave 0)
__device__ void chain_role(MixState* S, const Lds& L, const float* decay1, int nbits,
                           float* mix_out, bool prof_on, int lane, int dbg) {
  const int m = lane;
  const bool is0 = m < CMX_MIX0;
  const float smin = S->stretch_min, smax = S->stretch_max;
  const float cdec = 1.0f - 3.0e-6f;
  const float lr = is0 ? S->lr[m] : 0.0f;
  uint64_t tprev = __builtin_readcyclecounter();
  // phase timers accumulate in scalar registers and are written once at the end: a global
  // read-modify-write per timer would put an L2 round trip into every phase it measures
  uint64_t pacc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) pacc[i] = 0;
#define CPROF(k)                                                       \
  do {                                                                 \
    if (prof_on) {                                                     \
      uint64_t now_ = __builtin_readcyclecounter();                    \
      pacc[k] += now_ - tprev;                                         \
      tprev = now_;                                                    \
    }                                                                  \
  } while (0)
  __builtin_amdgcn_s_setprio(3);
  // Per-row state of lane m's current weight row stays in registers while the selector does not
  // change (like the producers' weights): the 0..25 extra weights (mixer.cpp:45-53), the row's step
  // counter (ContextData::steps) and the mixer's max_steps_. It is swapped only when the row changes.
  const int mm = is0 ? m : 0;
  float ew[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) ew[i] = 0.0f;
  uint64_t rsteps = 0;
  uint64_t mx = S->max_steps[mm];
  gptr<float> row0 = as_global(S->rows0);
  gptr<uint64_t> rsp = as_global(S->row_steps);
  auto store_row_state = [&]() {
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (4 * i < m) gstore4_async(row0 + CMX_ROW0_EXTRA + 4 * i, make_float4(ew[4 * i], ew[4 * i + 1], ew[4 * i + 2], ew[4 * i + 3]));
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" :: "v"(rsp), "v"(rsteps) : "memory");
  };
  for (int t = 0; t < nbits; ++t) {
    if (!wait_ge(L.ctl, &L.ctl->scout_epoch, t + 1, false)) return;
    CPROF(0);
    const BitRec* rec = L.rec + (t % L.rr);
    const int bit = rec->bit;
    if (is0 && rec->changed[mm] && !((dbg & 8) && t > 0)) {  // dbg&8: timing experiment only (the chain wave keeps its first row state)
      // asm stores: re-using ew[] for the incoming row must not make the compiler wait for their acks
      if (t > 0) store_row_state();
      row0 = as_global(S->rows0) + ((size_t)mm * CMX_ROWS_PER_MIXER + rec->rowidx[mm]) * CMX_ROW0_STRIDE;
      rsp = as_global(S->row_steps) + (size_t)mm * CMX_ROWS_PER_MIXER + rec->rowidx[mm];
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        float4 v = gload4(row0 + CMX_ROW0_EXTRA + 4 * i);
        ew[4 * i] = v.x; ew[4 * i + 1] = v.y; ew[4 * i + 2] = v.z; ew[4 * i + 3] = v.w;
      }
      rsteps = *rsp;
#pragma unroll
      for (int i = 0; i < 28; ++i)
        if (i >= m) ew[i] = 0.0f;  // only j < m are weights of mixer m (mixer.cpp:45-53); the rest is padding
    }
    const double d1 = (double)as_global(decay1)[t];  // (float)(0.9/pow(1e-7*steps_+0.8,0.8)), host libm

    float pm = 0.0f, dlr = 0.0f;
#pragma unroll 1
    for (int q = 0; q < 4; ++q) {
      const int g = 4 * t + q;
      if (!wait_ge(L.ctl, &L.ctl->staged[g & 1], NPROD * ((g >> 1) + 1), false)) return;
      if (q == 0) CPROF(1); else if (q == 1) CPROF(13); else if (q == 2) CPROF(14); else CPROF(15);
      pm = chain_seg(L.prod + (g & 1) * PBUF + mm * SEG, q == 3 ? CMX_IN0 - 1536 : 512, pm);
      st_rel(&L.ctl->consumed, g + 1);
      if (q == 0) {
        // decay * lr (mixer.cpp:58-60): double-precision, needs only state fetched above; done here,
        // where the row loads have landed and the next segment is usually still being staged
        float decay = (float)(d1 * (1.5 - ((1.0 * (double)rsteps) / (double)mx)));
        dlr = fmul(decay, lr);
      }
      CPROF(2);
    }
    // intra-layer chain (predictor.cpp:395-400, mixer.cpp:45-53): mixer j's clamped output is an
    // extra input of every later mixer. Serial over j, so each step is kept to add -> clamp -> SGPR
    // broadcast -> mul -> add with no branch and no per-step lane select:
    //  * ew[j] is exactly 0 for j >= m (zeroed when the row is loaded), so every lane can add
    //    oj*ew[j] unconditionally: e + (+-0) == e bit for bit (e is never -0: it starts at +0);
    //  * lane m's own p_ is pm + e after the loop (only terms j < m ever changed its e);
    //  * the clamp (mixer-input.cpp:23-27) is done per lane before the broadcast.
    float e = 0.0f;
#pragma unroll
    for (int j = 0; j < CMX_MIX0; ++j) {
      const float mine = clamp_med3(fadd(pm, e), smin, smax);
      const float oj = bcast_lane(mine, j);
      e = fadd(e, fmul(oj, ew[j]));
    }
    const float p_ = fadd(pm, e);
    const float myout = clamp_out(p_, smin, smax);
    CPROF(3);
    // Mixer::Perceive scalar (mixer.cpp:56-64)
    float uu = fmul(dlr, fsub(cmx_logistic_t(p_, L.exptab), (float)bit));
    ++rsteps;
    if (rsteps > mx) mx = rsteps;
    const bool dfl = (rsteps & 1023) == 0;
    if (is0) {
      L.upd[m] = uu;
      L.dflag[m] = dfl;
    }
    st_rel(&L.ctl->u_epoch, t + 1);
    CPROF(4);
    // hand the layer-0 outputs to the tail wave
    if (t >= 2 && !wait_ge(L.ctl, &L.ctl->tail_done, t - 1, false)) return;
    CPROF(12);
    TailRec* tr = L.trec + (t & 1);
    if (is0) tr->out0[m] = myout;
    if (m >= CMX_MIX0 && m < CMX_MIXERS) tr->rowidx[m - CMX_MIX0] = rec->rowidx[m];
    if (m < 3) tr->aux3[m] = rec->aux3[m];
    if (m == 0) { tr->lstm_p = rec->lstm_p; tr->bit = bit; }
    st_rel(&L.ctl->tail_in, t + 1);
    if (is0 && mix_out) as_global(mix_out)[(size_t)t * CMX_MIXERS + m] = p_;
    // extra weights: ew[j] -= u * out_j (mixer.cpp:67,70)
#pragma unroll
    for (int j = 0; j < CMX_MIX0; ++j) {
      float oj = bcast_lane(myout, j);
      if (j < m) {
        float v = fsub(ew[j], fmul(uu, oj));
        if (dfl) v = fmul(v, cdec);
        ew[j] = v;
      }
    }
    CPROF(5);
  }
  if (is0 && nbits > 0) {
    store_row_state();
    S->max_steps[m] = mx;
  }
  if (prof_on && lane == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < 6 || i >= 12) S->prof[i] += pacc[i];
  }
#undef CPROF
}

// ------------------------------------------------------------------ tail (wave 1)
// One bit behind the chain wave: layer 1, layer 2, squash, SSE, LSTM override, output, and the
// Perceive of those 21 mixers and of the SSE (predictor.cpp:402-418,432-437).
//
// Everything whose ADDRESS is known before the layer-0 outputs arrive is fetched before waiting for
// them: the layer-1 rows (kept in registers while a mixer's selector does not change, swapped with
// asm stores otherwise), and every SSE cell the bit can possibly touch -- the four SSE contexts
// depend on the final probability only through a 3-way / 4-way quantisation (sse.cpp:248-262), so the
// 3+3 interpolation cells and 4+3 mixer weights are all loaded up front and selected afterwards.
// The single layer-2 row lives in LDS for the whole chunk. The SSE context registers (sse.cpp:222-240)
// stay in registers. What remains serial per bit is arithmetic plus the t_st/t_sq lookups (L1/L2 hits).
struct U16x8 { unsigned w0, w1, w2, w3; };   // (named words: as an array the cell went through scratch memory)
__device__ __forceinline__ U16x8 load_cell(const uint16_t* p) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u v = *(gptr<const v4u>)as_global(p);
  U16x8 r; r.w0 = v.x; r.w1 = v.y; r.w2 = v.z; r.w3 = v.w;
  return r;
}
__device__ __forceinline__ void store_cell_async(uint16_t* p, const U16x8& c) {
  typedef unsigned v4u __attribute__((ext_vector_type(4)));
  v4u o = {c.w0, c.w1, c.w2, c.w3};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(as_global(p)), "v"(o) : "memory");
}
__device__ __forceinline__ int cell_get(const U16x8& c, int i) {  // entry i of 8 u16
  unsigned w = i < 2 ? c.w0 : i < 4 ? c.w1 : i < 6 ? c.w2 : c.w3;
  return (int)((i & 1) ? (w >> 16) : (w & 0xffffu));
}
__device__ __forceinline__ void cell_set(U16x8& c, int i, int v) {
  const unsigned m = (i & 1) ? 0x0000ffffu : 0xffff0000u, x = ((unsigned)v & 0xffffu) << ((i & 1) * 16);
  const int q = i >> 1;
  c.w0 = q == 0 ? (c.w0 & m) | x : c.w0;
  c.w1 = q == 1 ? (c.w1 & m) | x : c.w1;
  c.w2 = q == 2 ? (c.w2 & m) | x : c.w2;
  c.w3 = q == 3 ? (c.w3 & m) | x : c.w3;
}
__device__ __forceinline__ unsigned bcast_u(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 0); }
template <class T> __device__ __forceinline__ T pick3(int a, T x0, T x1, T x2) { return a == 0 ? x0 : a == 1 ? x1 : x2; }

struct SseInterp {  // SSEi<7>::SSE_Pred / SSE_Update on a cell held in registers (sse.cpp:37-62)
  int freq, sw, P;
  __device__ __forceinline__ int pred(const U16x8& c, int iP) {
    freq = (6 * iP) >> 15;
    sw = (6 * iP) & 32767;
    int f = (((32768 - sw) * cell_get(c, freq) + sw * cell_get(c, freq + 1)) >> 15) - 8192;
    if (f <= 0) f = 1;
    if (f >= 32768) f = 32767;
    P = f;
    return f;
  }
  __device__ __forceinline__ void update(U16x8& c, int bit, int wr0) {
    P = (P * (32768 - wr0)) >> 15;
    if (bit == 0) P += wr0;
    const int c0 = cell_get(c, freq), c1 = cell_get(c, freq + 1);
    const int dC = c0 - c1;
    const int sw_dC = (sw * dC + 32767) >> 15;
    cell_set(c, freq, (uint16_t)(P + sw_dC + 8192));
    cell_set(c, freq + 1, (uint16_t)(P - (dC - sw_dC) + 8192));
  }
};

__device__ void tail_role(MixState* S, const Lds& L, const float* decay1, int nbits, float* p_out,
                          float* mix_out, int lane, bool prof_on) {
  uint64_t tprev = __builtin_readcyclecounter();
  uint64_t pacc[6] = {0, 0, 0, 0, 0, 0};
#define TPROF(k)                                                       \
  do {                                                                 \
    if (prof_on) {                                                     \
      uint64_t now_ = __builtin_readcyclecounter();                    \
      pacc[k - 6] += now_ - tprev;                                     \
      tprev = now_;                                                    \
    }                                                                  \
  } while (0)
  const int k = lane;  // layer-1 mixer index
  const bool is1 = k < CMX_MIX1;
  const int kk = is1 ? k : 0;
  const float smin = S->stretch_min, smax = S->stretch_max;
  const float cdec = 1.0f - 3.0e-6f;
  const float lr1 = S->lr[CMX_MIX0 + kk], lr2 = S->lr[CMX_MIXERS - 1];
  const uint16_t* const t_st = S->t_st;
  const uint16_t* const t_sq = S->t_sq;
  uint16_t* const s6 = S->s6;
  uint16_t* const s7 = S->s7;
  int* const x1 = S->x1;
  int* const x2 = S->x2;
  const gptr<float> rows1 = as_global(S->rows1);
  const gptr<uint64_t> row_steps = as_global(S->row_steps);
  float* const w2 = L.w2;  // layer-2 row (predictor.cpp:354-356: a single weight set), LDS-resident
  const gptr<float> rows2 = as_global(S->rows2);
  const gptr<uint64_t> rsteps2 = row_steps + (size_t)(CMX_MIXERS - 1) * CMX_ROWS_PER_MIXER;
  // ---- chunk prologue ----
  uint32_t cur_row2 = 0xffffffffu;
  uint64_t rs2 = 0, mx2 = S->max_steps[CMX_MIXERS - 1];
  uint64_t mx1 = S->max_steps[CMX_MIX0 + kk], rs1 = 0;
  unsigned sj = S->sse_j, spc = S->sse_pc, sffl = S->sse_ffl;
  uint64_t steps_done = 0;
  // layer-1 row of mixer k: LDS-resident (the tail's register budget goes to the SSE arithmetic);
  // 68-float pitch keeps the rows 16-byte aligned and spreads them over the banks
  float* const w1 = L.w1 + kk * 68;
  uint32_t cur_row = 0xffffffffu;
  gptr<float> row1 = rows1;
  gptr<uint64_t> rsp1 = row_steps;
  auto store_row1 = [&]() {
#pragma unroll
    for (int i = 0; i < 13; ++i)
      gstore4_async(row1 + 4 * i, *reinterpret_cast<const float4*>(w1 + 4 * i));
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" :: "v"(rsp1), "v"(rs1) : "memory");
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int t = 0; t < nbits; ++t) {
    // ---- before the layer-0 outputs exist: rows and SSE cells of bit t ----
    TPROF(11);
    if (!wait_ge(L.ctl, &L.ctl->scout_epoch, t + 1, true)) return;
    const BitRec* rec = L.rec + (t % L.rr);
    const uint32_t newrow = rec->rowidx[CMX_MIX0 + kk];
    const double d1 = (double)as_global(decay1)[t];
    if (is1 && newrow != cur_row) {
      if (cur_row != 0xffffffffu) store_row1();
      cur_row = newrow;
      row1 = rows1 + ((size_t)kk * CMX_ROWS_PER_MIXER + newrow) * CMX_ROW1_STRIDE;
      rsp1 = row_steps + (size_t)(CMX_MIX0 + kk) * CMX_ROWS_PER_MIXER + newrow;
#pragma unroll
      for (int i = 0; i < 13; ++i) *reinterpret_cast<float4*>(w1 + 4 * i) = gload4(row1 + 4 * i);
      rs1 = *rsp1;
#pragma unroll
      for (int j = 0; j < CMX_MIX1; ++j)
        if (j >= k) w1[CMX_ROW1_EXTRA + j] = 0.0f;  // only j < k are extra weights of mixer k
    }
    {  // layer 2 has one weight set in cmix (selector = zero_context_); a changing key is still honoured
      const uint32_t newrow2 = rec->rowidx[CMX_MIXERS - 1];
      if (newrow2 != cur_row2) {
        if (cur_row2 != 0xffffffffu) {
          if (k < 16) gstore4(rows2 + (size_t)cur_row2 * CMX_ROW2_STRIDE + 4 * k, *reinterpret_cast<const float4*>(w2 + 4 * k));
          if (k == 0) rsteps2[cur_row2] = rs2;
        }
        cur_row2 = newrow2;
        if (k < 16) *reinterpret_cast<float4*>(w2 + 4 * k) = gload4(rows2 + (size_t)newrow2 * CMX_ROW2_STRIDE + 4 * k);
        rs2 = rsteps2[newrow2];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
    // pull the layer-1 rows of the NEXT bit towards the caches (2 lines per row), if the scout is there yet
    if (t + 1 < nbits && lds_poll(&L.ctl->scout_epoch) >= t + 2) {
      const uint32_t nr = L.rec[(t + 1) % L.rr].rowidx[CMX_MIX0 + kk];
      if (is1 && nr != cur_row) {
        const gptr<const float> nrow = rows1 + ((size_t)kk * CMX_ROWS_PER_MIXER + nr) * CMX_ROW1_STRIDE;
        touch_line(nrow, L.pfdump + 256 + 256 + 64);
        touch_line(nrow + 32, L.pfdump + 256 + 256 + 64);
      }
    }
    // SSE contexts for a = 0..2 / b = 0..3 (sse.cpp:248-262): lane i pulls candidate cell i towards the
    // caches (no register destination: LDS-DMA into the dump area); lane 0 reloads the selected ones below
    if (k < 13) {
      const unsigned j_ = bcast_u(sj), pc_ = bcast_u(spc), ffl_ = bcast_u(sffl);
      const int a = k % 3, b = k - 9;
      const float* addr;
      if (k < 3) addr = (const float*)(s6 + (size_t)(((((a << 7) + (int)(ffl_ & 127)) << 8) + (int)(pc_ & 255)) * 256 + (int)j_) * 8);
      else if (k < 6) addr = (const float*)(s7 + (size_t)(((((a << 5) + (int)(ffl_ & 31)) << 8) + (int)(pc_ & 255)) * 255 + (j_ ? (int)j_ - 1 : 0)) * 8);
      else if (k < 9) addr = (const float*)(x2 + (((((a << 1) + (int)(ffl_ & 1)) << 8) + (int)(pc_ & 255)) * 256 + (int)j_));
      else addr = (const float*)(x1 + (((((b << 8) + (int)(ffl_ & 255)) << 3) + (int)((pc_ >> 5) & 7)) * 79 + sse_mx1mask((int)j_)));
      touch_line(as_global(addr), L.pfdump + 256 + 256);  // own 64-byte dump slot behind w2 (not the producers' one)
    }
    // ---- layer-0 outputs of bit t ----
    TPROF(6);
    if (!wait_ge(L.ctl, &L.ctl->tail_in, t + 1, false)) return;
    TPROF(7);
    const TailRec* tr = L.trec + (t & 1);
    const int bit = tr->bit;
    // layer-2 inputs = 26 layer-0 outputs + 20 layer-1 outputs + 3 auxiliary stretches (predictor.cpp:
    // 397-411); the first 26 and the last 3 are also the layer-1 inputs, in this order (:402-406). They
    // are kept in LDS (in2) and read from there wherever they are needed, not held in registers.
    if (k < CMX_MIX0) L.in2[k] = tr->out0[k];
    if (k < 3) L.in2[CMX_MIX0 + CMX_MIX1 + k] = tr->aux3[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float pm = 0.0f;
#pragma unroll
    for (int i = 0; i < CMX_MIX0; ++i) pm = fadd(pm, fmul(L.in2[i], w1[i]));
#pragma unroll
    for (int i = 0; i < 3; ++i) pm = fadd(pm, fmul(L.in2[CMX_MIX0 + CMX_MIX1 + i], w1[CMX_MIX0 + i]));
    // intra-layer chain, branch-free (see chain_role): extra weights j >= k are exactly 0
    float e = 0.0f;
#pragma unroll
    for (int j = 0; j < CMX_MIX1; ++j) {
      const float mine = clamp_out(fadd(pm, e), smin, smax);
      const float oj = bcast_lane(mine, j);
      e = fadd(e, fmul(oj, w1[CMX_ROW1_EXTRA + j]));
    }
    const float p1_ = fadd(pm, e);
    const float myout = clamp_out(p1_, smin, smax);
    if (is1) L.in2[CMX_MIX0 + k] = myout;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float u2 = 0.0f;
    int df2 = 0;
    TPROF(8);
    if (k == 0) {
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const float4 xv = *reinterpret_cast<const float4*>(L.in2 + 4 * i);
        const float4 wv = *reinterpret_cast<const float4*>(w2 + 4 * i);
        acc = fadd(acc, fmul(xv.x, wv.x)); acc = fadd(acc, fmul(xv.y, wv.y));
        acc = fadd(acc, fmul(xv.z, wv.z)); acc = fadd(acc, fmul(xv.w, wv.w));
      }
      acc = fadd(acc, fmul(L.in2[48], w2[48]));
      const float p2_ = acc;
      const float sq = cmx_logistic(p2_);                  // predictor.cpp:413
      // ---- SSE::Predict (sse.cpp:243-290,320-324) on the pre-fetched cells ----
      const int p = (int)(1 + (1 - sq) * 32766);
      const unsigned prq = (unsigned)p >> 11;
      const int a = (prq > 0) + (prq > 14);
      const int b = (prq > 0) + (prq > 7) + (prq > 14);
      const int i6 = ((((a << 7) + (int)(sffl & 127)) << 8) + (int)(spc & 255)) * 256 + (int)sj;
      const int i7 = ((((a << 5) + (int)(sffl & 31)) << 8) + (int)(spc & 255)) * 255 + (sj ? (int)sj - 1 : 0);
      const int im2 = ((((a << 1) + (int)(sffl & 1)) << 8) + (int)(spc & 255)) * 256 + (int)sj;
      const int im1 = ((((b << 8) + (int)(sffl & 255)) << 3) + (int)((spc >> 5) & 7)) * 79 + sse_mx1mask((int)sj);
      U16x8 c6 = load_cell(s6 + (size_t)i6 * 8);
      U16x8 c7 = load_cell(s7 + (size_t)i7 * 8);
      const int w1x = as_global(x1)[im1];
      const int w2x = as_global(x2)[im2];
      SseInterp e6, e7;
      auto ST = [&](int i) -> int { return L.lst ? (int)L.lst[i] : (int)as_global(t_st)[i]; };
      auto SQ = [&](int i) -> int { return L.lsq ? (int)L.lsq[i] : (int)as_global(t_sq)[i]; };
      const int stp = ST(p);
      const int q6 = SQ(sse_extrap(stp, 10240)), q7 = SQ(sse_extrap(stp, 8200));
      const int pp1 = e6.pred(c6, q6);
      const int pp2 = e7.pred(c7, q7);
      const int s0 = sse_extrap(stp, 7935);
      const int s1 = sse_extrap(ST(pp1), 9592);
      const int s4 = sse_extrap(ST(pp2), 7677);
      const int s2 = sse_extrap(sse_mixup(w1x, s0, s1), 8092);
      const int mix1_p = SQ(s2);
      const int s5 = sse_extrap(sse_mixup(w2x, s2, s4), 8202);
      const int mix2_p = SQ(s5);
      float pf = (float)(1 - ((mix2_p - 1) / 32766.0));
      const float lp = tr->lstm_p;
      if (lp == 0.0f || lp == 1.0f) pf = lp;               // predictor.cpp:383,415-417
      as_global(p_out)[t] = pf;
      // ---- SSE::Perceive (sse.cpp:291-306,326-328) ----
      e6.update(c6, bit, 106);
      e7.update(c7, bit, 127);
      store_cell_async(s6 + (size_t)i6 * 8, c6);
      store_cell_async(s7 + (size_t)i7 * 8, c7);
      as_global(x1)[im1] = w1x + sse_wdelta(bit, s0, s1, 6202, mix1_p);
      as_global(x2)[im2] = w2x + sse_wdelta(bit, s2, s4, 8320, mix2_p);
      sj += sj + (unsigned)bit;
      if (sj >= 256) {
        sffl = (sffl * 2 + (spc >= 0x40)) & 255;
        spc = sj & 255;
        sj = 1;
      }
      // ---- Mixer::Perceive, layer 2 (mixer.cpp:56-72) ----
      const float decay = (float)(d1 * (1.5 - ((1.0 * (double)rs2) / (double)mx2)));
      u2 = fmul(fmul(decay, lr2), fsub(sq, (float)bit));
      ++rs2;
      if (rs2 > mx2) mx2 = rs2;
      df2 = (rs2 & 1023) == 0;
      if (mix_out) as_global(mix_out)[(size_t)t * CMX_MIXERS + CMX_MIXERS - 1] = p2_;
      ++steps_done;
    }
    TPROF(9);
    u2 = bcast_lane(u2, 0);
    df2 = __builtin_amdgcn_readlane(df2, 0);
    if (k < CMX_IN2) {  // layer-2 weights: one lane per weight
      float v = fsub(w2[k], fmul(u2, L.in2[k]));
      if (df2) v = fmul(v, cdec);
      w2[k] = v;
    }
    // ---- Mixer::Perceive, layer 1 ----
    if (is1) {
      const float decay = (float)(d1 * (1.5 - ((1.0 * (double)rs1) / (double)mx1)));
      const float u = fmul(fmul(decay, lr1), fsub(cmx_logistic(p1_), (float)bit));
      ++rs1;
      if (rs1 > mx1) mx1 = rs1;
      const bool df = (rs1 & 1023) == 0;
      // w -= u*x over the 29 weights and the k extra weights (mixer.cpp:66-71). All operands are read
      // into registers first and written back at the end: w1 and in2 are both LDS, and element-wise
      // read-modify-write through possibly aliasing pointers would serialise on an LDS round trip each.
      float4 xv[13], wv[13];
#pragma unroll
      for (int i = 0; i < 13; ++i) wv[i] = *reinterpret_cast<const float4*>(w1 + 4 * i);
#pragma unroll
      for (int i = 0; i < 6; ++i) xv[i] = *reinterpret_cast<const float4*>(L.in2 + 4 * i);          // out0[0..23]
      xv[6] = make_float4(L.in2[24], L.in2[25], L.in2[46], L.in2[47]);                               // out0[24,25], aux[0,1]
      xv[7] = make_float4(L.in2[48], 0.0f, 0.0f, 0.0f);                                              // aux[2], pad
#pragma unroll
      for (int i = 0; i < 5; ++i) {                                                                  // layer-1 outputs 0..19
        xv[8 + i] = make_float4(L.in2[CMX_MIX0 + 4 * i], L.in2[CMX_MIX0 + 4 * i + 1], L.in2[CMX_MIX0 + 4 * i + 2],
                                L.in2[CMX_MIX0 + 4 * i + 3]);
      }
      const float c = df ? cdec : 1.0f;  // w * 1.0f == w exactly
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        float4 v = f4sub_mul(wv[i], u, xv[i]);
        if (i >= 8) {  // extra weights: only j < k belong to mixer k, the rest stays exactly 0
          const int j0 = 4 * (i - 8);
          if (j0 + 0 >= k) v.x = 0.0f;
          if (j0 + 1 >= k) v.y = 0.0f;
          if (j0 + 2 >= k) v.z = 0.0f;
          if (j0 + 3 >= k) v.w = 0.0f;
        }
        wv[i] = f4scale(v, c);
      }
#pragma unroll
      for (int i = 0; i < 13; ++i) *reinterpret_cast<float4*>(w1 + 4 * i) = wv[i];
      if (mix_out) as_global(mix_out)[(size_t)t * CMX_MIXERS + CMX_MIX0 + k] = p1_;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    st_rel(&L.ctl->tail_done, t + 1);
    TPROF(10);
  }
  if (prof_on && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) S->prof[6 + i] += pacc[i];
  }
#undef TPROF
  // ---- chunk epilogue: registers / LDS -> HBM ----
  if (nbits > 0) {
    if (is1 && cur_row != 0xffffffffu) {
      store_row1();
      S->max_steps[CMX_MIX0 + k] = mx1;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (k < 16 && cur_row2 != 0xffffffffu)
      gstore4(rows2 + (size_t)cur_row2 * CMX_ROW2_STRIDE + 4 * k, *reinterpret_cast<const float4*>(w2 + 4 * k));
    if (k == 0) {
      if (cur_row2 != 0xffffffffu) rsteps2[cur_row2] = rs2;
      S->max_steps[CMX_MIXERS - 1] = mx2;
      S->sse_j = sj; S->sse_pc = spc; S->sse_ffl = sffl;
      S->steps = S->steps + steps_done;
    }
  }
}


// ------------------------------------------------------------------ the tail on two waves (cmx_mixnet_spec_kernel)
// tail_role's bit is layer 1 (dot products, intra-layer chain, its 20 updates) followed by layer 2 + SSE + output on one lane: 17 k clocks
// in a row, the longest role of the kernel once the layer-0 chains moved to the helper workgroups. Layer 1's learning needs nothing from
// layer 2 (every mixer learns from its own output, mixer.cpp:56-72), so the two halves are a pipeline: tail_a_role (wave 1) does layer 1 of
// bit t and hands the 49 layer-2 inputs over through LDS (Lds::h2, two slots); tail_b_role (wave 3) does layer 2, the SSE and the output of
// bit t while wave 1 is on bit t + 1. Same arithmetic, same order, per half as in tail_role.
template <bool LATE, bool JIT = false> __device__ void tail_a_role(MixState* S, const Lds& L, const float* decay1, int nbits, float* mix_out, int lane, bool prof_on) {
  uint64_t tprev = __builtin_readcyclecounter();
  uint64_t pacc[6] = {0, 0, 0, 0, 0, 0};
#define TPROF(k) do { if (prof_on) { uint64_t now_ = __builtin_readcyclecounter(); pacc[k - 6] += now_ - tprev; tprev = now_; } } while (0)
  const int k = lane;  // layer-1 mixer index
  const bool is1 = k < CMX_MIX1;
  const int kk = is1 ? k : 0;
  const float smin = S->stretch_min, smax = S->stretch_max;
  const float cdec = 1.0f - 3.0e-6f;
  const float lr1 = S->lr[CMX_MIX0 + kk];
  const gptr<float> rows1 = as_global(S->rows1);
  const gptr<uint64_t> row_steps = as_global(S->row_steps);
  uint64_t mx1 = S->max_steps[CMX_MIX0 + kk], rs1 = 0;
  float* const w1 = L.w1 + kk * 68;
  uint32_t cur_row = 0xffffffffu;
  gptr<float> row1 = rows1;
  gptr<uint64_t> rsp1 = row_steps;
  auto store_row1 = [&]() {
#pragma unroll
    for (int i = 0; i < 13; ++i) gstore4_async(row1 + 4 * i, *reinterpret_cast<const float4*>(w1 + 4 * i));
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" :: "v"(rsp1), "v"(rs1) : "memory");
  };
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  for (int t = 0; t < nbits; ++t) {
    TPROF(11);
    jitter_stall<JIT>(L.jit, t, 1);
    if (!wait_ge<LATE>(L.ctl, &L.ctl->scout_epoch, t + 1, true)) return;
    const BitRec* rec = L.rec + (t % L.rr);
    const uint32_t newrow = rec->rowidx[CMX_MIX0 + kk];
    const uint32_t row2 = rec->rowidx[CMX_MIXERS - 1];
    const double d1 = (double)as_global(decay1)[t];
    if (is1 && newrow != cur_row) {
      if (cur_row != 0xffffffffu) store_row1();
      cur_row = newrow;
      row1 = rows1 + ((size_t)kk * CMX_ROWS_PER_MIXER + newrow) * CMX_ROW1_STRIDE;
      rsp1 = row_steps + (size_t)(CMX_MIX0 + kk) * CMX_ROWS_PER_MIXER + newrow;
#pragma unroll
      for (int i = 0; i < 13; ++i) *reinterpret_cast<float4*>(w1 + 4 * i) = gload4(row1 + 4 * i);
      rs1 = *rsp1;
#pragma unroll
      for (int j = 0; j < CMX_MIX1; ++j)
        if (j >= k) w1[CMX_ROW1_EXTRA + j] = 0.0f;  // only j < k are extra weights of mixer k
    }
    if (t + 1 < nbits && lds_poll(&L.ctl->scout_epoch) >= t + 2) {   // the layer-1 rows of the NEXT bit towards the caches
      const uint32_t nr = L.rec[(t + 1) % L.rr].rowidx[CMX_MIX0 + kk];
      if (is1 && nr != cur_row) {
        const gptr<const float> nrow = rows1 + ((size_t)kk * CMX_ROWS_PER_MIXER + nr) * CMX_ROW1_STRIDE;
        touch_line(nrow, L.pfdump + 256 + 256 + 64);
        touch_line(nrow + 32, L.pfdump + 256 + 256 + 64);
      }
    }
    TPROF(6);
    if (t >= 2 && !wait_ge<LATE>(L.ctl, &L.ctl->b_done, t - 1, false)) return;   // the hand-over slot t & 1 is free
    if (!wait_ge<LATE>(L.ctl, &L.ctl->tail_in, t + 1, false)) return;
    TPROF(7);
    jitter_stall<JIT>(L.jit, t, 12);
    const TailRec* tr = L.trec + (t & 1);
    int bit = tr->bit;   // (late mode: not known yet -- awaited below, where layer 1 learns)
    float* const in2 = L.h2 + 64 * (t & 1);   // this bit's layer-2 inputs: built here, read by tail_b_role
    if (k < CMX_MIX0) in2[k] = tr->out0[k];
    if (k < 3) in2[CMX_MIX0 + CMX_MIX1 + k] = tr->aux3[k];
    if (k == 49) in2[49] = __int_as_float(bit);
    if (k == 50) in2[50] = tr->lstm_p;
    if (k == 51) in2[51] = __int_as_float((int)row2);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    st_rel(&L.ctl->tail_done, t + 1);   // the gather wave's record of this bit and the scout's are read: their slots may be rewritten
    float pm = 0.0f;
#pragma unroll
    for (int i = 0; i < CMX_MIX0; ++i) pm = fadd(pm, fmul(in2[i], w1[i]));
#pragma unroll
    for (int i = 0; i < 3; ++i) pm = fadd(pm, fmul(in2[CMX_MIX0 + CMX_MIX1 + i], w1[CMX_MIX0 + i]));
    float e = 0.0f;
#pragma unroll
    for (int j = 0; j < CMX_MIX1; ++j) {
      const float mine = clamp_out(fadd(pm, e), smin, smax);
      const float oj = bcast_lane(mine, j);
      e = fadd(e, fmul(oj, w1[CMX_ROW1_EXTRA + j]));
    }
    const float p1_ = fadd(pm, e);
    const float myout = clamp_out(p1_, smin, smax);
    if (is1) in2[CMX_MIX0 + k] = myout;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    st_rel(&L.ctl->b_in, t + 1);
    TPROF(8);
    if (LATE) {   // the decoder's bit: the output wave (tail_b_role) receives it after p has gone out
      if (!wait_ge<LATE>(L.ctl, &L.ctl->bit_epoch, t + 1, false)) return;
      bit = L.bitring[t & 7];
    }
    // ---- Mixer::Perceive, layer 1 ----
    if (is1) {
      const float decay = (float)(d1 * (1.5 - ((1.0 * (double)rs1) / (double)mx1)));
      const float u = fmul(fmul(decay, lr1), fsub(cmx_logistic(p1_), (float)bit));
      ++rs1;
      if (rs1 > mx1) mx1 = rs1;
      const bool df = (rs1 & 1023) == 0;
      float4 xv[13], wv[13];
#pragma unroll
      for (int i = 0; i < 13; ++i) wv[i] = *reinterpret_cast<const float4*>(w1 + 4 * i);
#pragma unroll
      for (int i = 0; i < 6; ++i) xv[i] = *reinterpret_cast<const float4*>(in2 + 4 * i);          // out0[0..23]
      xv[6] = make_float4(in2[24], in2[25], in2[46], in2[47]);                                     // out0[24,25], aux[0,1]
      xv[7] = make_float4(in2[48], 0.0f, 0.0f, 0.0f);                                              // aux[2], pad
#pragma unroll
      for (int i = 0; i < 5; ++i) xv[8 + i] = make_float4(in2[CMX_MIX0 + 4 * i], in2[CMX_MIX0 + 4 * i + 1], in2[CMX_MIX0 + 4 * i + 2], in2[CMX_MIX0 + 4 * i + 3]);
      const float c = df ? cdec : 1.0f;  // w * 1.0f == w exactly
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        float4 v = f4sub_mul(wv[i], u, xv[i]);
        if (i >= 8) {  // extra weights: only j < k belong to mixer k, the rest stays exactly 0
          const int j0 = 4 * (i - 8);
          if (j0 + 0 >= k) v.x = 0.0f;
          if (j0 + 1 >= k) v.y = 0.0f;
          if (j0 + 2 >= k) v.z = 0.0f;
          if (j0 + 3 >= k) v.w = 0.0f;
        }
        wv[i] = f4scale(v, c);
      }
#pragma unroll
      for (int i = 0; i < 13; ++i) *reinterpret_cast<float4*>(w1 + 4 * i) = wv[i];
      if (mix_out) as_global(mix_out)[(size_t)t * CMX_MIXERS + CMX_MIX0 + k] = p1_;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    TPROF(10);
  }
  if (prof_on && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) S->prof[6 + i] += pacc[i];
  }
#undef TPROF
  if (nbits > 0 && is1 && cur_row != 0xffffffffu) {
    store_row1();
    S->max_steps[CMX_MIX0 + k] = mx1;
  }
}

template <bool LATE, bool JIT = false> __device__ void tail_b_role(MixState* S, const Lds& L, const float* decay1, int nbits, float* p_out, float* mix_out, int lane, bool prof_on) {
  uint64_t tprev = __builtin_readcyclecounter();
  uint64_t pacc[3] = {0, 0, 0};
#define TPROF(k) do { if (prof_on) { uint64_t now_ = __builtin_readcyclecounter(); pacc[k] += now_ - tprev; tprev = now_; } } while (0)
  const int k = lane;
  const float cdec = 1.0f - 3.0e-6f;
  const float lr2 = S->lr[CMX_MIXERS - 1];
  const uint16_t* const t_st = S->t_st;
  const uint16_t* const t_sq = S->t_sq;
  uint16_t* const s6 = S->s6;
  uint16_t* const s7 = S->s7;
  int* const x1 = S->x1;
  int* const x2 = S->x2;
  const gptr<uint64_t> row_steps = as_global(S->row_steps);
  float* const w2 = L.w2;  // layer-2 row (predictor.cpp:354-356: a single weight set), LDS-resident
  const gptr<float> rows2 = as_global(S->rows2);
  const gptr<uint64_t> rsteps2 = row_steps + (size_t)(CMX_MIXERS - 1) * CMX_ROWS_PER_MIXER;
  uint32_t cur_row2 = 0xffffffffu;
  uint64_t rs2 = 0, mx2 = S->max_steps[CMX_MIXERS - 1];
  unsigned sj = S->sse_j, spc = S->sse_pc, sffl = S->sse_ffl;
  uint64_t steps_done = 0;
  for (int t = 0; t < nbits; ++t) {
    jitter_stall<JIT>(L.jit, t, 13);
    const double d1 = (double)as_global(decay1)[t];
    // SSE contexts for a = 0..2 / b = 0..3 (sse.cpp:248-262): lane i pulls candidate cell i towards the caches (see tail_role)
    if (k < 13) {
      const unsigned j_ = bcast_u(sj), pc_ = bcast_u(spc), ffl_ = bcast_u(sffl);
      const int a = k % 3, b = k - 9;
      const float* addr;
      if (k < 3) addr = (const float*)(s6 + (size_t)(((((a << 7) + (int)(ffl_ & 127)) << 8) + (int)(pc_ & 255)) * 256 + (int)j_) * 8);
      else if (k < 6) addr = (const float*)(s7 + (size_t)(((((a << 5) + (int)(ffl_ & 31)) << 8) + (int)(pc_ & 255)) * 255 + (j_ ? (int)j_ - 1 : 0)) * 8);
      else if (k < 9) addr = (const float*)(x2 + (((((a << 1) + (int)(ffl_ & 1)) << 8) + (int)(pc_ & 255)) * 256 + (int)j_));
      else addr = (const float*)(x1 + (((((b << 8) + (int)(ffl_ & 255)) << 3) + (int)((pc_ >> 5) & 7)) * 79 + sse_mx1mask((int)j_)));
      touch_line(as_global(addr), L.pfdump + 256 + 256);
    }
    if (!wait_ge<LATE>(L.ctl, &L.ctl->b_in, t + 1, false)) return;
    TPROF(0);
    if (LATE && lane == 0) late_stamp(L.late, 4);   // layer 1 done
    const float* const in2 = L.h2 + 64 * (t & 1);
    int bit = __float_as_int(in2[49]);
    {  // layer 2 has one weight set in cmix (selector = zero_context_); a changing key is still honoured
      const uint32_t newrow2 = (uint32_t)__float_as_int(in2[51]);
      if (newrow2 != cur_row2) {
        if (cur_row2 != 0xffffffffu) {
          if (k < 16) gstore4(rows2 + (size_t)cur_row2 * CMX_ROW2_STRIDE + 4 * k, *reinterpret_cast<const float4*>(w2 + 4 * k));
          if (k == 0) rsteps2[cur_row2] = rs2;
        }
        cur_row2 = newrow2;
        if (k < 16) *reinterpret_cast<float4*>(w2 + 4 * k) = gload4(rows2 + (size_t)newrow2 * CMX_ROW2_STRIDE + 4 * k);
        rs2 = rsteps2[newrow2];
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
      }
    }
    float u2 = 0.0f;
    int df2 = 0;
    if (k == 0) {
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < 12; ++i) {
        const float4 xv = *reinterpret_cast<const float4*>(in2 + 4 * i);
        const float4 wv = *reinterpret_cast<const float4*>(w2 + 4 * i);
        acc = fadd(acc, fmul(xv.x, wv.x)); acc = fadd(acc, fmul(xv.y, wv.y));
        acc = fadd(acc, fmul(xv.z, wv.z)); acc = fadd(acc, fmul(xv.w, wv.w));
      }
      acc = fadd(acc, fmul(in2[48], w2[48]));
      const float p2_ = acc;
      const float sq = cmx_logistic(p2_);                  // predictor.cpp:413
      // ---- SSE::Predict (sse.cpp:243-290,320-324) on the pre-fetched cells ----
      const int p = (int)(1 + (1 - sq) * 32766);
      const unsigned prq = (unsigned)p >> 11;
      const int a = (prq > 0) + (prq > 14);
      const int b = (prq > 0) + (prq > 7) + (prq > 14);
      const int i6 = ((((a << 7) + (int)(sffl & 127)) << 8) + (int)(spc & 255)) * 256 + (int)sj;
      const int i7 = ((((a << 5) + (int)(sffl & 31)) << 8) + (int)(spc & 255)) * 255 + (sj ? (int)sj - 1 : 0);
      const int im2 = ((((a << 1) + (int)(sffl & 1)) << 8) + (int)(spc & 255)) * 256 + (int)sj;
      const int im1 = ((((b << 8) + (int)(sffl & 255)) << 3) + (int)((spc >> 5) & 7)) * 79 + sse_mx1mask((int)sj);
      U16x8 c6 = load_cell(s6 + (size_t)i6 * 8);
      U16x8 c7 = load_cell(s7 + (size_t)i7 * 8);
      const int w1x = as_global(x1)[im1];
      const int w2x = as_global(x2)[im2];
      SseInterp e6, e7;
      auto ST = [&](int i) -> int { return L.lst ? (int)L.lst[i] : (int)as_global(t_st)[i]; };
      auto SQ = [&](int i) -> int { return L.lsq ? (int)L.lsq[i] : (int)as_global(t_sq)[i]; };
      const int stp = ST(p);
      const int q6 = SQ(sse_extrap(stp, 10240)), q7 = SQ(sse_extrap(stp, 8200));
      const int pp1 = e6.pred(c6, q6);
      const int pp2 = e7.pred(c7, q7);
      const int s0 = sse_extrap(stp, 7935);
      const int s1 = sse_extrap(ST(pp1), 9592);
      const int s4 = sse_extrap(ST(pp2), 7677);
      const int s2 = sse_extrap(sse_mixup(w1x, s0, s1), 8092);
      const int mix1_p = SQ(s2);
      const int s5 = sse_extrap(sse_mixup(w2x, s2, s4), 8202);
      const int mix2_p = SQ(s5);
      float pf = (float)(1 - ((mix2_p - 1) / 32766.0));
      const float lp = in2[50];
      if (lp == 0.0f || lp == 1.0f) pf = lp;               // predictor.cpp:383,415-417
      as_global(p_out)[t] = pf;
      if (LATE) {
        // Decoder::Decode (decoder.cpp:20-39): p goes to the host (value | tag, one 8-byte store into its mapped memory); the arithmetic
        // decoder turns it into the bit, which comes back through the box and is handed to the waves that learn from it
        __hip_atomic_store(&L.late.box->p_word[t % CMX_LATE_P_RING], ((unsigned long long)(unsigned)(t + 1) << 32) | (unsigned)__float_as_int(pf), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
        late_stamp(L.late, 1);   // p has gone out
        bit = late_y(L.late, t + 1);
        if (bit < 0) { lds_publish_store(&L.ctl->abort, 1); bit = 0; }
        L.bitring[t & 7] = bit;
        st_rel(&L.ctl->bit_epoch, t + 1);
      }
      // ---- SSE::Perceive (sse.cpp:291-306,326-328) ----
      e6.update(c6, bit, 106);
      e7.update(c7, bit, 127);
      store_cell_async(s6 + (size_t)i6 * 8, c6);
      store_cell_async(s7 + (size_t)i7 * 8, c7);
      as_global(x1)[im1] = w1x + sse_wdelta(bit, s0, s1, 6202, mix1_p);
      as_global(x2)[im2] = w2x + sse_wdelta(bit, s2, s4, 8320, mix2_p);
      sj += sj + (unsigned)bit;
      if (sj >= 256) {
        sffl = (sffl * 2 + (spc >= 0x40)) & 255;
        spc = sj & 255;
        sj = 1;
      }
      // ---- Mixer::Perceive, layer 2 (mixer.cpp:56-72) ----
      const float decay = (float)(d1 * (1.5 - ((1.0 * (double)rs2) / (double)mx2)));
      u2 = fmul(fmul(decay, lr2), fsub(sq, (float)bit));
      ++rs2;
      if (rs2 > mx2) mx2 = rs2;
      df2 = (rs2 & 1023) == 0;
      if (mix_out) as_global(mix_out)[(size_t)t * CMX_MIXERS + CMX_MIXERS - 1] = p2_;
      ++steps_done;
    }
    if (LATE && lds_poll(&L.ctl->abort)) return;   // (uniform: the decoder has left)
    TPROF(1);
    u2 = bcast_lane(u2, 0);
    df2 = __builtin_amdgcn_readlane(df2, 0);
    if (k < CMX_IN2) {  // layer-2 weights: one lane per weight
      float v = fsub(w2[k], fmul(u2, in2[k]));
      if (df2) v = fmul(v, cdec);
      w2[k] = v;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    st_rel(&L.ctl->b_done, t + 1);
    TPROF(2);
  }
  if (prof_on && lane == 0) {
#pragma unroll
    for (int i = 0; i < 3; ++i) S->prof[13 + i] += pacc[i];
  }
#undef TPROF
  if (nbits > 0) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (k < 16 && cur_row2 != 0xffffffffu)
      gstore4(rows2 + (size_t)cur_row2 * CMX_ROW2_STRIDE + 4 * k, *reinterpret_cast<const float4*>(w2 + 4 * k));
    if (k == 0) {
      if (cur_row2 != 0xffffffffu) rsteps2[cur_row2] = rs2;
      S->max_steps[CMX_MIXERS - 1] = mx2;
      S->sse_j = sj; S->sse_pc = spc; S->sse_ffl = sffl;
      S->steps = S->steps + steps_done;
    }
  }
}


// ==================================================================================================================================
// cmx_mixnet_spec_kernel: the layer-0 dot products on 26 helper workgroups, each cutting its 2078-term ordered chain into four
// segments that run AT THE SAME TIME -- segment 0 exactly, segments 1..3 speculatively from 64 candidate start values.
//
// s -> RN(s + x) is the only operation of the chain, so a segment is a function of its start value alone. The start of segment k
// is the f32 running sum after the preceding terms; an estimate of it (the f64 sum of those products, rounded) is available as soon
// as the products are, and the true value lies within a few dozen ulps of it (profiles/r03_spec_chain_study.txt: within +-32 ulp
// for 96-97 % of the segments on the bench text). Wave k of a helper therefore runs segment k from the 64 consecutive floats around
// the estimate, one per lane; when wave k-1 delivers the true start, the lane whose candidate has the same bit pattern holds the
// exact result (same operations on the same operands), otherwise the wave re-runs the segment from the true start (the serial
// path). Either way the value handed on is exactly the reference's (mixer.cpp:40-43), and the chain's latency drops from 2078
// dependent adds to ~520 plus the re-runs.
//
//   block 0 (main)        wave 0 gather: waits for the 26 sums, intra-layer extra-input chain, Mixer::Perceive scalars, publishes u
//                         wave 1 tail, wave 2 scout: as in cmx_mixnet_chunk_kernel (the scout also feeds the global ring)
//   block 1 + m (helper)  owns the selected row of layer-0 mixer m in registers (wave w: elements 512 w .. 512 w + 511 / 541):
//                         u of bit t-1 arrives -> w -= u x (mixer.cpp:66-71), row swap if the selector changed (the incoming row is
//                         already in registers: it was requested a bit ahead) -> products -> LDS -> four segment chains -> resolve
//                         -> the sum of bit t goes back to the gather wave.
// Two global hand-offs per bit (sum, u), 8-byte value|tag words, agent scope.
struct HelperLds {
  float prod[4][SEG];          // rounded products of the four segments (each read back as broadcast float4s by its own wave)
  double segsum[4];            // f64 sum of a segment's products
  float res[4];                // exact running sum after segment w
  int sum_epoch[4];            // bit + 1 for which segsum[w] is valid
  int res_epoch[4];            // bit + 1 for which res[w] is valid
  int abort;
  int pad_;
};

__device__ __forceinline__ unsigned long long ld_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_u64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned ld_u32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ld_f32(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
constexpr unsigned SPEC_SPIN = 1u << 24;

// float <-> integer with the same order (consecutive floats = consecutive integers, across zero too)
__device__ __forceinline__ int f2ord(float f) { int b = __float_as_int(f); return b ^ ((b >> 31) & 0x7fffffff); }
__device__ __forceinline__ float ord2f(int o) { return __int_as_float(o ^ ((o >> 31) & 0x7fffffff)); }

// sum of a double over the wavefront, in every lane (any order will do: it only centres the candidates). Cross-lane moves as DPP modifiers
// (quad permutes, rotations inside the rows of 16) and four readlanes instead of six LDS-crossbar round trips of two dwords each.
template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum_f64(double v) {
  v += dpp_f64<0xB1>(v);    // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);    // quad_perm [2,3,0,1]: every lane holds its quad's sum
  v += dpp_f64<0x124>(v);   // row_ror:4
  v += dpp_f64<0x128>(v);   // row_ror:8: every lane holds its row's sum
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

template <bool LATE, bool JIT = false> __device__ void helper_role(MixState* S, SpecXfer* X, HelperLds* H, int nbits, int m, int w, int lane, bool tol, CmxLateBox* LB, bool local, int jit,
                                                                  const CmxLate& late, const float* probs) {
  const gptr<float> rows0 = as_global(S->rows0);
  const gptr<const float> lut = as_global(S->logit_lut);
  // four waves cut the 2078-term chain: 4 x 512 (+ 30) terms (other cuts, candidate counts and re-run forms were measured and are slower:
  // scripts/study/mixnet_variants.patch)
  constexpr int NW = 4, SEGN = 2048 / NW, KS = SEGN / 64, LASTW = NW - 1;
  const int base = SEGN * w;                // first element of this wave's segment
  const int nseg = w == LASTW ? CMX_IN0 - SEGN * LASTW : SEGN;
  const bool tailk = w == LASTW && lane < 30;   // slot KS: elements 2048 + lane (lane < 30)
  const float cdec = 1.0f - 3.0e-6f;
  float W[KS + 1], Wn[KS + 1], xc[KS + 1], xp[KS + 1];
#pragma unroll
  for (int k = 0; k < KS + 1; ++k) { W[k] = 0.0f; Wn[k] = 0.0f; xc[k] = 0.0f; xp[k] = 0.0f; }
  unsigned long long n_spec = 0, n_hit = 0, n_miss = 0;
  uint32_t cur_base = 0;
  bool f_chg = false; uint32_t f_base = 0;   // of the bit fetched last: its selector changes, and to which row (read off the serial path)
  auto failed = [&]() { return lds_poll(&H->abort) != 0; };
  unsigned long long late_t0 = 0;
  // has a wait for the main workgroup run out? a compressor's: by spin count; a decoder's (the wait then includes the host): by its box
  auto spun_out = [&](unsigned spins) { return LATE ? late_expired(LB, late_t0) : spins > SPEC_SPIN; };
  auto give_up = [&]() { lds_publish_store(&H->abort, 1); __hip_atomic_store(&X->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  // fetch the inputs of bit t (this wave's slice) and, if its selector changes, the incoming row
  auto fetch = [&](int t) -> bool {
    unsigned spins = 0;
    // a decoder's chunk: the row selection of this mixer is published ahead of the stretched inputs (sel_epoch; the auxiliary-context mixer's with scout_epoch)
    const unsigned* const ep = LATE && m != CMX_AUX ? &X->sel_epoch : &X->scout_epoch;
    while (ld_u32(ep) < (unsigned)(t + 1)) {
      __builtin_amdgcn_s_sleep(1);
      if ((++spins & 1023u) == 0 && (spun_out(spins) || failed() || ld_u32(&X->fail))) { give_up(); return false; }
    }
    late_t0 = 0;
    const int slot = t % CMX_SPEC_RING;
    f_chg = ld_u32(&X->changed[slot][m]) != 0;
    if (f_chg) {
      const uint32_t nb = ((uint32_t)m * CMX_ROWS_PER_MIXER + ld_u32(&X->rowidx[slot][m])) * CMX_ROW0_STRIDE + (uint32_t)base;
      f_base = nb;
#pragma unroll
      for (int k = 0; k < KS; ++k) Wn[k] = rows0[nb + 64 * k + lane];
      Wn[KS] = tailk ? rows0[nb + SEGN + lane] : 0.0f;
    }
    if (LATE) {
      // round 6: this wave's slice of the bit's inputs straight from the producers' row (uncached device memory), as soon as the stages that write its columns have
      // counted the row -- MixerInput::SetInput + Sigmoid::Logit (mixer-input.cpp:11-15, sigmoid.cpp:12-17) with the stretch wave's arithmetic; no ring, no second hop.
      // columns: 0 Bracket, 1..2 + 2025..2075 contexts, 3..433 fxcm, 434..2024 paq8, 2076 PPMd, 2077 LSTM
      const unsigned need = w == 0 ? (1u << LC_CTX) | (1u << LC_BM0) | (1u << LC_FX) | (1u << LC_P8) : w == LASTW ? (1u << LC_P8) | (1u << LC_CTX) | (1u << LC_BM1) | (1u << LC_BM2) : (1u << LC_P8);
      bool ok = true;
      if (lane <= LC_P8 && ((need >> lane) & 1u)) ok = late_wait_cnt(late, lane, (uint32_t)(t + 1));
      if (__ballot(!ok)) { give_up(); return false; }
      const float* row = probs + (size_t)t * CMX_IN0 + base;
      float pv[KS + 1];
#pragma unroll
      for (int k = 0; k < KS; ++k) pv[k] = __hip_atomic_load(row + 64 * k + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      pv[KS] = tailk ? __hip_atomic_load(row + SEGN + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.5f;
#pragma unroll
      for (int k = 0; k < KS + 1; ++k) {
        float p = pv[k];
        if (p < 1.0e-4f) p = 1.0e-4f;
        else if (p > 1 - 1.0e-4f) p = 1 - 1.0e-4f;
        int idx = (int)(p * 100001.0f);
        if (idx >= 100001) idx = 100000;
        else if (idx < 0) idx = 0;
        xc[k] = lut[idx];
      }
      if (!tailk) xc[KS] = 0.0f;
    } else {
      const float* gx = X->xs[slot] + base;
#pragma unroll
      for (int k = 0; k < KS; ++k) xc[k] = ld_f32(gx + 64 * k + lane);
      xc[KS] = tailk ? ld_f32(gx + SEGN + lane) : 0.0f;
    }
    return true;
  };
  // u of bit tu - 1 -> W -= u x (mixer.cpp:66-71; xp = the inputs of that bit), the 1024-step decay
  auto apply_u = [&](int tu) -> bool {
    unsigned long long v;
    unsigned spins = 0;
    while ((unsigned)((v = ld_u64(&X->u[m])) >> 33) != (unsigned)tu) {
      if ((++spins & 1023u) == 0 && (spun_out(spins) || failed() || ld_u32(&X->fail))) { give_up(); return false; }
    }
    late_t0 = 0;
    const float u = __int_as_float((int)(unsigned)v);
    const bool df = ((v >> 32) & 1ull) != 0;
#pragma unroll
    for (int k = 0; k < KS + 1; ++k) {
      W[k] = fsub(W[k], fmul(u, xp[k]));
      if (df) W[k] = fmul(W[k], cdec);
    }
    return true;
  };
  if (nbits > 0 && !fetch(0)) return;
  for (int t = 0; t <= nbits; ++t) {
    const bool live = t < nbits;             // t == nbits: apply the last update and store the row
    jitter_stall<JIT>(jit, t, 16 + 4 * m + w);
    // ---- u of bit t-1 (the serial hand-off of the bit) ----
    // (a decoder's chunk: taken at the END of iteration t - 1, below -- the bit is known long before the next row is complete, so the poll's round trip and the
    // update run under the wait for the row instead of behind it)
    if (t > 0 && !LATE) { if (!apply_u(t)) return; }
    const bool chg = !live || f_chg;
    if (chg) {                               // outgoing row to HBM (16 B per 4 lanes: 64 consecutive floats per k), incoming row is in Wn
      if (t > 0) {
#pragma unroll
        for (int k = 0; k < KS; ++k) rows0[cur_base + 64 * k + lane] = W[k];
        if (tailk) rows0[cur_base + SEGN + lane] = W[KS];
      }
      if (live) {
        cur_base = f_base;
#pragma unroll
        for (int k = 0; k < KS + 1; ++k) W[k] = Wn[k];
      }
    }
    if (!live) break;
    // ---- products of bit t (mixer.cpp:41: in[i] * w[i], rounded) -> LDS, f64 sum of the segment ----
    float* pr = H->prod[w];
    double ds = 0.0;
#pragma unroll
    for (int k = 0; k < KS; ++k) { const float p = fmul(xc[k], W[k]); pr[64 * k + lane] = p; ds += (double)p; }
    if (w == LASTW) { const float p = tailk ? fmul(xc[KS], W[KS]) : 0.0f; if (lane < 36) pr[SEGN + lane] = p; ds += (double)p; }   // nseg .. nseg + 5: zero pad read by chain_seg_n
#pragma unroll
    for (int k = 0; k < KS + 1; ++k) xp[k] = xc[k];
    if (tol) {
      // TOLERANCE MODE (opt-in, cmx_mixnet_set_tolerance; NOT bit-exact): the dot product as a tree sum -- every wave reduces its segment's products in
      // f64 across its lanes, the last wave adds the four segment sums and rounds once. No ordered chain, no speculation. The value differs from the
      // reference's sequentially rounded f32 sum in the last bits (it is the more accurate one); streams coded with it are not the reference's.
      ds = wave_sum_f64(ds);
      if (w < LASTW) {
        if (lane == 0) H->segsum[w] = ds;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_publish_store(&H->sum_epoch[w], t + 1);
      } else {
        double tot = 0.0;
        for (int q = 0; q < LASTW; ++q) {
          unsigned spins = 0;
          while (lds_poll(&H->sum_epoch[q]) < t + 1)
            if ((++spins & 1023u) == 0 && (spins > SPEC_SPIN || failed())) { give_up(); return; }
          tot += H->segsum[q];
        }
        tot += ds;
        if (lane == 0) st_u64(&X->sum[m], ((unsigned long long)(unsigned)(t + 1) << 32) | (unsigned)__float_as_int((float)tot));
      }
      if (t + 1 < nbits && !fetch(t + 1)) return;
      continue;
    }
    if (w < LASTW) {                         // the later waves centre their candidates on the sum of what precedes them
      ds = wave_sum_f64(ds);
      if (lane == 0) H->segsum[w] = ds;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) lds_publish_store(&H->sum_epoch[w], t + 1);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    // ---- the segment's chain ----
    float start = 0.0f;
    if (w > 0) {
      double est = 0.0;
      for (int q = 0; q < w; ++q) {
        unsigned spins = 0;
        while (lds_poll(&H->sum_epoch[q]) < t + 1)
          if ((++spins & 1023u) == 0 && (spins > SPEC_SPIN || failed())) { give_up(); return; }
        est += H->segsum[q];
      }
      start = ord2f(f2ord((float)est) + lane - 32);
    }
    float r = chain_seg_n<16>(pr, nseg, start);
    // ---- resolve against the true start ----
    if (w > 0) {
      unsigned spins = 0;
      while (lds_poll(&H->res_epoch[w - 1]) < t + 1)
        if ((++spins & 1023u) == 0 && (spins > SPEC_SPIN || failed())) { give_up(); return; }
      const float s = H->res[w - 1];
      const unsigned long long hit = __ballot(__float_as_int(start) == __float_as_int(s));
      ++n_spec;
      if (hit) {
        r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(r), (int)__builtin_amdgcn_readfirstlane(__ffsll((long long)hit) - 1)));
        ++n_hit;
      } else {
        r = chain_seg_n<16>(pr, nseg, s);          // outside the candidates: the serial path
        ++n_miss;
      }
    }
    if (w < LASTW) {
      if (lane == 0) H->res[w] = r;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (lane == 0) lds_publish_store(&H->res_epoch[w], t + 1);
    } else if (lane == 0) {
      const unsigned long long word = ((unsigned long long)(unsigned)(t + 1) << 32) | (unsigned)__float_as_int(r);
      if (local) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(&X->sum[m]), "v"(word) : "memory");   // stays in this XCD's L2 (every reader is on this XCD, checked at launch)
      else st_u64(&X->sum[m], word);
    }
    // ---- while the gather wave works: the inputs / incoming row of bit t+1 ----
    jitter_stall<JIT>(jit, t, 144 + 4 * m + w);
    if (LATE && !apply_u(t + 1)) return;     // (a decoder: u of THIS bit first -- it arrives with the bit, the next row much later)
    if (t + 1 < nbits && !fetch(t + 1)) return;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && w > 0) {
    atomicAdd(&X->stat[0], n_spec); atomicAdd(&X->stat[1], n_hit); atomicAdd(&X->stat[1 + w], n_miss);   // ([2..4]: re-runs of segment 1, 2, 3)
  }
}

// ------------------------------------------------------------------ scout, split (main workgroup of cmx_mixnet_spec_kernel)
// The one-wave scout needs ~18 k clocks per bit (three memory round trips for the 2078 inputs -- load, logit table, publish -- and
// two more for the row selection), more than a whole bit of the helpers. Only Mixer::GetContextData is stateful; the stretch of a
// bit's inputs depends on nothing, so four stretch waves take the bits round robin (each bit still costs its ~12 k clocks, four
// are in flight) and one select wave follows them in order.
template <bool LATE, bool JIT = false> __device__ void stretch_role(MixState* S, const Lds& L, SpecXfer* X, const float* probs, const uint8_t* bits, int nbits, int sw, int lane) {
  const gptr<const float> lut = as_global(S->logit_lut);
  const gptr<const float> gprobs = as_global(probs);
  const float smin = S->stretch_min, smax = S->stretch_max;
  for (int t = sw; t < nbits; t += 4) {
    jitter_stall<JIT>(L.jit, t, 4 + sw);
    if (t >= L.lead && !wait_ge<LATE>(L.ctl, &L.ctl->consumed, 4 * (t - L.lead) + 1, true)) return;   // the gather wave has begun bit t - lead
    if (t >= L.rr && !wait_ge<LATE>(L.ctl, &L.ctl->tail_done, t - L.rr + 1, true)) return;             // rec slot t % rr is free (see scout_role)
    BitRec* rec = L.rec + (t % L.rr);
    const gptr<const float> pr = gprobs + (size_t)t * CMX_IN0;
    if (LATE) {   // a decoder: row t exists once every producing stage has counted it (cmx_late.h), one lane per counter
      bool ok = true;
      if (lane <= LC_P8) ok = late_wait_cnt(L.late, lane, (uint32_t)(t + 1));
      if (__ballot(!ok)) { lds_publish_store(&L.ctl->abort, 1); return; }
      if (lane == 0) late_stamp(L.late, 2);   // row t complete
    }
    float pv[33];
    if (LATE) {
      // `probs` is a const __restrict__ kernel argument: the compiler may treat its contents as invariant for the whole launch (merge a
      // load with an earlier one, move it above the wait). A decoder's rows are written WHILE this kernel runs: atomic loads, which
      // it has to perform where they stand.
#pragma unroll
      for (int r = 0; r < 33; ++r) {
        int i = r * 64 + lane;
        pv[r] = i < CMX_IN0 ? __hip_atomic_load(probs + (size_t)t * CMX_IN0 + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.5f;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 33; ++r) {
        int i = r * 64 + lane;
        pv[r] = i < CMX_IN0 ? pr[i] : 0.5f;
      }
    }
    const int bitv = LATE ? 0 : (int)bits[t];   // (late: not known yet; the waves that learn wait for it)
    const float lstm_raw = bcast_lane(pv[32], 29);   // probs[t][2077]
#pragma unroll
    for (int r = 0; r < 33; ++r) {   // MixerInput::SetInput (mixer-input.cpp:11-15) + Sigmoid::Logit (sigmoid.cpp:12-17)
      float p = pv[r];
      if (p < 1.0e-4f) p = 1.0e-4f;
      else if (p > 1 - 1.0e-4f) p = 1 - 1.0e-4f;
      int idx = (int)(p * 100001.0f);
      if (idx >= 100001) idx = 100000;
      else if (idx < 0) idx = 0;
      pv[r] = lut[idx];
    }
    if (!LATE) {   // (a decoder's helpers stretch their own slice of the row straight from the producers' memory, helper_role's fetch: one hop less on every bit's path)
      float* gx = X->xs[t % CMX_SPEC_RING];
#pragma unroll
      for (int r = 0; r < 33; ++r) {
        int i = r * 64 + lane;
        if (i < CMX_IN0) __hip_atomic_store(gx + i, pv[r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const float ax0 = bcast_lane(pv[6], 49), ax1 = bcast_lane(pv[31], 40), ax2 = bcast_lane(pv[32], 29);   // columns 433, 2024, 2077
    if (lane < 3) {
      float v = lane == 0 ? ax0 : lane == 1 ? ax1 : ax2;
      if (v > smax) v = smax;
      else if (v < smin) v = smin;
      rec->aux3[lane] = v;
    } else if (lane == 3) {  // predictor.cpp:388-393
      float avg = 0;
      avg = fadd(avg, cmx_logistic(ax0));
      avg = fadd(avg, cmx_logistic(ax1));
      avg = fadd(avg, cmx_logistic(ax2));
      avg = avg / 3.0f;
      rec->auxkey = (uint32_t)(unsigned long long)(avg * 15);
      rec->lstm_p = lstm_raw;
      rec->bit = bitv;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the ring slot and the record are complete
    __builtin_amdgcn_wave_barrier();
    jitter_stall<JIT>(L.jit, t, 8 + sw);
    if (lane == 0) st_rel(&L.sdone[t % L.rr], t + 1);
  }
}

template <bool LATE, bool JIT = false> __device__ void select_role(MixState* S, const Lds& L, SpecXfer* X, const uint32_t* sel, const float* probs, int nbits, int lane) {
  const gptr<const uint32_t> gsel = as_global(sel);
  const gptr<const float> lut = as_global(S->logit_lut);
  for (int t = 0; t < nbits; ++t) {
    jitter_stall<JIT>(L.jit, t, 2);
    uint32_t key = (!LATE && lane < CMX_MIXERS) ? gsel[(size_t)t * CMX_MIXERS + lane] : 0;
    BitRec* rec = L.rec + (t % L.rr);
    const BitRec* prev = L.rec + ((t + L.rr - 1) % L.rr);
    if (LATE) {
      // A decoder: the context stage counts row t (LC_CTX) ~13 us before the slowest producer does, and 46 of the 47 selections need
      // nothing else -- they are done while the stretch wave still waits for the row; only the auxiliary-context mixer (its key comes out
      // of three of the inputs) follows the stretch wave. `sel` is a const __restrict__ kernel argument (see stretch_role): atomic loads.
      if (t >= L.rr && !wait_ge<LATE>(L.ctl, &L.ctl->tail_done, t - L.rr + 1, true)) return;   // rec slot t % rr is free (as the stretch wave checks)
      bool ok = true;
      if (lane == 0) ok = late_wait_cnt(L.late, LC_CTX, (uint32_t)(t + 1));
      if (__ballot(!ok)) { lds_publish_store(&L.ctl->abort, 1); return; }
      if (lane < CMX_MIXERS && lane != CMX_AUX) {
        key = __hip_atomic_load(sel + (size_t)t * CMX_MIXERS + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t r = select_row(S, lane, key);
        rec->rowidx[lane] = r;
        const uint32_t chg = (t == 0) || (r != prev->rowidx[lane]);
        if (lane < CMX_MIX0) {
          rec->changed[lane] = chg;
          __hip_atomic_store(&X->rowidx[t % CMX_SPEC_RING][lane], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          __hip_atomic_store(&X->changed[t % CMX_SPEC_RING][lane], chg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
      // round 6: the 46 selections are out -- their helpers fetch the incoming rows and, as the producers count the row, their slices of it, without waiting
      // for the stretch wave; the auxiliary-context mixer's key (predictor.cpp:388-393: the mean of three of the bit's squashed stretched inputs) is formed
      // HERE from those three columns as soon as their producers have counted the row, with the stretch wave's own arithmetic
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_store(&X->sel_epoch, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      bool ok3 = true;
      if (lane < 3) ok3 = late_wait_cnt(L.late, lane == 0 ? LC_FX : lane == 1 ? LC_P8 : LC_BM2, (uint32_t)(t + 1));
      if (__ballot(!ok3)) { lds_publish_store(&L.ctl->abort, 1); return; }
      float ax = 0.0f;
      if (lane < 3) {
        float pa = __hip_atomic_load(probs + (size_t)t * CMX_IN0 + (lane == 0 ? 433 : lane == 1 ? 2024 : 2077), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (pa < 1.0e-4f) pa = 1.0e-4f;
        else if (pa > 1 - 1.0e-4f) pa = 1 - 1.0e-4f;
        int idx = (int)(pa * 100001.0f);
        if (idx >= 100001) idx = 100000;
        else if (idx < 0) idx = 0;
        ax = cmx_logistic(lut[idx]);
      }
      const float a0 = bcast_lane(ax, 0), a1 = bcast_lane(ax, 1), a2 = bcast_lane(ax, 2);
      if (lane == CMX_AUX) {
        float avg = 0;
        avg = fadd(avg, a0);
        avg = fadd(avg, a1);
        avg = fadd(avg, a2);
        avg = avg / 3.0f;
        key = (uint32_t)(unsigned long long)(avg * 15);
        const uint32_t r = select_row(S, lane, key);
        rec->rowidx[lane] = r;
        const uint32_t chg = (t == 0) || (r != prev->rowidx[lane]);
        rec->changed[lane] = chg;
        __hip_atomic_store(&X->rowidx[t % CMX_SPEC_RING][lane], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&X->changed[t % CMX_SPEC_RING][lane], chg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      if (lane == 0) __hip_atomic_store(&X->scout_epoch, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!wait_ge<LATE>(L.ctl, &L.sdone[t % L.rr], t + 1, true)) return;   // (the stretch wave's record of the bit: what the gather and tail waves read)
    if (lane == CMX_AUX) key = rec->auxkey;
    if (lane < CMX_MIXERS && !LATE) {
      uint32_t r = select_row(S, lane, key);
      rec->rowidx[lane] = r;
      const uint32_t chg = (t == 0) || (r != prev->rowidx[lane]);
      if (lane < CMX_MIX0) {
        rec->changed[lane] = chg;
        __hip_atomic_store(&X->rowidx[t % CMX_SPEC_RING][lane], r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&X->changed[t % CMX_SPEC_RING][lane], chg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    jitter_stall<JIT>(L.jit, t, 3);
    if (!LATE && lane == 0) __hip_atomic_store(&X->scout_epoch, (unsigned)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (a decoder: published above, ahead of the stretch wave)
    st_rel(&L.ctl->scout_epoch, t + 1);
    if (LATE && lane == 0) late_stamp(L.late, 5);   // inputs and rows published to the helpers
  }
}

// ------------------------------------------------------------------ gather (main workgroup, wave 0)
// chain_role with the 26 ordered sums arriving from the helpers instead of being added up here.
template <bool LATE, bool JIT = false> __device__ void gather_role(MixState* S, const Lds& L, SpecXfer* X, const float* decay1, int nbits,
                            float* mix_out, bool prof_on, int lane, bool local) {
  const int m = lane;
  const bool is0 = m < CMX_MIX0;
  const float smin = S->stretch_min, smax = S->stretch_max;
  const float cdec = 1.0f - 3.0e-6f;
  const float lr = is0 ? S->lr[m] : 0.0f;
  uint64_t tprev = __builtin_readcyclecounter();
  uint64_t pacc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) pacc[i] = 0;
#define GPROF(k)                                                       \
  do {                                                                 \
    if (prof_on) {                                                     \
      uint64_t now_ = __builtin_readcyclecounter();                    \
      pacc[k] += now_ - tprev;                                         \
      tprev = now_;                                                    \
    }                                                                  \
  } while (0)
  __builtin_amdgcn_s_setprio(3);
  const int mm = is0 ? m : 0;
  float ew[28];
#pragma unroll
  for (int i = 0; i < 28; ++i) ew[i] = 0.0f;
  uint64_t rsteps = 0;
  uint64_t mx = S->max_steps[mm];
  gptr<float> row0 = as_global(S->rows0);
  gptr<uint64_t> rsp = as_global(S->row_steps);
  auto store_row_state = [&]() {
#pragma unroll
    for (int i = 0; i < 7; ++i)
      if (4 * i < m) gstore4_async(row0 + CMX_ROW0_EXTRA + 4 * i, make_float4(ew[4 * i], ew[4 * i + 1], ew[4 * i + 2], ew[4 * i + 3]));
    asm volatile("global_store_dwordx2 %0, %1, off\n\ts_nop 0" :: "v"(rsp), "v"(rsteps) : "memory");
  };
  // row state of the NEXT bit, requested at the end of a bit (the scout is ahead) so that the loads run under the helpers' work:
  // a row that changes comes from a different row than the current one, whose last stores this wave issued earlier (in order)
  float4 ewn[7];
  uint64_t rsn = 0;
  int pf_t = -1, d1_t = -1;
  float d1n = 0.0f;
#pragma unroll
  for (int i = 0; i < 7; ++i) ewn[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  for (int t = 0; t < nbits; ++t) {
    jitter_stall<JIT>(L.jit, t, 0);
    if (!wait_ge<LATE>(L.ctl, &L.ctl->scout_epoch, t + 1, false)) return;
    st_rel(&L.ctl->consumed, 4 * t + 1);      // the scout may go on to bit t + 1 (it waits for consumed >= 4 t)
    GPROF(0);
    const BitRec* rec = L.rec + (t % L.rr);
    const int bit = rec->bit;
    if (is0 && rec->changed[mm]) {
      if (t > 0) store_row_state();
      row0 = as_global(S->rows0) + ((size_t)mm * CMX_ROWS_PER_MIXER + rec->rowidx[mm]) * CMX_ROW0_STRIDE;
      rsp = as_global(S->row_steps) + (size_t)mm * CMX_ROWS_PER_MIXER + rec->rowidx[mm];
      if (pf_t == t) {          // requested while the helpers worked on the previous bit
#pragma unroll
        for (int i = 0; i < 7; ++i) { ew[4 * i] = ewn[i].x; ew[4 * i + 1] = ewn[i].y; ew[4 * i + 2] = ewn[i].z; ew[4 * i + 3] = ewn[i].w; }
        rsteps = rsn;
      } else {
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          float4 v = gload4(row0 + CMX_ROW0_EXTRA + 4 * i);
          ew[4 * i] = v.x; ew[4 * i + 1] = v.y; ew[4 * i + 2] = v.z; ew[4 * i + 3] = v.w;
        }
        rsteps = *rsp;
      }
#pragma unroll
      for (int i = 0; i < 28; ++i)
        if (i >= m) ew[i] = 0.0f;
    }
    const double d1 = (double)(d1_t == t ? d1n : as_global(decay1)[t]);  // (float)(0.9/pow(1e-7*steps_+0.8,0.8)), host libm; fetched a bit ahead
    const float decay = (float)(d1 * (1.5 - ((1.0 * (double)rsteps) / (double)mx)));   // mixer.cpp:58-60
    const float dlr = fmul(decay, lr);
    GPROF(1);
    // ---- the 26 ordered sums of bit t ----
    float pm = 0.0f;
    {
      bool have = !is0;
      unsigned spins = 0;
      unsigned long long gt0 = 0;
      while (true) {
        if (!have) {
          const unsigned long long v = ld_u64(&X->sum[mm]);
          if ((unsigned)(v >> 32) == (unsigned)(t + 1)) { pm = __int_as_float((int)(unsigned)v); have = true; }
        }
        if (__ballot(!have) == 0) break;
        if ((++spins & 1023u) == 0 && ((LATE ? late_expired(L.late.box, gt0) : spins > SPEC_SPIN) || lds_poll(&L.ctl->abort) || ld_u32(&X->fail))) {
          lds_publish_store(&L.ctl->abort, 1);
          __hip_atomic_store(&X->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          return;
        }
      }
    }
    GPROF(2);
    if (LATE && lane == 0) late_stamp(L.late, 3);   // the 26 sums are there
    float e = 0.0f;
#pragma unroll
    for (int j = 0; j < CMX_MIX0; ++j) {      // intra-layer chain (predictor.cpp:395-400), see chain_role
      const float mine = clamp_med3(fadd(pm, e), smin, smax);
      const float oj = bcast_lane(mine, j);
      e = fadd(e, fmul(oj, ew[j]));
    }
    const float p_ = fadd(pm, e);
    const float myout = clamp_out(p_, smin, smax);
    GPROF(3);
    int bitv = bit;
    auto hand_to_tail = [&]() -> bool {
      if (t >= 2 && !wait_ge<LATE>(L.ctl, &L.ctl->tail_done, t - 1, false)) return false;
      TailRec* tr = L.trec + (t & 1);
      if (is0) tr->out0[m] = myout;
      if (m >= CMX_MIX0 && m < CMX_MIXERS) tr->rowidx[m - CMX_MIX0] = rec->rowidx[m];
      if (m < 3) tr->aux3[m] = rec->aux3[m];
      if (m == 0) { tr->lstm_p = rec->lstm_p; tr->bit = bit; }
      st_rel(&L.ctl->tail_in, t + 1);
      return true;
    };
    if (LATE) {   // a decoder: the bit is an output of the arithmetic decoder, which needs p -- the tail waves first, then the wait
      if (!hand_to_tail()) return;
      if (!wait_ge<LATE>(L.ctl, &L.ctl->bit_epoch, t + 1, false)) return;
      bitv = L.bitring[t & 7];
    }
    float uu = fmul(dlr, fsub(cmx_logistic_t(p_, L.exptab), (float)bitv));   // Mixer::Perceive scalar (mixer.cpp:56-64)
    ++rsteps;
    if (rsteps > mx) mx = rsteps;
    const bool dfl = (rsteps & 1023) == 0;
    if (is0) {
      const unsigned long long uw = ((unsigned long long)(2u * (unsigned)(t + 1) + (dfl ? 1u : 0u)) << 32) | (unsigned)__float_as_int(uu);
      if (local) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(&X->u[m]), "v"(uw) : "memory");   // stays in this XCD's L2: every helper is on this XCD (checked at launch)
      else st_u64(&X->u[m], uw);
    }
    GPROF(4);
    if (!LATE && !hand_to_tail()) return;
    GPROF(12);
    if (is0 && mix_out) as_global(mix_out)[(size_t)t * CMX_MIXERS + m] = p_;
#pragma unroll
    for (int j = 0; j < CMX_MIX0; ++j) {      // extra weights: ew[j] -= u * out_j (mixer.cpp:67,70)
      float oj = bcast_lane(myout, j);
      if (j < m) {
        float v = fsub(ew[j], fmul(uu, oj));
        if (dfl) v = fmul(v, cdec);
        ew[j] = v;
      }
    }
    if (t + 1 < nbits) { d1n = as_global(decay1)[t + 1]; d1_t = t + 1; }
    if (t + 1 < nbits && lds_poll(&L.ctl->scout_epoch) >= t + 2) {
      const BitRec* nx = L.rec + ((t + 1) % L.rr);
      if (is0 && nx->changed[mm]) {
        const gptr<float> r1 = as_global(S->rows0) + ((size_t)mm * CMX_ROWS_PER_MIXER + nx->rowidx[mm]) * CMX_ROW0_STRIDE;
#pragma unroll
        for (int i = 0; i < 7; ++i) ewn[i] = gload4(r1 + CMX_ROW0_EXTRA + 4 * i);
        rsn = (as_global(S->row_steps) + (size_t)mm * CMX_ROWS_PER_MIXER + nx->rowidx[mm])[0];
      }
      pf_t = t + 1;
    }
    GPROF(5);
  }
  if (is0 && nbits > 0) {
    store_row_state();
    S->max_steps[m] = mx;
  }
  if (prof_on && lane == 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i)
      if (i < 6 || i >= 12) S->prof[i] += pacc[i];
  }
#undef GPROF
}

}  // namespace

extern "C" __global__ __launch_bounds__(NTHREADS) void cmx_mixnet_chunk_kernel(
    MixState* __restrict__ S, const float* __restrict__ probs, const uint32_t* __restrict__ sel,
    const uint8_t* __restrict__ bits, const float* __restrict__ decay1, int nbits,
    float* __restrict__ p_out, float* __restrict__ mix_out, int mode) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  {  // optional XCD placement: mode bits 8..11 = 1 + the block that does the work (the others only occupy a slot)
    const int want = (mode >> 8) & 15;
    if (want && (int)blockIdx.x != want - 1) return;
    mode &= 0xff;
  }
  Lds L;
  L.prod = smem;                                                  // 2 * PBUF
  L.xs = L.prod + 2 * PBUF;                                       // 3 * XS
  L.rec = reinterpret_cast<BitRec*>(L.xs + 3 * XS);               // 3
  L.rr = 3; L.lead = 2; L.lst = nullptr; L.lsq = nullptr; L.sdone = nullptr; L.h2 = nullptr; L.bitring = nullptr; L.late = CmxLate(); L.jit = 0;
  L.trec = reinterpret_cast<TailRec*>(L.rec + 3);                 // 2
  L.upd = reinterpret_cast<float*>(L.trec + 2);                   // 32
  L.dflag = reinterpret_cast<uint32_t*>(L.upd + 32);              // 32
  L.in2 = reinterpret_cast<float*>(L.dflag + 32);                 // 64
  L.ctl = reinterpret_cast<Ctl*>(L.in2 + 64);
  L.pfdump = (unsigned)(size_t)(lds_int*)(reinterpret_cast<int*>(L.ctl) + 16);   // 256 B behind the control block
  L.w2 = reinterpret_cast<float*>(L.ctl) + 16 + 64;                               // 256 B behind the dump area
  L.w1 = L.w2 + 64 + 16 + 64;                                                     // behind w2 and the tail's two dump slots
  L.exptab = reinterpret_cast<uint64_t*>(L.w1 + 20 * 68);                         // 256 B
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * XS; i += NTHREADS) L.xs[i] = 0.0f;    // incl. the zero padding 2078..2111
  for (int i = tid; i < 2 * PBUF; i += NTHREADS) L.prod[i] = 0.0f;
  if (tid < 32) { L.upd[tid] = 0.0f; L.dflag[tid] = 0; L.exptab[tid] = cmx_exp2f_tab[tid]; }
  if (tid < (int)(sizeof(Ctl) / 4)) reinterpret_cast<int*>(L.ctl)[tid] = 0;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // Waves w, w+4, w+8 share a SIMD. The chain wave's SIMD-mates are two producers: they are busy mainly in the u -> segment-0
  // window, when the chain wave waits anyway, and idle (sleeping polls) for most of the chain; the scout and the tail do
  // 10-14 k clocks of VALU / memory work per bit spread over the whole bit and would compete with the dependent add chain
  // for issue slots (ubench: 6.7 clk/add alone, 7.5 with busy SIMD-mates).
  if (wave == 0) chain_role(S, L, decay1, nbits, mix_out, (mode & 4) != 0, lane, mode >> 4);
  else if (wave == 1) tail_role(S, L, decay1, nbits, p_out, mix_out, lane, (mode & 4) != 0 && ((mode >> 4) & 4) != 0);
  else if (wave == 2) scout_role(S, L, probs, sel, bits, nbits, lane, (mode & 4) != 0 && ((mode >> 4) & 2) != 0);
  else producer_role(S, L, nbits, wave - 3, lane, (mode & 4) != 0, mode >> 4);
  __syncthreads();
  if (tid == 0 && L.ctl->abort) S->error = 1;
}

// LATE: a decoder's chunk (cmx_late.h) -- `bits` is unused, rows / selectors arrive as their stages count them, p goes to the box. A compile-time
// switch: the compressor's kernel carries none of the decoder's state (125 VGPRs / 71 spilled SGPRs as before the decoder existed, against
// 167 / 135 when the two forms shared one kernel body at run time; the measured time per bit is the same either way, 6.8 us in the pipeline).
template <bool LATE, bool JIT = false> __device__ __forceinline__ void spec_kernel_body(
    MixState* __restrict__ S, SpecXfer* __restrict__ X, const float* __restrict__ probs, const uint32_t* __restrict__ sel,
    const uint8_t* __restrict__ bits, const float* __restrict__ decay1, int nbits,
    float* __restrict__ p_out, float* __restrict__ mix_out, int mode, const CmxLate& box) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  // Placement (CMX_MIXNET_XCD, mode bit 0x20000): the grid is 8 x 27 workgroups and only those with blockIdx % 8 == k work (observed: block b runs
  // on XCD b % 8) -- the 27 roles then share ONE L2 and the per-bit exchange of the u / sum words goes through it (plain store, L1-bypassing load:
  // 0.37 us per all-gather instead of ~1.1 through the fabric, profiles/r05_allgather.txt). The mapping is an observation, not a contract: every
  // workgroup publishes the XCC it really runs on and the fast form is used only if all 27 agree; the words are value|tag, so a wrong guess could
  // only ever time out, never deliver a stale value.
  // Test hook (CMX_MIXNET_ROTATE, mode bits 8..10): the first `rot` workgroups of the grid leave at once and the roles move up by that many -- block b runs on
  // XCD b mod 8, so with rot = launch number mod 8 every role changes its XCD (its L2) from one launch of a stream to the next. A stream's state crosses launches
  // through memory only (weight rows, row state, SSE cells); results must not depend on WHERE a role ran last time.
  int role = (int)blockIdx.x - ((mode >> 8) & 7);
  bool local = false;
  if (role < 0) return;
  if (!(mode & 0x20000) && role > CMX_SPEC_HELPERS) return;   // diagnostic launches with a padded grid (CMX_MIXNET_PADGRID): the surplus workgroups leave at once
  if (mode & 0x20000) {
    if (!(mode & 0x1000000)) {   // (0x1000000: the stream carries a compute-unit mask that does the placement -- grid 27, every block has a role. NOT a bit of
                                 // 20..22, the XCD number's field: as 0x400000 it made XCD 4..7 take this branch -- 216 workgroups with roles up to 215,
                                 // out-of-bounds rows, the memory fault of profiles/r05_xcd_fault.txt)
      if (((int)blockIdx.x & 7) != ((mode >> 20) & 7)) return;
      role = (int)blockIdx.x >> 3;
    }
    __shared__ int s_local;
    if (tid == 0) {
      unsigned id;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
      __hip_atomic_store(&X->xcc[role], (id & 15u) + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {
      unsigned mine = 0, spins = 0;
      bool ok = true;
      for (;;) {
        if (lane <= CMX_SPEC_HELPERS && !mine) mine = ld_u32(&X->xcc[lane]);
        if (__ballot(lane <= CMX_SPEC_HELPERS && !mine) == 0) break;
        __builtin_amdgcn_s_sleep(4);
        if (++spins > (1u << 22) || ld_u32(&X->fail)) { ok = false; break; }   // (some workgroup never started: the launch fails through the usual flag below)
      }
      const unsigned first = (unsigned)__builtin_amdgcn_readlane((int)mine, 0);
      const bool same = __ballot(lane <= CMX_SPEC_HELPERS && mine != first) == 0;
      if (lane == 0) { s_local = ok && same; if (!ok) __hip_atomic_store(&X->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    __syncthreads();
    local = s_local != 0 && !(mode & 0x800000);   // (0x800000: diagnostic, CMX_MIXNET_XCD_NOLOCAL -- the placement without the L2 form of the hand-off)
  }
  if (role > 0) {
    HelperLds* H = reinterpret_cast<HelperLds*>(smem);
    for (int i = tid; i < (int)(sizeof(HelperLds) / 4); i += CMX_SPEC_THREADS) reinterpret_cast<int*>(H)[i] = 0;
    __syncthreads();
    CmxLateBox* const lb = LATE ? box.box : nullptr;
    const bool tol = (mode & 0x1000) != 0;
    if (wave < 4) helper_role<LATE, JIT>(S, X, H, nbits, role - 1, wave, lane, tol, lb, local, JIT ? ((mode >> 26) & 15) | 16 : 0, box, probs);
    return;
  }
  Lds L;
  L.prod = smem;                                                  // unused here (no producers): 16 floats
  L.xs = L.prod;                                                  // unused (the inputs go to the helpers through the global ring)
  L.rec = reinterpret_cast<BitRec*>(L.prod + 16);                 // 8: the scout waves run up to 5 bits ahead of the gather wave
  L.rr = 8; L.lead = 5;
  L.trec = reinterpret_cast<TailRec*>(L.rec + 8);                 // 2
  L.upd = reinterpret_cast<float*>(L.trec + 2);                   // 32
  L.dflag = reinterpret_cast<uint32_t*>(L.upd + 32);              // 32
  L.in2 = reinterpret_cast<float*>(L.dflag + 32);                 // 64
  L.ctl = reinterpret_cast<Ctl*>(L.in2 + 64);
  L.pfdump = (unsigned)(size_t)(lds_int*)(reinterpret_cast<int*>(L.ctl) + 16);
  L.w2 = reinterpret_cast<float*>(L.ctl) + 16 + 64;
  L.w1 = L.w2 + 64 + 16 + 64;
  L.exptab = reinterpret_cast<uint64_t*>(L.w1 + 20 * 68);
  L.sdone = reinterpret_cast<int*>(L.exptab + 32);                // 8
  uint16_t* lst = reinterpret_cast<uint16_t*>(L.sdone + 8);       // 2 x 32768 u16: the SSE's t_st / t_sq on chip
  L.lst = lst; L.lsq = lst + 32768;
  L.h2 = reinterpret_cast<float*>(lst + 65536);                   // 2 x 64
  L.bitring = reinterpret_cast<int*>(L.h2 + 128);                 // 8
  L.late = LATE ? box : CmxLate();
  L.jit = JIT ? ((mode >> 26) & 15) | 16 : 0;
  for (int i = tid; i < 32768; i += CMX_SPEC_THREADS) reinterpret_cast<uint32_t*>(lst)[i] = i < 16384 ? reinterpret_cast<const uint32_t*>(S->t_st)[i] : reinterpret_cast<const uint32_t*>(S->t_sq)[i - 16384];
  if (tid < 32) { L.upd[tid] = 0.0f; L.dflag[tid] = 0; L.exptab[tid] = cmx_exp2f_tab[tid]; }
  if (tid < 8) L.sdone[tid] = 0;
  if (tid < (int)(sizeof(Ctl) / 4)) reinterpret_cast<int*>(L.ctl)[tid] = 0;
  __syncthreads();
  if (LATE && tid == 0) { L.ctl->late_lo = (unsigned)(unsigned long long)box.box; L.ctl->late_hi = (unsigned)((unsigned long long)box.box >> 32); }   // (0 otherwise: cleared above)
  __syncthreads();
  const bool prof = (mode & 4) != 0;
  if (wave == 0) gather_role<LATE, JIT>(S, L, X, decay1, nbits, mix_out, prof, lane, local);
  else if (wave == 1) tail_a_role<LATE, JIT>(S, L, decay1, nbits, mix_out, lane, prof && ((mode >> 4) & 4) != 0);
  else if (wave == 3) tail_b_role<LATE, JIT>(S, L, decay1, nbits, p_out, mix_out, lane, prof && ((mode >> 4) & 4) != 0);
  else if (wave == 2) select_role<LATE, JIT>(S, L, X, sel, probs, nbits, lane);
  else if (wave >= 4) stretch_role<LATE, JIT>(S, L, X, probs, bits, nbits, wave - 4, lane);
  __syncthreads();
  if (tid == 0 && (L.ctl->abort || ld_u32(&X->fail))) S->error = 1;
}

// Grid: 1 + 26 workgroups of 512 threads (x 8 with the one-XCD placement); all of them must be resident at once (they hand values to each other inside
// the launch; every wait is bounded and a time-out sets SpecXfer::fail / MixState::error instead of hanging).
extern "C" __global__ __launch_bounds__(CMX_SPEC_THREADS) void cmx_mixnet_spec_kernel(
    MixState* __restrict__ S, SpecXfer* __restrict__ X, const float* __restrict__ probs, const uint32_t* __restrict__ sel,
    const uint8_t* __restrict__ bits, const float* __restrict__ decay1, int nbits,
    float* __restrict__ p_out, float* __restrict__ mix_out, int mode) {
  spec_kernel_body<false>(S, X, probs, sel, bits, decay1, nbits, p_out, mix_out, mode, CmxLate());
}
// test hook (CMX_MIXNET_JITTER, jitter_stall): the same roles with pseudo-random stalls -- an instantiation of its own, so the product kernel carries none of it
extern "C" __global__ __launch_bounds__(CMX_SPEC_THREADS) void cmx_mixnet_spec_jitter_kernel(
    MixState* __restrict__ S, SpecXfer* __restrict__ X, const float* __restrict__ probs, const uint32_t* __restrict__ sel,
    const uint8_t* __restrict__ bits, const float* __restrict__ decay1, int nbits,
    float* __restrict__ p_out, float* __restrict__ mix_out, int mode) {
  spec_kernel_body<false, true>(S, X, probs, sel, bits, decay1, nbits, p_out, mix_out, mode, CmxLate());
}
// the decoder's form (engine mode 3, cmx_late.h): same grid, same roles, patient
extern "C" __global__ __launch_bounds__(CMX_SPEC_THREADS) void cmx_mixnet_spec_late_kernel(
    MixState* __restrict__ S, SpecXfer* __restrict__ X, const float* __restrict__ probs, const uint32_t* __restrict__ sel,
    const float* __restrict__ decay1, int nbits, float* __restrict__ p_out, float* __restrict__ mix_out, int mode, CmxLate box) {
  spec_kernel_body<true>(S, X, probs, sel, nullptr, decay1, nbits, p_out, mix_out, mode, box);
}
