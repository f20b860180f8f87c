// mixnet_chunk.hip -- look-ahead (chunk mode) kernel of the final mixing network.
//
// Same arithmetic, same HBM state and same results as the bit-synchronous kernel in
// mixnet_kernels.hip, restructured as a 12-wave software pipeline inside ONE persistent
// workgroup, because in compression every bit of the chunk is already known:
//
//   wave 0  chain    lane m = layer-0 mixer m: the 26 ordered 2078-term add chains
//                    (mixer.cpp:40-43) in four LDS-staged segments, the intra-layer
//                    extra-input chain (predictor.cpp:395-400), Mixer::Perceive scalars.
//   wave 4  tail     one bit behind: layer 1, layer 2, squash, SSE, override, output
//                    (predictor.cpp:402-418) and the updates of those rows.
//   wave 8  scout    up to two bits ahead: MixerInput stretch of the 2078 inputs, aux
//                    context, Mixer::GetContextData row selection for all 47 mixers.
//   other 9 waves    producers p=0..8: own the selected layer-0 rows of mixers p, p+9,
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
};

struct BitRec {            // written by the scout for bit t (slot t % 3)
  uint32_t rowidx[48];
  uint32_t changed[32];    // layer-0 row differs from the previous bit's
  float aux3[4];           // clamped stretch of the three auxiliary inputs
  float lstm_p;            // raw probs[t][2077] (override test)
  int bit;
  int pad[2];
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
  BitRec* rec;     // [3]
  TailRec* trec;   // [2]
  float* upd;      // [32]
  uint32_t* dflag; // [32]
  float* in2;      // [64] (tail wave scratch)
  Ctl* ctl;
  unsigned pfdump; // LDS byte offset of a 256-byte dump area for the row-prefetch LDS-DMA loads
};

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
__device__ __forceinline__ bool wait_ge(Ctl* ctl, const int* p, int target, bool sleepy) {
  unsigned spins = 0;
  while (lds_poll(p) < target) {
    if (sleepy) __builtin_amdgcn_s_sleep(2);
    if ((++spins & 1023u) == 0) {
      if (lds_poll(&ctl->abort) || spins > SPIN_LIMIT) {
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
__device__ __forceinline__ float chain_seg(const float* rowp, int n, float p) {
  const float4* row = reinterpret_cast<const float4*>(__builtin_assume_aligned(rowp, 16));
  float4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = row[i];
#pragma unroll 1
  for (int bi = 0; bi < 16; bi += 2) {
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
    for (int i = 0; i < 8; ++i) a[i] = row[(bi + 2) * 8 + i];   // bi == 14: the remainder batch
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) p = add4(p, b[i]);
  }
  __builtin_amdgcn_sched_barrier(0);
  const int rem4 = (n >> 2) - 128;  // 0 or 7
#pragma unroll
  for (int i = 0; i < 7; ++i)
    if (i < rem4) p = add4(p, a[i]);
  if (n & 2) {                      // 542 = 135 * 4 + 2
    p = fadd(p, a[7].x);
    p = fadd(p, a[7].y);
  }
  return p;
}

// ------------------------------------------------------------------ scout (wave 2)
__device__ void scout_role(MixState* S, const Lds& L, const float* probs, const uint32_t* sel,
                           const uint8_t* bits, int nbits, int lane, bool prof_on) {
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
    if (t >= 2 && !wait_ge(L.ctl, &L.ctl->consumed, 4 * t - 4, true)) return;
    SPROF(6);
    float* xs = L.xs + (t % 3) * XS;
    BitRec* rec = L.rec + (t % 3);
    const BitRec* prev = L.rec + ((t + 2) % 3);
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
#pragma unroll
    for (int r = 0; r < 33; ++r) {
      int i = r * 64 + lane;
      if (i < CMX_IN0) xs[i] = pv[r];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    SPROF(7);
    if (lane == CMX_AUX) {  // predictor.cpp:388-393
      float avg = 0;
      avg = fadd(avg, cmx_logistic(xs[433]));
      avg = fadd(avg, cmx_logistic(xs[2024]));
      avg = fadd(avg, cmx_logistic(xs[2077]));
      avg = avg / 3.0f;
      key = (uint32_t)(unsigned long long)(avg * 15);
    }
    if (lane < CMX_MIXERS) {
      uint32_t r = select_row(S, lane, key);
      rec->rowidx[lane] = r;
      if (lane < CMX_MIX0) rec->changed[lane] = (t == 0) || (r != prev->rowidx[lane]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    SPROF(8);
    SPROF(9);
    if (lane < 3) {
      float v = xs[lane == 0 ? 433 : lane == 1 ? 2024 : 2077];
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
    if (prof_on && p == 0 && !(dbg & 2)) {                             \
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
    const BitRec* rec = L.rec + (t % 3);
    const BitRec* prev = L.rec + ((t + 2) % 3);
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
      const BitRec* nxt = L.rec + ((t + 1) % 3);
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
  if (prof_on && p == 0 && !(dbg & 2) && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) S->prof[6 + i] += pacc[i];
  }
#undef PPROF
}

// ------------------------------------------------------------------ chain (wave 0)
__device__ void chain_role(MixState* S, const Lds& L, const float* decay1, int nbits,
                           float* mix_out, bool prof_on, int lane) {
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
    const BitRec* rec = L.rec + (t % 3);
    const int bit = rec->bit;
    if (is0 && rec->changed[mm]) {
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
      const float mine = clamp_out(fadd(pm, e), smin, smax);
      const float oj = bcast_lane(mine, j);
      e = fadd(e, fmul(oj, ew[j]));
    }
    const float p_ = fadd(pm, e);
    const float myout = clamp_out(p_, smin, smax);
    CPROF(3);
    // Mixer::Perceive scalar (mixer.cpp:56-64)
    float uu = fmul(dlr, fsub(cmx_logistic(p_), (float)bit));
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
      if (i < 6 || i > 12) S->prof[i] += pacc[i];
  }
#undef CPROF
}

// ------------------------------------------------------------------ tail (wave 1)
__device__ void tail_role(MixState* S, const Lds& L, const float* decay1, int nbits, float* p_out,
                          float* mix_out, int lane) {
  const int k = lane;  // layer-1 mixer index
  const bool is1 = k < CMX_MIX1;
  const float smin = S->stretch_min, smax = S->stretch_max;
  const float cdec = 1.0f - 3.0e-6f;
  for (int t = 0; t < nbits; ++t) {
    if (!wait_ge(L.ctl, &L.ctl->tail_in, t + 1, true)) return;
    const TailRec* tr = L.trec + (t & 1);
    const int bit = tr->bit;
    const double d1 = (double)as_global(decay1)[t];
    // layer-1 inputs: 26 clamped layer-0 outputs + 3 auxiliary stretches (predictor.cpp:397-406)
    float in1[CMX_IN1];
#pragma unroll
    for (int i = 0; i < CMX_MIX0; ++i) in1[i] = tr->out0[i];
#pragma unroll
    for (int i = 0; i < 3; ++i) in1[CMX_MIX0 + i] = tr->aux3[i];
    // rows: 64 floats = 29 weights, pad, 20 extra weights
    const int kk = is1 ? k : 0;
    const gptr<float> row1 = as_global(S->rows1) + ((size_t)kk * CMX_ROWS_PER_MIXER + tr->rowidx[kk]) * CMX_ROW1_STRIDE;
    float w1[64];
    {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float4 v = gload4(row1 + 4 * i);
        w1[4 * i] = v.x; w1[4 * i + 1] = v.y; w1[4 * i + 2] = v.z; w1[4 * i + 3] = v.w;
      }
    }
    const gptr<uint64_t> rsp1 = as_global(S->row_steps) + (size_t)(CMX_MIX0 + kk) * CMX_ROWS_PER_MIXER + tr->rowidx[kk];
    uint64_t rs1 = *rsp1, mx1 = S->max_steps[CMX_MIX0 + kk];
    const gptr<float> row2 = as_global(S->rows2) + (size_t)tr->rowidx[CMX_MIX1] * CMX_ROW2_STRIDE;
    const gptr<uint64_t> rsp2 = as_global(S->row_steps) + (size_t)(CMX_MIXERS - 1) * CMX_ROWS_PER_MIXER + tr->rowidx[CMX_MIX1];

    float pm = 0.0f;
#pragma unroll
    for (int i = 0; i < CMX_IN1; ++i) pm = fadd(pm, fmul(in1[i], w1[i]));
    float e = 0.0f, p1_ = 0.0f, myout = 0.0f;
#pragma unroll
    for (int j = 0; j < CMX_MIX1; ++j) {
      float mine = fadd(pm, e);
      float oj = bcast_lane(mine, j);
      if (oj > smax) oj = smax;
      else if (oj < smin) oj = smin;
      if (k == j) { p1_ = mine; myout = oj; }
      if (k > j) e = fadd(e, fmul(oj, w1[CMX_ROW1_EXTRA + j]));
    }
    // layer-2 inputs = 26 + 20 + 3 (predictor.cpp:397-411)
    if (k < CMX_MIX0) L.in2[k] = tr->out0[k];
    if (is1) L.in2[CMX_MIX0 + k] = myout;
    if (k < 3) L.in2[CMX_MIX0 + CMX_MIX1 + k] = tr->aux3[k];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    float p2_ = 0.0f;
    if (k == 0) {
      float w2[52];
#pragma unroll
      for (int i = 0; i < 13; ++i) {
        float4 v = gload4(row2 + 4 * i);
        w2[4 * i] = v.x; w2[4 * i + 1] = v.y; w2[4 * i + 2] = v.z; w2[4 * i + 3] = v.w;
      }
      uint64_t rs2 = *rsp2, mx2 = S->max_steps[CMX_MIXERS - 1];
      float acc = 0.0f;
#pragma unroll
      for (int i = 0; i < CMX_IN2; ++i) acc = fadd(acc, fmul(L.in2[i], w2[i]));
      p2_ = acc;
      float pf = sse_step(S, cmx_logistic(p2_), bit, true);  // predictor.cpp:413-414
      float lp = tr->lstm_p;
      if (lp == 0.0f || lp == 1.0f) pf = lp;                 // predictor.cpp:383,415-417
      as_global(p_out)[t] = pf;
      // Mixer::Perceive, layer 2 (mixer.cpp:56-72)
      float decay = (float)(d1 * (1.5 - ((1.0 * (double)rs2) / (double)mx2)));
      float u = fmul(fmul(decay, S->lr[CMX_MIXERS - 1]), fsub(cmx_logistic(p2_), (float)bit));
      ++rs2;
      *rsp2 = rs2;
      if (rs2 > mx2) S->max_steps[CMX_MIXERS - 1] = rs2;
      const bool df = (rs2 & 1023) == 0;
#pragma unroll
      for (int i = 0; i < CMX_IN2; ++i) {
        float v = fsub(w2[i], fmul(u, L.in2[i]));
        if (df) v = fmul(v, cdec);
        w2[i] = v;
      }
#pragma unroll
      for (int i = 0; i < 13; ++i) gstore4(row2 + 4 * i, make_float4(w2[4 * i], w2[4 * i + 1], w2[4 * i + 2], w2[4 * i + 3]));
      if (mix_out) mix_out[(size_t)t * CMX_MIXERS + CMX_MIXERS - 1] = p2_;
      S->steps = S->steps + 1;
    }
    // Mixer::Perceive, layer 1
    if (is1) {
      float decay = (float)(d1 * (1.5 - ((1.0 * (double)rs1) / (double)mx1)));
      float u = fmul(fmul(decay, S->lr[CMX_MIX0 + k]), fsub(cmx_logistic(p1_), (float)bit));
      ++rs1;
      *rsp1 = rs1;
      if (rs1 > mx1) S->max_steps[CMX_MIX0 + k] = rs1;
      const bool df = (rs1 & 1023) == 0;
#pragma unroll
      for (int i = 0; i < CMX_IN1; ++i) {
        float v = fsub(w1[i], fmul(u, in1[i]));
        if (df) v = fmul(v, cdec);
        w1[i] = v;
      }
#pragma unroll
      for (int j = 0; j < CMX_MIX1; ++j) {
        if (j < k) {
          float v = fsub(w1[CMX_ROW1_EXTRA + j], fmul(u, L.in2[CMX_MIX0 + j]));
          if (df) v = fmul(v, cdec);
          w1[CMX_ROW1_EXTRA + j] = v;
        }
      }
#pragma unroll
      for (int i = 0; i < 13; ++i) gstore4(row1 + 4 * i, make_float4(w1[4 * i], w1[4 * i + 1], w1[4 * i + 2], w1[4 * i + 3]));
      if (mix_out) mix_out[(size_t)t * CMX_MIXERS + CMX_MIX0 + k] = p1_;
    }
    st_rel(&L.ctl->tail_done, t + 1);
  }
}

}  // namespace

extern "C" __global__ __launch_bounds__(NTHREADS) void cmx_mixnet_chunk_kernel(
    MixState* __restrict__ S, const float* __restrict__ probs, const uint32_t* __restrict__ sel,
    const uint8_t* __restrict__ bits, const float* __restrict__ decay1, int nbits,
    float* __restrict__ p_out, float* __restrict__ mix_out, int mode) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  Lds L;
  L.prod = smem;                                                  // 2 * PBUF
  L.xs = L.prod + 2 * PBUF;                                       // 3 * XS
  L.rec = reinterpret_cast<BitRec*>(L.xs + 3 * XS);               // 3
  L.trec = reinterpret_cast<TailRec*>(L.rec + 3);                 // 2
  L.upd = reinterpret_cast<float*>(L.trec + 2);                   // 32
  L.dflag = reinterpret_cast<uint32_t*>(L.upd + 32);              // 32
  L.in2 = reinterpret_cast<float*>(L.dflag + 32);                 // 64
  L.ctl = reinterpret_cast<Ctl*>(L.in2 + 64);
  L.pfdump = (unsigned)(size_t)(lds_int*)(reinterpret_cast<int*>(L.ctl) + 16);   // 256 B behind the control block
  const int tid = threadIdx.x;
  for (int i = tid; i < 3 * XS; i += NTHREADS) L.xs[i] = 0.0f;    // incl. the zero padding 2078..2111
  for (int i = tid; i < 2 * PBUF; i += NTHREADS) L.prod[i] = 0.0f;
  if (tid < 32) { L.upd[tid] = 0.0f; L.dflag[tid] = 0; }
  if (tid < (int)(sizeof(Ctl) / 4)) reinterpret_cast<int*>(L.ctl)[tid] = 0;
  __syncthreads();
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
  if ((mode & 4) && lane == 0) {  // profiling: which SIMD does each wave sit on (HW_ID[5:4])
    unsigned hwid = __builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    atomicOr((unsigned long long*)&S->prof[12], (unsigned long long)((hwid >> 4) & 3) << (4 * wave));
  }
  // Waves w, w+4, w+8 share a SIMD. The chain wave's SIMD-mates are the two latency-tolerant,
  // mostly-sleeping roles, so nothing competes with its dependent add chain for issue slots.
  if (wave == 0) chain_role(S, L, decay1, nbits, mix_out, (mode & 4) != 0, lane);
  else if (wave == 4) tail_role(S, L, decay1, nbits, p_out, mix_out, lane);
  else if (wave == 8) scout_role(S, L, probs, sel, bits, nbits, lane, (mode & 4) != 0 && ((mode >> 4) & 2) != 0);
  else producer_role(S, L, nbits, wave - 1 - (wave > 4) - (wave > 8), lane, (mode & 4) != 0, mode >> 4);
  __syncthreads();
  if (tid == 0 && L.ctl->abort) S->error = 1;
}
