// mixnet_dev.h -- device helpers shared by the mixing-network kernels:
// strict f32 arithmetic, Mixer::GetContextData row selection, the SSE stage.
#ifndef CMX_MIXNET_DEV_H
#define CMX_MIXNET_DEV_H
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cmx_libm.h"
#include "mixnet_state.h"

namespace {

// Pointers read out of MixState are generic ("flat") to the compiler; flat accesses tick the LDS
// counter as well as the vector-memory one. These casts pin the hot ones to the global address space.
template <class T> using gptr = __attribute__((address_space(1))) T*;
template <class T> __device__ __forceinline__ gptr<T> as_global(T* p) { return (gptr<T>)p; }
template <class T> __device__ __forceinline__ gptr<const T> as_global(const T* p) { return (gptr<const T>)p; }

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 gload4(gptr<const float> p) {
  v4f v = *(gptr<const v4f>)p;
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float4 gload4(gptr<float> p) { return gload4((gptr<const float>)p); }
// 16-byte global store issued from an asm statement. hipcc keeps no waitcnt bookkeeping for it, so
// re-using the data registers afterwards (loading the next weight row into them) does not make
// the compiler drain vmcnt(0) -- i.e. wait for the store's acknowledgement from L2 -- the way a
// compiler-visible store does. The hardware has read the data registers once the trailing
// s_nop 1 has passed. Vector memory operations of one wave are performed in issue order, so a later
// load of the same address by this wave still observes the stored value.
__device__ __forceinline__ void gstore4_async(gptr<float> p, float4 v) {
  v4f o = {v.x, v.y, v.z, v.w};
  asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" :: "v"(p), "v"(o) : "memory");
}
__device__ __forceinline__ void gstore4(gptr<float> p, float4 v) {
  v4f o = {v.x, v.y, v.z, v.w};
  *(gptr<v4f>)p = o;
}

__device__ __forceinline__ float fmul(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float fadd(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ float fsub(float a, float b) { return __fsub_rn(a, b); }

// Mixer::GetContextData, ref mixer.cpp:16-36. One lane per mixer; tables are
// private to a mixer so there are no races. Returns the row index.
__device__ uint32_t select_row(MixState* S, int m, uint32_t key) {
  uint32_t* keys = S->map_keys + (size_t)m * CMX_MAP_SLOTS;
  uint32_t* vals = S->map_vals + (size_t)m * CMX_MAP_SLOTS;
  uint32_t n = S->n_rows[m];
  uint32_t h = (key * 2654435761u) >> 17;
  while (vals[h] != 0 && keys[h] != key) h = (h + 1) & (CMX_MAP_SLOTS - 1);
  if (vals[h] == 0 && n >= CMX_ROW_LIMIT) {  // unseen key after the cap: shared overflow row
    key = 0xDEADBEEFu;
    h = (key * 2654435761u) >> 17;
    while (vals[h] != 0 && keys[h] != key) h = (h + 1) & (CMX_MAP_SLOTS - 1);
  }
  if (vals[h] == 0) {  // first touch: rows are zero-initialised memory
    keys[h] = key;
    vals[h] = ++n;
    S->n_rows[m] = n;
  }
  return vals[h] - 1;
}

// SSE stage, single lane. ref sse.cpp:138-143
__device__ __forceinline__ int sse_extrap(int p1, int C) {
  p1 = (((p1 - 16384) * C) >> 13) + 16384;
  if (p1 < 1) p1 = 1;
  if (p1 > 32767) p1 = 32767;
  return p1;
}
__device__ __forceinline__ int sse_rdiv(int x, int a, int d) {
  return x >= 0 ? (x + a) >> d : -((-x + a) >> d);
}
__device__ __forceinline__ int sse_mixup(int w, int s1, int s0) {  // ref sse.cpp:166-170
  int x = s1 + sse_rdiv((w - 16384) * (s0 - s1), 1 << 14, 15);
  return (x > 0) ? ((x < 32768) ? x : 32767) : 1;
}
__device__ __forceinline__ int sse_wdelta(int y, int p0, int p1, int wq, int pm) {  // ref :172-178
  int e = (32768 - (y << 15)) - pm;
  int d = sse_rdiv(e * (p0 - p1), 1 << 14, 15);
  return sse_rdiv(d * wq, 1 << 14, 15);
}
__device__ __forceinline__ int sse_mx1mask(int j) {  // ref sse.cpp:154
  if (j < 32) return j ? j - 1 : 0;
  if (j < 64) return 31 + (j - 32) / 2;
  if (j < 128) return 47 + (j - 64) / 4;
  return 63 + (j - 128) / 8;
}

struct SseCell {  // SSEi<7>::SSE_Pred / SSE_Update, ref sse.cpp:37-62
  uint16_t* C1;
  int sw, P;
  __device__ __forceinline__ int pred(uint16_t* P7, int iP) {
    int freq = (6 * iP) >> 15;
    sw = (6 * iP) & 32767;
    C1 = P7 + freq;
    int f = (((32768 - sw) * (int)C1[0] + sw * (int)C1[1]) >> 15) - 8192;
    if (f <= 0) f = 1;
    if (f >= 32768) f = 32767;
    P = f;
    return f;
  }
  __device__ __forceinline__ void update(int c, int wr0) {
    P = (P * (32768 - wr0)) >> 15;
    if (c == 0) P += wr0;
    int dC = (int)C1[0] - (int)C1[1];
    int sw_dC = (sw * dC + 32767) >> 15;
    C1[0] = (uint16_t)(P + sw_dC + 8192);
    C1[1] = (uint16_t)(P - (dC - sw_dC) + 8192);
  }
};

// SSE::Predict + SSE::Perceive fused (the bit is known in chunk mode; in
// bit-synchronous mode the two halves run in separate launches and the cell
// state in between is re-derived, which is exact because Predict has no side
// effects on the tables). ref sse.cpp:243-328
__device__ float sse_step(MixState* S, float input, int bit, bool do_update) {
  const uint16_t* t_st = S->t_st;
  const uint16_t* t_sq = S->t_sq;
  int p = (int)(1 + (1 - input) * 32766);
  unsigned j = S->sse_j, pc = S->sse_pc, ffl = S->sse_ffl, prq = (unsigned)p >> 11;
  int a = (prq > 0) + (prq > 14);
  int b = (prq > 0) + (prq > 7) + (prq > 14);
  int sm7x = ((((a << 5) + (int)(ffl & 31)) << 8) + (int)(pc & 255)) * 255 + (j ? (int)j - 1 : 0);
  int mix2 = ((((a << 1) + (int)(ffl & 1)) << 8) + (int)(pc & 255)) * 256 + (int)j;
  int sm6x = ((((a << 7) + (int)(ffl & 127)) << 8) + (int)(pc & 255)) * 256 + (int)j;
  int mix1 = ((((b << 8) + (int)(ffl & 255)) << 3) + (int)((pc >> 5) & 7)) * 79 + sse_mx1mask((int)j);

  SseCell c6, c7;
  int stp = t_st[p];
  int p1 = c6.pred(S->s6 + (size_t)sm6x * 8, t_sq[sse_extrap(stp, 10240)]);
  int s0 = sse_extrap(stp, 7935);
  int s1 = sse_extrap(t_st[p1], 9592);
  int w1 = S->x1[mix1];
  int s2 = sse_extrap(sse_mixup(w1, s0, s1), 8092);
  int mix1_p = t_sq[s2];
  int p2 = c7.pred(S->s7 + (size_t)sm7x * 8, t_sq[sse_extrap(stp, 8200)]);
  int s4 = sse_extrap(t_st[p2], 7677);
  int w2 = S->x2[mix2];
  int s5 = sse_extrap(sse_mixup(w2, s2, s4), 8202);
  int mix2_p = t_sq[s5];
  float out = (float)(1 - ((mix2_p - 1) / 32766.0));

  if (do_update) {
    c6.update(bit, 106);
    S->x1[mix1] = w1 + sse_wdelta(bit, s0, s1, 6202, mix1_p);
    c7.update(bit, 127);
    S->x2[mix2] = w2 + sse_wdelta(bit, s2, s4, 8320, mix2_p);
    j += j + (unsigned)bit;
    if (j >= 256) {
      ffl = (ffl * 2 + (pc >= 0x40)) & 255;
      pc = j & 255;
      j = 1;
    }
    S->sse_j = j;
    S->sse_pc = pc;
    S->sse_ffl = ffl;
  }
  return out;
}


}  // namespace
#endif
