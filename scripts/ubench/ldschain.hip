// Microbenchmark: ordered add chain fed from LDS (the mixer chain wave's inner loop), variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define SEG 548
__device__ __forceinline__ float add4(float p, float4 v) {
  p = __fadd_rn(p, v.x); p = __fadd_rn(p, v.y); p = __fadd_rn(p, v.z); p = __fadd_rn(p, v.w); return p;
}
// variant 0: compiler waitcnt ladder (as in mixnet_chunk.hip now)
__device__ __forceinline__ float seg_v0(const float* rowp, float p) {
  const float4* row = reinterpret_cast<const float4*>(__builtin_assume_aligned(rowp, 16));
  float4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = row[i];
#pragma unroll 1
  for (int bi = 0; bi < 16; bi += 2) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = row[(bi + 1) * 8 + i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) p = add4(p, a[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = row[(bi + 2) * 8 + i];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) p = add4(p, b[i]);
  }
  return p;
}
// variant 1: inline-asm reads, ONE wait per 32 adds
typedef __attribute__((address_space(3))) const float lds_cf;
#define RD8(dst, addr, off)                                                                      \
  asm volatile("ds_read_b128 %0, %8 offset:%9\n\tds_read_b128 %1, %8 offset:%9+16\n\t"          \
               "ds_read_b128 %2, %8 offset:%9+32\n\tds_read_b128 %3, %8 offset:%9+48\n\t"        \
               "ds_read_b128 %4, %8 offset:%9+64\n\tds_read_b128 %5, %8 offset:%9+80\n\t"        \
               "ds_read_b128 %6, %8 offset:%9+96\n\tds_read_b128 %7, %8 offset:%9+112"          \
               : "=v"(dst[0]), "=v"(dst[1]), "=v"(dst[2]), "=v"(dst[3]), "=v"(dst[4]), "=v"(dst[5]), \
                 "=v"(dst[6]), "=v"(dst[7])                                                      \
               : "v"(addr), "i"(off))
typedef float v4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float add4v(float p, v4 v) {
  p = __fadd_rn(p, v.x); p = __fadd_rn(p, v.y); p = __fadd_rn(p, v.z); p = __fadd_rn(p, v.w); return p;
}
__device__ __forceinline__ float seg_v1(const float* rowp, float p) {
  lds_cf* base = (lds_cf*)rowp;
  v4 a[8], b[8];
  RD8(a, base, 0);
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]));
#pragma unroll 1
  for (int bi = 0; bi < 16; bi += 2) {
    RD8(b, base, 128);
#pragma unroll
    for (int i = 0; i < 8; ++i) p = add4v(p, a[i]);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(b[0]), "+v"(b[1]), "+v"(b[2]), "+v"(b[3]), "+v"(b[4]), "+v"(b[5]), "+v"(b[6]), "+v"(b[7]), "+v"(p));
    RD8(a, base, 256);
#pragma unroll
    for (int i = 0; i < 8; ++i) p = add4v(p, b[i]);
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]), "+v"(p));
    base += 64;
  }
  return p;
}
// variant 2: two lanes per chain -- lane L (L & 8 == 0) owns the chain, lane L + 8 of the same row of 16 fetches the
// other half of every 8 consecutive terms; the owner adds its partner's four values with DPP (row_shl:8) adds, so one
// ds_read_b128 of the wave feeds EIGHT terms of every chain instead of four.
__device__ __forceinline__ float dpp_add(float p, float x) {
  asm volatile("v_add_f32_dpp %0, %1, %0 row_shl:8 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(p) : "v"(x));
  return p;
}
__device__ __forceinline__ float add8(float p, float4 v) {
  p = add4(p, v);
  p = dpp_add(p, v.x); p = dpp_add(p, v.y); p = dpp_add(p, v.z); p = dpp_add(p, v.w);
  return p;
}
// rowp: the lane's own view (owner: row, partner: row + 4 floats); 512 terms = 64 reads of 8 terms
__device__ __forceinline__ float seg_v2(const float* rowp, float p) {
  const float4* row = reinterpret_cast<const float4*>(__builtin_assume_aligned(rowp, 16));
  float4 a[8], b[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = row[2 * i];
#pragma unroll 1
  for (int bi = 0; bi < 8; bi += 2) {
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = row[2 * ((bi + 1) * 8 + i)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) p = add8(p, a[i]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = row[2 * ((bi + 2) * 8 + i)];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 8; ++i) p = add8(p, b[i]);
  }
  return p;
}
template <int V, int ACTIVE, int EXTRA_WAVES>
__global__ void k(float* out, uint64_t* t) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  for (int i = threadIdx.x; i < 26 * SEG + 64; i += blockDim.x) sm[i] = 1.0f + (i & 7);
  __syncthreads();
  if (threadIdx.x >= 64) {  // co-resident waves: sleep-poll like the real kernel
    volatile float* f = sm;
    for (int i = 0; i < 4000; ++i) { __builtin_amdgcn_s_sleep(2); if (f[0] < 0) break; }
    return;
  }
  int m = threadIdx.x < 26 ? threadIdx.x : 0;
  const int L = threadIdx.x, m2 = (L >> 4) * 8 + (L & 7), part = (L >> 3) & 1;
  float p = 0;
  __builtin_amdgcn_s_setprio(3);
  uint64_t c0 = __builtin_readcyclecounter();
  if ((int)threadIdx.x < ACTIVE) {
#pragma unroll 1
    for (int r = 0; r < 64; ++r) p = V == 0 ? seg_v0(sm + m * SEG, p) : V == 1 ? seg_v1(sm + m * SEG, p) : seg_v2(sm + (m2 < 26 ? m2 : 0) * SEG + 4 * part, p);
  }
  uint64_t c1 = __builtin_readcyclecounter();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0) t[0] = c1 - c0;
}
template <int V, int ACTIVE, int EW>
void run(const char* name, float* out, uint64_t* t) {
  uint64_t h;
  for (int rep = 0; rep < 2; ++rep) {
    k<V, ACTIVE, EW><<<1, 64 * (1 + EW), 26 * SEG * 4 + 256>>>(out, t);
    hipDeviceSynchronize();
    hipMemcpy(&h, t, 8, hipMemcpyDeviceToHost);
  }
  printf("%-44s %.2f ticks/add\n", name, h / (64.0 * 512));
}
int main() {
  float* out; uint64_t* t;
  hipMalloc(&out, 4096); hipMalloc(&t, 64);
  run<0, 64, 0>("v0 compiler ladder, 64 lanes, alone", out, t);
  run<0, 26, 0>("v0 compiler ladder, 26 lanes, alone", out, t);
  run<1, 64, 0>("v1 asm one-wait, 64 lanes, alone", out, t);
  run<1, 26, 0>("v1 asm one-wait, 26 lanes, alone", out, t);
  run<1, 26, 11>("v1 asm one-wait, 26 lanes, +11 sleeping waves", out, t);
  run<0, 26, 11>("v0 ladder, 26 lanes, +11 sleeping waves", out, t);
  run<2, 64, 0>("v2 owner+partner DPP, 64 lanes, alone", out, t);
  run<2, 64, 11>("v2 owner+partner DPP, +11 sleeping waves", out, t);
  return 0;
}
