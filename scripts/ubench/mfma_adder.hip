// Microbenchmark + exactness check: v_mfma_f32_4x4x1_16b_f32 used as a BROADCAST ADDER for the strict ordered sums.
//
// D[b][i][j] = C[b][i][j] + A[b][i] * B[b][j] with K = 1 is ONE fused multiply-add per element; with B = 1.0 the product is exact, so
// D = RN(C + A): the reference's separately rounded `sum += product` (mixer.cpp:40-43) -- IF the unit rounds to nearest even, keeps
// denormals and has no other surprise. This program checks that claim on the hardware (random operands over cancellation, binade
// crossings, denormals, signed zeros), prints the operand layout (which lane's A / B lands in which lane / register of D), and times a
// dependent chain of such MFMAs with the A operand fed by a DPP row broadcast (row_newbcast) from the lanes that own the products.
//
//   hipcc --offload-arch=gfx950 -O3 -o mfma_adder mfma_adder.hip && ./mfma_adder
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>

typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}
// a float with sign / mantissa from h and the exponent field e (0 = denormal or zero)
__device__ __forceinline__ float mkf(uint32_t h, int e) {
  if (e < 0) e = 0;
  if (e > 254) e = 254;
  return __uint_as_float((h & 0x807fffffu) | ((uint32_t)e << 23));
}

// layout: out[lane][0..3] = A-source lane of D register v, out[lane][4..7] = B-source lane
__global__ void layout_kernel(int* out) {
  const int lane = threadIdx.x;
  f4 z = {0, 0, 0, 0};
  f4 da = __builtin_amdgcn_mfma_f32_4x4x1f32((float)lane, 1.0f, z, 0, 0, 0);
  f4 db = __builtin_amdgcn_mfma_f32_4x4x1f32(1.0f, (float)lane, z, 0, 0, 0);
  for (int v = 0; v < 4; ++v) { out[lane * 8 + v] = (int)da[v]; out[lane * 8 + 4 + v] = (int)db[v]; }
}

// exactness: mode 0 = operands of comparable size (cancellation), 1 = wide exponent spread, 2 = denormal range, 3 = signed zeros / equal magnitudes
__global__ void exact_kernel(unsigned long long* bad, unsigned long long* total, uint32_t* example, int iters, int mode) {
  const int lane = threadIdx.x & 63;
  const uint32_t gid = blockIdx.x * blockDim.x + threadIdx.x;
  f4 z = {0, 0, 0, 0};
  f4 la = __builtin_amdgcn_mfma_f32_4x4x1f32((float)lane, 1.0f, z, 0, 0, 0);   // A-source lane per D register
  unsigned long long nbad = 0;
  for (int it = 0; it < iters; ++it) {
    const uint32_t h0 = mix32(gid * 2654435761u + it * 40503u + mode * 977u);
    int ec, ea;
    if (mode == 0) { ec = 100 + (h0 & 31); ea = ec + (int)((h0 >> 5) & 7) - 5; }
    else if (mode == 1) { ec = 60 + (h0 & 127); ea = ec + (int)((h0 >> 7) & 63) - 40; }
    else if (mode == 2) { ec = (h0 & 3); ea = ((h0 >> 2) & 3); }
    else { ec = 120; ea = 120; }
    const float a = mode == 3 ? mkf(mix32(h0) & 0x80000003u, (h0 & 1) ? 120 : 0) : mkf(mix32(h0 + 1), ea);
    f4 c;
    for (int v = 0; v < 4; ++v) {
      const uint32_t hv = mix32(h0 + 7 * v + 3);
      c[v] = mode == 3 ? mkf(hv & 0x80000003u, (hv & 4) ? 120 : 0) : mkf(hv, ec + (int)(hv >> 29) - 3);
    }
    f4 d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, 1.0f, c, 0, 0, 0);
    for (int v = 0; v < 4; ++v) {
      const float asrc = __shfl(a, (int)la[v], 64);
      const float want = __fadd_rn(c[v], asrc);
      if (__float_as_uint(want) != __float_as_uint(d[v])) {
        if (nbad == 0 && atomicAdd(&example[0], 1u) == 0) {
          example[1] = __float_as_uint(asrc); example[2] = __float_as_uint(c[v]); example[3] = __float_as_uint(want); example[4] = __float_as_uint(d[v]); example[5] = mode;
        }
        ++nbad;
      }
    }
  }
  atomicAdd(bad, nbad);
  atomicAdd(total, 4ull * iters);
}

// a chain: acc = mfma(A_t, 1, acc), t = 0..n-1, against the same chain with v_add_f32; A_t = product t of the row's segment, broadcast from lane t % 16 of
// each row of 16 lanes (DPP row_newbcast) out of register t / 16 -- the layout the mixing network's helpers would use (16 lanes own 144 products of a segment)
template <int R> __device__ __forceinline__ float row_bcast(float src) {   // DPP row_newbcast:R (gfx90a+): lane R of each row of 16 lanes to the whole row
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(src), 0x150 + R, 0xf, 0xf, true));
}

template <bool USE_MFMA, int CHAINS> __global__ void chain_kernel(float* out, unsigned long long* ticks, int reps, unsigned long long* mism) {
  const int lane = threadIdx.x & 63;
  float P[9];
  for (int k = 0; k < 9; ++k) P[k] = mkf(mix32(lane * 9 + k + 1), 118 + (int)(mix32(lane + 31 * k) & 7));
  f4 acc[CHAINS];
  for (int c = 0; c < CHAINS; ++c) { acc[c][0] = 0.5f * lane + c; acc[c][1] = 1.0f + c; acc[c][2] = -3.0f; acc[c][3] = 0.25f * lane; }
  const unsigned long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int rep = 0; rep < reps; ++rep) {
#define STEP(k, r)                                                                                                   \
    {                                                                                                                \
      const float a = row_bcast<r>(P[k]);                                                                            \
      _Pragma("unroll") for (int c = 0; c < CHAINS; ++c) {                                                           \
        if (USE_MFMA) acc[c] = __builtin_amdgcn_mfma_f32_4x4x1f32(a, 1.0f, acc[c], 0, 0, 0);                         \
        else { acc[c][0] = __fadd_rn(acc[c][0], a); acc[c][1] = __fadd_rn(acc[c][1], a); acc[c][2] = __fadd_rn(acc[c][2], a); acc[c][3] = __fadd_rn(acc[c][3], a); } \
      }                                                                                                              \
    }
#define ROW(k) STEP(k, 0) STEP(k, 1) STEP(k, 2) STEP(k, 3) STEP(k, 4) STEP(k, 5) STEP(k, 6) STEP(k, 7) STEP(k, 8) STEP(k, 9) STEP(k, 10) STEP(k, 11) STEP(k, 12) STEP(k, 13) STEP(k, 14) STEP(k, 15)
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5) ROW(6) ROW(7) ROW(8)
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  for (int c = 0; c < CHAINS; ++c)
    for (int v = 0; v < 4; ++v) out[((blockIdx.x * blockDim.x + threadIdx.x) * CHAINS + c) * 4 + v] = acc[c][v];
  if (threadIdx.x == 0 && blockIdx.x == 0) ticks[0] = t1 - t0;
  (void)mism;
}

int main() {
  int* d_lay; hipMalloc(&d_lay, 64 * 8 * 4);
  layout_kernel<<<1, 64>>>(d_lay);
  int lay[64 * 8];
  hipMemcpy(lay, d_lay, sizeof lay, hipMemcpyDeviceToHost);
  printf("layout of v_mfma_f32_4x4x1_16b_f32: D register v of lane l = C + A[lane a] * B[lane b]\n");
  for (int l = 0; l < 64; l += 1) {
    if (l < 8 || l == 16 || l == 21 || l == 63) {
      printf("  lane %2d:", l);
      for (int v = 0; v < 4; ++v) printf("  v%d <- A lane %2d, B lane %2d;", v, lay[l * 8 + v], lay[l * 8 + 4 + v]);
      printf("\n");
    }
  }
  bool regular = true;
  for (int l = 0; l < 64; ++l)
    for (int v = 0; v < 4; ++v)
      if (lay[l * 8 + v] != (l / 4) * 4 + v || lay[l * 8 + 4 + v] != l) regular = false;
  printf("layout is { block = lane / 4, row i = register, column j = lane %% 4 ; A from lane 4 * block + i, B from the lane itself }: %s\n", regular ? "yes" : "NO");

  unsigned long long *d_bad, *d_tot; uint32_t* d_ex;
  hipMalloc(&d_bad, 8); hipMalloc(&d_tot, 8); hipMalloc(&d_ex, 32);
  for (int mode = 0; mode < 4; ++mode) {
    hipMemset(d_bad, 0, 8); hipMemset(d_tot, 0, 8); hipMemset(d_ex, 0, 32);
    exact_kernel<<<1024, 256>>>(d_bad, d_tot, d_ex, 2000, mode);
    hipDeviceSynchronize();
    unsigned long long bad, tot; uint32_t ex[8];
    hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost); hipMemcpy(&tot, d_tot, 8, hipMemcpyDeviceToHost); hipMemcpy(ex, d_ex, 32, hipMemcpyDeviceToHost);
    const char* names[4] = {"comparable magnitudes (cancellation)", "wide exponent spread", "denormal range", "signed zeros / equal magnitudes"};
    printf("exactness, %-38s: %llu adds, %llu differ from v_add_f32", names[mode], tot, bad);
    if (bad) printf("  e.g. a=%08x c=%08x add=%08x mfma=%08x", ex[1], ex[2], ex[3], ex[4]);
    printf("\n");
  }

  float* d_out; unsigned long long* d_t;
  hipMalloc(&d_out, 256 * 4 * 4 * 4 * 64); hipMalloc(&d_t, 8);
  const int reps = 200;
  auto report = [&](const char* name, int chains, int waves) {
    hipDeviceSynchronize();
    unsigned long long t; hipMemcpy(&t, d_t, 8, hipMemcpyDeviceToHost);
    printf("%-64s %6.2f ticks per step (%d chains x 256 accumulators, %d wave(s) per SIMD)\n", name, (double)t / (reps * 144.0), chains, waves);
  };
  for (int r = 0; r < 2; ++r) {
    chain_kernel<true, 1><<<1, 64>>>(d_out, d_t, reps, nullptr); report("dependent 4x4x1 MFMA chain, A by DPP row broadcast, 1 wave", 1, 1);
    chain_kernel<true, 2><<<1, 64>>>(d_out, d_t, reps, nullptr); report("two interleaved MFMA chains, same A, 1 wave", 2, 1);
    chain_kernel<true, 4><<<1, 64>>>(d_out, d_t, reps, nullptr); report("four interleaved MFMA chains, same A, 1 wave", 4, 1);
    chain_kernel<true, 1><<<1, 256>>>(d_out, d_t, reps, nullptr); report("dependent MFMA chain, 4 waves (one per SIMD)", 1, 1);
    chain_kernel<true, 1><<<1, 512>>>(d_out, d_t, reps, nullptr); report("dependent MFMA chain, 8 waves (two per SIMD)", 1, 2);
    chain_kernel<false, 1><<<1, 64>>>(d_out, d_t, reps, nullptr); report("the same with 4 v_add_f32 per step (VALU), 1 wave", 1, 1);
  }
  // the MFMA chain and the VALU chain give the same bits (whole chains, not single adds)
  {
    float *o1, *o2; hipMalloc(&o1, 64 * 16); hipMalloc(&o2, 64 * 16);
    chain_kernel<true, 1><<<1, 64>>>(o1, d_t, 3, nullptr);
    chain_kernel<false, 1><<<1, 64>>>(o2, d_t, 3, nullptr);
    hipDeviceSynchronize();
    uint32_t h1[256], h2[256];
    hipMemcpy(h1, o1, sizeof h1, hipMemcpyDeviceToHost); hipMemcpy(h2, o2, sizeof h2, hipMemcpyDeviceToHost);
    // VALU lane l register v adds P of ITS OWN row broadcast; MFMA D[v] of lane l gets A of lane 4 * (l / 4) + v, the same row's broadcast: equal
    int diff = 0;
    for (int i = 0; i < 256; ++i) diff += h1[i] != h2[i];
    printf("432-step chains, MFMA vs VALU: %d of 256 accumulators differ\n", diff);
  }
  return 0;
}
