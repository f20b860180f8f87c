// sgpr_chain.hip -- can the mixing network's ordered add chain be fed WITHOUT the LDS pipe? (DESIGN.md 4.1's open item, round-5 review item 6)
//
// Today a helper's four chain waves read their segment's 520 products back from LDS as broadcast ds_read_b128 (130 per wave): a wave-wide read occupies the compute
// unit's one LDS return path for ~8 clocks whatever its width, so four waves need ~4.2 k clocks for what the adder could do in 2.3 k (520 x 4.5). The alternative
// measured here: the products are stored to global memory (they reach L2), the chain wave invalidates the scalar cache and reads them back with s_load_dwordx16 into
// SGPRs; `v_add_f32 v, s, v` takes the operand straight from the SGPR -- one VALU issue per term, no LDS, no cross-lane move.
//   variant L: LDS-fed chain (the product kernel's chain_seg_n), 512 terms per wave
//   variant S<D>: SGPR-fed chain, D s_load_dwordx16 in flight ahead of the adds (D x 16 SGPRs), 512 terms per wave
// each with 1 and with 4 chain waves on the compute unit, per round: store products (as the helper does) -> [variant's path] -> chain. Reports shader clocks per round.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o sgpr_chain sgpr_chain.hip && ./sgpr_chain
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define SEGF 544
typedef float f16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float add4(float p, float4 v) { p = __fadd_rn(p, v.x); p = __fadd_rn(p, v.y); p = __fadd_rn(p, v.z); p = __fadd_rn(p, v.w); return p; }
__device__ __forceinline__ float chain_lds(const float* rowp, float p) {   // chain_seg_n<16> of mixnet_chunk.hip, 512 terms
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
#define SLOAD(dst, base, off) asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(dst) : "s"(base), "i"(off) : "memory")
#define SWAIT() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
__device__ __forceinline__ float add16(float p, f16 x) {
#pragma unroll
  for (int i = 0; i < 16; ++i) p = __fadd_rn(p, x[i]);
  return p;
}
// 512 terms = 32 loads of 16; D loads are issued ahead. SMEM loads may return out of order: the only safe wait is lgkmcnt(0), so the loop works in groups of D.
template <int D> __device__ __forceinline__ float chain_sgpr(const float* gp, float p) {
  const unsigned long long gpv = (unsigned long long)gp;
  const unsigned long long base = ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(gpv >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)gpv);   // into an SGPR pair
  f16 a[D], b[D];
#pragma unroll
  for (int i = 0; i < D; ++i) SLOAD(a[i], base, i * 64);
  SWAIT();
#pragma unroll 1
  for (int g = 0; g < 32 / D; g += 2) {
    const unsigned long long nb = base + (unsigned long long)(g + 1) * D * 64;
#pragma unroll
    for (int i = 0; i < D; ++i) SLOAD(b[i], nb, i * 64);            // next group in flight under this group's adds
#pragma unroll
    for (int i = 0; i < D; ++i) p = add16(p, a[i]);
    SWAIT();
    const unsigned long long nb2 = base + (unsigned long long)(g + 2) * D * 64;   // (the last iteration reads one group past the segment: inside the padded buffer)
#pragma unroll
    for (int i = 0; i < D; ++i) SLOAD(a[i], nb2, i * 64);
#pragma unroll
    for (int i = 0; i < D; ++i) p = add16(p, b[i]);
    SWAIT();
  }
  return p;
}

// mode 0: LDS-fed; 1: SGPR-fed with D in flight. nw chain waves (1 or 4) of the 256-thread workgroup take part; out[w] = result, clk[w] = clocks of R rounds
template <int D> __global__ __launch_bounds__(256) void bench(int mode, int nw, int R, float* gbuf, const float* x, const float* wts, float* out, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) float prod[4][SEGF + 64];
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  if (w >= nw) return;
  float* gp = gbuf + (size_t)w * (SEGF + 64 + 512);
  float xs[8], ws[8];
  for (int k = 0; k < 8; ++k) { xs[k] = x[w * 512 + 64 * k + lane]; ws[k] = wts[w * 512 + 64 * k + lane]; }
  float acc = 0.0f;
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int r = 0; r < R; ++r) {
    // the products of this "bit" (the weights move a little every round, as after an update)
    for (int k = 0; k < 8; ++k) { ws[k] = __fadd_rn(ws[k], 1e-7f * (float)(r & 3)); }
    if (mode == 0) {
      for (int k = 0; k < 8; ++k) prod[w][64 * k + lane] = __fmul_rn(xs[k], ws[k]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_wave_barrier();
      const float start = __int_as_float(__float_as_int(acc * 1e-3f) + lane);   // 64 candidate starts, as the speculative waves
      acc = chain_lds(prod[w], start);
    } else {
      for (int k = 0; k < 8; ++k) gp[64 * k + lane] = __fmul_rn(xs[k], ws[k]);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the stores have reached L2
      asm volatile("s_dcache_inv\n\ts_waitcnt lgkmcnt(0)" ::: "memory");   // the scalar cache holds the previous round's lines
      __builtin_amdgcn_wave_barrier();
      const float start = __int_as_float(__float_as_int(acc * 1e-3f) + lane);
      acc = chain_sgpr<D>(gp, start);
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (lane == 32) { out[w] = acc; clk[w] = t1 - t0; }
}

int main() {
  const int R = 2000;
  float *gbuf, *x, *wts, *out; unsigned long long* clk;
  hipMalloc((void**)&gbuf, 4 * (SEGF + 64 + 512) * 4 + 4096); hipMemset(gbuf, 0, 4 * (SEGF + 64 + 512) * 4 + 4096);
  hipMalloc((void**)&x, 2048 * 4); hipMalloc((void**)&wts, 2048 * 4); hipMalloc((void**)&out, 16); hipMalloc((void**)&clk, 32);
  std::vector<float> hx(2048), hw(2048);
  srand(5);
  for (int i = 0; i < 2048; ++i) { hx[i] = (rand() % 2001 - 1000) * 1e-3f; hw[i] = (rand() % 2001 - 1000) * 1e-4f; }
  hipMemcpy(x, hx.data(), 8192, hipMemcpyHostToDevice); hipMemcpy(wts, hw.data(), 8192, hipMemcpyHostToDevice);
  float ref[4] = {0, 0, 0, 0};
  auto run = [&](const char* name, int mode, int nw, int d) {
    hipMemset(out, 0, 16); hipMemset(clk, 0, 32);
    if (d == 2) hipLaunchKernelGGL(bench<2>, dim3(1), dim3(256), 0, 0, mode, nw, R, gbuf, x, wts, out, clk);
    else if (d == 4) hipLaunchKernelGGL(bench<4>, dim3(1), dim3(256), 0, 0, mode, nw, R, gbuf, x, wts, out, clk);
    else hipLaunchKernelGGL(bench<1>, dim3(1), dim3(256), 0, 0, mode, nw, R, gbuf, x, wts, out, clk);
    hipDeviceSynchronize();
    float ho[4]; unsigned long long hc[4];
    hipMemcpy(ho, out, 16, hipMemcpyDeviceToHost); hipMemcpy(hc, clk, 32, hipMemcpyDeviceToHost);
    unsigned long long mx = 0; for (int w = 0; w < nw; ++w) if (hc[w] > mx) mx = hc[w];
    bool same = true;
    if (mode == 0 && nw == 4) for (int w = 0; w < 4; ++w) ref[w] = ho[w];
    else for (int w = 0; w < nw; ++w) same = same && (ref[w] == 0.0f || ho[w] == ref[w]);
    printf("%-44s %d wave(s): %8.0f clocks per round (512-term chain + products%s)  result %s\n", name, nw, (double)mx / R, mode ? " + store + s_dcache_inv" : "", same ? "== LDS-fed" : "DIFFERS");
  };
  run("L  LDS-fed (ds_read_b128 broadcast)", 0, 4, 1);
  run("L  LDS-fed (ds_read_b128 broadcast)", 0, 1, 1);
  for (int d : {1, 2, 4}) {
    char nm[64]; snprintf(nm, sizeof nm, "S%d SGPR-fed, %d x s_load_dwordx16 in flight", d, d);
    run(nm, 1, 1, d);
    run(nm, 1, 4, d);
  }
  return 0;
}
