// Microbenchmark: the mixing network's per-bit exchange as an ALL-GATHER of 8-byte value|tag words between 27 workgroups of one launch
// (26 helper sums -> every workgroup), by placement (spread over the XCDs as dispatched / all on one XCD), store flavour (agent-scope
// write-through `sc1` / plain store that stays in the XCD's L2), word layout (contiguous / one 128-byte line per word) and poll style.
// A same-XCD plain store + `sc1` (L1-bypassing, L2-served) load is coherent ONLY inside one XCD: the value|tag protocol cannot return a
// wrong value, a misplaced workgroup shows as a time-out (reported), never as stale data.
//
//   hipcc --offload-arch=gfx950 -O3 -o allgather allgather.hip && ./allgather
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

__device__ __forceinline__ unsigned xcc_id() {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  return v & 0xf;
}
template <int ST> __device__ __forceinline__ void st_word(unsigned long long* p, unsigned long long v) {
  if (ST == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                 // global_store_dwordx2 ... sc1
  else if (ST == 1) asm volatile("global_store_dwordx2 %0, %1, off" :: "v"(p), "v"(v) : "memory");   // plain: the line stays in this XCD's L2
  else asm volatile("global_store_dwordx2 %0, %1, off sc0" :: "v"(p), "v"(v) : "memory");
}
__device__ __forceinline__ unsigned long long ld_word(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);                            // global_load_dwordx2 ... sc1: bypasses L1, served by L2
}

// participants: blocks first, first + stride, ... (P of them); word k of round r at words[((r & 1) * 32 + k) * pitch] -- two slots by parity: a participant
// can be one round ahead of the slowest reader of its previous word, never two (it needs that reader's word of the round in between)
template <int ST> __global__ void allgather_kernel(unsigned long long* words, int pitch, int first, int stride, int P, int rounds, int sleepy,
                                                   long long* out, unsigned* xcc, int work) {
  const int b = blockIdx.x;
  if (b < first || (b - first) % stride != 0 || (b - first) / stride >= P) return;
  const int k = (b - first) / stride, lane = threadIdx.x;
  if (lane == 0) xcc[k] = xcc_id();
  float acc = (float)lane;
  const long long t0 = wall_clock64();
  unsigned fail = 0;
  for (int r = 1; r <= rounds; ++r) {
    for (int i = 0; i < work; ++i) acc = __fadd_rn(acc, 1.25f);   // the helper's own work between two exchanges (dependent adds)
    if (lane == 0) st_word<ST>(words + (size_t)((r & 1) * 32 + k) * pitch, ((unsigned long long)(unsigned)r << 32) | (unsigned)(k + r));
    bool have = lane >= P;
    unsigned spins = 0;
    for (;;) {
      if (!have) {
        const unsigned long long v = ld_word(words + (size_t)((r & 1) * 32 + lane) * pitch);
        if ((unsigned)(v >> 32) == (unsigned)r) { have = true; if ((unsigned)v != (unsigned)(lane + r)) fail = 2; }
      }
      if (__ballot(!have) == 0) break;
      if (sleepy) __builtin_amdgcn_s_sleep(1);
      if (++spins > (1u << 18)) { fail = 1; break; }
    }
    if (__ballot(fail != 0)) break;
  }
  const long long t1 = wall_clock64();
  if (k == 0 && lane == 0) { out[0] = t1 - t0; out[1] = (long long)__ballot(fail != 0); }
  if (acc == 12345.678f) out[3] = 1;
  if (__ballot(fail != 0) && lane == 0) atomicMax((unsigned long long*)&out[2], 1ull);
}

template <int ST> static void run(const char* name, int pitch, int first, int stride, int sleepy, int work) {
  const int P = 27, rounds = 4000, grid = 27 * 8;
  unsigned long long* words; long long* d_out; unsigned* d_xcc;
  hipMalloc(&words, 64 * 128 * 8); hipMalloc(&d_out, 64); hipMalloc(&d_xcc, 32 * 4);
  hipMemset(words, 0, 64 * 128 * 8); hipMemset(d_out, 0, 64); hipMemset(d_xcc, 0xff, 32 * 4);
  hipLaunchKernelGGL((allgather_kernel<ST>), dim3(grid), dim3(64), 0, 0, words, pitch, first, stride, P, rounds, sleepy, d_out, d_xcc, work);
  hipDeviceSynchronize();
  long long h[4]; unsigned x[32];
  hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(x, d_xcc, sizeof x, hipMemcpyDeviceToHost);
  int nx = 0; unsigned seen = 0;
  for (int i = 0; i < P; ++i) if (!(seen >> x[i] & 1)) { seen |= 1u << x[i]; ++nx; }
  printf("%-74s %7.3f us per round  (XCDs used %d%s)\n", name, h[0] / 100.0 / rounds, nx, h[2] ? "; TIME-OUT / wrong value: not coherent" : "");
  hipFree(words); hipFree(d_out); hipFree(d_xcc);
}

int main() {
  for (int rep = 0; rep < 2; ++rep) {
    run<0>("spread (blocks 0..26), sc1 store, contiguous words", 1, 0, 1, 0, 0);
    run<0>("spread, sc1 store, one 128-byte line per word", 16, 0, 1, 0, 0);
    run<0>("spread, sc1 store, line per word, s_sleep 1 in the poll", 16, 0, 1, 1, 0);
    run<0>("one XCD (blocks 0, 8, .., 208), sc1 store, contiguous", 1, 0, 8, 0, 0);
    run<0>("one XCD, sc1 store, line per word", 16, 0, 8, 0, 0);
    run<1>("one XCD, PLAIN store + sc1 load, contiguous", 1, 0, 8, 0, 0);
    run<1>("one XCD, PLAIN store + sc1 load, line per word", 16, 0, 8, 0, 0);
    run<1>("one XCD, PLAIN store + sc1 load, line per word, s_sleep 1", 16, 0, 8, 1, 0);
    run<2>("one XCD, sc0 store + sc1 load, line per word", 16, 0, 8, 0, 0);
    run<1>("spread, PLAIN store + sc1 load (must time out: L2s are not coherent)", 16, 0, 1, 0, 0);
    run<0>("spread, sc1 store, line per word, 1000 dependent adds between exchanges", 16, 0, 1, 0, 1000);
    run<1>("one XCD, PLAIN store, line per word, 1000 dependent adds between exchanges", 16, 0, 8, 0, 1000);
  }
  return 0;
}
