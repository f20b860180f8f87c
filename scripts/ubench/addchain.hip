// Microbenchmark: latency of a dependent v_add_f32 chain on one wave (the floor of the strict
// mixer dot product), and the ratio of s_memtime ticks to the 100 MHz wall clock.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int ACTIVE>
__global__ void chain(float* out, uint64_t* t, float x) {
  float p = threadIdx.x;
  uint64_t w0 = wall_clock64();
  uint64_t c0 = __builtin_readcyclecounter();
  if ((int)threadIdx.x < ACTIVE) {
#pragma unroll 1
    for (int it = 0; it < 1024; ++it) {
#pragma unroll
      for (int k = 0; k < 64; ++k) p = __fadd_rn(p, x);
    }
  }
  uint64_t c1 = __builtin_readcyclecounter();
  uint64_t w1 = wall_clock64();
  out[threadIdx.x] = p;
  if (threadIdx.x == 0) { t[0] = c1 - c0; t[1] = w1 - w0; }
}

__global__ void indep(float* out, uint64_t* t, float x) {
  float p0 = threadIdx.x, p1 = 1, p2 = 2, p3 = 3, p4 = 4, p5 = 5, p6 = 6, p7 = 7;
  uint64_t c0 = __builtin_readcyclecounter();
#pragma unroll 1
  for (int it = 0; it < 1024; ++it) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      p0 = __fadd_rn(p0, x); p1 = __fadd_rn(p1, x); p2 = __fadd_rn(p2, x); p3 = __fadd_rn(p3, x);
      p4 = __fadd_rn(p4, x); p5 = __fadd_rn(p5, x); p6 = __fadd_rn(p6, x); p7 = __fadd_rn(p7, x);
    }
  }
  uint64_t c1 = __builtin_readcyclecounter();
  out[threadIdx.x] = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
  if (threadIdx.x == 0) t[0] = c1 - c0;
}

int main() {
  float* out; uint64_t* t;
  hipMalloc(&out, 4096); hipMalloc(&t, 64);
  uint64_t h[2];
  for (int rep = 0; rep < 3; ++rep) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); chain<64><<<1, 64>>>(out, t, 1.0f); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("dep chain 64 lanes : %.2f ticks/add, wall100MHz %.2f ns/add, event %.2f ns/add -> memtime freq %.3f GHz\n",
           h[0] / 65536.0, h[1] * 10.0 / 65536.0, ms * 1e6 / 65536.0, h[0] / (h[1] * 10.0));
    chain<26><<<1, 64>>>(out, t, 1.0f); hipDeviceSynchronize(); hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("dep chain 26 lanes : %.2f ticks/add, %.2f ns/add\n", h[0] / 65536.0, h[1] * 10.0 / 65536.0);
    chain<32><<<1, 32>>>(out, t, 1.0f); hipDeviceSynchronize(); hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("dep chain 32-thread block: %.2f ticks/add, %.2f ns/add\n", h[0] / 65536.0, h[1] * 10.0 / 65536.0);
    indep<<<1, 64>>>(out, t, 1.0f); hipDeviceSynchronize(); hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("8 independent chains: %.2f ticks/add\n", h[0] / 65536.0);
    // same with the whole chip busy (clock behaviour under load)
    chain<64><<<1024, 256>>>(out, t, 1.0f); hipDeviceSynchronize(); hipMemcpy(h, t, 16, hipMemcpyDeviceToHost);
    printf("dep chain, 1024x256 grid: %.2f ticks/add, %.2f ns/add\n", h[0] / 65536.0, h[1] * 10.0 / 65536.0);
  }
  return 0;
}
