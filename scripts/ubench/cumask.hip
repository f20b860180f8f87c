// Which XCC do the workgroups of a stream with a compute-unit mask run on? (hipExtStreamCreateWithCUMask; bit numbering is not documented for the
// eight-XCC parts.) For each candidate mask: 64 workgroups, histogram of HW_REG_XCC_ID.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
__global__ void probe(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) out[blockIdx.x] = id & 15;
  for (volatile int i = 0; i < 20000; ++i) {}
}
static void run(const char* name, const uint32_t m[8]) {
  hipStream_t st;
  if (hipExtStreamCreateWithCUMask(&st, 8, m) != hipSuccess) { printf("%-40s stream creation failed\n", name); return; }
  unsigned* d; hipMalloc(&d, 64 * 4);
  hipLaunchKernelGGL(probe, dim3(64), dim3(64), 0, st, d);
  hipStreamSynchronize(st);
  unsigned h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  int hist[16] = {0};
  for (int i = 0; i < 64; ++i) hist[h[i]]++;
  printf("%-40s XCC histogram:", name);
  for (int i = 0; i < 8; ++i) printf(" %2d", hist[i]);
  printf("   first blocks:");
  for (int i = 0; i < 12; ++i) printf(" %u", h[i]);
  printf("\n");
  hipFree(d); hipStreamDestroy(st);
}
int main() {
  uint32_t m[8];
  for (int x = 0; x < 8; x += 7) {
    for (int w = 0; w < 8; ++w) { m[w] = 0; for (int b = 0; b < 32; ++b) if (((32 * w + b) & 7) == x) m[w] |= 1u << b; }
    char nm[64]; snprintf(nm, sizeof nm, "bits i with i %% 8 == %d", x); run(nm, m);
    for (int w = 0; w < 8; ++w) m[w] = w == x ? 0xffffffffu : 0u;
    snprintf(nm, sizeof nm, "word %d only (bits %d..%d)", x, 32 * x, 32 * x + 31); run(nm, m);
  }
  for (int w = 0; w < 8; ++w) m[w] = 0x0000000fu;
  run("bits 0..3 of every word", m);
  for (int w = 0; w < 8; ++w) m[w] = w < 4 ? 0xffffffffu : 0u;
  run("words 0..3 (bits 0..127)", m);
  for (int w = 0; w < 8; ++w) m[w] = 0xffffffffu;
  run("all bits", m);
  return 0;
}
