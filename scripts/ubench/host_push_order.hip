// host_push_order.hip -- when the decoder thread PUSHES a step (bit, records, then the step counter) into uncached device memory through the PCIe BAR, can a kernel
// that has seen the counter still read the PREVIOUS contents of a record?   (DESIGN.md 4.10 round 6 "host push"; the one wrong round trip of profiles/r06_pytest_gpu.txt)
//
// PCIe keeps posted writes in order up to the device; behind it the records and the counter live in different memory channels. cmx_late.h's consumers poll the
// counter and then read the step's records at once. This program does that tens of millions of times: per round the host stores one byte into a bit array, a 64-byte
// row and a 512-byte row (three separate uncached allocations, rows walking through 1 MB each so that they change channel every round), sfence, then the counter in a
// fourth allocation, sfence; the kernel polls the counter without sleeping, reads the byte and both rows with system-scope loads, counts every word that is not this
// round's, and acknowledges in host memory. `pollers` extra workgroups poll the same counter all the time, as the decoder's ~60 stage workgroups do.
//   hipcc --offload-arch=gfx950 -O2 -o host_push_order host_push_order.hip && ./host_push_order [rounds] [pollers]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <immintrin.h>

static double now_s() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + t.tv_nsec * 1e-9; }
#define ROWS 2048   // a row ring of 2048 entries: 128 KB of 64-byte rows, 1 MB of 512-byte rows

__global__ __launch_bounds__(64) void consumer(const uint32_t* cnt, const uint8_t* bits, const uint32_t* r64, const uint32_t* r512, uint32_t* ack, long n, unsigned long long* stale, uint32_t* first) {
  const int lane = threadIdx.x;
  if (blockIdx.x > 0) {   // pollers: hammer the counter's line until the last round
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (uint32_t)n) __builtin_amdgcn_s_sleep(2);
    return;
  }
  unsigned long long bad = 0;
  for (long i = 1; i <= n; ++i) {
    unsigned spins = 0;
    while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < (uint32_t)i)
      if (++spins > 400000000u) { if (lane == 0) __hip_atomic_store(ack, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
    const int slot = (int)(i % ROWS);
    const uint32_t want = (uint32_t)i * 2654435761u;
    unsigned b = 0;
    if (lane < 16) { const uint32_t v = __hip_atomic_load(r64 + slot * 16 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); if (v != want + lane) { ++b; if (atomicCAS(&first[0], 0u, 1u) == 0u) { first[1] = (uint32_t)i; first[2] = 64; first[3] = lane; first[4] = v; } } }
    for (int k = lane; k < 128; k += 64) { const uint32_t v = __hip_atomic_load(r512 + slot * 128 + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); if (v != want + 16 + k) { ++b; if (atomicCAS(&first[0], 0u, 1u) == 0u) { first[1] = (uint32_t)i; first[2] = 512; first[3] = k; first[4] = v; } } }
    if (lane == 0) { const uint8_t v = *(volatile const uint8_t*)(bits + (i % 65536)); if (v != (uint8_t)(want >> 7)) { ++b; if (atomicCAS(&first[0], 0u, 1u) == 0u) { first[1] = (uint32_t)i; first[2] = 1; first[3] = 0; first[4] = v; } } }
    bad += b;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(ack, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (bad) atomicAdd(stale, bad);
}

int main(int argc, char** argv) {
  const long N = argc > 1 ? atol(argv[1]) : 20000000;
  const int pollers = argc > 2 ? atoi(argv[2]) : 60;
  uint32_t *cnt, *r64, *r512, *ack, *first; uint8_t* bits; unsigned long long* stale;
  if (hipExtMallocWithFlags((void**)&cnt, 4096, hipDeviceMallocUncached) || hipExtMallocWithFlags((void**)&bits, 65536, hipDeviceMallocUncached) ||
      hipExtMallocWithFlags((void**)&r64, ROWS * 64, hipDeviceMallocUncached) || hipExtMallocWithFlags((void**)&r512, ROWS * 512, hipDeviceMallocUncached)) { printf("allocation failed\n"); return 1; }
  hipHostMalloc((void**)&ack, 4096, hipHostMallocDefault);
  hipMalloc((void**)&stale, 8); hipMalloc((void**)&first, 32);
  hipMemset(cnt, 0, 4096); hipMemset(bits, 0, 65536); hipMemset(r64, 0, ROWS * 64); hipMemset(r512, 0, ROWS * 512); hipMemset(stale, 0, 8); hipMemset(first, 0, 32);
  *(volatile uint32_t*)ack = 0;
  hipDeviceSynchronize();
  hipLaunchKernelGGL(consumer, dim3(1 + pollers), dim3(64), 0, 0, cnt, bits, r64, r512, ack, N, stale, first);
  const double t0 = now_s();
  long done = N;
  for (long i = 1; i <= N; ++i) {
    const int slot = (int)(i % ROWS);
    const uint32_t want = (uint32_t)i * 2654435761u;
    bits[i % 65536] = (uint8_t)(want >> 7);
    uint32_t row[144];
    for (int k = 0; k < 144; ++k) row[k] = want + k;
    __builtin_memcpy((void*)(r64 + slot * 16), row, 64);
    __builtin_memcpy((void*)(r512 + slot * 128), row + 16, 512);
    _mm_sfence();
    *(volatile uint32_t*)cnt = (uint32_t)i;
    _mm_sfence();
    uint32_t a;
    while ((a = *(volatile uint32_t*)ack) != (uint32_t)i) if (a == 0xFFFFFFFFu) { printf("the kernel gave up at round %ld\n", i); done = i; i = N + 1; break; }
  }
  const double dt = now_s() - t0;
  *(volatile uint32_t*)cnt = (uint32_t)N; _mm_sfence();
  hipDeviceSynchronize();
  unsigned long long st = 0; uint32_t hf[8] = {0};
  hipMemcpy(&st, stale, 8, hipMemcpyDeviceToHost); hipMemcpy(hf, first, 32, hipMemcpyDeviceToHost);
  printf("%ld pushed steps (1 byte + 64 B + 512 B, sfence, counter; %d workgroups polling the counter): %.2f us per round trip, %llu stale words", done, pollers, dt * 1e6 / done, st);
  if (st) printf("; first: round %u, the %u-byte record, word %u, value %08x (this round's %08x, the slot's previous use %08x)", hf[1], hf[2], hf[3], hf[4],
                 hf[1] * 2654435761u + (hf[2] == 512 ? 16 : 0) + hf[3], (hf[1] - ROWS) * 2654435761u + (hf[2] == 512 ? 16 : 0) + hf[3]);
  printf("\n");
  return 0;
}
