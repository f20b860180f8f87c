// xcd_flag_data.hip -- the mixing network's in-launch hand-off of a bit's stretched inputs, alone: is "agent-scope stores, s_waitcnt vmcnt(0), then an agent-scope
// epoch word" enough for a workgroup on ANOTHER XCD that has seen the epoch to read the new row?   
//
// cmx_mixnet_spec_kernel: four stretch waves of the main workgroup store the 2078 inputs of bits t, t + 1, .. into a ring of rows in global memory (sc1 stores),
// each waits for ITS stores (vmcnt(0)) and raises an LDS word; the select wave then stores the epoch t + 1 (sc1); 26 helper workgroups on other XCDs poll the epoch
// with sc1 loads and read their slice of the row with sc1 loads. If the acknowledgement of an sc1 store could come back before the write is visible to the other
// XCDs' sc1 loads, a helper would -- rarely -- add up a value of the row's previous use: other final probabilities from identical columns, which is what two digest
// runs showed. This program runs exactly that protocol, as fast as it goes, with and without a kernel that saturates HBM, and counts stale words.
//   hipcc --offload-arch=gfx950 -O3 -o xcd_flag_data xcd_flag_data.hip && ./xcd_flag_data [seconds] [load 0|1]
// STATE: written at the very end of round 6 and NOT yet run to a result (the round's last 45 GPU-seconds ended inside its first run, no output): the first thing to
// run next. A meaningful run is long: the engine hands over 8 rows per stream byte, so the two sightings in ~0.2 GB are ~1 event per 10^9 rows; at a few us per row this
// program needs hours (or a higher rate under its harsher conditions) to say anything, and only a non-zero count is conclusive.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <chrono>

#define NIN 2078
#define ROWF 2112
#define RING 4
#define NC 26
struct Xfer {
  unsigned epoch, fail, stop, pad[29];
  unsigned long long ack[32];          // consumer c has read round ack[c] - 1
  unsigned long long rounds, stale;    // results
  unsigned first[8];
  float xs[RING][ROWF];
};
__device__ __forceinline__ unsigned ldu(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ unsigned long long ldq(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float val(unsigned t, int i) { return __uint_as_float(0x3f000000u | ((t * 2654435761u + (unsigned)i * 40503u) & 0x7fffffu)); }

__global__ __launch_bounds__(320) void handoff(Xfer* X) {
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  __shared__ int sdone[8], got[4];
  if (tid < 8) sdone[tid] = 0;
  if (tid < 4) got[tid] = 0;
  __syncthreads();
  if (blockIdx.x == 0) {
    if (wave == 0) {   // the select wave: publishes the epochs in order
      for (unsigned t = 0;; ++t) {
        unsigned spins = 0;
        while (__hip_atomic_load(&sdone[t & 7], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (int)(t + 1)) {
          if ((++spins & 4095u) == 0 && (ldu(&X->stop) || ldu(&X->fail) || spins > (1u << 28))) { if (lane == 0) { X->rounds = t; __hip_atomic_store(&X->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } return; }
        }
        if (lane == 0) __hip_atomic_store(&X->epoch, t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    } else {           // a stretch wave: rounds wave - 1, wave + 3, ..
      for (unsigned t = (unsigned)wave - 1;; t += 4) {
        if (t >= RING) {   // the ring slot is free when every consumer has read round t - RING
          unsigned spins = 0;
          for (;;) {
            const unsigned long long a = lane < NC ? ldq(&X->ack[lane]) : ~0ull;
            if (__ballot(a < (unsigned long long)(t - RING + 1)) == 0) break;
            if ((++spins & 1023u) == 0 && (ldu(&X->stop) || ldu(&X->fail) || spins > (1u << 26))) return;
          }
        }
        float* row = X->xs[t % RING];
#pragma unroll
        for (int r = 0; r < 33; ++r) { const int i = r * 64 + lane; if (i < NIN) __hip_atomic_store(row + i, val(t, i), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // this wave's stores are acknowledged
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(&sdone[t & 7], (int)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
    }
  } else if (wave < 4) {   // a helper workgroup: four waves, each its 512-term slice (+ the tail in the last)
    const int c = blockIdx.x - 1, base = wave * 512;
    unsigned long long bad = 0;
    for (unsigned t = 0;; ++t) {
      unsigned spins = 0;
      while (ldu(&X->epoch) < t + 1)
        if ((++spins & 4095u) == 0 && (ldu(&X->stop) || ldu(&X->fail) || spins > (1u << 28))) { if (bad) atomicAdd(&X->stale, bad); return; }
      const float* row = X->xs[t % RING];
      unsigned b = 0;
#pragma unroll
      for (int k = 0; k < 9; ++k) {
        const int i = base + 64 * k + lane;
        if (k < 8 || (wave == 3 && i < NIN)) {
          const float v = __hip_atomic_load(row + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (v != val(t, i)) { ++b; if (atomicCAS(&X->first[0], 0u, 1u) == 0u) { X->first[1] = t; X->first[2] = (unsigned)i; X->first[3] = __float_as_uint(v); X->first[4] = (unsigned)c; } }
        }
      }
      bad += b;
      // the workgroup's four waves have all read the round: wave 0 acknowledges
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (lane == 0) __hip_atomic_store(&got[wave], (int)(t + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (wave == 0) {
        unsigned sp = 0;
        for (;;) {
          const int g = lane < 4 ? __hip_atomic_load(&got[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) : 0x7fffffff;
          if (__ballot(g < (int)(t + 1)) == 0) break;
          if ((++sp & 4095u) == 0 && (ldu(&X->stop) || ldu(&X->fail))) break;
        }
        if (lane == 0) __hip_atomic_store(&X->ack[c], (unsigned long long)t + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
  }
}
__global__ __launch_bounds__(256) void churn(float4* dst, const float4* src, size_t n) {
  for (size_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) dst[k] = src[k];
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 20.0;
  const int load = argc > 2 ? atoi(argv[2]) : 1;
  Xfer* X; hipMalloc((void**)&X, sizeof(Xfer)); hipMemset(X, 0, sizeof(Xfer));
  unsigned* hstop; hipHostMalloc((void**)&hstop, 64, hipHostMallocDefault);
  hipStream_t K, L; hipStreamCreateWithFlags(&K, hipStreamNonBlocking); hipStreamCreateWithFlags(&L, hipStreamNonBlocking);
  float4 *ld = nullptr, *ls = nullptr; const size_t ln = (size_t)16 << 20;
  if (load) { hipMalloc((void**)&ld, ln * 16); hipMalloc((void**)&ls, ln * 16); hipMemset(ls, 1, ln * 16); }
  hipDeviceSynchronize();
  hipLaunchKernelGGL(handoff, dim3(1 + NC), dim3(320), 0, K, X);
  const auto t0 = std::chrono::steady_clock::now();
  long nload = 0;
  hipEvent_t e[2]; hipEventCreate(&e[0]); hipEventCreate(&e[1]);
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    if (load) {
      if (nload >= 2) hipEventSynchronize(e[nload & 1]);
      hipLaunchKernelGGL(churn, dim3(1024), dim3(256), 0, L, ld, ls, ln);   // (1024 workgroups: the 27 of the hand-off keep their compute units)
      hipEventRecord(e[nload & 1], L);
      ++nload;
    } else {
      struct timespec ts = {0, 2000000}; nanosleep(&ts, nullptr);
    }
  }
  const unsigned one = 1;
  hipMemcpyAsync(&X->stop, &one, 4, hipMemcpyHostToDevice, L);   // the kernel's roles poll it
  hipStreamSynchronize(L);
  hipStreamSynchronize(K);
  Xfer h; hipMemcpy(&h, X, sizeof(unsigned) * 32 + 8 * 32 + 16 + 32, hipMemcpyDeviceToHost);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  unsigned long long rounds = h.rounds ? h.rounds : h.epoch;
  printf("%llu rows of %d inputs handed from one workgroup to %d on other XCDs in %.1f s (%.2f us per row), %s: %llu stale words%s",
         rounds, NIN, NC, dt, dt * 1e6 / (rounds ? rounds : 1), load ? "HBM saturated by copy kernels" : "idle device", h.stale, h.fail && !h.rounds ? " (a wait ran out)" : "");
  if (h.stale) printf("; first: row %u, input %u, helper %u, value %08x (this row's %08x, the slot's previous use %08x)", h.first[1], h.first[2], h.first[4], h.first[3],
                      0x3f000000u | ((h.first[1] * 2654435761u + h.first[2] * 40503u) & 0x7fffffu), 0x3f000000u | (((h.first[1] - RING) * 2654435761u + h.first[2] * 40503u) & 0x7fffffu));
  printf("\n");
  return 0;
}
