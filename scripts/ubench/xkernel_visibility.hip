// xkernel_visibility.hip -- does a kernel ALWAYS see what an event-ordered kernel (or upload) on another stream wrote, under load?  
//
// The engine's mixing network reads, with plain loads, slot buffers that other stages' kernels (on other streams, other XCDs) and host-to-device copies filled a moment
// before; the only ordering is hipEventRecord on the producer's stream + hipStreamWaitEvent on the consumer's, and the buffers are recycled every 8 chunks. If a
// consumer could -- rarely -- read a line of the PREVIOUS use of a buffer (a write-back of the producer's L2 not finished, an invalidate of the consumer's L2
// skipped), the engine would compute other final probabilities from columns that every later read of memory shows correct: what was observed twice in ~0.2 GB.
// This program models that hand-off in isolation, as often as the engine does it in a few hundred MB:
//   per iteration i: 2 x P producer workgroups (streams A, A2; they share every line) fill a buffer with i (plain stores); a pinned host buffer with i goes up on stream U;
//                    C consumer workgroups (stream B, after both events) read both with plain loads and count every word != i;
//                    A and U wait for the consumer before iteration i + 1 reuses the buffers (the engine: ev_mix1).
//   meanwhile, on stream L: copy kernels that keep HBM and all compute units busy (the digest load of scripts/gpu_stage_hashes.py).
//   hipcc --offload-arch=gfx950 -O3 -o xkernel_visibility xkernel_visibility.hip && ./xkernel_visibility [iterations] [words] [load 0|1]
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <chrono>

#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// two producers on two streams share every 128-byte line of the buffer (16 words each, alternating): the layer-0 row's column ranges of different stages meet inside lines
__global__ __launch_bounds__(256) void produce(uint32_t* buf, size_t words, uint32_t it, int phase) {
  const size_t per = (words + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < words ? lo + per : words;
  for (size_t k = lo + threadIdx.x; k < hi; k += 256)
    if ((int)((k >> 4) & 1) == phase) buf[k] = it + (uint32_t)k;   // plain stores: dirty bytes in this XCD's L2 until the kernel's release
}
__global__ __launch_bounds__(256) void consume(const uint32_t* __restrict__ buf, size_t words, const uint32_t* __restrict__ up, size_t upwords, uint32_t it,
                                               unsigned long long* bad, uint32_t* first) {
  const size_t per = (words + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * per, hi = lo + per < words ? lo + per : words;
  unsigned n = 0;
  for (size_t k = lo + threadIdx.x; k < hi; k += 256) {
    const uint32_t v = buf[k];
    if (v != it + (uint32_t)k) { if (!n && atomicCAS(&first[0], 0u, 1u) == 0u) { first[1] = it; first[2] = (uint32_t)k; first[3] = v; first[4] = 0; } ++n; }
  }
  for (size_t k = blockIdx.x * 256 + threadIdx.x; k < upwords; k += (size_t)gridDim.x * 256) {
    const uint32_t v = up[k];
    if (v != it * 3u + (uint32_t)k) { if (!n && atomicCAS(&first[0], 0u, 1u) == 0u) { first[1] = it; first[2] = (uint32_t)k; first[3] = v; first[4] = 1; } ++n; }
  }
  if (n) atomicAdd(bad, (unsigned long long)n);
}
__global__ __launch_bounds__(256) void churn(float4* dst, const float4* src, size_t n) {
  for (size_t k = blockIdx.x * 256 + threadIdx.x; k < n; k += (size_t)gridDim.x * 256) dst[k] = src[k];
}

int main(int argc, char** argv) {
  const long iters = argc > 1 ? atol(argv[1]) : 200000;
  const size_t words = argc > 2 ? (size_t)atol(argv[2]) : (1u << 20);   // 4 MB per hand-off by default
  const int load = argc > 3 ? atoi(argv[3]) : 1;
  const size_t upwords = 8192;                                          // 32 KB: the coded bits of a 4 KB chunk
  const int P = 4, C = 27;                                              // paq8's mixer writes from 4 workgroups; the network's 27 read
  hipStream_t A, A2, B, U, L;
  OK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&A2, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  OK(hipStreamCreateWithFlags(&U, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&L, hipStreamNonBlocking));
  const int SLOTS = 8;
  uint32_t *buf[SLOTS], *up[SLOTS], *hup[SLOTS], *first; unsigned long long* bad;
  hipEvent_t eP[SLOTS], eP2[SLOTS], eU[SLOTS], eC[SLOTS];
  for (int s = 0; s < SLOTS; ++s) {
    OK(hipMalloc((void**)&buf[s], words * 4)); OK(hipMemset(buf[s], 0xff, words * 4));
    OK(hipMalloc((void**)&up[s], upwords * 4)); OK(hipHostMalloc((void**)&hup[s], upwords * 4, hipHostMallocDefault));
    OK(hipEventCreateWithFlags(&eP[s], hipEventDisableTiming)); OK(hipEventCreateWithFlags(&eP2[s], hipEventDisableTiming)); OK(hipEventCreateWithFlags(&eU[s], hipEventDisableTiming)); OK(hipEventCreateWithFlags(&eC[s], hipEventDisableTiming));
  }
  OK(hipMalloc((void**)&bad, 8)); OK(hipMemset(bad, 0, 8)); OK(hipMalloc((void**)&first, 32)); OK(hipMemset(first, 0, 32));
  float4 *ld = nullptr, *ls = nullptr; const size_t ln = (size_t)16 << 20;   // 256 MB each
  hipEvent_t eL[2];
  if (load) { OK(hipMalloc((void**)&ld, ln * 16)); OK(hipMalloc((void**)&ls, ln * 16)); OK(hipMemset(ls, 1, ln * 16)); OK(hipEventCreate(&eL[0])); OK(hipEventCreate(&eL[1])); }
  OK(hipDeviceSynchronize());
  const auto t0 = std::chrono::steady_clock::now();
  long nload = 0;
  for (long i = 1; i <= iters; ++i) {
    const int s = (int)(i % SLOTS);
    if (i > SLOTS) OK(hipEventSynchronize(eC[s]));                    // the consumer that last read this slot (the engine's begin() waits for ev_mix1 on the host)
    for (size_t k = 0; k < upwords; ++k) hup[s][k] = (uint32_t)i * 3u + (uint32_t)k;
    OK(hipMemcpyAsync(up[s], hup[s], upwords * 4, hipMemcpyHostToDevice, U));
    OK(hipEventRecord(eU[s], U));
    hipLaunchKernelGGL(produce, dim3(P), dim3(256), 0, A, buf[s], words, (uint32_t)i, 0);
    OK(hipEventRecord(eP[s], A));
    hipLaunchKernelGGL(produce, dim3(P), dim3(256), 0, A2, buf[s], words, (uint32_t)i, 1);
    OK(hipEventRecord(eP2[s], A2));
    OK(hipStreamWaitEvent(B, eP[s], 0)); OK(hipStreamWaitEvent(B, eP2[s], 0)); OK(hipStreamWaitEvent(B, eU[s], 0));
    hipLaunchKernelGGL(consume, dim3(C), dim3(256), 0, B, buf[s], words, up[s], upwords, (uint32_t)i, bad, first);
    OK(hipEventRecord(eC[s], B));
    if (load && (i % 4) == 0) {                                       // keep two 512 MB copy kernels in flight on the load stream
      if (nload >= 2) OK(hipEventSynchronize(eL[nload & 1]));
      hipLaunchKernelGGL(churn, dim3(2048), dim3(256), 0, L, ld, ls, ln);
      OK(hipEventRecord(eL[nload & 1], L));
      ++nload;
    }
  }
  OK(hipDeviceSynchronize());
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  unsigned long long hb = 0; uint32_t hf[8] = {0};
  OK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost)); OK(hipMemcpy(hf, first, 32, hipMemcpyDeviceToHost));
  printf("%ld hand-offs of %zu KB (kernel -> kernel, %d -> %d workgroups) + %zu KB (upload -> kernel), %s load: %.1f s (%.1f us each), %llu stale words",
         iters, words * 4 / 1024, P, C, upwords * 4 / 1024, load ? "copy kernels as" : "no", dt, dt * 1e6 / iters, hb);
  if (hb) printf("; first: iteration %u, word %u of the %s buffer, value %u (expected %u)", hf[1], hf[2], hf[4] ? "uploaded" : "kernel-written", hf[3], hf[4] ? hf[1] * 3u + hf[2] : hf[1] + hf[2]);
  printf("\n");
  return 0;
}
