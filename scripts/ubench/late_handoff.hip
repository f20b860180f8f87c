// late_handoff.hip -- microbenchmark behind the late-bit protocol's choice of memory (cmix_amd/csrc/cmx_late.h): a producer kernel and a
// consumer kernel, launched on two streams and running at the same time on DIFFERENT XCDs, hand 1 KB rows to each other through a buffer
// of a given kind: producer = plain stores of the row, s_waitcnt vmcnt(0), counter store (system-scope atomic); consumer = poll the
// counter, plain loads of the row, compare. Reports, per kind of memory, the rows that arrived stale and the time per hand-off.
//   hipcc --offload-arch=gfx950 -O3 -o late_handoff late_handoff.hip && ./late_handoff
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int ROWF = 256, RING = 64;

__device__ __forceinline__ uint32_t ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

// mode 0: plain stores / loads; 1: agent-scope atomic stores / loads of the data
__global__ void producer(float* buf, uint32_t* cnt, uint32_t* ack, int n, int xcd, int mode) {
  if ((int)blockIdx.x != xcd) return;
  const int lane = threadIdx.x;
  for (int i = 0; i < n; i++) {
    while (i >= RING && (int)ld(ack) < i - RING + 1) __builtin_amdgcn_s_sleep(1);
    float* row = buf + (size_t)(i % RING) * ROWF;
    for (int k = 0; k < 4; k++) {
      if (mode == 1) __hip_atomic_store(row + 64 * k + lane, (float)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else row[64 * k + lane] = (float)i;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) st(cnt, (uint32_t)(i + 1));
  }
}
__global__ void consumer(const float* buf, const uint32_t* cnt, uint32_t* ack, int n, int xcd, int mode, unsigned long long* out) {
  if ((int)blockIdx.x != xcd) return;
  const int lane = threadIdx.x;
  unsigned long long bad = 0;
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; i++) {
    while ((int)ld(cnt) < i + 1) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    const float* row = buf + (size_t)(i % RING) * ROWF;
    int wrong = 0;
    for (int k = 0; k < 4; k++) {
      const float v = mode == 1 ? __hip_atomic_load(row + 64 * k + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
                      : mode == 2 ? row[64 * k + lane]                                   // an ordinary load (what shared step functions contain)
                                  : *(volatile const float*)(row + 64 * k + lane);      // volatile: the compiler marks it system-coherent

      wrong |= v != (float)i;
    }
    if (__ballot(wrong)) bad++;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) st(ack, (uint32_t)(i + 1));
  }
  if (lane == 0) { out[0] = bad; out[1] = (unsigned long long)(wall_clock64() - t0); }
}

// the host-producer variant's consumer: hand-off i brings 16 floats (64 bytes: HALF a 128-byte line) at row (i / 16) % RING, offset 16 * (i % 16); the
// lanes first touch the whole row (as a kernel walking consecutive records touches the lines around the record it needs), then check the part
__global__ void consumer_part(const float* buf, const uint32_t* cnt, uint32_t* ack, int n, int xcd, int mode, unsigned long long* out) {
  if ((int)blockIdx.x != xcd) return;
  const int lane = threadIdx.x;
  unsigned long long bad = 0;
  const long long t0 = wall_clock64();
  for (int i = 0; i < n; i++) {
    while ((int)ld(cnt) < i + 1) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
    const float* row = buf + (size_t)((i / 8) % RING) * ROWF;
    const int k = 32 * (i % 8) + (lane & 31);
    const float v = mode == 1 ? __hip_atomic_load(row + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : mode == 2 ? row[k] : *(volatile const float*)(row + k);
    // touch the next part's line too (not yet written by the host): a later ordinary load of it may be served from a cache
    const int k2 = (k + 32) % ROWF;
    const float w = mode == 1 ? __hip_atomic_load(row + k2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : mode == 2 ? row[k2] : *(volatile const float*)(row + k2);
    if (__ballot(v != (float)i || w == -1.0f)) bad++;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) st(ack, (uint32_t)(i + 1));
  }
  if (lane == 0) { out[0] = bad; out[1] = (unsigned long long)(wall_clock64() - t0); }
}

int main() {
  const int n = 200000;
  hipStream_t s0, s1;
  CHECK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking));
  CHECK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  unsigned long long* out;
  CHECK(hipHostMalloc((void**)&out, 16, hipHostMallocDefault));
  const char* kinds[] = {"hipMalloc (coarse-grained device)", "hipExtMallocWithFlags(hipDeviceMallocFinegrained)", "hipExtMallocWithFlags(hipDeviceMallocUncached)",
                         "hipHostMalloc(Mapped | Coherent)"};
  for (int kind = 0; kind < 4; kind++) {
    for (int cntkind = 0; cntkind < 2; cntkind++) {   // the counters: with the data, or in host-coherent memory
      for (int mode = 0; mode < 3; mode++) {
        float* buf = nullptr; uint32_t* cnt = nullptr;
        const size_t bytes = (size_t)RING * ROWF * 4;
        hipError_t e = hipSuccess;
        if (kind == 0) e = hipMalloc((void**)&buf, bytes + 256);
        else if (kind == 1) e = hipExtMallocWithFlags((void**)&buf, bytes + 256, hipDeviceMallocFinegrained);
        else if (kind == 2) e = hipExtMallocWithFlags((void**)&buf, bytes + 256, hipDeviceMallocUncached);
        else e = hipHostMalloc((void**)&buf, bytes + 256, hipHostMallocMapped | hipHostMallocCoherent);
        if (e != hipSuccess) { printf("%-52s allocation failed: %s\n", kinds[kind], hipGetErrorString(e)); (void)hipGetLastError(); break; }
        uint32_t* hostcnt = nullptr;
        if (cntkind) { CHECK(hipHostMalloc((void**)&hostcnt, 256, hipHostMallocMapped | hipHostMallocCoherent)); cnt = hostcnt; }
        else cnt = (uint32_t*)((char*)buf + bytes);
        if (kind == 3) memset(buf, 0, bytes + 256); else CHECK(hipMemset(buf, 0, bytes + 256));
        if (hostcnt) memset(hostcnt, 0, 256);
        CHECK(hipDeviceSynchronize());
        out[0] = out[1] = 0;
        hipLaunchKernelGGL(consumer, dim3(8), dim3(64), 0, s1, buf, cnt, cnt + 32, n, 5, mode, out);
        hipLaunchKernelGGL(producer, dim3(8), dim3(64), 0, s0, buf, cnt, cnt + 32, n, 2, mode);
        CHECK(hipDeviceSynchronize());
        printf("%-52s counters %-6s data %-22s: %8llu of %d rows stale, %.2f us per hand-off (round trip)\n", kinds[kind], cntkind ? "host" : "same", mode == 1 ? "agent atomics" : mode == 2 ? "plain st, ordinary ld" : "plain st, volatile ld", out[0], n,
               (double)out[1] / 100.0 / n);
        if (kind == 3) (void)hipHostFree(buf); else (void)hipFree(buf);
        if (hostcnt) (void)hipHostFree(hostcnt);
      }
    }
  }
  // ---- the HOST writes rows (then the counter), a kernel reads them: which host allocation lets ordinary loads see them? ----
  const unsigned hflags[] = {hipHostMallocMapped | hipHostMallocCoherent, hipHostMallocMapped | hipHostMallocUncached};
  const char* hnames[] = {"hipHostMalloc(Mapped | Coherent)", "hipHostMalloc(Mapped | Uncached)"};
  for (int hk = 0; hk < 2; hk++) {
    for (int mode = 0; mode < 3; mode++) {
      float* buf = nullptr;
      const size_t bytes = (size_t)RING * ROWF * 4;
      // (below the host fills only a QUARTER of a row per hand-off -- 64 floats at offset 64 * (i % 4) -- and the consumer checks that quarter:
      //  the other quarters of the same 128-byte lines were read a hand-off earlier, as consecutive records of a stream are)
      if (hipHostMalloc((void**)&buf, bytes + 256, hflags[hk]) != hipSuccess) { printf("%-52s allocation failed\n", hnames[hk]); (void)hipGetLastError(); break; }
      memset(buf, 0, bytes + 256);
      volatile uint32_t* cnt = (volatile uint32_t*)((char*)buf + bytes);
      volatile uint32_t* ack = cnt + 32;
      const int nh = 20000;
      out[0] = out[1] = 0;
      hipLaunchKernelGGL(consumer_part, dim3(8), dim3(64), 0, s1, buf, (const uint32_t*)cnt, (uint32_t*)ack, nh, 5, mode, out);
      for (int i = 0; i < nh; i++) {
        while (i >= 8 && (int)*ack < i - 8 + 1) {}
        float* row = buf + (size_t)((i / 8) % RING) * ROWF;
        for (int k = 0; k < 32; k++) row[32 * (i % 8) + k] = (float)i;   // 128 bytes: the next eighth of the row (lines are 128 bytes, so every line is written once, but the lines of a row are re-read 8 times)
        __sync_synchronize();
        *cnt = (uint32_t)(i + 1);
      }
      CHECK(hipDeviceSynchronize());
      printf("%-52s HOST producer   data %-22s: %8llu of %d rows stale, %.2f us per hand-off (round trip)\n", hnames[hk], mode == 1 ? "agent-atomic ld" : mode == 2 ? "ordinary ld" : "volatile ld", out[0], nh,
             (double)out[1] / 100.0 / nh);
      (void)hipHostFree(buf);
    }
  }
  return 0;
}
