// host_push.hip -- can the decoder's HOST thread write straight into device memory (large BAR), and what does a host -> device -> host round trip cost that way?
// Today (cmx_late.h) the device PULLS: a relay wave polls a word in host-coherent memory across PCIe (a non-posted read per poll), then copies the step's records
// across (another read round trip) and counts the step in uncached device memory. If the host can PUSH -- store the records and the counter into uncached device
// memory itself (posted writes through the BAR) -- both read round trips leave a decoded bit's path.
//   hipcc --offload-arch=gfx950 -O2 -o host_push host_push.hip && ./host_push
// Prints: whether a host store into each kind of device allocation works (a fault is caught by a signal handler), and the
// round-trip time host -> kernel -> host for (a) pull: kernel polls host memory, (b) push: host stores into device memory, the kernel polls that.
#include <hip/hip_runtime.h>
#include <signal.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/wait.h>
#include <time.h>
#include <unistd.h>
#include <immintrin.h>

static double now_us() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; }

// the kernel: for i = 1 .. n: wait until *in == i (system-scope loads), copy `rec` words from src to dst (the "records"), then *out = i (system-scope store)
__global__ void echo(const volatile uint32_t* in, volatile uint32_t* out, const uint32_t* src, uint32_t* dst, int rec, int n, uint32_t* stale) {
  const int lane = threadIdx.x;
  uint32_t bad = 0;
  for (int i = 1; i <= n; ++i) {
    __shared__ int gone;
    if (lane == 0) {
      unsigned spins = 0;
      gone = 0;
      while (__hip_atomic_load((const uint32_t*)in, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != (uint32_t)i) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > 30000000u) { gone = 1; break; }   // ~2 s without the host: leave (a kernel that never ends would hang the box)
      }
    }
    __syncthreads();
    if (gone) { if (lane == 0) __hip_atomic_store((uint32_t*)out, 0xFFFFFFFFu, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); return; }
    for (int k = lane; k < rec; k += 64) {
      const uint32_t v = __hip_atomic_load(src + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (v != (uint32_t)i * 1000u + (uint32_t)k) ++bad;    // the record of round i must be there when the counter says so
      if (dst) dst[k] = v;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (lane == 0) __hip_atomic_store((uint32_t*)out, (uint32_t)i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  if (bad) atomicAdd(stale, bad);
}

#include <setjmp.h>
static sigjmp_buf g_jb;
static void on_fault(int) { siglongjmp(g_jb, 1); }
static bool host_can_store(void* p) {   // in-process (a device mapping is not inherited by a forked child): a fault comes back through the handler
  struct sigaction sa, old_segv, old_bus;
  memset(&sa, 0, sizeof sa);
  sa.sa_handler = on_fault;
  sigaction(SIGSEGV, &sa, &old_segv);
  sigaction(SIGBUS, &sa, &old_bus);
  bool ok = false;
  if (sigsetjmp(g_jb, 1) == 0) {
    *(volatile uint32_t*)p = 0x12345678u;
    _mm_sfence();
    ok = *(volatile uint32_t*)p == 0x12345678u;
  }
  sigaction(SIGSEGV, &old_segv, nullptr);
  sigaction(SIGBUS, &old_bus, nullptr);
  return ok;
}

int main() {
  const int N = 20000, REC = 256;   // 1 KB of records per round
  int large_bar = -1;
  hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, 0);
  printf("hipDeviceAttributeIsLargeBar = %d\n", large_bar);
  uint32_t *h_in, *h_out, *h_rec, *d_stale;
  hipHostMalloc((void**)&h_in, 4096, hipHostMallocDefault);
  hipHostMalloc((void**)&h_out, 4096, hipHostMallocDefault);
  hipHostMalloc((void**)&h_rec, REC * 4, hipHostMallocDefault);
  hipMalloc((void**)&d_stale, 4);
  struct { const char* name; unsigned flag; int kind; } kinds[] = {{"hipMalloc", 0, 0}, {"hipExtMallocWithFlags(Uncached)", hipDeviceMallocUncached, 1}, {"hipExtMallocWithFlags(Finegrained)", hipDeviceMallocFinegrained, 1}};
  // (a) pull: the kernel polls host memory and reads the records from host memory
  {
    memset(h_in, 0, 4096); memset(h_out, 0, 4096); hipMemset(d_stale, 0, 4);
    uint32_t* d_dst; hipMalloc((void**)&d_dst, REC * 4);
    hipLaunchKernelGGL(echo, dim3(1), dim3(64), 0, 0, h_in, h_out, h_rec, d_dst, REC, N, d_stale);
    const double t0 = now_us();
    for (int i = 1; i <= N; ++i) {
      for (int k = 0; k < REC; ++k) h_rec[k] = (uint32_t)i * 1000u + (uint32_t)k;
      __sync_synchronize();
      *(volatile uint32_t*)h_in = (uint32_t)i;
      while (*(volatile uint32_t*)h_out != (uint32_t)i) { if (*(volatile uint32_t*)h_out == 0xFFFFFFFFu) { printf("pull: the kernel gave up at round %d\n", i); i = N + 1; break; } }
    }
    const double dt = now_us() - t0;
    hipDeviceSynchronize();
    uint32_t st = 0; hipMemcpy(&st, d_stale, 4, hipMemcpyDeviceToHost);
    printf("pull (kernel polls host memory, reads 1 KB of records from host memory): %.2f us per round trip, stale record words %u\n", dt / N, st);
    hipFree(d_dst);
  }
  for (auto& K : kinds) {
    uint32_t* d = nullptr;
    hipError_t e = K.kind ? hipExtMallocWithFlags((void**)&d, 8192, K.flag) : hipMalloc((void**)&d, 8192);
    if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", K.name, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    hipMemset(d, 0, 8192);
    hipDeviceSynchronize();
    const bool ok = host_can_store(d);
    printf("%s: host store into it %s\n", K.name, ok ? "WORKS" : "faults / does not arrive");
    if (!ok) { hipFree(d); continue; }
    // (b) push: records at d[64 ..], counter at d[0]
    memset(h_out, 0, 4096); hipMemset(d, 0, 8192); hipMemset(d_stale, 0, 4); hipDeviceSynchronize();
    hipLaunchKernelGGL(echo, dim3(1), dim3(64), 0, 0, d, h_out, d + 64, (uint32_t*)nullptr, REC, N, d_stale);
    const double t0 = now_us();
    for (int i = 1; i <= N; ++i) {
      for (int k = 0; k < REC; ++k) ((volatile uint32_t*)d)[64 + k] = (uint32_t)i * 1000u + (uint32_t)k;
      _mm_sfence();
      *(volatile uint32_t*)d = (uint32_t)i;
      _mm_sfence();
      while (*(volatile uint32_t*)h_out != (uint32_t)i) { if (*(volatile uint32_t*)h_out == 0xFFFFFFFFu) { printf("push: the kernel gave up at round %d (the host's stores do not arrive)\n", i); i = N + 1; break; } }
    }
    const double dt = now_us() - t0;
    hipDeviceSynchronize();
    uint32_t st = 0; hipMemcpy(&st, d_stale, 4, hipMemcpyDeviceToHost);
    printf("push into %s (host stores 1 KB of records + the counter, kernel polls device memory): %.2f us per round trip, stale record words %u\n", K.name, dt / N, st);
    hipFree(d);
  }
  return 0;
}
