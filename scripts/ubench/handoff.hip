// Cross-workgroup hand-off latency on MI355X: ping-pong of one value between two workgroups of one launch.
// variants: protocol (0 = data store, wait, counter add | poll counter, load data ; 1 = 8-byte {value, tag} store | poll it)
//           scope of the accesses (agent / system), partner on the same XCD (block 0 <-> 8) or another (0 <-> 1)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

template <int SCOPE> __device__ __forceinline__ unsigned long long ld64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE> __device__ __forceinline__ void st64(unsigned long long* p, unsigned long long v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE> __device__ __forceinline__ unsigned ld32(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, SCOPE); }
template <int SCOPE> __device__ __forceinline__ void st32(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, SCOPE); }

struct Area { unsigned long long ll[2][32]; unsigned cnt[2][32]; unsigned data[2][32]; };

template <int PROTO, int SCOPE>
__global__ void pingpong(Area* A, int a, int b, int iters, long long* out) {
  const int me = blockIdx.x == a ? 0 : (blockIdx.x == b ? 1 : -1);
  if (me < 0 || threadIdx.x != 0) return;
  const int other = me ^ 1;
  long long t0 = wall_clock64();
  unsigned v = 0;
  for (int i = 1; i <= iters; ++i) {
    if (me == 0) {
      // send i, wait for the echo
      if (PROTO == 1) {
        st64<SCOPE>(&A->ll[0][0], ((unsigned long long)i << 32) | (v + 1));
        unsigned long long r;
        { unsigned g = 0; do { r = ld64<SCOPE>(&A->ll[1][0]); } while ((unsigned)(r >> 32) != (unsigned)i && ++g < (1u << 20)); if (g >= (1u << 20)) { out[0] = -1; return; } }
        v = (unsigned)r;
      } else {
        st32<SCOPE>(&A->data[0][0], v + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&A->cnt[0][0], 1u, __ATOMIC_RELAXED, SCOPE);
        { unsigned g = 0; while (ld32<SCOPE>(&A->cnt[1][0]) < (unsigned)i && ++g < (1u << 20)) {} if (g >= (1u << 20)) { out[0] = -1; return; } }
        v = ld32<SCOPE>(&A->data[1][0]);
      }
    } else {
      if (PROTO == 1) {
        unsigned long long r;
        { unsigned g = 0; do { r = ld64<SCOPE>(&A->ll[0][0]); } while ((unsigned)(r >> 32) != (unsigned)i && ++g < (1u << 20)); if (g >= (1u << 20)) return; }
        st64<SCOPE>(&A->ll[1][0], ((unsigned long long)i << 32) | ((unsigned)r + 1));
      } else {
        { unsigned g = 0; while (ld32<SCOPE>(&A->cnt[0][0]) < (unsigned)i && ++g < (1u << 20)) {} if (g >= (1u << 20)) return; }
        unsigned r = ld32<SCOPE>(&A->data[0][0]);
        st32<SCOPE>(&A->data[1][0], r + 1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(&A->cnt[1][0], 1u, __ATOMIC_RELAXED, SCOPE);
      }
    }
  }
  long long t1 = wall_clock64();
  if (me == 0) { out[0] = t1 - t0; out[1] = v; }
  (void)other;
}

template <int PROTO, int SCOPE> void run(const char* name, Area* A, long long* d_out, int a, int b) {
  const int iters = 2000;
  hipMemset(A, 0, sizeof(Area));
  hipLaunchKernelGGL((pingpong<PROTO, SCOPE>), dim3(16), dim3(64), 0, 0, A, a, b, iters, d_out);
  hipDeviceSynchronize();
  long long h[2];
  hipMemcpy(h, d_out, sizeof h, hipMemcpyDeviceToHost);
  // wall_clock64: 100 MHz constant clock
  printf("%-44s blocks %d<->%d : %8.3f us per round trip (one way %.3f us), check %lld\n", name, a, b, h[0] / 100.0 / iters, h[0] / 200.0 / iters, h[1]);
}

int main() {
  Area* A; long long* d_out;
  hipMalloc(&A, sizeof(Area)); hipMalloc(&d_out, 16);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, __HIP_MEMORY_SCOPE_AGENT>("counter+data, agent scope", A, d_out, 0, 8);
    run<0, __HIP_MEMORY_SCOPE_AGENT>("counter+data, agent scope", A, d_out, 0, 1);
    run<1, __HIP_MEMORY_SCOPE_AGENT>("8-byte value|tag, agent scope", A, d_out, 0, 8);
    run<1, __HIP_MEMORY_SCOPE_AGENT>("8-byte value|tag, agent scope", A, d_out, 0, 1);
    run<1, __HIP_MEMORY_SCOPE_AGENT>("8-byte value|tag, agent scope", A, d_out, 0, 4);
    run<1, __HIP_MEMORY_SCOPE_SYSTEM>("8-byte value|tag, system scope", A, d_out, 0, 8);
    run<1, __HIP_MEMORY_SCOPE_SYSTEM>("8-byte value|tag, system scope", A, d_out, 0, 1);
    run<1, __HIP_MEMORY_SCOPE_WORKGROUP>("8-byte value|tag, workgroup scope (same XCD)", A, d_out, 0, 8);
  }
  return 0;
}
