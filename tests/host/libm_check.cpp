// Exhaustive host-side check of cmix_amd/csrc/cmx_libm.h against the host glibc
// (the libm the -O3 reference binary resolves to). Usage: libm_check [stride]
// stride 1 = all 2^32 float bit patterns. Exit code 0 iff bit-identical.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>
#include <atomic>
#include "../../cmix_amd/csrc/cmx_libm.h"

static inline bool same(float a, float b) {
  if (a != a && b != b) return true;
  return cmx_f2u(a) == cmx_f2u(b);
}

int main(int argc, char** argv) {
  uint64_t stride = argc > 1 ? strtoull(argv[1], 0, 10) : 1;
  unsigned nt = std::thread::hardware_concurrency();
  if (!nt) nt = 4;
  std::atomic<uint64_t> bad_exp{0}, bad_tanh{0}, bad_log{0};
  std::vector<std::thread> th;
  for (unsigned t = 0; t < nt; ++t)
    th.emplace_back([&, t] {
      uint64_t be = 0, bt = 0, bl = 0;
      for (uint64_t u = t * stride; u < (1ull << 32); u += nt * stride) {
        float x = cmx_u2f((uint32_t)u);
        volatile float xv = x;
        float e0 = expf(xv), e1 = cmx_expf(x);
        if (!same(e0, e1)) { if (be++ < 3) fprintf(stderr, "expf(%a)=%a mine %a\n", x, e0, e1); }
        float t0 = tanhf(xv), t1 = cmx_tanhf(x);
        if (!same(t0, t1)) { if (bt++ < 3) fprintf(stderr, "tanhf(%a)=%a mine %a\n", x, t0, t1); }
        float l0 = 1 / (1 + expf(-xv)), l1 = cmx_logistic(x);
        if (!same(l0, l1)) { if (bl++ < 3) fprintf(stderr, "logistic(%a)=%a mine %a\n", x, l0, l1); }
      }
      bad_exp += be; bad_tanh += bt; bad_log += bl;
    });
  for (auto& x : th) x.join();
  printf("stride %llu: expf mismatches %llu, tanhf mismatches %llu, logistic mismatches %llu\n",
         (unsigned long long)stride, (unsigned long long)bad_exp.load(),
         (unsigned long long)bad_tanh.load(), (unsigned long long)bad_log.load());
  return (bad_exp || bad_tanh || bad_log) ? 1 : 0;
}
