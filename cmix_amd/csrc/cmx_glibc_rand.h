// cmx_glibc_rand.h -- glibc rand() after srand(seed): TYPE_3 additive-feedback generator
// (glibc stdlib/random_r.c), r[i] = r[i-3] + r[i-31], 310 warm-up draws, result >> 1.
// The reference seeds it with 0xDEADBEEF (predictor.cpp:26) and draws from it in construction
// order: one value per Indirect model (indirect.cpp:10), then the LSTM weights
// (lstm-layer.cpp:52-59). Re-implemented so the library never touches the process-global state.
#ifndef CMX_GLIBC_RAND_H
#define CMX_GLIBC_RAND_H
#include <stddef.h>
#include <stdint.h>
#include <vector>

struct GlibcRand {
  std::vector<int32_t> r;
  size_t k = 344;
  explicit GlibcRand(uint32_t seed) : r(344) {
    int32_t word = (int32_t)seed;
    if (word == 0) word = 1;
    r[0] = word;
    for (int i = 1; i < 31; ++i) {
      long hi = word / 127773, lo = word % 127773;
      long w = 16807 * lo - 2836 * hi;
      if (w < 0) w += 2147483647;
      word = (int32_t)w;
      r[i] = word;
    }
    for (int i = 31; i < 34; ++i) r[i] = r[i - 31];
    for (int i = 34; i < 344; ++i) r[i] = (int32_t)((uint32_t)r[i - 31] + (uint32_t)r[i - 3]);
  }
  int next() {
    int32_t v = (int32_t)((uint32_t)r[k - 31] + (uint32_t)r[k - 3]);
    r.push_back(v);
    ++k;
    return (int)((uint32_t)v >> 1);
  }
};

#endif
