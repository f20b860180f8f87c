// p8dmc_build.h -- host-side construction of a P8DmcDev (p8dmc_dev.h): dmcForest's constructor (reference
// src/models/paq8.cpp:7777-7795: ten graphs sized from MEM() = 0x10000 << level, dmcparams / dmcmem) and each
// dmcModel's first resetstategraph; memory from a policy object as in p8cm2_build.h.
#ifndef CMX_P8DMC_BUILD_H
#define CMX_P8DMC_BUILD_H
#include <cstring>
#include <vector>

#include "p8dmc_dev.h"

namespace p8b {
template <class Policy>
void build_dmc(P8DmcDev& h, Policy& P, int level, const uint8_t* nex1024, const int16_t* stretch4096) {
  static const uint32_t kParams[P8DMC_N] = {2, 32, 64, 4, 128, 8, 256, 16, 1024, 1536};   // dmcparams :7780
  static const uint64_t kMem[P8DMC_N] = {6, 10, 11, 7, 12, 8, 13, 9, 2, 2};                // dmcmem :7781
  memset(&h, 0, sizeof h);
  auto up = [&](const void* src, size_t bytes) { void* p = P.zalloc(bytes); P.upload(p, src, bytes); return p; };
  h.nex = (const uint8_t*)up(nex1024, 1024);
  h.stretch = (const int16_t*)up(stretch4096, 4096 * 2);
  uint32_t sm[256];
  for (int i = 0; i < 256; ++i) {   // StateMap32(256) :651-656
    uint32_t n0 = nex1024[4 * i + 2], n1 = nex1024[4 * i + 3];
    if (n0 == 0) n1 *= 64;
    if (n1 == 0) n0 *= 64;
    sm[i] = ((n1 << 16) / (n0 + n1 + 1)) << 16;
  }
  const uint64_t mem = 0x10000ull << level;   // MEM() :190-192
  std::vector<P8DmcNode> fresh(P8DMC_BASE);
  for (int k = 0; k < P8DMC_N; k++) {
    uint64_t nodes = (mem >> 2) / kMem[k] + P8DMC_BASE;
    const uint64_t cap = (1ull << 31) / 12;   // DMC_NODES_MAX
    if (nodes > cap) nodes = cap;
    P8DmcModel& M = h.m[k];
    M.size = (uint32_t)nodes; M.th_start = kParams[k];
    M.t = (P8DmcNode*)P.zalloc((size_t)nodes * sizeof(P8DmcNode));
    for (uint32_t q = 0; q < P8DMC_BASE; q++) fresh[q] = p8d_dmc_fresh(q, M.th_start);
    P.upload(M.t, fresh.data(), fresh.size() * sizeof(P8DmcNode));
    M.top = P8DMC_BASE; M.threshold = M.th_start; M.threshold_fine = M.th_start << 11;
    M.sm = (uint32_t*)up(sm, sizeof sm);
  }
}
}  // namespace p8b
#endif
