// p8cm_build.h -- host-side construction of a P8CmDev (p8cm_dev.h): ContextMap's constructor (reference
// src/models/paq8.cpp:1049-1062), StateMap's (:626-635) and Random's (:154-157); memory from a policy object as in
// p8cm2_build.h (whose hash helper ContextMap::set shares, :1064-1069).
#ifndef CMX_P8CM_BUILD_H
#define CMX_P8CM_BUILD_H
#include "p8cm2_build.h"
#include "p8cm_dev.h"

namespace p8b {
template <class Policy>
bool build_family(P8CmDev& h, Policy& P, int ninst, const uint64_t* size_bytes, const int* counts, const uint8_t* nex1024, const int16_t* stretch4096,
                  const uint8_t* ilog257) {
  memset(&h, 0, sizeof h);
  if (ninst <= 0 || ninst > P8CM_MAXI) return false;
  auto up = [&](const void* src, size_t bytes) { void* p = P.zalloc(bytes); P.upload(p, src, bytes); return p; };
  h.ninst = ninst; h.slot_parallel = 1;
  int s = 0;
  for (int k = 0; k < ninst; k++) {
    const uint64_t sz = size_bytes[k];
    if (counts[k] <= 0 || s + counts[k] > P8CM_MAXS || sz < 4096 || (sz & (sz - 1)) || (sz >> 6) > 0x4000000ull) return false;
    h.inst[k].table = (uint8_t*)P.zalloc((size_t)sz);
    h.inst[k].mask = (uint32_t)((sz >> 6) - 1);
    h.inst[k].first = s; h.inst[k].count = counts[k];
    for (int i = 0; i < counts[k]; i++, s++) {
      h.slot_inst[s] = (uint8_t)k;
      h.regs.cp0[s] = h.regs.cp[s] = P8_B_STATE;   // &t[0].bh[0][0]
      h.regs.runp[s] = P8_B_STATE + 3;
    }
  }
  h.nslots = s;
  h.row_stride = 5 * s; h.order_slot = -1;
  for (int i = 0; i < s; i++) h.slot_off[i] = (int16_t)(5 * i);
  h.nex = (const uint8_t*)up(nex1024, 1024);
  h.stretch = (const int16_t*)up(stretch4096, 4096 * 2);
  h.ilog = (const uint8_t*)up(ilog257, 257);
  std::vector<uint16_t> sm((size_t)s * 256);
  for (size_t i = 0; i < sm.size(); ++i) {
    int n0 = nex1024[4 * (i & 255) + 2], n1 = nex1024[4 * (i & 255) + 3];
    if (n0 == 0) n1 *= 64;
    if (n1 == 0) n0 *= 64;
    sm[i] = (uint16_t)(65536 * (n1 + 1) / (n0 + n1 + 2));
  }
  h.sm = (uint16_t*)up(sm.data(), sm.size() * 2);
  h.rnd.table[0] = 123456789; h.rnd.table[1] = 987654321;
  for (int j = 0; j < 62; ++j) h.rnd.table[j + 2] = h.rnd.table[j + 1] * 11 + h.rnd.table[j] * 23 / 16;
  h.rnd.i = 0;
  return true;
}
}  // namespace p8b
#endif
