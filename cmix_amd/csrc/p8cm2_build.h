// p8cm2_build.h -- host-side construction of a P8Cm2Dev (p8cm2_dev.h) following ContextMap2's constructor (reference
// src/models/paq8.cpp:1283-1303) and StateMap32's (:648-659); memory from a policy object (device: p8cm2.hip, host run
// of the kernel body: tests/host/p8cm2_emul.cpp), as in fxcm_build.h.
#ifndef CMX_P8CM2_BUILD_H
#define CMX_P8CM2_BUILD_H
#include <cstring>
#include <vector>

#include "p8cm2_dev.h"

namespace p8b {
// ContextMap2::set (:1305-1310): hash(ctx, index) -> bucket index bits + 16-bit checksum. hashbits = log2(size_bytes / 64).
inline void hash(uint64_t ctx, uint32_t index, int hashbits, uint32_t* ctx32, uint16_t* chk16) {
  const uint64_t h = (ctx + 1) * 0x9E3779B97F4A7C15ull + ((uint64_t)index + 1) * 0x993DDEFFB1462949ull;   // hash(a, b) :726
  *ctx32 = (uint32_t)(h >> (64 - hashbits));                                                               // finalize64 :746
  *chk16 = (uint16_t)((h >> (64 - hashbits - 16)) & 0xffff);                                               // checksum64 :752
}
inline int hashbits(uint64_t size_bytes) { int n = 0; for (uint64_t x = size_bytes >> 6; x > 1; x >>= 1) ++n; return n; }

template <class Policy>
bool build(P8Cm2Dev& h, Policy& P, uint64_t size_bytes, int count, const uint8_t* nex1024, const int16_t* stretch4096, const uint8_t* ilog257) {
  memset(&h, 0, sizeof h);
  if (count <= 0 || count > P8CM2_MAXC || size_bytes < 64 * 1024 || (size_bytes & (size_bytes - 1)) || (size_bytes >> 6) > 0x4000000ull) return false;
  auto up = [&](const void* src, size_t bytes) { void* p = P.zalloc(bytes); P.upload(p, src, bytes); return p; };
  h.C = count; h.slot_parallel = 1;
  h.mask = (uint32_t)((size_bytes >> 6) - 1);
  h.table = (uint8_t*)P.zalloc((size_t)size_bytes);
  h.nex = (const uint8_t*)up(nex1024, 1024);
  h.stretch = (const int16_t*)up(stretch4096, 4096 * 2);
  h.ilog = (const uint8_t*)up(ilog257, 257);
  std::vector<uint32_t> m8((size_t)count * P8_M8), m((size_t)count * P8_M12, 1u << 31);
  for (size_t i = 0; i < m8.size(); ++i) {   // StateMap32(256): the state's own counts (:651-656)
    uint32_t n0 = nex1024[4 * (i & 255) + 2], n1 = nex1024[4 * (i & 255) + 3];
    if (n0 == 0) n1 *= 64;
    if (n1 == 0) n0 *= 64;
    m8[i] = ((n1 << 16) / (n0 + n1 + 1)) << 16;
  }
  h.m8 = (uint32_t*)up(m8.data(), m8.size() * 4);
  h.m12 = (uint32_t*)up(m.data(), (size_t)count * P8_M12 * 4);
  h.m6 = (uint32_t*)up(m.data(), (size_t)count * P8_M6 * 4);
  for (int i = 0; i < count; ++i) {
    h.regs.bit_state[i] = h.regs.bit_state0[i] = (uint32_t)i * P8_B_SIZE + P8_B_STATE;   // &Table[i].BitState[0][0]
    h.regs.byte_hist[i] = h.regs.bit_state[i] + 3;
  }
  h.bits = 1;
  h.row_stride = 7 * count; h.out_off = 0;
  return true;
}
}  // namespace p8b
#endif
