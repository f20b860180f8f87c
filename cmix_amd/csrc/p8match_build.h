// p8match_build.h -- host-side construction of a P8MatchDev (p8match_dev.h): MatchModel's and SparseMatchModel's
// constructors (reference src/models/paq8.cpp:3536-3543, :3710-3721) and those of the maps they own; memory from a policy
// object as in p8cm2_build.h.
#ifndef CMX_P8MATCH_BUILD_H
#define CMX_P8MATCH_BUILD_H
#include <cstring>
#include <vector>

#include "p8match_dev.h"

namespace p8b {
template <class Policy>
void dmap_new(P8DMap& m, Policy& P, int kind, int bits_of_context, int bits_per_context, int rate, const uint8_t* nex1024) {
  memset(&m, 0, sizeof m);
  m.kind = kind; m.mask = (1 << bits_of_context) - 1; m.maskbits = bits_of_context;
  m.stride = (1 << bits_per_context) - 1; m.btotal = bits_per_context;
  const size_t n = ((size_t)1 << bits_of_context) * (size_t)m.stride;
  if (kind == 0) { std::vector<uint16_t> v(n, 0x7FFF); m.d16 = (uint16_t*)P.zalloc(n * 2); P.upload(m.d16, v.data(), n * 2); }
  else if (kind == 1) {
    std::vector<uint32_t> v(n, (0x7FFu << 20) | (uint32_t)(rate < 1023 ? rate : 1023));
    m.d32 = (uint32_t*)P.zalloc(n * 4); P.upload(m.d32, v.data(), n * 4);
  } else {
    m.d8 = (uint8_t*)P.zalloc(n);
    uint32_t sm[256];
    for (int i = 0; i < 256; ++i) {
      uint32_t n0 = nex1024[4 * i + 2], n1 = nex1024[4 * i + 3];
      if (n0 == 0) n1 *= 64;
      if (n1 == 0) n0 *= 64;
      sm[i] = ((n1 << 16) / (n0 + n1 + 1)) << 16;
    }
    m.sm = (uint32_t*)P.zalloc(sizeof sm); P.upload(m.sm, sm, sizeof sm);
  }
}
inline int log2u(uint64_t x) { int n = 0; while (x > 1) { x >>= 1; ++n; } return n; }
// match_bytes / sparse_bytes: the two position tables' sizes in bytes (powers of two); hist_log2: the history ring
template <class Policy>
bool build_match(P8MatchDev& h, Policy& P, uint64_t match_bytes, uint64_t sparse_bytes, int hist_log2, const uint8_t* nex1024, const int16_t* stretch4096,
                 const uint8_t* ilog65536) {
  memset(&h, 0, sizeof h);
  if (match_bytes < 1024 || (match_bytes & (match_bytes - 1)) || sparse_bytes < 1024 || (sparse_bytes & (sparse_bytes - 1)) || hist_log2 < 12 || hist_log2 > 30) return false;
  auto up = [&](const void* src, size_t bytes) { void* p = P.zalloc(bytes); P.upload(p, src, bytes); return p; };
  h.nex = (const uint8_t*)up(nex1024, 1024);
  h.stretch = (const int16_t*)up(stretch4096, 4096 * 2);
  h.ilog = (const uint8_t*)up(ilog65536, 65536);
  h.hist = (uint8_t*)P.zalloc((size_t)1 << hist_log2);
  h.bmask = (uint32_t)(((uint64_t)1 << hist_log2) - 1);
  h.m_table = (uint32_t*)P.zalloc((size_t)match_bytes);
  h.m_mask = (uint32_t)(match_bytes / 4 - 1);
  h.m_hashbits = log2u((uint64_t)h.m_mask + 1);
  static const size_t sm_n[3] = {56 * 256, 8 * 256 * 256 + 1, 256 * 256};
  for (int i = 0; i < 3; i++) {
    std::vector<uint32_t> v(sm_n[i], 1u << 31);
    h.m_sm[i] = (uint32_t*)up(v.data(), v.size() * 4);
  }
  dmap_new(h.m_scm[0], P, 0, 8, 8, 0, nex1024); dmap_new(h.m_scm[1], P, 0, 11, 1, 0, nex1024); dmap_new(h.m_scm[2], P, 0, 8, 8, 0, nex1024);
  dmap_new(h.m_maps[0], P, 1, 16, 8, 0, nex1024); dmap_new(h.m_maps[1], P, 1, 22, 1, 0, nex1024); dmap_new(h.m_maps[2], P, 1, 4, 1, 0, nex1024);
  h.m_ictx = (uint8_t*)P.zalloc((size_t)1 << 19);
  h.s_table = (uint32_t*)P.zalloc((size_t)sparse_bytes);
  h.s_mask = (uint32_t)(sparse_bytes / 4 - 1);
  h.s_hashbits = log2u((uint64_t)h.s_mask + 1);
  dmap_new(h.s_maps[0], P, 1, 22, 1, 0, nex1024); dmap_new(h.s_maps[1], P, 1, 14, 4, 0, nex1024);
  dmap_new(h.s_maps[2], P, 1, 8, 1, 0, nex1024); dmap_new(h.s_maps[3], P, 1, 19, 1, 0, nex1024);
  h.s_ictx8 = (uint8_t*)P.zalloc((size_t)1 << 19);
  h.s_ictx16 = (uint16_t*)P.zalloc(((size_t)1 << 16) * 2);
  for (int i = 0; i < 4; ++i) {
    h.s_prev[i] = i - 1; h.s_next[i] = i + 1;
    h.sparse[i] = P8SparseCfg{0, 1, 0, 3, 0xFF};
  }
  h.s_next[3] = -1;
  h.sparse[0].minLen = 5; h.sparse[0].bitMask = 0xDF;
  h.sparse[1].offset = 1; h.sparse[1].minLen = 4;
  h.sparse[2].stride = 2; h.sparse[2].minLen = 4; h.sparse[2].bitMask = 0xDF;
  h.sparse[3].minLen = 5; h.sparse[3].bitMask = 0xF;
  return true;
}
}  // namespace p8b
#endif
