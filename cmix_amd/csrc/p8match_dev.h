// p8match_dev.h -- paq8's two match predictors (reference src/models/paq8.cpp:3520-3692 MatchModel, :3694-3843
// SparseMatchModel with MTFList :1498-1528) and the three direct-lookup map flavours they read out through
// (SmallStationaryContextMap :891-933, StationaryMap :935-974, IndirectMap :976-1008; the same maps serve recordModel,
// sparseModel1, the linear-prediction model ...). Input: the byte stream only. Two lanes of one workgroup: lane 0 appends
// the finished byte to the history ring, then lane 0 = MatchModel and lane 1 = SparseMatchModel (they share nothing but
// the ring, which neither writes). 17 + 11 mixer inputs per bit, the sparse model's two weight-set selectors, and the
// match statistics other models read (length, expected byte). Single source: tests/host/p8match_emul.cpp.
#ifndef CMX_P8MATCH_DEV_H
#define CMX_P8MATCH_DEV_H
#include <stdint.h>

#include "p8cm2_dev.h"   // P8_HD

struct P8DMap {            // kind 0 SSCM (u16 cells), 1 StationaryMap (u32), 2 IndirectMap (u8 histories + StateMap32(256))
  int kind, mask, maskbits, stride, context, bcount, btotal, B;
  uint32_t cp;
  uint16_t* d16; uint32_t* d32; uint8_t* d8; uint32_t* sm; int sm_cxt;
};
struct P8SparseCfg { uint32_t offset, stride, deletions, minLen, bitMask; };
struct P8MatchDev {
  const uint8_t* nex; const int16_t* stretch; const uint8_t* ilog;   // ilog(x), x in [0, 65535]
  uint8_t* hist; uint32_t bmask; int pos;                            // Buf (:169-187)
  // MatchModel
  uint32_t* m_table; uint32_t m_mask, m_hashes[3], m_length, m_index; int m_hashbits; uint8_t m_expected, m_delta;
  uint32_t* m_sm[3]; int m_sm_cxt[3];
  P8DMap m_scm[3], m_maps[3];
  uint8_t* m_ictx; uint32_t m_ictx_cur;
  // SparseMatchModel
  uint32_t* s_table; uint32_t s_mask, s_hashes[4], s_hashIndex, s_length, s_index; int s_hashbits; uint8_t s_expected, s_valid;
  P8DMap s_maps[4];
  uint8_t* s_ictx8; uint32_t s_ictx8_cur; uint16_t* s_ictx16; uint32_t s_ictx16_cur;
  int s_root, s_idx, s_prev[4], s_next[4];
  P8SparseCfg sparse[4];
  int last_y;
};

P8_HD uint64_t p8d_combine64(uint64_t seed, uint64_t x) { return (seed + x + 1) * 0x9E3779B97F4A7C15ull; }
P8_HD uint32_t p8d_finalize64(uint64_t h, int bits) { return (uint32_t)(h >> (64 - bits)); }
P8_HD uint64_t p8d_hash5(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e) {   // hash(a..e) :742-773
  return (a + 1) * 0x9E3779B97F4A7C15ull + (b + 1) * 0x993DDEFFB1462949ull + (c + 1) * 0xE9C91DC159AB0D2Dull + (d + 1) * 0x83D6A14F1B0CED73ull +
         (e + 1) * 0xA14F1B0CED5A841Full;
}
P8_HD unsigned p8d_ilog2(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }
P8_HD int p8d_sm32_lim(uint32_t* t, int* cxt, int y, int cx, int limit) {   // StateMap32::p (:660-672)
  uint32_t p0 = t[*cxt];
  const int n = p0 & 1023, pr = p0 >> 10;
  if (n < limit) ++p0; else p0 = (p0 & 0xfffffc00u) | (uint32_t)limit;
  p0 += (uint32_t)((((y << 22) - pr) >> 3) * (16384 / (n + n + 3))) & 0xfffffc00u;
  t[*cxt] = p0;
  *cxt = cx;
  return (int)(t[cx] >> 20);
}
P8_HD void p8d_dmap_set_direct(P8DMap* m, uint32_t ctx) { m->context = (int)(ctx & (uint32_t)m->mask) * m->stride; m->bcount = m->B = 0; }
P8_HD void p8d_dmap_set(P8DMap* m, uint64_t ctx) { m->context = (int)(p8d_finalize64(ctx, m->maskbits) & (uint32_t)m->mask) * m->stride; m->bcount = m->B = 0; }
// mix(): a = rate (SSCM) or Limit (the other two); two inputs
P8_HD int p8d_dmap_mix(const P8MatchDev* d, P8DMap* m, int y, int a, int mul, int div, int16_t* out) {
  int pred;
  if (m->kind == 0) {
    m->d16[m->cp] = (uint16_t)(m->d16[m->cp] + (((y << 16) - m->d16[m->cp] + (1 << (a - 1))) >> a));
    m->B += (y && m->B > 0);
    m->cp = (uint32_t)(m->context + m->B);
    pred = m->d16[m->cp] >> 4;
  } else if (m->kind == 1) {
    const uint32_t v = m->d32[m->cp];
    const int lim = a < 0x3FF ? a : 0x3FF;
    const uint32_t count = (uint32_t)lim < (v & 0x3FF) + 1 ? (uint32_t)lim : (v & 0x3FF) + 1;
    int p = (int)(v >> 10), err = (y << 22) - p;
    err = ((err / 8) * (16384 / (int)(count + count + 3))) / 1024;   // dt[Count]
    p = p + err; p = p < 0 ? 0 : p > 0x3FFFFF ? 0x3FFFFF : p;
    m->d32[m->cp] = ((uint32_t)p << 10) | count;
    m->B += (y && m->B > 0);
    m->cp = (uint32_t)(m->context + m->B);
    pred = (int)(m->d32[m->cp] >> 20);
  } else {
    m->d8[m->cp] = d->nex[4 * m->d8[m->cp] + y];
    m->B += (y && m->B > 0);
    m->cp = (uint32_t)(m->context + m->B);
    pred = p8d_sm32_lim(m->sm, &m->sm_cxt, y, m->d8[m->cp], a);
  }
  out[0] = (int16_t)((d->stretch[pred] * mul) / div);
  out[1] = (int16_t)(((pred - 2048) * mul) / (div * 2));
  m->bcount++; m->B += m->B + 1;
  if (m->bcount == m->btotal) m->bcount = m->B = 0;
  return 2;
}
#define P8BUFB(i) ((uint32_t)d->hist[((uint32_t)d->pos - (uint32_t)(i)) & d->bmask])
#define P8BUFA(i) ((uint32_t)d->hist[(uint32_t)(i) & d->bmask])
enum { P8M_MaxLen = 0xFFFF, P8M_MinLen = 5, P8M_StepSize = 2, P8M_DeltaLen = 5, P8M_OUT = 17, P8S_OUT = 11 };

// MatchModel::Predict (+ Update at bpos 0) :3544-3691. out: 17 inputs; stats[0] = length, stats[1] = expected byte at bpos 0 (else -1)
P8_HD void p8d_match(P8MatchDev* d, int y, int bpos, int c0, int16_t* out, int* stats) {
  if (bpos == 0) {
    d->m_delta = 0;
    unsigned minLen = P8M_MinLen + 2 * P8M_StepSize;
    for (unsigned i = 0; i < 3; i++, minLen -= P8M_StepSize) {
      uint64_t h = 0;
      for (unsigned j = minLen; j > 0; j--) h = p8d_combine64(h, P8BUFB(j));
      d->m_hashes[i] = p8d_finalize64(h, d->m_hashbits);
    }
    if (d->m_length) {
      d->m_index++;
      if (d->m_length < P8M_MaxLen) d->m_length++;
    } else {
      unsigned bestLen = 0, bestIndex = 0;
      minLen = P8M_MinLen + 2 * P8M_StepSize;
      for (unsigned i = 0; i < 3 && d->m_length < minLen; i++, minLen -= P8M_StepSize) {
        d->m_index = d->m_table[d->m_hashes[i]];
        if (d->m_index > 0) {
          d->m_length = 0;
          while (d->m_length < minLen && P8BUFB(d->m_length + 1) == P8BUFA(d->m_index - d->m_length - 1)) d->m_length++;
          if (d->m_length > bestLen) { bestLen = d->m_length; bestIndex = d->m_index; }
        }
      }
      if (bestLen >= P8M_MinLen) { d->m_length = bestLen - (P8M_MinLen - 1); d->m_index = bestIndex; }
      else d->m_length = d->m_index = 0;
    }
    for (unsigned i = 0; i < 3; i++) d->m_table[d->m_hashes[i]] = (uint32_t)d->pos;
    d->m_expected = (uint8_t)P8BUFA(d->m_index);
    d->m_ictx[d->m_ictx_cur] = (uint8_t)((d->m_ictx[d->m_ictx_cur] << 1) | (y & 1));
    d->m_ictx_cur = ((P8BUFB(1) << 8) | d->m_expected) & ((1u << 19) - 1);
    p8d_dmap_set_direct(&d->m_scm[0], d->m_expected);
    p8d_dmap_set_direct(&d->m_scm[1], d->m_expected);
    p8d_dmap_set_direct(&d->m_scm[2], (uint32_t)d->pos);
    p8d_dmap_set_direct(&d->m_maps[0], ((uint32_t)d->m_expected << 8) | P8BUFB(1));
    { const unsigned lg = p8d_ilog2(d->m_length + 1); p8d_dmap_set(&d->m_maps[1], p8d_hash5(d->m_expected, (uint64_t)c0, P8BUFB(1), P8BUFB(2), lg < 3 ? lg : 3)); }
    p8d_dmap_set_direct(&d->m_maps[2], d->m_ictx[d->m_ictx_cur]);
  } else {
    const uint8_t B = (uint8_t)(c0 << (8 - bpos));
    p8d_dmap_set_direct(&d->m_scm[1], ((uint32_t)bpos << 8) | (uint32_t)(d->m_expected ^ B));
    { const unsigned lg = p8d_ilog2(d->m_length + 1); p8d_dmap_set(&d->m_maps[1], p8d_hash5(d->m_expected, (uint64_t)c0, P8BUFB(1), P8BUFB(2), lg < 3 ? lg : 3)); }
    d->m_ictx[d->m_ictx_cur] = (uint8_t)((d->m_ictx[d->m_ictx_cur] << 1) | (y & 1));
    d->m_ictx_cur = (((uint32_t)bpos << 16) | (P8BUFB(1) << 8) | (uint32_t)(d->m_expected ^ B)) & ((1u << 19) - 1);
    p8d_dmap_set_direct(&d->m_maps[2], d->m_ictx[d->m_ictx_cur]);
  }
  stats[1] = bpos == 0 ? (d->m_length > 0 ? d->m_expected : 0) : -1;
  const int expectedBit = (d->m_expected >> (7 - bpos)) & 1;
  if (d->m_length > 0) {
    const int isMatch = bpos == 0 ? (P8BUFB(1) == P8BUFA(d->m_index - 1)) : (((d->m_expected + 256) >> (8 - bpos)) == c0);
    if (!isMatch) { d->m_delta = (d->m_length + P8M_MinLen) > P8M_DeltaLen; d->m_length = 0; }
  }
  uint32_t ctx[3] = {0, 0, 0};
  int n = 0;
  if (d->m_length > 0) {
    if (d->m_length <= 16) ctx[0] = (d->m_length - 1) * 2 + (uint32_t)expectedBit;
    else ctx[0] = 24 + (((d->m_length - 1) < 63 ? (d->m_length - 1) : 63) >> 2) * 2 + (uint32_t)expectedBit;
    ctx[0] = (ctx[0] << 8) | (uint32_t)c0;
    ctx[1] = (((uint32_t)d->m_expected << 11) | ((uint32_t)bpos << 8) | P8BUFB(1)) + 1;
    const int sign = 2 * expectedBit - 1;
    out[n++] = (int16_t)(sign * (int)((d->m_length < 32 ? d->m_length : 32) << 5));
    out[n++] = (int16_t)(sign * (d->ilog[d->m_length & 0xffff] << 2));
  } else { out[n++] = 0; out[n++] = 0; }
  if (d->m_delta) ctx[2] = ((uint32_t)d->m_expected << 8) | (uint32_t)c0;
  for (int i = 0; i < 3; i++) {
    const int p = p8d_sm32_lim(d->m_sm[i], &d->m_sm_cxt[i], y, (int)ctx[i], 1023);
    out[n++] = (int16_t)(ctx[i] != 0 ? (d->stretch[p] + 1) >> 1 : 0);
  }
  n += p8d_dmap_mix(d, &d->m_scm[0], y, 7, 1, 4, out + n);
  n += p8d_dmap_mix(d, &d->m_scm[1], y, 6, 1, 4, out + n);
  n += p8d_dmap_mix(d, &d->m_scm[2], y, 5, 1, 4, out + n);
  n += p8d_dmap_mix(d, &d->m_maps[0], y, 255, 1, 4, out + n);
  n += p8d_dmap_mix(d, &d->m_maps[1], y, 1023, 1, 4, out + n);
  n += p8d_dmap_mix(d, &d->m_maps[2], y, 1023, 1, 4, out + n);
  stats[0] = (int)d->m_length;
}
P8_HD void p8d_smtf_front(P8MatchDev* d, int i) {   // MTFList::MoveToFront
  d->s_idx = i;
  if (i == d->s_root) return;
  const int p = d->s_prev[i], n = d->s_next[i];
  if (p >= 0) d->s_next[p] = d->s_next[i];
  if (n >= 0) d->s_prev[n] = d->s_prev[i];
  d->s_prev[d->s_root] = i;
  d->s_next[i] = d->s_root;
  d->s_root = i;
  d->s_prev[d->s_root] = -1;
}
// SparseMatchModel::Predict (+ Update) :3722-3843. out: 11 inputs; stats[2] = length; sets[0..1] = the two mixer selectors
P8_HD void p8d_sparse(P8MatchDev* d, int y, int bpos, int c0, int16_t* out, int* stats, int* sets) {
  const uint8_t B = (uint8_t)(c0 << (8 - bpos));
  int n = 0;
  if (bpos == 0) {
    for (unsigned i = 0; i < 4; i++) {
      uint64_t h = 0;
      for (unsigned j = 0, k = d->sparse[i].offset + 1; j < d->sparse[i].minLen; j++, k += d->sparse[i].stride) h = p8d_combine64(h, P8BUFB(k) & d->sparse[i].bitMask);
      d->s_hashes[i] = p8d_finalize64(h, d->s_hashbits);
    }
    if (d->s_length) {
      d->s_index++;
      if (d->s_length < 0xFFFF) d->s_length++;
    } else {
      for (int i = (d->s_idx = d->s_root); i >= 0; i = (d->s_idx >= 0 ? (d->s_idx = d->s_next[d->s_idx]) : d->s_idx)) {
        d->s_index = d->s_table[d->s_hashes[i]];
        if (d->s_index > 0) {
          uint32_t offset = d->sparse[i].offset + 1;
          while (d->s_length < d->sparse[i].minLen && ((P8BUFB(offset) ^ P8BUFA(d->s_index - offset)) & d->sparse[i].bitMask) == 0) {
            d->s_length++;
            offset += d->sparse[i].stride;
          }
          if (d->s_length >= d->sparse[i].minLen) {
            d->s_length -= (d->sparse[i].minLen - 1);
            d->s_index += d->sparse[i].deletions;
            d->s_hashIndex = (uint32_t)i;
            p8d_smtf_front(d, i);
            break;
          }
        }
        d->s_length = d->s_index = 0;
      }
    }
    for (unsigned i = 0; i < 4; i++) d->s_table[d->s_hashes[i]] = (uint32_t)d->pos;
    d->s_expected = (uint8_t)P8BUFA(d->s_index);
    if (d->s_valid) {
      d->s_ictx8[d->s_ictx8_cur] = (uint8_t)((d->s_ictx8[d->s_ictx8_cur] << 1) | (y & 1));
      d->s_ictx16[d->s_ictx16_cur] = (uint16_t)((d->s_ictx16[d->s_ictx16_cur] << 8) | (P8BUFB(1) & 0xff));
    }
    d->s_valid = d->s_length > 1;
    if (d->s_valid) {
      p8d_dmap_set(&d->s_maps[0], p8d_hash5(d->s_expected, (uint64_t)c0, P8BUFB(1), P8BUFB(2), p8d_ilog2(d->s_length + 1) * 4 + d->s_hashIndex));
      p8d_dmap_set_direct(&d->s_maps[1], ((uint32_t)d->s_expected << 8) | P8BUFB(1));
      d->s_ictx8_cur = ((P8BUFB(1) << 8) | d->s_expected) & ((1u << 19) - 1);
      d->s_ictx16_cur = ((P8BUFB(1) << 8) | d->s_expected) & 0xffff;
      p8d_dmap_set_direct(&d->s_maps[2], d->s_ictx8[d->s_ictx8_cur]);
      p8d_dmap_set_direct(&d->s_maps[3], d->s_ictx16[d->s_ictx16_cur]);
    }
  } else if (d->s_valid) {
    p8d_dmap_set(&d->s_maps[0], p8d_hash5(d->s_expected, (uint64_t)c0, P8BUFB(1), P8BUFB(2), p8d_ilog2(d->s_length + 1) * 4 + d->s_hashIndex));
    if (bpos == 4) p8d_dmap_set_direct(&d->s_maps[1], 0x10000u | ((uint32_t)(d->s_expected ^ (uint8_t)(c0 << 4)) << 8) | P8BUFB(1));
    d->s_ictx8[d->s_ictx8_cur] = (uint8_t)((d->s_ictx8[d->s_ictx8_cur] << 1) | (y & 1));
    d->s_ictx8_cur = (((uint32_t)bpos << 16) | (P8BUFB(1) << 8) | (uint32_t)(d->s_expected ^ B)) & ((1u << 19) - 1);
    p8d_dmap_set_direct(&d->s_maps[2], d->s_ictx8[d->s_ictx8_cur]);
    p8d_dmap_set_direct(&d->s_maps[3], ((uint32_t)bpos << 16) | (d->s_ictx16[d->s_ictx16_cur] ^ (uint32_t)(B | (B << 8))));
  }
  if (d->s_length > 0 && (((d->s_expected ^ B) & d->sparse[d->s_hashIndex].bitMask) >> (8 - bpos)) != 0) d->s_length = 0;
  if (d->s_valid) {
    if (d->s_length > 1 && ((d->sparse[d->s_hashIndex].bitMask >> (7 - bpos)) & 1) > 0) {
      const int expectedBit = (d->s_expected >> (7 - bpos)) & 1, sign = 2 * expectedBit - 1;
      const uint32_t l1 = d->s_length - 1, l2 = d->s_length - 2;
      out[n++] = (int16_t)(sign * (int)((l1 < 64 ? l1 : 64) << 4));
      out[n++] = (int16_t)((sign * (1 << (l2 < 3 ? l2 : 3)) * (int)(l1 < 8 ? l1 : 8)) * 16);
      out[n++] = (int16_t)(sign * 512);
    } else { out[n++] = 0; out[n++] = 0; out[n++] = 0; }
    for (int i = 0; i < 4; i++) n += p8d_dmap_mix(d, &d->s_maps[i], y, 1023, 1, 2, out + n);
  } else {
    for (int i = 0; i < P8S_OUT; i++) out[n++] = 0;
  }
  const uint32_t l7 = d->s_length < 7 ? d->s_length : 7, lg = p8d_ilog2(d->s_length + 1);
  sets[0] = (int)((d->s_hashIndex << 6) | ((uint32_t)bpos << 3) | l7);
  sets[1] = 4 * 64 + (int)((d->s_hashIndex << 11) | ((lg < 7 ? lg : 7) << 8) | ((uint32_t)c0 ^ (uint32_t)(d->s_expected >> (8 - bpos))));
  stats[2] = (int)d->s_length;
}
// one bit, two steps: (1) lane 0 appends the byte finished by the previous bit; (2) lane 0 = match, lane 1 = sparse match
P8_HD void p8d_match_step1(P8MatchDev* d, int tid, int push, int byte) { if (tid == 0 && push) { d->hist[(uint32_t)d->pos & d->bmask] = (uint8_t)byte; d->pos++; } }
P8_HD void p8d_match_step2(P8MatchDev* d, int tid, int y, int bpos, int c0, int16_t* out28, int* stats3, int* sets2) {
  if (tid == 0) p8d_match(d, y, bpos, c0, out28, stats3);
  else if (tid == 1) p8d_sparse(d, y, bpos, c0, out28 + P8M_OUT, stats3, sets2);
}
#undef P8BUFB
#undef P8BUFA
#endif
