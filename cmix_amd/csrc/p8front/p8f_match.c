/* p8front/p8f_match.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's MatchModel (reference src/models/paq8.cpp:3520-3692): the longest-match predictor over the
 * whole byte history -- three hashes of the last 9 / 7 / 5 bytes into a position table, match verification and
 * extension, "delta" mode after a miss, and its read-out through three StateMap32s, three SmallStationaryContextMaps
 * and three StationaryMaps plus an IndirectContext<U8>(19, 1). Parity: tests/test_p8stage_host.py (stage vs columns 434..2024 of reference traces). hist[] is the reference's Buf: a ring of bmask + 1 bytes, hist[(pos-1) & bmask] = the
 * last byte. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int p8f_stretch(int p);
int p8f_ilog(int x);
uint64_t p8f_combine64(uint64_t seed, uint64_t x);
uint32_t p8f_finalize64(uint64_t h, int bits);
uint64_t p8f_hash5(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e);
typedef struct DMap DMap;
DMap* p8f_dmap_new(int kind, int bits_of_context, int bits_per_context, int rate);
void p8f_dmap_set_direct(DMap* m, uint32_t ctx);
void p8f_dmap_set(DMap* m, uint64_t ctx);
int p8f_dmap_mix(DMap* m, int y, int a, int mul, int div, int16_t* out);
int p8f_dmap_skip(DMap* m, int a, int mul, int div, int16_t* out);
typedef struct P8fStateMap32 P8fStateMap32;
P8fStateMap32* p8f_statemap32_new(int n);
void p8f_statemap32_emit(P8fStateMap32* s, int cx, int zero, int16_t* out);

enum { MaxLen = 0xFFFF, MinLen = 5, StepSize = 2, DeltaLen = 5, NumCtxs = 3, NumHashes = 3 };
static unsigned ilog2u(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }

typedef struct {
  uint32_t* table;
  uint32_t mask, hashes[NumHashes], length, index;
  int hashbits;
  uint8_t expected, delta;
  P8fStateMap32* sm[NumCtxs];
  DMap *scm[3], *maps[3];
  uint8_t* ictx;       /* IndirectContext<U8>(19, 1): 1 << 19 cells, 1 input bit */
  uint32_t ictx_cur;
} Match;

Match* p8f_match_new(uint32_t size) {
  Match* m = (Match*)calloc(1, sizeof *m);
  m->table = (uint32_t*)calloc(size / 4, 4);
  m->mask = size / 4 - 1;
  m->hashbits = (int)ilog2u(m->mask + 1);
  m->sm[0] = p8f_statemap32_new(56 * 256);
  m->sm[1] = p8f_statemap32_new(8 * 256 * 256 + 1);
  m->sm[2] = p8f_statemap32_new(256 * 256);
  m->scm[0] = p8f_dmap_new(0, 8, 8, 0);
  m->scm[1] = p8f_dmap_new(0, 11, 1, 0);
  m->scm[2] = p8f_dmap_new(0, 8, 8, 0);
  m->maps[0] = p8f_dmap_new(1, 16, 8, 0);
  m->maps[1] = p8f_dmap_new(1, 22, 1, 0);
  m->maps[2] = p8f_dmap_new(1, 4, 1, 0);
  m->ictx = (uint8_t*)calloc(1 << 19, 1);
  return m;
}
/* Buf (:169-187): a ring of 2^k bytes; buffer(i) = i-th last byte, buffer[i] = absolute position, both wrapped */
#define BUFB(i) ((uint32_t)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
#define BUFA(i) ((uint32_t)hist[(uint32_t)(i) & bmask])
static void ictx_add_set(Match* m, int y, uint32_t next) {  /* iCtx += y, iCtx = next  (:1484-1490) */
  m->ictx[m->ictx_cur] = (uint8_t)((m->ictx[m->ictx_cur] << 1) | (y & 1));
  m->ictx_cur = next & ((1u << 19) - 1);
}
static uint64_t maps1_ctx(Match* m, int c0, const uint8_t* hist, int pos, uint32_t bmask) {
  const unsigned lg = ilog2u(m->length + 1);
  return p8f_hash5(m->expected, (uint64_t)c0, BUFB(1), BUFB(2), lg < 3 ? lg : 3);
}
static void match_update(Match* m, int y, int c0, const uint8_t* hist, int pos, uint32_t bmask) {  /* Update() :3544-3596 */
  m->delta = 0;
  unsigned minLen = MinLen + (NumHashes - 1) * StepSize;
  for (unsigned i = 0; i < NumHashes; i++, minLen -= StepSize) {
    uint64_t h = 0;
    for (unsigned j = minLen; j > 0; j--) h = p8f_combine64(h, BUFB(j));
    m->hashes[i] = p8f_finalize64(h, m->hashbits);
  }
  if (m->length) {
    m->index++;
    if (m->length < MaxLen) m->length++;
  } else {
    unsigned bestLen = 0, bestIndex = 0;
    minLen = MinLen + (NumHashes - 1) * StepSize;
    for (unsigned i = 0; i < NumHashes && m->length < minLen; i++, minLen -= StepSize) {
      m->index = m->table[m->hashes[i]];
      if (m->index > 0) {
        m->length = 0;
        while (m->length < minLen && BUFB(m->length + 1) == BUFA(m->index - m->length - 1)) m->length++;
        if (m->length > bestLen) { bestLen = m->length; bestIndex = m->index; }
      }
    }
    if (bestLen >= MinLen) { m->length = bestLen - (MinLen - 1); m->index = bestIndex; }
    else m->length = m->index = 0;
  }
  for (unsigned i = 0; i < NumHashes; i++) m->table[m->hashes[i]] = (uint32_t)pos;
  m->expected = (uint8_t)BUFA(m->index);
  ictx_add_set(m, y, (BUFB(1) << 8) | m->expected);
  p8f_dmap_set_direct(m->scm[0], m->expected);
  p8f_dmap_set_direct(m->scm[1], m->expected);
  p8f_dmap_set_direct(m->scm[2], (uint32_t)pos);
  p8f_dmap_set_direct(m->maps[0], ((uint32_t)m->expected << 8) | BUFB(1));
  p8f_dmap_set(m->maps[1], maps1_ctx(m, c0, hist, pos, bmask));
  p8f_dmap_set_direct(m->maps[2], m->ictx[m->ictx_cur]);
}
/* Predict() :3630-3691. Returns the match length; *expected_out = Stats->Match.expectedByte. */
int p8f_match_step(Match* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out,
                      int* nout, int* expected_out) {
  int n = 0;
  if (bpos == 0) match_update(m, y, c0, hist, pos, bmask);
  else {
    const uint8_t B = (uint8_t)(c0 << (8 - bpos));
    p8f_dmap_set_direct(m->scm[1], ((uint32_t)bpos << 8) | (uint32_t)(m->expected ^ B));
    p8f_dmap_set(m->maps[1], maps1_ctx(m, c0, hist, pos, bmask));
    ictx_add_set(m, y, ((uint32_t)bpos << 16) | (BUFB(1) << 8) | (uint32_t)(m->expected ^ B));
    p8f_dmap_set_direct(m->maps[2], m->ictx[m->ictx_cur]);
  }
  if (bpos == 0) *expected_out = m->length > 0 ? m->expected : 0;
  const int expectedBit = (m->expected >> (7 - bpos)) & 1;
  if (m->length > 0) {
    const int isMatch = bpos == 0 ? (BUFB(1) == BUFA(m->index - 1)) : (((m->expected + 256) >> (8 - bpos)) == c0);
    if (!isMatch) { m->delta = (m->length + MinLen) > DeltaLen; m->length = 0; }
  }
  uint32_t ctx[NumCtxs] = {0, 0, 0};
  if (m->length > 0) {
    if (m->length <= 16) ctx[0] = (m->length - 1) * 2 + (uint32_t)expectedBit;
    else ctx[0] = 24 + (((m->length - 1) < 63 ? (m->length - 1) : 63) >> 2) * 2 + (uint32_t)expectedBit;
    ctx[0] = (ctx[0] << 8) | (uint32_t)c0;
    ctx[1] = (((uint32_t)m->expected << 11) | ((uint32_t)bpos << 8) | BUFB(1)) + 1;
    const int sign = 2 * expectedBit - 1;
    out[n++] = (int16_t)(sign * (int)((m->length < 32 ? m->length : 32) << 5));
    out[n++] = (int16_t)(sign * (p8f_ilog((int)(m->length & 0xffff)) << 2));
  } else { out[n++] = 0; out[n++] = 0; }
  if (m->delta) ctx[2] = ((uint32_t)m->expected << 8) | (uint32_t)c0;
  for (int i = 0; i < NumCtxs; i++) {
    p8f_statemap32_emit(m->sm[i], (int)ctx[i], ctx[i] == 0, out + n);   /* ctx != 0 ? (stretch(sm.p(ctx)) + 1) >> 1 : 0, the map learns either way */
    n++;
  }
  n += p8f_dmap_mix(m->scm[0], y, 7, 1, 4, out + n);
  n += p8f_dmap_mix(m->scm[1], y, 6, 1, 4, out + n);
  n += p8f_dmap_mix(m->scm[2], y, 5, 1, 4, out + n);
  n += p8f_dmap_mix(m->maps[0], y, 255, 1, 4, out + n);
  n += p8f_dmap_mix(m->maps[1], y, 1023, 1, 4, out + n);
  n += p8f_dmap_mix(m->maps[2], y, 1023, 1, 4, out + n);
  if (bpos != 0) *expected_out = -1;
  *nout = n;
  return (int)m->length;
}

/* ---- SparseMatchModel (:3694-3843): four "sparse" match finders (byte masks, skipped bytes, strides) tried in
 * move-to-front order (MTFList :1498-1528), read out through four StationaryMaps and two indirect contexts, plus two
 * mixer weight-set selectors. ---- */
typedef struct { uint32_t offset, stride, deletions, minLen, bitMask; } SparseCfg;
typedef struct {
  uint32_t* table;
  uint32_t mask, hashes[4], hashIndex, length, index;
  int hashbits;
  uint8_t expected, valid;
  DMap* maps[4];
  uint8_t* ictx8; uint32_t ictx8_cur;     /* IndirectContext<U8>(19, 1) */
  uint16_t* ictx16; uint32_t ictx16_cur;  /* IndirectContext<U16>(16) */
  int root, idx, prev[4], next[4];        /* MTFList(4) */
  SparseCfg sparse[4];
} SMatch;

SMatch* p8f_sparsematch_new(uint64_t size) {
  SMatch* m = (SMatch*)calloc(1, sizeof *m);
  m->table = (uint32_t*)calloc(size / 4, 4);
  m->mask = (uint32_t)(size / 4 - 1);
  m->hashbits = (int)ilog2u(m->mask + 1);
  m->maps[0] = p8f_dmap_new(1, 22, 1, 0);
  m->maps[1] = p8f_dmap_new(1, 14, 4, 0);
  m->maps[2] = p8f_dmap_new(1, 8, 1, 0);
  m->maps[3] = p8f_dmap_new(1, 19, 1, 0);
  m->ictx8 = (uint8_t*)calloc(1 << 19, 1);
  m->ictx16 = (uint16_t*)calloc(1 << 16, 2);
  for (int i = 0; i < 4; ++i) {
    m->prev[i] = i - 1; m->next[i] = i + 1;
    m->sparse[i] = (SparseCfg){0, 1, 0, 3, 0xFF};
  }
  m->next[3] = -1;
  m->sparse[0].minLen = 5; m->sparse[0].bitMask = 0xDF;
  m->sparse[1].offset = 1; m->sparse[1].minLen = 4;
  m->sparse[2].stride = 2; m->sparse[2].minLen = 4; m->sparse[2].bitMask = 0xDF;
  m->sparse[3].minLen = 5; m->sparse[3].bitMask = 0xF;
  return m;
}
static void mtf_front(SMatch* m, int i) {  /* MTFList::MoveToFront */
  m->idx = i;
  if (i == m->root) return;
  const int p = m->prev[i], n = m->next[i];
  if (p >= 0) m->next[p] = m->next[i];
  if (n >= 0) m->prev[n] = m->prev[i];
  m->prev[m->root] = i;
  m->next[i] = m->root;
  m->root = i;
  m->prev[m->root] = -1;
}
static uint64_t smaps0_ctx(SMatch* m, int c0, const uint8_t* hist, int pos, uint32_t bmask) {
  return p8f_hash5(m->expected, (uint64_t)c0, BUFB(1), BUFB(2), ilog2u(m->length + 1) * 4 + m->hashIndex);
}
static void smatch_update(SMatch* m, int y, int c0, const uint8_t* hist, int pos, uint32_t bmask) {
  for (unsigned i = 0; i < 4; i++) {
    uint64_t h = 0;
    for (unsigned j = 0, k = m->sparse[i].offset + 1; j < m->sparse[i].minLen; j++, k += m->sparse[i].stride)
      h = p8f_combine64(h, BUFB(k) & m->sparse[i].bitMask);
    m->hashes[i] = p8f_finalize64(h, m->hashbits);
  }
  if (m->length) {
    m->index++;
    if (m->length < 0xFFFF) m->length++;
  } else {
    for (int i = (m->idx = m->root); i >= 0; i = (m->idx >= 0 ? (m->idx = m->next[m->idx]) : m->idx)) {
      m->index = m->table[m->hashes[i]];
      if (m->index > 0) {
        uint32_t offset = m->sparse[i].offset + 1;
        while (m->length < m->sparse[i].minLen && ((BUFB(offset) ^ BUFA(m->index - offset)) & m->sparse[i].bitMask) == 0) {
          m->length++;
          offset += m->sparse[i].stride;
        }
        if (m->length >= m->sparse[i].minLen) {
          m->length -= (m->sparse[i].minLen - 1);
          m->index += m->sparse[i].deletions;
          m->hashIndex = (uint32_t)i;
          mtf_front(m, i);
          break;
        }
      }
      m->length = m->index = 0;
    }
  }
  for (unsigned i = 0; i < 4; i++) m->table[m->hashes[i]] = (uint32_t)pos;
  m->expected = (uint8_t)BUFA(m->index);
  if (m->valid) {
    m->ictx8[m->ictx8_cur] = (uint8_t)((m->ictx8[m->ictx8_cur] << 1) | (y & 1));
    m->ictx16[m->ictx16_cur] = (uint16_t)((m->ictx16[m->ictx16_cur] << 8) | (BUFB(1) & 0xff));
  }
  m->valid = m->length > 1;
  if (m->valid) {
    p8f_dmap_set(m->maps[0], smaps0_ctx(m, c0, hist, pos, bmask));
    p8f_dmap_set_direct(m->maps[1], ((uint32_t)m->expected << 8) | BUFB(1));
    m->ictx8_cur = ((BUFB(1) << 8) | m->expected) & ((1u << 19) - 1);
    m->ictx16_cur = ((BUFB(1) << 8) | m->expected) & 0xffff;
    p8f_dmap_set_direct(m->maps[2], m->ictx8[m->ictx8_cur]);
    p8f_dmap_set_direct(m->maps[3], m->ictx16[m->ictx16_cur]);
  }
}
int p8f_sparsematch_step(SMatch* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out,
                            int* nout, int* sets) {
  const uint8_t B = (uint8_t)(c0 << (8 - bpos));
  int n = 0;
  if (bpos == 0) smatch_update(m, y, c0, hist, pos, bmask);
  else if (m->valid) {
    p8f_dmap_set(m->maps[0], smaps0_ctx(m, c0, hist, pos, bmask));
    if (bpos == 4) p8f_dmap_set_direct(m->maps[1], 0x10000u | ((uint32_t)(m->expected ^ (uint8_t)(c0 << 4)) << 8) | BUFB(1));
    m->ictx8[m->ictx8_cur] = (uint8_t)((m->ictx8[m->ictx8_cur] << 1) | (y & 1));
    m->ictx8_cur = (((uint32_t)bpos << 16) | (BUFB(1) << 8) | (uint32_t)(m->expected ^ B)) & ((1u << 19) - 1);
    p8f_dmap_set_direct(m->maps[2], m->ictx8[m->ictx8_cur]);
    p8f_dmap_set_direct(m->maps[3], ((uint32_t)bpos << 16) | (m->ictx16[m->ictx16_cur] ^ (uint32_t)(B | (B << 8))));
  }
  if (m->length > 0 && (((m->expected ^ B) & m->sparse[m->hashIndex].bitMask) >> (8 - bpos)) != 0) m->length = 0;
  if (m->valid) {
    if (m->length > 1 && ((m->sparse[m->hashIndex].bitMask >> (7 - bpos)) & 1) > 0) {
      const int expectedBit = (m->expected >> (7 - bpos)) & 1, sign = 2 * expectedBit - 1;
      const uint32_t l1 = m->length - 1, l2 = m->length - 2;
      out[n++] = (int16_t)(sign * (int)((l1 < 64 ? l1 : 64) << 4));
      out[n++] = (int16_t)((sign * (1 << (l2 < 3 ? l2 : 3)) * (int)(l1 < 8 ? l1 : 8)) << 4);
      out[n++] = (int16_t)(sign * 512);
    } else { out[n++] = 0; out[n++] = 0; out[n++] = 0; }
    for (int i = 0; i < 4; i++) n += p8f_dmap_mix(m->maps[i], y, 1023, 1, 2, out + n);
  } else {   /* 11 zero inputs; the four maps are not touched this step */
    for (int i = 0; i < 3; i++) out[n++] = 0;
    for (int i = 0; i < 4; i++) n += p8f_dmap_skip(m->maps[i], 1023, 1, 2, out + n);
  }
  const uint32_t l7 = m->length < 7 ? m->length : 7, lg = ilog2u(m->length + 1);
  sets[0] = (int)((m->hashIndex << 6) | ((uint32_t)bpos << 3) | l7);
  sets[1] = 4 * 64 + (int)((m->hashIndex << 11) | ((lg < 7 ? lg : 7) << 8) | ((uint32_t)c0 ^ (uint32_t)(m->expected >> (8 - bpos))));
  *nout = n;
  return (int)m->length;
}
