/* oracle/fxcm_match.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of fxcm's SparseMatchModel (reference src/models/fxcmv1.cpp:1742-1829; MTFList :1694-1732): four
 * hashes over the last 3 / 4 / 6-with-stride-2 / 5 bytes index a table of positions; the finder that hit last is
 * tried first; a found match predicts the next byte until a bit disagrees. Two inputs per bit. Pinned against the
 * reference's own struct in tests/test_oracle_fxcmcore.py. hist/mask/pos: the 16 MB byte history ring (:3251-3253). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fxcm_core.h"

enum { NumHashes = 4, MaxLen = 64 };
typedef struct { int root, index, prev[NumHashes], next[NumHashes]; } Mtf;
static void mtf_front(Mtf* l, int i) {
  if ((l->index = i) == l->root) return;
  const int p = l->prev[i], n = l->next[i];
  if (p >= 0) l->next[p] = l->next[i];
  if (n >= 0) l->prev[n] = l->prev[i];
  l->prev[l->root] = i;
  l->next[i] = l->root;
  l->root = i;
  l->prev[l->root] = -1;
}
typedef struct {
  uint32_t* table;               /* 1 M positions */
  Mtf list;
  uint32_t hashes[NumHashes], hashIndex, length, index;
  uint8_t expectedByte, valid;
} FxSparseMatch;
static const struct { uint32_t stride, minLen; } kSparse[NumHashes] = {{1, 3}, {1, 4}, {2, 6}, {1, 5}};

FxSparseMatch* orc_fx_sparsematch_new(void) {
  FxSparseMatch* m = (FxSparseMatch*)calloc(1, sizeof *m);
  m->table = (uint32_t*)calloc(1024 * 1024, 4);
  for (int i = 0; i < NumHashes; i++) { m->list.prev[i] = i - 1; m->list.next[i] = i + 1; }
  m->list.next[NumHashes - 1] = -1;
  return m;
}
#define BUF(i) ((uint32_t)hist[((uint32_t)pos - (uint32_t)(i)) & mask])
#define BUFR(i) ((uint32_t)hist[(uint32_t)(i) & mask])
static void sm_update(FxSparseMatch* m, const uint8_t* hist, uint32_t mask, int pos) {
  const uint32_t tmask = 1024 * 1024 - 1;
  for (uint32_t i = 0; i < NumHashes; i++) {
    uint32_t h = (i + 1) * 191;
    for (uint32_t j = 0, k = 1; j < kSparse[i].minLen; j++, k += kSparse[i].stride) h = h * 191 + (BUF(k) << i);
    m->hashes[i] = h & tmask;
  }
  if (m->length) {
    m->index++;
    if (m->length < MaxLen) m->length++;
  } else {
    for (int i = m->list.index = m->list.root; i >= 0; i = (m->list.index >= 0 ? (m->list.index = m->list.next[m->list.index]) : m->list.index)) {
      m->index = m->table[m->hashes[i]];
      if (m->index > 0) {
        uint32_t offset = 1;
        while (m->length < kSparse[i].minLen && (BUF(offset) ^ BUFR(m->index - offset)) == 0) { m->length++; offset += kSparse[i].stride; }
        if (m->length >= kSparse[i].minLen) {
          m->length -= kSparse[i].minLen - 1;
          m->hashIndex = (uint32_t)i;
          mtf_front(&m->list, i);
          break;
        }
      }
      m->length = m->index = 0;
    }
  }
  for (uint32_t i = 0; i < NumHashes; i++) m->table[m->hashes[i]] = (uint32_t)pos;
  m->expectedByte = (uint8_t)BUFR(m->index);
  m->valid = m->length > 1;
}
int fx_sparsematch_p(FxSparseMatch* m, FxSink* s, int bpos, int c0, const uint8_t* hist, uint32_t mask, int pos) {
  const uint8_t B = (uint8_t)(c0 << (8 - bpos));
  if (bpos == 0) sm_update(m, hist, mask, pos);
  if (m->length > 0 && ((m->expectedByte ^ B) >> (8 - bpos)) != 0) m->length = 0;
  if (m->valid && m->length > 1) {
    const int expectedBit = (m->expectedByte >> (7 - bpos)) & 1, sign = 2 * expectedBit - 1;
    const int l1 = (int)m->length - 1, l2 = (int)m->length - 2;
    fx_add(s, sign * ((l1 < 32 ? l1 : 32) << 5));
    fx_add(s, sign * (1 << (l2 < 3 ? l2 : 3)) * (l1 < 8 ? l1 : 8) << 4);
  } else { fx_add(s, 0); fx_add(s, 0); }
  return (int)m->length;
}
int orc_fx_sparsematch_p(FxSparseMatch* m, int bpos, int c0, const uint8_t* hist, uint32_t mask, int pos, int16_t* out, int* state4) {
  FxSink s;
  s.ncount = s.pidx = 0;
  const int r = fx_sparsematch_p(m, &s, bpos, c0, hist, mask, pos);
  memcpy(out, s.n, (size_t)s.ncount * 2);
  state4[0] = (int)m->hashIndex; state4[1] = (int)m->index; state4[2] = m->expectedByte; state4[3] = m->valid;
  return r;
}
