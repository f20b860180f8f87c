/* oracle/paq8_dmc.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of paq8's dynamic Markov coding models: dmcModel (reference src/models/paq8.cpp:7637-7775: a bit-level
 * state graph grown by cloning, fixed-point counts, a bit-history byte per node mapped through a StateMap32) and
 * dmcForest (:7777-7822: ten of them, eight reset when full). Inputs: the coded bit and bpos only. Pinned against the
 * reference's own classes in tests/test_oracle_paq8core.py.
 *
 * A node is 12 bytes as in the reference: two u16 counts and two u32 whose upper 28 bits are the successors and whose
 * lower 4 + 4 bits hold the bit-history state. */
#include <stdint.h>
#include <stdlib.h>

#include "paq8_tables.h"

int orc_p8_stretch(int p);
#define NEX(s, k) P8_STATE[4 * (s) + (k)]

typedef struct { uint16_t c0, c1; uint32_t nx0, nx1; } Node;
typedef struct { int cxt; uint32_t t[256]; } Sm;   /* StateMap32(256), :645-690 */
typedef struct {
  Node* t;
  uint64_t size;
  Sm sm;
  uint32_t top, curr, threshold, threshold_fine, extra;
} Dmc;

static void sm_init(Sm* s) {
  s->cxt = 0;
  for (int i = 0; i < 256; ++i) {
    uint32_t n0 = NEX(i, 2), n1 = NEX(i, 3);
    if (n0 == 0) n1 *= 64;
    if (n1 == 0) n0 *= 64;
    s->t[i] = ((n1 << 16) / (n0 + n1 + 1)) << 16;
  }
}
static int sm_p(Sm* s, int y, int cx, int limit) {
  uint32_t p0 = s->t[s->cxt];
  const int n = p0 & 1023, pr = p0 >> 10;
  if (n < limit) ++p0; else p0 = (p0 & 0xfffffc00u) | (uint32_t)limit;
  p0 += (uint32_t)((((y << 22) - pr) >> 3) * (16384 / (n + n + 3))) & 0xfffffc00u;
  s->t[s->cxt] = p0;
  return s->t[s->cxt = cx] >> 20;
}
static uint8_t nd_state(const Node* n) { return (uint8_t)(((n->nx0 & 0xf) << 4) | (n->nx1 & 0xf)); }
static void nd_set_state(Node* n, uint8_t s) { n->nx0 = (n->nx0 & 0xfffffff0u) | (s >> 4); n->nx1 = (n->nx1 & 0xfffffff0u) | (s & 0xf); }
static void nd_set_nx0(Node* n, uint32_t v) { n->nx0 = (n->nx0 & 0xf) | (v << 4); }
static void nd_set_nx1(Node* n, uint32_t v) { n->nx1 = (n->nx1 & 0xf) | (v << 4); }

static void dmc_reset(Dmc* d, uint32_t th_start) {  /* resetstategraph :7664-7684: 256 byte trees of 255 nodes */
  d->top = d->curr = d->extra = 0;
  d->threshold = th_start;
  d->threshold_fine = th_start << 11;
  for (int j = 0; j < 256; ++j)
    for (int i = 0; i < 255; ++i) {
      Node* n = &d->t[d->top];
      if (i < 127) { nd_set_nx0(n, d->top + i + 1); nd_set_nx1(n, d->top + i + 2); }
      else { const int root = (i - 127) * 2 * 255; nd_set_nx0(n, (uint32_t)root); nd_set_nx1(n, (uint32_t)root + 255); }
      n->c0 = n->c1 = th_start < 1024 ? 2048 : 512;
      nd_set_state(n, 0);
      d->top++;
    }
}
static uint32_t inc_counter(uint32_t x, uint32_t inc) { return (((x << 6) - x) >> 6) + (inc << 10); }
static void dmc_update(Dmc* d, int y) {  /* :7687-7735 */
  Node* cur = &d->t[d->curr];
  uint32_t c0 = cur->c0, c1 = cur->c1;
  const uint32_t n = y == 0 ? c0 : c1;
  cur->c0 = (uint16_t)inc_counter(c0, 1 - y);
  cur->c1 = (uint16_t)inc_counter(c1, y);
  nd_set_state(cur, NEX(nd_state(cur), y));
  if (n > d->threshold) {
    const uint32_t next = y == 0 ? cur->nx0 >> 4 : cur->nx1 >> 4;
    Node* nx = &d->t[next];
    c0 = nx->c0; c1 = nx->c1;
    const uint32_t nn = c0 + c1;
    if (nn > n + d->threshold) {
      if (d->top != d->size) {
        const uint32_t c0_top = (uint32_t)((uint64_t)c0 * n / nn), c1_top = (uint32_t)((uint64_t)c1 * n / nn);
        Node* tp = &d->t[d->top];
        tp->c0 = (uint16_t)c0_top; tp->c1 = (uint16_t)c1_top;
        nx->c0 = (uint16_t)(c0 - c0_top); nx->c1 = (uint16_t)(c1 - c1_top);
        tp->nx0 = tp->nx1 = 0;
        nd_set_nx0(tp, nx->nx0 >> 4); nd_set_nx1(tp, nx->nx1 >> 4);
        nd_set_state(tp, nd_state(nx));
        if (y == 0) nd_set_nx0(cur, d->top); else nd_set_nx1(cur, d->top);
        ++d->top;
        if (d->threshold < 8 * 1024) d->threshold = (++d->threshold_fine) >> 11;
      } else d->extra += nn >> 10;
    }
  }
  d->curr = y == 0 ? d->t[d->curr].nx0 >> 4 : d->t[d->curr].nx1 >> 4;
}
static int dmc_st(Dmc* d, int y) {  /* st() :7748-7751 */
  dmc_update(d, y);
  const Node* c = &d->t[d->curr];
  const uint32_t n0 = c->c0 + 1u, n1 = c->c1 + 1u;
  const int pr1 = (int)((n1 << 12) / (n0 + n1));
  const int pr2 = sm_p(&d->sm, y, nd_state(c), 256);
  return orc_p8_stretch(pr1) + orc_p8_stretch(pr2);
}

typedef struct { Dmc m[10]; } Forest;
static const uint32_t kParams[10] = {2, 32, 64, 4, 128, 8, 256, 16, 1024, 1536};   /* dmcparams :7780 */
static const uint64_t kMem[10] = {6, 10, 11, 7, 12, 8, 13, 9, 2, 2};                /* dmcmem :7781 */

Forest* orc_p8_dmc_new(int level) {
  Forest* f = (Forest*)calloc(1, sizeof *f);
  const uint64_t mem = 0x10000ull << level;  /* MEM() :190-192 */
  for (int i = 9; i >= 0; --i) {
    uint64_t nodes = (mem >> 2) / kMem[i] + 255 * 256;           /* DMC_NODES_BASE */
    const uint64_t cap = (1ull << 31) / 12;                      /* DMC_NODES_MAX */
    if (nodes > cap) nodes = cap;
    f->m[i].size = nodes;
    f->m[i].t = (Node*)calloc(nodes, sizeof(Node));
    sm_init(&f->m[i].sm);
    dmc_reset(&f->m[i], kParams[i]);
  }
  return f;
}
void orc_p8_dmc_free(Forest* f) {
  if (!f) return;
  for (int i = 0; i < 10; ++i) free(f->m[i].t);
  free(f);
}
int orc_p8_dmc_mix(Forest* f, int y, int bpos, int16_t* out) {  /* dmcForest::mix :7796-7815 */
  int i = 10, n = 0;
  out[n++] = (int16_t)(dmc_st(&f->m[--i], y) >> 3);
  out[n++] = (int16_t)(dmc_st(&f->m[--i], y) >> 3);
  while (i > 0) {
    const int pr1 = dmc_st(&f->m[--i], y);
    const int pr2 = dmc_st(&f->m[--i], y);
    out[n++] = (int16_t)((pr1 + pr2) >> 4);
  }
  if (bpos == 0)
    for (int k = 7; k >= 0; --k)
      if ((f->m[k].extra >> 7) > (uint32_t)f->m[k].size) dmc_reset(&f->m[k], kParams[k]);
  return n;
}
