/* oracle/paq8_maps.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of paq8's context-to-prediction structures (SURVEY.md 8a'): the hash helpers (reference
 * src/models/paq8.cpp:714-776), ilog (:253-266), ContextMap2 with its 64-byte bucket (:1164-1358),
 * SmallStationaryContextMap (:891-933), StationaryMap (:935-974), IndirectMap (:976-1008). Outputs go where the
 * reference hands them to Mixer::add(): an int16 sink in call order. Pinned against the reference's own classes
 * (oracle/ref_paq8core.cpp) in tests/test_oracle_paq8core.py.
 *
 * Representation: the reference keeps raw pointers into buckets (BitState / BitState0 / ByteHistory); here they are
 * byte offsets into the one table, 0xFFFFFFFF for nullptr. A bucket is 64 bytes: 7 x u16 checksums, 1 MRU byte,
 * 7 x 7 state bytes -- the reference's field order, which its pointer arithmetic (BitState0 + 3 = run stats and byte
 * history of slot) relies on. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "paq8_tables.h"

int orc_p8_squash(int d);
int orc_p8_stretch(int p);

#define NEX(s, k) P8_STATE[4 * (s) + (k)]

static uint8_t g_ilog[65536];
static int g_ilog_ready;
static void ilog_init(void) {  /* Ilog::Ilog :260-266 */
  if (g_ilog_ready) return;
  uint32_t x = 14155776;
  for (int i = 2; i < 65536; ++i) {
    x += 774541002 / (i * 2 - 1);
    g_ilog[i] = (uint8_t)(x >> 24);
  }
  g_ilog_ready = 1;
}
int orc_p8_ilog(int x) { ilog_init(); return g_ilog[x & 0xffff]; }

static unsigned ilog2u(unsigned x) {  /* :245-252: index of the highest set bit */
  unsigned n = 0;
  while (x > 1) { x >>= 1; ++n; }
  return n;
}

/* ---- hash helpers :714-776 ---- */
#define PHI64 0x9E3779B97F4A7C15ull
#define MUL64_1 0x993DDEFFB1462949ull
uint64_t orc_p8_hash2(uint64_t a, uint64_t b) { return (a + 1) * PHI64 + (b + 1) * MUL64_1; }
uint64_t orc_p8_combine64(uint64_t seed, uint64_t x) { return (seed + x + 1) * PHI64; }
uint32_t orc_p8_finalize64(uint64_t h, int bits) { return (uint32_t)(h >> (64 - bits)); }
uint64_t orc_p8_checksum64(uint64_t h, int hashbits, int checksumbits) { return h >> (64 - hashbits - checksumbits); }

typedef struct { int16_t* out; int n; } Sink;
static void sink_add(Sink* s, int v) { s->out[s->n++] = (int16_t)v; }

/* ---- StateMap32 (shared with paq8_core.c; local copy of the two operations used here) :645-690 ---- */
typedef struct { int N, cxt; uint32_t* t; } SM32;
static SM32* sm32_new(int n) {
  SM32* s = (SM32*)calloc(1, sizeof *s);
  s->N = n;
  s->t = (uint32_t*)malloc((size_t)n * 4);
  if (n == 256) {
    for (int i = 0; i < n; ++i) {
      uint32_t n0 = NEX(i, 2), n1 = NEX(i, 3);
      if (n0 == 0) n1 *= 64;
      if (n1 == 0) n0 *= 64;
      s->t[i] = ((n1 << 16) / (n0 + n1 + 1)) << 16;
    }
  } else {
    for (int i = 0; i < n; ++i) s->t[i] = 1u << 31;
  }
  return s;
}
static int sm32_p(SM32* s, int y, int cx, int limit) {
  uint32_t p0 = s->t[s->cxt];
  const int n = p0 & 1023, pr = p0 >> 10;
  if (n < limit) ++p0; else p0 = (p0 & 0xfffffc00u) | (uint32_t)limit;
  const int delta = (((y << 22) - pr) >> 3) * (16384 / (n + n + 3));
  p0 += (uint32_t)delta & 0xfffffc00u;
  s->t[s->cxt] = p0;
  return s->t[s->cxt = cx] >> 20;
}

/* ---- ContextMap2 :1164-1358 ---- */
#define NIL 0xFFFFFFFFu
enum { B_CHK = 0, B_MRU = 14, B_STATE = 15, B_SIZE = 64 };  /* bucket field offsets */
typedef struct {
  uint32_t C, mask, index, bits;
  int hashbits;
  uint8_t* table;                 /* [mask+1][64] */
  uint32_t *bit_state, *bit_state0, *byte_hist;  /* byte offsets into table */
  uint32_t* contexts;
  uint16_t* chk;
  uint8_t* has_history;
  SM32 **m6, **m8, **m12;
  uint8_t last_byte, last_bit, bit_pos;
} CM2;

/* Bucket::Find :1173-1189 -> byte offset of BitState[slot][0] */
static uint32_t bucket_find(CM2* c, uint32_t b, uint16_t checksum) {
  uint8_t* p = c->table + (size_t)b * B_SIZE;
  uint16_t* cs = (uint16_t*)(p + B_CHK);
  uint8_t* mru = p + B_MRU;
  if (cs[*mru & 15] == checksum) return b * B_SIZE + B_STATE + 7 * (*mru & 15);
  int worst = 0xFFFF, index = 0;
  for (int i = 0; i < 7; ++i) {
    if (cs[i] == checksum) { *mru = (uint8_t)(*mru << 4 | i); return b * B_SIZE + B_STATE + 7 * i; }
    if (p[B_STATE + 7 * i] < worst && (*mru & 15) != i && *mru >> 4 != i) { worst = p[B_STATE + 7 * i]; index = i; }
  }
  *mru = (uint8_t)(0xF0 | index);
  cs[index] = checksum;
  memset(p + B_STATE + 7 * index, 0, 7);
  return b * B_SIZE + B_STATE + 7 * index;
}

CM2* orc_p8_cm2_new(uint64_t size_bytes, uint32_t count) {
  ilog_init();
  CM2* c = (CM2*)calloc(1, sizeof *c);
  c->C = count;
  const uint64_t nb = size_bytes >> 6;
  c->mask = (uint32_t)(nb - 1);
  c->hashbits = (int)ilog2u(c->mask + 1);
  c->table = (uint8_t*)calloc(nb, B_SIZE);
  c->bit_state = (uint32_t*)malloc(count * 4); c->bit_state0 = (uint32_t*)malloc(count * 4);
  c->byte_hist = (uint32_t*)malloc(count * 4); c->contexts = (uint32_t*)calloc(count, 4);
  c->chk = (uint16_t*)calloc(count, 2); c->has_history = (uint8_t*)calloc(count, 1);
  c->m6 = (SM32**)malloc(count * sizeof(SM32*)); c->m8 = (SM32**)malloc(count * sizeof(SM32*));
  c->m12 = (SM32**)malloc(count * sizeof(SM32*));
  for (uint32_t i = 0; i < count; ++i) {
    c->m6[i] = sm32_new((1 << 6) + 8);
    c->m8[i] = sm32_new(1 << 8);
    c->m12[i] = sm32_new((1 << 12) + (1 << 9));
    c->bit_state[i] = c->bit_state0[i] = i * B_SIZE + B_STATE;  /* &Table[i].BitState[0][0] */
    c->byte_hist[i] = c->bit_state[i] + 3;
  }
  c->bits = 1;
  return c;
}
void orc_p8_cm2_free(CM2* c) {
  if (!c) return;
  for (uint32_t i = 0; i < c->C; ++i) {
    free(c->m6[i]->t); free(c->m6[i]); free(c->m8[i]->t); free(c->m8[i]); free(c->m12[i]->t); free(c->m12[i]);
  }
  free(c->m6); free(c->m8); free(c->m12); free(c->table); free(c->bit_state); free(c->bit_state0); free(c->byte_hist);
  free(c->contexts); free(c->chk); free(c->has_history); free(c);
}
static void cm2_set(CM2* c, uint64_t ctx) {  /* :1305-1310 */
  ctx = orc_p8_hash2(ctx, c->index);
  c->contexts[c->index] = orc_p8_finalize64(ctx, c->hashbits);
  c->chk[c->index] = (uint16_t)(orc_p8_checksum64(ctx, c->hashbits, 16) & 0xffff);
  c->index++;
}
static void cm2_update(CM2* c) {  /* :1209-1266 */
  uint8_t* T = c->table;
  for (uint32_t i = 0; i < c->index; ++i) {
    if (c->bit_state[i] != NIL) T[c->bit_state[i]] = NEX(T[c->bit_state[i]], c->last_bit);
    if (c->bit_pos > 1 && T[c->byte_hist[i]] == 0) { c->bit_state[i] = NIL; continue; }
    switch (c->bit_pos) {
      case 0: {
        const uint16_t chk = c->chk[i];
        const uint32_t ctx = c->contexts[i];
        c->bit_state[i] = c->bit_state0[i] = bucket_find(c, (ctx + c->bits) & c->mask, chk);
        uint8_t* s0 = T + c->bit_state0[i];
        if (s0[3] == 2) {  /* pending bit histories for bits 2-7 */
          const int cc = s0[4] + 256;
          uint8_t* p = T + bucket_find(c, (ctx + (cc >> 6)) & c->mask, chk);
          p[0] = (uint8_t)(1 + ((cc >> 5) & 1));
          p[1 + ((cc >> 5) & 1)] = (uint8_t)(1 + ((cc >> 4) & 1));
          p[3 + ((cc >> 4) & 3)] = (uint8_t)(1 + ((cc >> 3) & 1));
          p = T + bucket_find(c, (ctx + (cc >> 3)) & c->mask, chk);
          p[0] = (uint8_t)(1 + ((cc >> 2) & 1));
          p[1 + ((cc >> 2) & 1)] = (uint8_t)(1 + ((cc >> 1) & 1));
          p[3 + ((cc >> 1) & 3)] = (uint8_t)(1 + (cc & 1));
          s0[6] = 0;
        }
        uint8_t* bh = T + c->byte_hist[i];  /* byte history of the PREVIOUS context */
        bh[3] = bh[2];
        bh[2] = bh[1];
        if (bh[0] == 0) { bh[0] = 2; bh[1] = c->last_byte; }
        else if (bh[1] != c->last_byte) { bh[0] = 1; bh[1] = c->last_byte; }
        else if (bh[0] < 254) bh[0] += 2;
        else if (bh[0] == 255) bh[0] = 128;
        c->byte_hist[i] = c->bit_state0[i] + 3;
        c->has_history[i] = T[c->bit_state0[i]] > 15;
        break;
      }
      case 2: case 5:
        c->bit_state[i] = c->bit_state0[i] = bucket_find(c, (c->contexts[i] + c->bits) & c->mask, c->chk[i]);
        break;
      case 1: case 3: case 6: c->bit_state[i] = c->bit_state0[i] + 1 + c->last_bit; break;
      default: c->bit_state[i] = c->bit_state0[i] + 3 + (c->bits & 3); break;  /* 4, 7 */
    }
  }
}
/* One coded bit: at bpos == 0 the nset byte contexts first (contextModel2 :8139-8153), then mix() :1321-1357. */
int orc_p8_cm2_step(CM2* c, int y_prev, int bpos, const uint64_t* ctx, int nset, int16_t* out, int* nout) {
  Sink s = {out, 0};
  if (bpos == 0) for (int i = 0; i < nset; ++i) cm2_set(c, ctx[i]);
  int result = 0;
  c->last_bit = (uint8_t)y_prev;
  c->bit_pos = (uint8_t)bpos;
  c->bits += c->bits + c->last_bit;
  c->last_byte = (uint8_t)(c->bits & 0xFF);
  if (c->bit_pos == 0) c->bits = 1;
  cm2_update(c);
  const uint8_t* T = c->table;
  for (uint32_t i = 0; i < c->index; ++i) {
    int state = c->bit_state[i] != NIL ? T[c->bit_state[i]] : 0;
    result += state > 0;
    int p1 = sm32_p(c->m8[i], y_prev, state, 1023);
    int n0 = NEX(state, 2), n1 = NEX(state, 3), k = n1 + 1;   /* -~n1 */
    k = (k * 64) / (k + n0 + 1);                               /* k - ~n0 */
    n0 = -!n0; n1 = -!n1;
    const uint8_t* bh = T + c->byte_hist[i];
    if ((uint32_t)((bh[1] + 256) >> (8 - c->bit_pos)) == c->bits) {
      const int run = bh[0];
      const int sign = ((bh[1] >> (7 - c->bit_pos)) & 1) * 2 - 1;
      sink_add(&s, sign * (g_ilog[run + 1] << (3 - (run & 1))));
    } else if (c->bit_pos > 0 && (bh[0] & 1) > 0) {
      if ((uint32_t)((bh[2] + 256) >> (8 - c->bit_pos)) == c->bits) sink_add(&s, (((bh[2] >> (7 - c->bit_pos)) & 1) * 2 - 1) * 128);
      else if (c->has_history[i] && (uint32_t)((bh[3] + 256) >> (8 - c->bit_pos)) == c->bits)
        sink_add(&s, (((bh[3] >> (7 - c->bit_pos)) & 1) * 2 - 1) * 128);
      else sink_add(&s, 0);
    } else sink_add(&s, 0);
    if (c->has_history[i]) {
      state = (bh[1] >> (7 - c->bit_pos)) & 1;
      state |= ((bh[2] >> (7 - c->bit_pos)) & 1) * 2;
      state |= ((bh[3] >> (7 - c->bit_pos)) & 1) * 4;
    } else state = 8;
    const int st = orc_p8_stretch(p1) >> 2;
    sink_add(&s, st);
    sink_add(&s, (p1 - 2047) >> 3);
    p1 >>= 4;
    const int p0 = 255 - p1;
    sink_add(&s, st * abs(n1 - n0));
    sink_add(&s, (p1 & n0) - (p0 & n1));
    sink_add(&s, orc_p8_stretch(sm32_p(c->m12[i], y_prev, (state << 9) | (c->bit_pos << 6) | k, 1023)) >> 2);
    sink_add(&s, orc_p8_stretch(sm32_p(c->m6[i], y_prev, (state << 3) | c->bit_pos, 1023)) >> 2);
  }
  if (c->bit_pos == 7) c->index = 0;
  *nout = s.n;
  return result;
}

/* ---- SmallStationaryContextMap :891-933, StationaryMap :935-974, IndirectMap :976-1008 ---- */
typedef struct {
  int kind;                 /* 0 SSCM (u16), 1 StationaryMap (u32), 2 IndirectMap (u8 + StateMap32) */
  int mask, maskbits, stride, context, bcount, btotal, B;
  size_t cp, n;
  uint16_t* d16; uint32_t* d32; uint8_t* d8;
  SM32* map;
} DMap;
DMap* orc_p8_dmap_new(int kind, int bits_of_context, int bits_per_context, int rate) {
  DMap* m = (DMap*)calloc(1, sizeof *m);
  m->kind = kind; m->mask = (1 << bits_of_context) - 1; m->maskbits = bits_of_context;
  m->stride = (1 << bits_per_context) - 1; m->btotal = bits_per_context;
  m->n = ((size_t)1 << bits_of_context) * (size_t)m->stride;
  if (kind == 0) { m->d16 = (uint16_t*)malloc(m->n * 2); for (size_t i = 0; i < m->n; ++i) m->d16[i] = 0x7FFF; }
  else if (kind == 1) {
    m->d32 = (uint32_t*)malloc(m->n * 4);
    for (size_t i = 0; i < m->n; ++i) m->d32[i] = (0x7FFu << 20) | (uint32_t)(rate < 1023 ? rate : 1023);
  } else { m->d8 = (uint8_t*)calloc(m->n, 1); m->map = sm32_new(256); }
  return m;
}
void orc_p8_dmap_set_direct(DMap* m, uint32_t ctx) { m->context = (int)(ctx & (uint32_t)m->mask) * m->stride; m->bcount = m->B = 0; }
void orc_p8_dmap_set(DMap* m, uint64_t ctx) {
  m->context = (int)(orc_p8_finalize64(ctx, m->maskbits) & (uint32_t)m->mask) * m->stride;
  m->bcount = m->B = 0;
}
/* mix(): a = rate (SSCM) or Limit (the other two) */
int orc_p8_dmap_mix(DMap* m, int y, int a, int mul, int div, int16_t* out) {
  Sink s = {out, 0};
  int pred;
  if (m->kind == 0) {
    m->d16[m->cp] += ((y << 16) - m->d16[m->cp] + (1 << (a - 1))) >> a;
    m->B += (y && m->B > 0);
    m->cp = (size_t)(m->context + m->B);
    pred = m->d16[m->cp] >> 4;
  } else if (m->kind == 1) {
    const uint32_t v = m->d32[m->cp];
    const int lim = a < 0x3FF ? a : 0x3FF;
    const uint32_t count = (uint32_t)lim < (v & 0x3FF) + 1 ? (uint32_t)lim : (v & 0x3FF) + 1;
    int p = v >> 10, err = (y << 22) - p;
    err = ((err / 8) * (16384 / (int)(count + count + 3))) / 1024;   /* dt[Count] */
    p = p + err; p = p < 0 ? 0 : p > 0x3FFFFF ? 0x3FFFFF : p;
    m->d32[m->cp] = ((uint32_t)p << 10) | count;
    m->B += (y && m->B > 0);
    m->cp = (size_t)(m->context + m->B);
    pred = m->d32[m->cp] >> 20;
  } else {
    m->d8[m->cp] = NEX(m->d8[m->cp], y);
    m->B += (y && m->B > 0);
    m->cp = (size_t)(m->context + m->B);
    pred = sm32_p(m->map, y, m->d8[m->cp], a);
  }
  sink_add(&s, (orc_p8_stretch(pred) * mul) / div);
  sink_add(&s, ((pred - 2048) * mul) / (div * 2));
  m->bcount++; m->B += m->B + 1;
  if (m->bcount == m->btotal) m->bcount = m->B = 0;
  return s.n;
}

/* ---- paq8's process-global pseudo-random generator (reference :152-165), used by ContextMap's state decay. Every
 * ContextMap of every sub-model draws from the ONE sequence, and only when a state >= 204 comes up: the order of
 * draws is data-dependent and global. A device stage has to walk those contexts in the reference's order. ---- */
typedef struct { uint32_t table[64]; int i; } P8Rnd;
static P8Rnd g_rnd;
void orc_p8_rnd_reset(void) {
  g_rnd.table[0] = 123456789;
  g_rnd.table[1] = 987654321;
  for (int j = 0; j < 62; ++j) g_rnd.table[j + 2] = g_rnd.table[j + 1] * 11 + g_rnd.table[j] * 23 / 16;
  g_rnd.i = 0;
}
uint32_t orc_p8_rnd_next(void) {
  ++g_rnd.i;
  return g_rnd.table[g_rnd.i & 63] = g_rnd.table[(g_rnd.i - 24) & 63] ^ g_rnd.table[(g_rnd.i - 55) & 63];
}

/* ---- StateMap (u16, :623-645) local copy ---- */
typedef struct { int cxt; uint16_t t[256]; } SM16;
static void sm16_init(SM16* s) {
  s->cxt = 0;
  for (int i = 0; i < 256; ++i) {
    int n0 = NEX(i, 2), n1 = NEX(i, 3);
    if (n0 == 0) n1 *= 64;
    if (n1 == 0) n0 *= 64;
    s->t[i] = (uint16_t)(65536 * (n1 + 1) / (n0 + n1 + 2));
  }
}
static int sm16_p(SM16* s, int y, int cx) {
  s->t[s->cxt] += ((y << 16) - s->t[s->cxt] + 128) >> 8;
  return s->t[s->cxt = cx] >> 4;
}

/* ---- ContextMap (:1010-1145): same 64-byte bucket as ContextMap2 (chk[7], last, bh[7][7]) ---- */
typedef struct {
  int C, cn, hashbits;
  uint32_t mask;
  uint8_t* table;
  uint32_t *cp, *cp0, *runp, *cxt;
  uint16_t* chk;
  SM16* sm;
  CM2 find_view;  /* bucket_find() only needs table; reuse it through a view */
} CM1;
CM1* orc_p8_cm_new(uint64_t size_bytes, int count) {
  ilog_init();
  CM1* c = (CM1*)calloc(1, sizeof *c);
  c->C = count;
  const uint64_t nb = size_bytes >> 6;
  c->mask = (uint32_t)(nb - 1);
  c->hashbits = (int)ilog2u(c->mask + 1);
  c->table = (uint8_t*)calloc(nb, B_SIZE);
  c->find_view.table = c->table;
  c->cp = (uint32_t*)malloc(count * 4); c->cp0 = (uint32_t*)malloc(count * 4); c->runp = (uint32_t*)malloc(count * 4);
  c->cxt = (uint32_t*)calloc(count, 4); c->chk = (uint16_t*)calloc(count, 2);
  c->sm = (SM16*)malloc(count * sizeof(SM16));
  for (int i = 0; i < count; ++i) {
    sm16_init(&c->sm[i]);
    c->cp0[i] = c->cp[i] = B_STATE;  /* &t[0].bh[0][0] */
    c->runp[i] = c->cp[i] + 3;
  }
  return c;
}
void orc_p8_cm_free(CM1* c) {
  if (!c) return;
  free(c->table); free(c->cp); free(c->cp0); free(c->runp); free(c->cxt); free(c->chk); free(c->sm); free(c);
}
/* One coded bit: set() at bpos == 0 (the callers' pattern), then mix1(m, c0, bpos, buf(1), y) :1072-1145 */
int orc_p8_cm_step(CM1* c, int y1, int bp, int c0, int c1, const uint64_t* ctx, int nset, int16_t* out, int* nout) {
  Sink s = {out, 0};
  if (bp == 0)
    for (int i = 0; i < nset; ++i) {  /* ContextMap::set :1064-1069 */
      const uint64_t h = orc_p8_hash2(ctx[i], (uint64_t)c->cn);
      c->cxt[c->cn] = orc_p8_finalize64(h, c->hashbits);
      c->chk[c->cn] = (uint16_t)(orc_p8_checksum64(h, c->hashbits, 16) & 0xffff);
      c->cn++;
    }
  uint8_t* T = c->table;
  int result = 0;
  for (int i = 0; i < c->cn; ++i) {
    if (c->cp[i] != NIL) {
      int ns = NEX(T[c->cp[i]], y1);
      if (ns >= 204 && (uint32_t)(orc_p8_rnd_next() << ((452 - ns) >> 3))) ns -= 4;  /* the draw happens only for ns >= 204 */
      T[c->cp[i]] = (uint8_t)ns;
    }
    if (bp > 1 && T[c->runp[i]] == 0) c->cp[i] = NIL;
    else {
      switch (bp) {
        case 1: case 3: case 6: c->cp[i] = c->cp0[i] + 1 + (c0 & 1); break;
        case 4: case 7: c->cp[i] = c->cp0[i] + 3 + (c0 & 3); break;
        case 2: case 5: c->cp0[i] = c->cp[i] = bucket_find(&c->find_view, (c->cxt[i] + (uint32_t)c0) & c->mask, c->chk[i]); break;
        default: {
          const uint16_t checksum = c->chk[i];
          const uint32_t cx = c->cxt[i];
          c->cp0[i] = c->cp[i] = bucket_find(&c->find_view, (cx + (uint32_t)c0) & c->mask, checksum);
          uint8_t* s0 = T + c->cp0[i];
          if (s0[3] == 2) {
            const int cc = s0[4] + 256;
            uint8_t* p = T + bucket_find(&c->find_view, (cx + (cc >> 6)) & c->mask, checksum);
            p[0] = (uint8_t)(1 + ((cc >> 5) & 1));
            p[1 + ((cc >> 5) & 1)] = (uint8_t)(1 + ((cc >> 4) & 1));
            p[3 + ((cc >> 4) & 3)] = (uint8_t)(1 + ((cc >> 3) & 1));
            p = T + bucket_find(&c->find_view, (cx + (cc >> 3)) & c->mask, checksum);
            p[0] = (uint8_t)(1 + ((cc >> 2) & 1));
            p[1 + ((cc >> 2) & 1)] = (uint8_t)(1 + ((cc >> 1) & 1));
            p[3 + ((cc >> 1) & 3)] = (uint8_t)(1 + (cc & 1));
            s0[6] = 0;
          }
          uint8_t* rp = T + c->runp[i];  /* run count of the previous context */
          if (rp[0] == 0) { rp[0] = 2; rp[1] = (uint8_t)c1; }
          else if (rp[1] != c1) { rp[0] = 1; rp[1] = (uint8_t)c1; }
          else if (rp[0] < 254) rp[0] += 2;
          else if (rp[0] == 255) rp[0] = 128;
          c->runp[i] = c->cp0[i] + 3;
        } break;
      }
    }
    const uint8_t* rp = T + c->runp[i];
    const int rc = rp[0];
    if ((rp[1] + 256) >> (8 - bp) == c0) {
      const int b = ((rp[1] >> (7 - bp)) & 1) * 2 - 1;
      sink_add(&s, b * (g_ilog[rc + 1] << (2 + (~rc & 1))));
    } else sink_add(&s, 0);
    const int st8 = c->cp[i] != NIL ? T[c->cp[i]] : 0;
    const int p1 = sm16_p(&c->sm[i], y1, st8);
    const int st = (orc_p8_stretch(p1) + (1 << 1)) >> 2;
    sink_add(&s, st);
    sink_add(&s, (p1 - 2047 + (1 << 2)) >> 3);
    const int n0 = -!NEX(st8, 2), n1 = -!NEX(st8, 3);
    sink_add(&s, st * abs(n1 - n0));
    const int p0 = 4095 - p1;
    sink_add(&s, ((p1 & n0) - (p0 & n1) + (1 << 3)) >> 4);
    result += st8 > 0;
  }
  if (bp == 7) c->cn = 0;
  *nout = s.n;
  return result;
}

/* ---- BH<4> (:778-813) and RunContextMap (:857-889) ---- */
typedef struct { uint8_t* t; uint32_t mask; int hashbits; uint32_t cp; } RCM;  /* cp: byte offset of the current element + 1 */
static uint32_t bh4_get(RCM* r, uint64_t ctx) {  /* BH<4>::operator[] -> offset of element byte 1 */
  enum { Bsz = 4, Mlim = 8 };
  const uint16_t chk = (uint16_t)(orc_p8_checksum64(ctx, r->hashbits, 16) & 0xffff);
  const uint32_t i = (orc_p8_finalize64(ctx, r->hashbits) * Mlim) & r->mask;
  uint8_t* t = r->t;
  int j;
  uint32_t p = 0;
  for (j = 0; j < Mlim; ++j) {
    p = (i + j) * Bsz;
    uint16_t cur;
    memcpy(&cur, t + p, 2);
    if (t[p + 2] == 0) { memcpy(t + p, &chk, 2); break; }  /* empty slot */
    if (cur == chk) break;                                    /* found */
  }
  if (j == 0) return p + 1;  /* front */
  uint8_t tmp[Bsz];
  if (j == Mlim) {
    --j;
    memset(tmp, 0, Bsz);
    memcpy(tmp, &chk, 2);
    if (Mlim > 2 && t[(i + j) * Bsz + 2] > t[(i + j - 1) * Bsz + 2]) --j;
  } else memcpy(tmp, t + p, Bsz);
  memmove(t + (i + 1) * Bsz, t + i * Bsz, (size_t)j * Bsz);
  memcpy(t + i * Bsz, tmp, Bsz);
  return i * Bsz + 1;
}
RCM* orc_p8_rcm_new(int m) {
  ilog_init();
  RCM* r = (RCM*)calloc(1, sizeof *r);
  const int n = m / 4;             /* BH<4> t(m/4): i elements of B bytes */
  r->t = (uint8_t*)calloc((size_t)n * 4 + 64, 1);
  r->mask = (uint32_t)(n - 1);
  r->hashbits = (int)ilog2u(r->mask + 1);
  r->cp = bh4_get(r, 0) + 1;       /* cp = t[0] + 1 */
  return r;
}
void orc_p8_rcm_set(RCM* r, uint64_t cx, int c1) {
  uint8_t* cp = r->t + r->cp;
  if (cp[0] == 0 || cp[1] != c1) { cp[0] = 1; cp[1] = (uint8_t)c1; }
  else if (cp[0] < 255) ++cp[0];
  r->cp = bh4_get(r, cx) + 1;
}
int orc_p8_rcm_mix(RCM* r, int bpos, int c0, int16_t* out) {
  const uint8_t* cp = r->t + r->cp;
  out[0] = (int16_t)(((cp[1] + 256) >> (8 - bpos)) == c0 ? (((cp[1] >> (7 - bpos)) & 1) * 2 - 1) * g_ilog[cp[0] + 1] * 8 : 0);
  return cp[0] != 0;
}

/* the nex() state table as data, for tests that hand it to the device building blocks (cmx_p8cm2_create) */
void orc_p8_state_table(uint8_t* out1024) { memcpy(out1024, P8_STATE, 1024); }
