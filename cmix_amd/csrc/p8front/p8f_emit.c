/* p8f_emit.c -- recording implementation of the table interface (p8f_emit.h) + the helpers the sub-models share:
 * hash functions (reference src/models/paq8.cpp:714-776), ilog (:253-266), squash / stretch tables, and the recorders of
 * the tables that are walked by one lane on the device (RunContextMap over BH<4> :778-889, im4bitModel's HashTable<16>, jpegModel's BH<9>). */
#include "p8f_emit.h"

#include <stdio.h>
#include <stdlib.h>
#include <pthread.h>
#include <string.h>

#include "p8f_tables.h"

__thread P8Emit* p8f_cur;

/* ---- helpers ---- */
int p8f_squash(int d) { return d > 2047 ? 4095 : d < -2047 ? 0 : P8_SQUASH[d + 2048]; }
int p8f_stretch(int p) { return P8_STRETCH[p]; }

static uint8_t g_ilog[65536];
static pthread_once_t g_ilog_once = PTHREAD_ONCE_INIT;   /* several streams are created concurrently (one thread per GPU) */
static void ilog_fill(void) {  /* Ilog::Ilog :260-266 */
  uint32_t x = 14155776;
  for (int i = 2; i < 65536; ++i) {
    x += 774541002 / (i * 2 - 1);
    g_ilog[i] = (uint8_t)(x >> 24);
  }
}
static void ilog_init(void) { pthread_once(&g_ilog_once, ilog_fill); }
int p8f_ilog(int x) { ilog_init(); return g_ilog[x & 0xffff]; }
const uint8_t* p8f_ilog_table(void) { ilog_init(); return g_ilog; }
static unsigned ilog2u(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }

#define PHI64 0x9E3779B97F4A7C15ull
#define MUL64_1 0x993DDEFFB1462949ull
#define MUL64_2 0xE9C91DC159AB0D2Dull
#define MUL64_3 0x83D6A14F1B0CED73ull
#define MUL64_4 0xA14F1B0CED5A841Full
#define MUL64_5 0xC0E51314A614F4EFull
#define MUL64_6 0xDA9CC2600AE45A27ull
#define MUL64_7 0x826797AA04A65737ull
uint64_t p8f_hash2(uint64_t a, uint64_t b) { return (a + 1) * PHI64 + (b + 1) * MUL64_1; }
uint64_t p8f_combine64(uint64_t seed, uint64_t x) { return (seed + x + 1) * PHI64; }
uint32_t p8f_finalize64(uint64_t h, int bits) { return (uint32_t)(h >> (64 - bits)); }
uint64_t p8f_checksum64(uint64_t h, int hashbits, int checksumbits) { return h >> (64 - hashbits - checksumbits); }
static uint64_t hashn(int n, const uint64_t* v) {  /* hash(a, b, ...) :724-758 */
  static const uint64_t mul[8] = {PHI64, MUL64_1, MUL64_2, MUL64_3, MUL64_4, MUL64_5, MUL64_6, MUL64_7};
  uint64_t h = 0;
  for (int i = 0; i < n; ++i) h += (v[i] + 1) * mul[i];
  return h;
}
uint64_t p8f_hash3(uint64_t a, uint64_t b, uint64_t c) { return hashn(3, (const uint64_t[]){a, b, c}); }
uint64_t p8f_hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d) { return hashn(4, (const uint64_t[]){a, b, c, d}); }
uint64_t p8f_hash5(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e) { return hashn(5, (const uint64_t[]){a, b, c, d, e}); }
uint64_t p8f_hash6(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e, uint64_t f) { return hashn(6, (const uint64_t[]){a, b, c, d, e, f}); }

static void fail(const char* what) {
  if (p8f_cur && !p8f_cur->err) { p8f_cur->err = 1; fprintf(stderr, "paq8 front end: %s\n", what); }
}

void p8f_emit_begin_step(P8Emit* e, int16_t* in_base, P8Chunk* chunk, size_t byte_row, size_t step_row, int full) {
  e->in_base = in_base; e->chunk = chunk; e->byte_row = byte_row; e->step_row = step_row; e->full = full;
  e->fam_calls = e->cm2_calls = 0;
  e->model = 0; e->step_model = 0;
  if (chunk) memset(chunk->ops + step_row * P8_NLANE, 0, P8_NLANE * sizeof(uint32_t));
}
void p8f_emit_model(P8Emit* e, int m) { e->model = m; }
void p8f_emit_step_model(P8Emit* e, int m) {
  e->step_model = m;
  if (m && e->chunk && e->chunk->xops) memset(e->chunk->xops + e->step_row * P8_XL_NLANE, 0, P8_XL_NLANE * sizeof(uint32_t));
}
/* discovery: input `compact` (add() order) of the current step's model sits at `pos` of the 1552-vector */
static void xmap(P8Emit* e, int compact, int pos, int n) {
  if (!e->xdiscovering || !e->step_model) return;
  P8XLayout* X = &e->L.xl[e->step_model - 1];
  for (int j = 0; j < n; j++) if (compact + j >= 0 && compact + j < P8_NX) X->map[compact + j] = (int16_t)(pos + j);
}

static int claim(P8Emit* e, const int16_t* out, int n) {  /* discovery: remember who produces which input positions */
  const int off = (int)(out - e->in_base);
  if (off < 0 || off + n > P8_NX) { fail("input position out of range"); return 0; }
  if (e->discovering && !e->model && !e->step_model) memset((e->full ? e->claimed : e->claimed0) + off, 1, (size_t)n);
  return off;
}

/* ---- ContextMap family (:1010-1145): set() hashes, mix() is the device's ---- */
typedef struct CM1 { int inst, first, count, cn, hashbits; uint64_t size; int order_idx; int model; } CM1;
CM1* p8f_cm_new(uint64_t size_bytes, int count) {
  CM1* c = (CM1*)calloc(1, sizeof *c);
  c->inst = -1; c->count = count; c->size = size_bytes; c->order_idx = -1;
  c->model = p8f_cur ? p8f_cur->model : 0;
  if (c->model && p8f_cur->xdiscovering) {   /* a model's own ContextMap: its slots come first in the model's family, whenever it is first called */
    P8XLayout* X = &p8f_cur->L.xl[c->model - 1];
    X->fam_size = size_bytes; X->fam_count = count; if (X->nslots < count) X->nslots = count;
  }
  c->hashbits = (int)ilog2u((unsigned)(size_bytes >> 6));
  return c;
}
void p8f_cm_free(CM1* c) { free(c); }
/* the context at position idx of this instance is hash(seed, order-N map's return value): sparseModel :4513 */
void p8f_cm_order_slot(CM1* c, int idx, uint64_t seed) {
  P8Emit* e = p8f_cur;
  if (c->order_idx >= 0 || !e->discovering) { c->order_idx = idx; return; }
  c->order_idx = idx;
  for (int o = 0; o <= P8_ORDER_MAX; ++o) {
    const uint64_t h = p8f_hash2(p8f_hash2(seed, (uint64_t)o), (uint64_t)idx);
    e->L.order_ctx[o] = p8f_finalize64(h, c->hashbits);
    e->L.order_chk[o] = (uint16_t)(p8f_checksum64(h, c->hashbits, 16) & 0xffff);
  }
}
/* an image model's one ContextMap: its contexts go to the chunk's xfam_* arrays, its offsets to the model's P8XLayout */
static int xcm_step(CM1* c, int bp, const uint64_t* ctx, int nset, int16_t* out, int* nout) {
  P8Emit* e = p8f_cur;
  P8XLayout* X = &e->L.xl[c->model - 1];
  if (e->model != c->model) { fail("an image model's ContextMap called outside its model"); return 0; }
  if (c->inst < 0) {
    c->inst = 0;
    if (X->fam_size != c->size) { fail("an image model's ContextMap changed"); return 0; }
  }
  if (bp == 0) {
    if (c->cn + nset > c->count || c->cn + nset > P8_XL_MAXS - 2) { fail("too many contexts set (image model)"); return 0; }
    if (c->cn + nset > X->fam_count) { fail("an image model's ContextMap got more contexts than in the layout pass"); return 0; }
    for (int i = 0; i < nset; ++i, ++c->cn) {
      const uint64_t h = p8f_hash2(ctx[i], (uint64_t)c->cn);
      if (e->chunk && e->chunk->xfam_ctx) {
        e->chunk->xfam_ctx[e->byte_row * (size_t)P8_XL_MAXS + (size_t)c->cn] = p8f_finalize64(h, c->hashbits);
        e->chunk->xfam_chk[e->byte_row * (size_t)P8_XL_MAXS + (size_t)c->cn] = (uint16_t)(p8f_checksum64(h, c->hashbits, 16) & 0xffff);
      }
    }
    /* how many contexts the byte has set (im8bitModel: 25 for a grayscale image, 52 for a palette one): the row's last cell */
    if (e->chunk && e->chunk->xfam_ctx) {
      e->chunk->xfam_ctx[e->byte_row * (size_t)P8_XL_MAXS + (size_t)(P8_XL_MAXS - 2)] = 0;
      e->chunk->xfam_ctx[e->byte_row * (size_t)P8_XL_MAXS + (size_t)(P8_XL_MAXS - 1)] = (uint32_t)c->cn;
    }
  }
  const int n = 5 * c->cn;
  if (n) {
    const int off = claim(e, out, n);
    if (e->xdiscovering) {
      for (int i = 0; i < c->cn; ++i) X->fam_off[i] = (int16_t)(off + 5 * i);
      xmap(e, off, off, n);
      X->opt_lo = off; if (n > X->opt_n) X->opt_n = n;
    }
  }
  if (bp == 7) c->cn = 0;
  *nout = n;
  return 0;
}
/* a generic ContextMap called in a model's step (recordModel under the audio models): it joins the model's family -- its contexts also go
 * to the model's family row, at the slots the layout pass gave the instance -- and its inputs stay at their generic positions */
static int gcm_in_model(CM1* c, int bp, const uint64_t* ctx, int nset, int16_t* out, int* nout) {
  P8Emit* e = p8f_cur;
  P8Layout* L = &e->L;
  P8XLayout* X = &L->xl[e->step_model - 1];
  if (c->inst < 0) { fail("a generic ContextMap first called inside a model's step"); return 0; }
  int g = -1;
  for (int k = 0; k < X->ngen; k++) if (X->gen_inst[k] == c->inst) g = k;
  if (g < 0) {
    if (!e->xdiscovering || X->ngen >= P8_XL_MAXG) { fail("a generic ContextMap the model's layout does not know"); return 0; }
    g = X->ngen++;
    X->gen_inst[g] = c->inst;
    X->gen_first[g] = g ? X->gen_first[g - 1] + L->fam_count[X->gen_inst[g - 1]] : X->fam_count;
    X->nslots = X->gen_first[g] + L->fam_count[c->inst];
    if (X->nslots > P8_XL_MAXS - 2) { fail("a model's ContextMap family is too large"); return 0; }
  }
  int first = 0;
  for (int i = 0; i < c->inst; ++i) first += L->fam_count[i];
  if (bp == 0) {
    if (c->cn + nset > L->fam_count[c->inst]) { fail("too many contexts set"); return 0; }
    for (int i = 0; i < nset; ++i, ++c->cn) {
      const uint64_t h = p8f_hash2(ctx[i], (uint64_t)c->cn);
      if (e->chunk && e->chunk->xfam_ctx) {
        const size_t at = e->byte_row * (size_t)P8_XL_MAXS + (size_t)(X->gen_first[g] + c->cn);
        e->chunk->xfam_ctx[at] = p8f_finalize64(h, c->hashbits);
        e->chunk->xfam_chk[at] = (uint16_t)(p8f_checksum64(h, c->hashbits, 16) & 0xffff);
      }
    }
  }
  const int n = 5 * c->cn;
  if (n) {
    const int compact = (int)(out - e->in_base);
    if (e->xdiscovering) for (int i = 0; i < c->cn; ++i) { xmap(e, compact + 5 * i, L->fam_off[first + i], 5); X->fam_off[X->gen_first[g] + i] = L->fam_off[first + i]; }
  }
  if (bp == 7) c->cn = 0;
  *nout = n;
  return 0;
}
int p8f_cm_step(CM1* c, int y1, int bp, int c0, int c1, const uint64_t* ctx, int nset, int16_t* out, int* nout) {
  P8Emit* e = p8f_cur;
  (void)y1; (void)c0; (void)c1;
  if (c->model) return xcm_step(c, bp, ctx, nset, out, nout);
  if (e->step_model) return gcm_in_model(c, bp, ctx, nset, out, nout);
  P8Layout* L = &e->L;
  const int k = e->fam_calls++;
  if (c->inst < 0) {  /* first call: the instance takes the next place in the walk */
    c->inst = k;
    if (e->discovering) {
      if (k >= P8_FAM_MAXI) { fail("ContextMap family too large"); return 0; }
      L->fam_size[k] = c->size; L->fam_count[k] = 0; L->fam_ninst = k + 1;
    } else if (k >= L->fam_ninst || L->fam_size[k] != c->size || L->fam_count[k] > c->count) { fail("ContextMap family changed"); return 0; }
  }
  if (c->inst != k) { fail("ContextMap instances called out of order"); return 0; }
  int first = 0;
  for (int i = 0; i < k; ++i) first += L->fam_count[i];
  c->first = first;
  if (bp == 0) {
    if (c->cn + nset > c->count) { fail("too many contexts set"); return 0; }
    if (e->discovering) {   /* a model may use fewer contexts than it allocates: the contexts it sets are the instance */
      if (first + c->cn + nset > P8_FAM_MAXS) { fail("ContextMap family too large"); return 0; }
      L->fam_count[k] = c->cn + nset; L->fam_slots = first + c->cn + nset;
    } else if (c->cn + nset > L->fam_count[k]) { fail("a ContextMap got more contexts than in the layout pass"); return 0; }
    for (int i = 0; i < nset; ++i, ++c->cn) {
      const uint64_t h = p8f_hash2(ctx[i], (uint64_t)c->cn);
      if (e->chunk) {
        e->chunk->fam_ctx[e->byte_row * (size_t)L->fam_slots + (size_t)(first + c->cn)] = p8f_finalize64(h, c->hashbits);
        e->chunk->fam_chk[e->byte_row * (size_t)L->fam_slots + (size_t)(first + c->cn)] = (uint16_t)(p8f_checksum64(h, c->hashbits, 16) & 0xffff);
      }
    }
    if (nset && c->order_idx >= 0 && e->discovering) L->order_slot = first + c->order_idx;
  }
  if (c->cn != 0 && c->cn != L->fam_count[k]) { fail("a ContextMap got fewer contexts than in the layout pass"); return 0; }
  const int n = 5 * c->cn;
  if (n) {
    const int off = claim(e, out, n);
    if (e->discovering) for (int i = 0; i < c->cn; ++i) L->fam_off[first + i] = (int16_t)(off + 5 * i);
  }
  if (bp == 7) c->cn = 0;
  *nout = n;
  return 0;
}

/* ---- ContextMap2 (:1164-1358) ---- */
typedef struct CM2 { int k, count, hashbits, index; uint64_t size; } CM2;
CM2* p8f_cm2_new(uint64_t size_bytes, uint32_t count) {
  CM2* c = (CM2*)calloc(1, sizeof *c);
  c->k = -1; c->count = (int)count; c->size = size_bytes;
  c->hashbits = (int)ilog2u((unsigned)(size_bytes >> 6));
  return c;
}
void p8f_cm2_free(CM2* c) { free(c); }
int p8f_cm2_step(CM2* c, int y_prev, int bpos, const uint64_t* ctx, int nset, int16_t* out, int* nout) {
  P8Emit* e = p8f_cur;
  (void)y_prev;
  P8Layout* L = &e->L;
  const int k = e->cm2_calls++;
  if (c->k < 0) {
    c->k = k;
    if (e->discovering) {
      if (k >= P8_NCM2) { fail("too many ContextMap2 instances"); return 0; }
      L->cm2_size[k] = c->size; L->cm2_count[k] = 0;
    } else if (k >= P8_NCM2 || L->cm2_size[k] != c->size || L->cm2_count[k] > c->count) { fail("ContextMap2 instances changed"); return 0; }
  }
  if (c->k != k) { fail("ContextMap2 instances called out of order"); return 0; }
  if (bpos == 0) {
    if (c->index + nset > c->count) { fail("too many contexts set"); return 0; }
    if (e->discovering) L->cm2_count[k] = c->index + nset;
    else if (c->index + nset > L->cm2_count[k]) { fail("a ContextMap2 got more contexts than in the layout pass"); return 0; }
    const size_t C = (size_t)L->cm2_count[k];
    for (int i = 0; i < nset; ++i, ++c->index) {
      const uint64_t h = p8f_hash2(ctx[i], (uint64_t)c->index);
      if (e->chunk) {
        e->chunk->cm2_ctx[k][e->byte_row * C + (size_t)c->index] = p8f_finalize64(h, c->hashbits);
        e->chunk->cm2_chk[k][e->byte_row * C + (size_t)c->index] = (uint16_t)(p8f_checksum64(h, c->hashbits, 16) & 0xffff);
      }
    }
  }
  if (c->index != 0 && c->index != L->cm2_count[k]) { fail("a ContextMap2 got fewer contexts than in the layout pass"); return 0; }
  const int n = 7 * c->index;
  if (n) {
    const int off = claim(e, out, n);
    if (e->discovering) L->cm2_off[k] = (int16_t)off;
  }
  if (bpos == 7) c->index = 0;
  *nout = n;
  return 0;   /* ContextMap2::mix's return value (the "order") exists on the device only */
}

/* ---- small maps: one lane each ---- */
static int new_lane(int kind, uint32_t cells, uint32_t init) {
  P8Emit* e = p8f_cur;
  if (e->model) {   /* a map of an image model: the model's own table */
    const int m = e->model - 1, l = e->xlane_objs[m]++;
    if (l >= P8_XL_NLANE) { fail("too many small maps in an image model"); return 0; }
    P8XLayout* X = &e->L.xl[m];
    if (e->xdiscovering) {
      P8Lane* q = &X->lane[l];
      memset(q, 0, sizeof *q);
      q->kind = (uint8_t)kind; q->cells = cells; q->init = init; q->off = -1;
      X->nlanes = l + 1;
    } else if (X->lane[l].kind != kind || X->lane[l].cells != cells) fail("small maps of an image model changed");
    return l;
  }
  const int l = e->lane_objs++;
  if (l >= P8_NLANE) { fail("too many small maps"); return 0; }
  if (e->discovering) {
    P8Lane* q = &e->L.lane[l];
    memset(q, 0, sizeof *q);
    q->kind = (uint8_t)kind; q->cells = cells; q->init = init; q->off = -1;
    e->L.nlanes = l + 1;
  } else if (e->L.lane[l].kind != kind || e->L.lane[l].cells != cells) fail("small maps changed");
  return l;
}
static void lane_out(int model, int l, const int16_t* out, int nout, int a, int mul, int div, int limit, int bpc) {
  P8Emit* e = p8f_cur;
  const int off = claim(e, out, nout);
  if (model) {
    if (e->model != model) { fail("a map of an image model called outside its model"); return; }
    if (!e->xdiscovering) return;
    P8Lane* q = &e->L.xl[model - 1].lane[l];
    q->off = (int16_t)off; q->nout = (int16_t)nout; q->a = (uint8_t)a; q->mul = (uint8_t)mul; q->div = (uint8_t)div; q->limit = (uint16_t)limit; q->bits_per_ctx = (uint16_t)bpc;
    xmap(e, off, off, nout);
    return;
  }
  if (e->step_model) {   /* a generic map called in a model's step: it runs at its generic place */
    if (e->xdiscovering) { e->L.lane[l].modes |= 1u << e->step_model; xmap(e, off, e->L.lane[l].off, nout); }
    return;
  }
  if (!e->discovering) return;
  P8Lane* q = &e->L.lane[l];
  if (e->full) { q->off = (int16_t)off; q->nout = (int16_t)nout; q->a = (uint8_t)a; q->mul = (uint8_t)mul; q->div = (uint8_t)div; q->limit = (uint16_t)limit; q->bits_per_ctx = (uint16_t)bpc; }
  else e->lane_off0[l] = (int16_t)off;
}
static void put_op(int model, int l, uint32_t op) {
  P8Emit* e = p8f_cur;
  if (!e->chunk) return;
  if (model) { if (e->chunk->xops) e->chunk->xops[e->step_row * P8_XL_NLANE + (size_t)l] = op; }
  else e->chunk->ops[e->step_row * P8_NLANE + (size_t)l] = op;
}

typedef struct DMap { int lane, kind, mask, maskbits, stride, btotal, pending, order, model; uint32_t ctx; } DMap;
DMap* p8f_dmap_new(int kind, int bits_of_context, int bits_per_context, int rate) {
  DMap* m = (DMap*)calloc(1, sizeof *m);
  m->model = p8f_cur ? p8f_cur->model : 0;
  m->kind = kind; m->mask = (1 << bits_of_context) - 1; m->maskbits = bits_of_context;
  m->stride = (1 << bits_per_context) - 1; m->btotal = bits_per_context;
  const uint32_t cells = ((uint32_t)1 << bits_of_context) * (uint32_t)m->stride;
  if (kind == 0) m->lane = new_lane(P8L_SSCM, cells, 0x7FFF);
  else if (kind == 1) m->lane = new_lane(P8L_STAT, cells, (0x7FFu << 20) | (uint32_t)(rate < 1023 ? rate : 1023));
  else m->lane = new_lane(P8L_IND, cells, 0);
  return m;
}
void p8f_dmap_set_direct(DMap* m, uint32_t ctx) { m->ctx = (ctx & (uint32_t)m->mask) * (uint32_t)m->stride; m->pending = 1; m->order = 0; }
void p8f_dmap_set(DMap* m, uint64_t ctx) { m->ctx = (p8f_finalize64(ctx, m->maskbits) & (uint32_t)m->mask) * (uint32_t)m->stride; m->pending = 1; m->order = 0; }
void p8f_dmap_set_order(DMap* m) { m->ctx = 0; m->pending = 1; m->order = 1; }   /* set(order-N map's return value) */
int p8f_dmap_mix(DMap* m, int y, int a, int mul, int div, int16_t* out) {
  (void)y;
  lane_out(m->model, m->lane, out, 2, a > 255 ? 255 : a, mul, div, a < 0x3FF ? a : 0x3FF, m->btotal);
  uint32_t op = P8OP_MIX;
  if (m->pending) op |= P8OP_SET | (m->order ? P8OP_ORDER : 0) | (m->ctx & P8OP_CTX);
  m->pending = 0;
  put_op(m->model, m->lane, op);
  return 2;
}
/* a step in which the model does not call mix(): the map is untouched, its two inputs are 0 (SparseMatchModel :3817-3822) */
int p8f_dmap_skip(DMap* m, int a, int mul, int div, int16_t* out) {
  lane_out(m->model, m->lane, out, 2, a > 255 ? 255 : a, mul, div, a < 0x3FF ? a : 0x3FF, m->btotal);
  put_op(m->model, m->lane, 0);
  return 2;
}

typedef struct P8fStateMap32 { int lane; } P8fStateMap32;
P8fStateMap32* p8f_statemap32_new(int n) {
  P8fStateMap32* s = (P8fStateMap32*)calloc(1, sizeof *s);
  s->lane = new_lane(P8L_SM32, (uint32_t)n, 1u << 31);
  return s;
}
/* StateMap32::p(cx, 1023) read out as one input (stretch(p) + 1) >> 1, or 0 when zero is set */
void p8f_statemap32_emit(P8fStateMap32* s, int cx, int zero, int16_t* out) {
  lane_out(0, s->lane, out, 1, 0, 1, 1, 1023, 0);
  put_op(0, s->lane, P8OP_MIX | P8OP_SET | (zero ? P8OP_ZERO : 0) | ((uint32_t)cx & P8OP_CTX));
}

/* picModel's three maps (:3844-3864): a bit-history byte per context in one shared array + a u16 StateMap each */
typedef struct P8fPic { int lane[3]; } P8fPic;
P8fPic* p8f_pic_new(void) {
  P8fPic* p = (P8fPic*)calloc(1, sizeof *p);
  for (int i = 0; i < 3; ++i) p->lane[i] = new_lane(P8L_PIC, 0x10200, 0);
  return p;
}
void p8f_pic_emit(P8fPic* p, int i, int cxt, int first, int16_t* out) {
  lane_out(0, p->lane[i], out, 1, i, 1, 1, 0, 0);   /* a = which of the three */
  put_op(0, p->lane[i], P8OP_MIX | P8OP_SET | (first ? P8OP_ZERO : 0) | ((uint32_t)cxt & P8OP_CTX));
}

/* im1bitModel's eleven maps (:4634-4673): a bit-history byte per context in one shared array -- disjoint context ranges, so one array per map --
 * and a u16 StateMap each; at the model's first call all of them sit on cell 0, which belongs to map 0's range: it takes their updates too */
typedef struct P8fBitMaps { int n, pair, lane[16]; } P8fBitMaps;
/* pair: maps pair, pair + 1 have overlapping context ranges (im1bitModel's cxt[6] = 0xC00 + up to 0xC3F and cxt[7] = 0x1000 + ...): one lane runs
 * both on one array, in the reference's order (P8L_PIC2); -1: none */
P8fBitMaps* p8f_bitmaps_new(int n, uint32_t cells, int pair) {
  P8fBitMaps* p = (P8fBitMaps*)calloc(1, sizeof *p);
  p->n = n; p->pair = pair;
  for (int i = 0; i < n; ++i) p->lane[i] = new_lane(i == pair ? P8L_PIC2 : i == pair + 1 && pair >= 0 ? P8L_NONE : P8L_PIC, cells, 0);
  return p;
}
void p8f_bitmaps_emit(P8fBitMaps* p, int i, int cxt, int first, int16_t* out) {
  P8Emit* e = p8f_cur;
  if (p->pair >= 0 && i == p->pair + 1) { (void)claim(e, out, 1); put_op(e->model, p->lane[i], (uint32_t)cxt & P8OP_CTX); return; }   /* the pair's second context: only its op word */
  lane_out(e->model, p->lane[i], out, i == p->pair ? 2 : 1, i, 1, 1, p->n - 1, 0);   /* a = which map; limit = the extra updates of cell 0 at the first call */
  put_op(e->model, p->lane[i], P8OP_MIX | P8OP_SET | (first ? P8OP_ZERO : 0) | ((uint32_t)cxt & P8OP_CTX));
}

/* im4bitModel's HashTable<16> with its 14 contexts (:4675-4742): one worker lane + 14 lanes that only carry the nibble's hashed contexts */
typedef struct P8fHt16 { int lane[15]; int hashbits; } P8fHt16;
P8fHt16* p8f_ht16_new(uint32_t table_bytes) {
  P8fHt16* p = (P8fHt16*)calloc(1, sizeof *p);
  p->hashbits = (int)ilog2u(table_bytes);
  p->lane[0] = new_lane(P8L_HT16, table_bytes, 0);
  for (int i = 1; i < 15; ++i) p->lane[i] = new_lane(P8L_NONE, 0, 0);
  return p;
}
/* keys: the 14 contexts of the nibble (NULL between nibble boundaries) */
int p8f_ht16_step(P8fHt16* p, const uint64_t* keys, int16_t* out) {
  P8Emit* e = p8f_cur;
  lane_out(e->model, p->lane[0], out, 42, 0, 1, 1, 0, 0);
  put_op(e->model, p->lane[0], P8OP_MIX);
  if (keys)
    for (int i = 0; i < 14; ++i) {   /* HashTable<B>::operator[] :841-843: 8-bit checksum, item index (the table has 2^hashbits bytes, 16 per item) */
      const uint32_t chk = (uint32_t)(p8f_checksum64(keys[i], p->hashbits, 8) & 0xff);
      const uint32_t item = (uint32_t)((((uint64_t)p8f_finalize64(keys[i], p->hashbits) * 16) & (((uint64_t)1 << p->hashbits) - 1)) >> 4);
      put_op(e->model, p->lane[1 + i], (chk << 22) | item);
    }
  return 42;
}
/* a StateMap32 read out as stretch(p) >> 1 (im4bitModel :4735) */
typedef struct P8fSm32b { int lane; } P8fSm32b;
P8fSm32b* p8f_sm32b_new(int n) {
  P8fSm32b* s = (P8fSm32b*)calloc(1, sizeof *s);
  s->lane = new_lane(P8L_SM32, (uint32_t)n, 1u << 31);
  return s;
}
void p8f_sm32b_emit(P8fSm32b* s, int cx, int16_t* out) {
  P8Emit* e = p8f_cur;
  lane_out(e->model, s->lane, out, 1, 1, 1, 1, 1023, 0);   /* a = 1: no rounding bit */
  put_op(e->model, s->lane, P8OP_MIX | P8OP_SET | ((uint32_t)cx & P8OP_CTX));
}

/* jpegModel's learning half (p8_rec.h P8L_JPG): the worker lane, 64 + 5 lanes that carry its raw words */
typedef struct P8fJpg { int lane[70]; int hashbits; } P8fJpg;
P8fJpg* p8f_jpg_new(uint64_t table_items) {
  P8fJpg* j = (P8fJpg*)calloc(1, sizeof *j);
  j->hashbits = (int)ilog2u((unsigned)table_items);
  j->lane[0] = new_lane(P8L_JPG, (uint32_t)table_items, 0);
  for (int i = 1; i < 70; ++i) j->lane[i] = new_lane(P8L_NONE, 0, 0);
  return j;
}
/* a coded step: cxt = the 32 context hashes (NULL unless hbcount == 0); 70 inputs at out[0..70), the export-only values at out[70..74) */
int p8f_jpg_step(P8fJpg* j, int hbcount, int hc_low, const uint64_t* cxt, const int* m1sel, int a1ctx, int a2ctx, int16_t* out) {
  P8Emit* e = p8f_cur;
  lane_out(e->model, j->lane[0], out, 74, 0, 1, 1, 0, 0);
  put_op(e->model, j->lane[0], P8OP_MIX | (uint32_t)(hbcount & 3) | ((uint32_t)(hc_low & 1) << 2));
  if (cxt)
    for (int i = 0; i < 32; ++i) {   /* BH<9>::operator[] :790-793: 16-bit checksum, first item of the 8-item neighbourhood */
      const uint32_t chk = (uint32_t)(p8f_checksum64(cxt[i], j->hashbits, 16) & 0xffff);
      const uint32_t item = (uint32_t)(((uint64_t)p8f_finalize64(cxt[i], j->hashbits) * 8) & (((uint64_t)1 << j->hashbits) - 1));
      put_op(e->model, j->lane[1 + 2 * i], chk);
      put_op(e->model, j->lane[2 + 2 * i], item);
    }
  for (int i = 0; i < 3; ++i) put_op(e->model, j->lane[65 + i], (uint32_t)m1sel[i]);
  put_op(e->model, j->lane[68], (uint32_t)a1ctx);
  put_op(e->model, j->lane[69], (uint32_t)a2ctx);
  return 70;
}

/* dmcForest (:7777-7822): bits only, lives on the device */
typedef struct Forest { int level; } Forest;
Forest* p8f_dmc_new(int level) { Forest* f = (Forest*)calloc(1, sizeof *f); f->level = level; return f; }
int p8f_dmc_mix(Forest* f, int y, int bpos, int16_t* out) {
  P8Emit* e = p8f_cur;
  (void)f; (void)y; (void)bpos;
  const int off = claim(e, out, 6);
  if (e->discovering) { if (e->full) e->L.dmc_off = (int16_t)off; else e->dmc_off0 = (int16_t)off; }
  return 6;
}

/* ---- end of the layout pass: host-computed inputs get DIRECT lanes, the first byte's compaction map is built ---- */
int p8f_emit_finish_discovery(P8Emit* e, int nx_first, int nx_full) {
  P8Layout* L = &e->L;
  if (nx_full != P8_NX || nx_first > P8_NX) { fail("unexpected number of mixer inputs"); return 1; }
  int direct0[P8_NX], nd0 = 0;
  for (int i = 0; i < nx_first; ++i) if (!e->claimed0[i]) direct0[nd0++] = i;
  for (int i = 0; i < P8_NX; ++i) L->first_map[i] = -1;
  int nd = 0;
  for (int i = 0; i < P8_NX; ++i) {
    if (e->claimed[i]) continue;
    const int l = L->nlanes;
    if (l >= P8_NLANE || nd >= nd0) { fail("host-computed inputs do not fit the lanes"); return 1; }
    P8Lane* q = &L->lane[l];
    memset(q, 0, sizeof *q);
    q->kind = P8L_DIRECT; q->off = (int16_t)i; q->nout = 1;
    L->nlanes = l + 1;
    L->first_map[direct0[nd++]] = (int16_t)i;
  }
  if (nd != nd0) { fail("host-computed inputs differ between the first byte and later ones"); return 1; }
  for (int l = 0; l < L->nlanes; ++l) {
    const P8Lane* q = &L->lane[l];
    if (q->kind == P8L_DIRECT || q->kind == P8L_NONE) continue;   /* (NONE: a lane that only carries another lane's second op word) */
    if (q->off < 0) { fail("a small map was never called"); return 1; }
    for (int j = 0; j < q->nout; ++j) L->first_map[e->lane_off0[l] + j] = (int16_t)(q->off + j);
  }
  for (int j = 0; j < 6; ++j) L->first_map[e->dmc_off0 + j] = (int16_t)(L->dmc_off + j);
  for (int i = 0; i < nx_first; ++i) if (L->first_map[i] < 0) { fail("first-byte input without a source"); return 1; }
  L->nx_first = nx_first;
  return e->err;
}

void p8f_emit_directs(P8Emit* e, int lim_off) {
  if (!e->chunk) return;
  const P8Layout* L = &e->L;
  uint32_t* ops = e->chunk->ops + e->step_row * P8_NLANE;
  if (e->full) {
    for (int l = 0; l < L->nlanes; ++l)
      if (L->lane[l].kind == P8L_DIRECT && L->lane[l].off < lim_off) ops[l] = P8OP_MIX | (uint32_t)(uint16_t)e->in_base[L->lane[l].off];
  } else {  /* first byte: the models wrote the compact vector */
    for (int i = 0; i < L->nx_first; ++i) {
      const int full = L->first_map[i];
      for (int l = 0; l < L->nlanes; ++l)
        if (L->lane[l].kind == P8L_DIRECT && L->lane[l].off == full) ops[l] = P8OP_MIX | (uint32_t)(uint16_t)e->in_base[i];
    }
  }
}

/* ---- RunContextMap (:857-889) over BH<4> (:778-813): the table is the device's (P8L_RCM); the front end hands it the hashed context per byte ---- */
typedef struct RCM { int lane, lane2, hashbits, pending; uint32_t chk, item, c1; } RCM;
RCM* p8f_rcm_new(int m) {
  ilog_init();
  RCM* r = (RCM*)calloc(1, sizeof *r);
  r->hashbits = (int)ilog2u((unsigned)(m / 4));
  r->lane = new_lane(P8L_RCM, (uint32_t)m, 0);
  r->lane2 = new_lane(P8L_NONE, 0, 0);
  return r;
}
void p8f_rcm_free(RCM* r) { free(r); }
void p8f_rcm_set(RCM* r, uint64_t cx, int c1) {   /* set(cx) :866-872 happens on the device at the byte's first step, with the byte just coded */
  r->chk = (uint32_t)(p8f_checksum64(cx, r->hashbits, 16) & 0xffff);
  r->item = (uint32_t)(((uint64_t)p8f_finalize64(cx, r->hashbits) * 8) & (((uint64_t)1 << r->hashbits) - 1));
  r->c1 = (uint32_t)c1 & 0xff;
  r->pending = 1;
}
int p8f_rcm_mix(RCM* r, int bpos, int c0, int16_t* out) {
  (void)bpos; (void)c0;
  lane_out(0, r->lane, out, 1, 0, 1, 1, 0, 0);
  put_op(0, r->lane, P8OP_MIX | (r->pending ? P8OP_SET | (r->chk << 8) | r->c1 : 0));
  put_op(0, r->lane2, r->item);
  r->pending = 0;
  return 0;
}
