/* oracle/ctxmodels_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's context plumbing and its 54 small native models:
 *   ContextManager                     reference src/context-manager.cpp:3-94
 *   54 byte contexts, 8 bit contexts   reference src/contexts/{bracket-context,sparse,context-hash,
 *                                      indirect-hash,interval,interval-hash,combined-context,
 *                                      bit-context}.cpp
 *   state machines                     reference src/states/{nonstationary,run-map}.cpp
 *   Direct, DirectHash, Indirect,      reference src/models/{direct,direct-hash,indirect,match,
 *   Match, Bracket (+ByteModel)        bracket,byte-model}.cpp
 *   construction order / parameters    reference src/predictor.cpp:24-36,90-178,199-356
 *   per-bit call order                 reference src/predictor.cpp:361-369,421-446,468
 * Output columns of the layer-0 vector: 0,1,2 and 2025..2075 (54 values),
 * plus the 47 mixer selectors (Appendix A.2; entry 12 = auxiliary_context_ is not produced here).
 *
 * The 256x2 transition table of the Nonstationary state machine (nonstationary.cpp:3) and the two
 * literal byte-class maps of predictor.cpp:255-272,285-302 are DATA of the reference; they are not
 * transcribed by hand but dumped from the reference build by scripts/gen_ref_tables.py into
 * oracle/ref_tables.h.
 *
 * Compiled with -ffp-contract=off; every float expression below has the operand types of the
 * reference expression it cites (int/float/double promotions matter for the last bit).
 */
#include "cmix_oracle.h"
#include "ref_tables.h"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define HISTORY_SIZE 100000000ull   /* context-manager.cpp:3 */
#define SHARED_MAP_SIZE 2048000000ull /* 256*8000000, context-manager.cpp:4 */
#define N_CTX 54
#define N_BITCTX 8
#define N_MODELS 54

/* ---- glibc rand() (TYPE_3 additive feedback, stdlib/random_r.c), so the oracle does not
 * disturb or depend on the process-global generator ---- */
typedef struct { int32_t r[34 + 310 + 64]; int k; } grand_t;
static void grand_seed(grand_t* g, uint32_t seed) {
  int32_t word = (int32_t)seed;
  if (word == 0) word = 1;
  g->r[0] = word;
  for (int i = 1; i < 31; ++i) {
    long hi = word / 127773, lo = word % 127773;
    long w = 16807 * lo - 2836 * hi;
    if (w < 0) w += 2147483647;
    word = (int32_t)w;
    g->r[i] = word;
  }
  for (int i = 31; i < 34; ++i) g->r[i] = g->r[i - 31];
  for (int i = 34; i < 344; ++i) g->r[i] = (int32_t)((uint32_t)g->r[i - 31] + (uint32_t)g->r[i - 3]);
  g->k = 344;
}
static int grand_next(grand_t* g) {
  int32_t v = (int32_t)((uint32_t)g->r[g->k - 31] + (uint32_t)g->r[g->k - 3]);
  g->r[g->k++] = v;
  return (int)((uint32_t)v >> 1);
}

/* ---- contexts ---- */
enum { C_BRACKET, C_SPARSE, C_HASH, C_INDIRECT, C_INTERVAL, C_INTERVALHASH, C_COMBINED };
typedef struct {
  int type;
  uint64_t context, size;
  /* sparse */
  int n_orders; unsigned orders[6];
  /* context-hash / interval-hash / indirect-hash */
  unsigned hash_size, hash_size1; uint64_t size1, context1; uint64_t* hashes;
  /* interval */
  const int* map; uint64_t mask; unsigned shift, interval;
  /* combined: indices into recent_bytes_ */
  int r1, r2; unsigned cshift;
} ctx_t;

enum { M_BRACKET, M_DIRECT, M_DIRECTHASH, M_INDIRECT, M_MATCH };
typedef struct {
  int type, ctx, col;
  float out;
  /* direct / direct-hash / match */
  int limit; float delta, divisor; uint64_t size, index;
  float* pred; uint8_t* cnt; uint64_t* chk;
  /* indirect */
  uint64_t map_index, map_offset; int run_map; float ipred[256];
  /* match */
  uint64_t m_history_pos, cur_match, map_size; uint8_t cur_byte, bit_pos, match_length;
  uint32_t* map; float mpred[256]; int mcnt[256];
} model_t;

struct orc_ctx {
  /* ContextManager, context-manager.h:21-31 */
  uint32_t bit_context, wrt_state;
  uint64_t long_bit_context, zero_context, history_pos, line_break, longest_match, wrt_context;
  uint8_t* history; uint8_t* shared_map;
  uint64_t words[8], recent[8];
  ctx_t ctx[N_CTX];
  uint64_t bitctx[N_BITCTX];
  uint8_t nonstat[256][2], runmap[512];
  /* BracketContext stack (bracket-context.h:21) -- unbounded in the reference */
  uint32_t* bc_active; uint32_t* bc_distance; size_t bc_n, bc_cap;
  /* Bracket model (bracket.h:17-22) + its ByteModel (byte-model.h:19-22) */
  uint32_t br_active[16], br_distance[16]; int br_n;
  uint32_t (*br_stats)[200][2];
  float br_probs[256]; int br_top, br_bot, br_ex;
  uint8_t vocab[256];
  int interval_maps[5][256];
  model_t m[N_MODELS];
};

static const int BITCTX_SRC[N_BITCTX] = {19, 20, 41, 42, 45, 48, 51, -1}; /* -1: recent_bytes_[1] */

static int bracket_close(unsigned c) { /* bracket-context.cpp:6 */
  switch (c) { case '(': return ')'; case '{': return '}'; case '[': return ']'; case '<': return '>'; }
  return -1;
}
static int bracket_close6(unsigned c) { /* bracket.cpp:9-10 */
  switch (c) { case '(': return ')'; case '{': return '}'; case '[': return ']'; case '<': return '>';
               case '\'': return '\''; case '"': return '"'; }
  return -1;
}

static void add_hash(ctx_t* c, unsigned order, unsigned hash_size) { /* context-hash.cpp:3-7 */
  memset(c, 0, sizeof *c);
  c->type = C_HASH; c->hash_size = hash_size; c->size = 1ull << (hash_size * order);
}
static void add_sparse(ctx_t* c, int n, const unsigned* o) { /* sparse.cpp:5-15 */
  memset(c, 0, sizeof *c);
  c->type = C_SPARSE; c->n_orders = n; memcpy(c->orders, o, n * sizeof *o); c->size = ~0ull;
}
static void add_indirect(ctx_t* c, unsigned o1, unsigned h1, unsigned o2, unsigned h2) { /* indirect-hash.cpp:3-11 */
  memset(c, 0, sizeof *c);
  c->type = C_INDIRECT; c->hash_size1 = h1; c->hash_size = h2;
  c->size1 = (uint32_t)(1ull << (h1 * o1)); c->size = 1ull << (h2 * o2);
  c->hashes = calloc(c->size1, sizeof(uint64_t));
}
static void add_interval(ctx_t* c, const int* map, unsigned num_bits) { /* interval.cpp:3-15 */
  memset(c, 0, sizeof *c);
  c->type = C_INTERVAL; c->map = map;
  int mx = 0; for (int i = 0; i < 256; ++i) if (map[i] > mx) mx = map[i];
  c->shift = 1; while ((1 << c->shift) <= mx) ++c->shift;
  c->size = 1ull << num_bits; c->mask = c->size - 1;
}
static void add_interval_hash(ctx_t* c, const int* map, unsigned num_bits, unsigned order, unsigned hash_size) {
  memset(c, 0, sizeof *c); /* interval-hash.cpp:3-16 */
  c->type = C_INTERVALHASH; c->map = map; c->hash_size = hash_size;
  int mx = 0; for (int i = 0; i < 256; ++i) if (map[i] > mx) mx = map[i];
  c->shift = 1; while ((1 << c->shift) <= mx) ++c->shift;
  c->mask = (1ull << num_bits) - 1; c->size = 1ull << (hash_size * order);
}
static void add_combined(ctx_t* c, int r1, int r2) { /* combined-context.cpp:3-11, sizes 256,256 */
  memset(c, 0, sizeof *c);
  c->type = C_COMBINED; c->r1 = r1; c->r2 = r2; c->size = 256 * 256;
  c->cshift = 1; while ((uint64_t)(1 << c->cshift) < 256) ++c->cshift;
}

static void init_direct(model_t* m, int type, int ctx, int col, int limit, float delta, uint64_t size) {
  memset(m, 0, sizeof *m); /* direct.cpp:3-13, direct-hash.cpp:3-14 */
  m->type = type; m->ctx = ctx; m->col = col; m->limit = limit; m->delta = delta; m->size = size;
  m->divisor = 1.0 / (limit + delta);
  m->pred = malloc(size * 256 * sizeof(float));
  for (uint64_t i = 0; i < size * 256; ++i) m->pred[i] = 0.5;
  m->cnt = calloc(size * 256, 1);
  if (type == M_DIRECTHASH) m->chk = calloc(size, sizeof(uint64_t));
  m->out = 0.5;
}
static void init_indirect(orc_ctx* o, model_t* m, int ctx, int col, float delta, int run_map, grand_t* g) {
  memset(m, 0, sizeof *m); /* indirect.cpp:4-14 */
  m->type = M_INDIRECT; m->ctx = ctx; m->col = col; m->run_map = run_map;
  m->divisor = 1.0 / delta;
  m->map_offset = (uint64_t)grand_next(g) % (SHARED_MAP_SIZE - 257);
  for (int i = 0; i < 256; ++i) {
    if (!run_map) m->ipred[i] = 0.5;                       /* nonstationary.cpp:9-11 */
    else if (i < 128) m->ipred[i] = (128.0 - i) / 256;     /* run-map.cpp:17-20 */
    else m->ipred[i] = i / 256.0;
  }
  m->out = 0.5;
  (void)o;
}
static void init_match(model_t* m, int ctx, int col, int limit, float delta, uint64_t map_size) {
  memset(m, 0, sizeof *m); /* match.cpp:3-15 */
  m->type = M_MATCH; m->ctx = ctx; m->col = col; m->limit = limit; m->delta = delta;
  m->divisor = 1.0 / (limit + delta);
  m->bit_pos = 128; m->map_size = map_size;
  m->map = calloc(map_size, sizeof(uint32_t));
  for (int i = 0; i < 256; ++i) m->mpred[i] = 0.5 + (i + 0.5) / 512;
  m->out = 0.5;
}

orc_ctx* orc_ctx_create(const uint8_t* vocab256) {
  orc_ctx* o = calloc(1, sizeof *o);
  memcpy(o->vocab, vocab256, 256);
  memcpy(o->nonstat, REF_NONSTATIONARY, 512);
  for (int i = 0; i < 512; ++i) { /* run-map.cpp:3-15 */
    int state = i / 2;
    if (i % 2 == 0) { if (state < 127) ++state; else if (state >= 128) state = 0; }
    else { if (state < 128) state = 128; else if (state < 255) ++state; }
    o->runmap[i] = state;
  }
  o->bit_context = 1; o->long_bit_context = 1;
  o->history = calloc(HISTORY_SIZE, 1);
  o->shared_map = calloc(SHARED_MAP_SIZE, 1);
  o->bc_cap = 1024; o->bc_active = malloc(o->bc_cap * 4); o->bc_distance = malloc(o->bc_cap * 4);
  o->br_stats = malloc(256 * sizeof *o->br_stats);
  for (int i = 0; i < 256; ++i) for (int d = 0; d < 200; ++d) { o->br_stats[i][d][0] = 1; o->br_stats[i][d][1] = 256; }
  for (int i = 0; i < 256; ++i) o->br_probs[i] = 1.0 / 256; /* byte-model.cpp:5-6 */
  o->br_top = 255; o->br_bot = 0; o->br_ex = 0;

  grand_t g; grand_seed(&g, 0xDEADBEEFu); /* predictor.cpp:26 */
  int nm = 0, nc = 0;
  ctx_t* C = o->ctx; model_t* M = o->m;
  /* AddBracket, predictor.cpp:90-98 */
  memset(&C[nc], 0, sizeof C[nc]); C[nc].type = C_BRACKET; C[nc].size = 257 * 256; nc++;       /* ctx 0 */
  memset(&M[nm], 0, sizeof M[nm]); M[nm].type = M_BRACKET; M[nm].col = 0; M[nm].out = 0.5; nm++;
  init_direct(&M[nm++], M_DIRECT, 0, 1, 30, 0, C[0].size);
  init_indirect(o, &M[nm++], 0, 2, 300, 0, &g);
  /* AddWord, predictor.cpp:104-131 */
  static const unsigned P1[18][7] = {{1,0},{2,0,1},{2,7,2},{1,7},{1,1},{2,1,2},{3,1,2,3},{2,1,3},{2,1,4},{2,1,5},
      {2,2,3},{2,3,4},{3,1,2,4},{4,1,2,3,4},{3,2,3,4},{1,2},{5,1,2,3,4,5},{6,1,2,3,4,5,6}};
  int col = 2025;
  for (int i = 0; i < 18; ++i) {
    add_sparse(&C[nc], P1[i][0], &P1[i][1]);
    init_indirect(o, &M[nm++], nc, col++, 200, 0, &g);
    nc++;
  }
  /* model_params2 de-duplicates against the contexts above (context-manager.cpp:6-12, sparse.cpp:24-33):
   * {0}->ctx1 {1}->ctx5 {7}->ctx4 {1,3}->ctx8 {1,2,3}->ctx7 {7,2}->ctx3 */
  static const int P2CTX[6] = {1, 5, 4, 8, 7, 3};
  for (int i = 0; i < 6; ++i) {
    init_match(&M[nm++], P2CTX[i], col++, 200, 0.5, 10000000);
    if (i == 1) {
      init_indirect(o, &M[nm++], P2CTX[i], col++, 200, 1, &g);
      init_direct(&M[nm++], M_DIRECTHASH, P2CTX[i], col++, 30, 0, 500000);
    }
  }
  /* AddDirect, predictor.cpp:133-148 */
  for (unsigned ord = 0; ord < 4; ++ord) {
    add_hash(&C[nc], ord, 8);
    if (ord < 3) init_direct(&M[nm++], M_DIRECT, nc, col++, 30, 0, C[nc].size);
    else init_direct(&M[nm++], M_DIRECTHASH, nc, col++, 30, 0, 100000);
    nc++;
  }
  /* AddMatch, predictor.cpp:150-164 */
  static const unsigned PM[10][2] = {{0,8},{1,8},{2,8},{7,4},{11,3},{13,2},{15,2},{17,2},{20,1},{25,1}};
  for (int i = 0; i < 10; ++i) {
    int c;
    if (i < 3) c = 19 + i; /* equal size_ and hash_size_ => shared (context-hash.cpp:13-18) */
    else { add_hash(&C[nc], PM[i][0], PM[i][1]); c = nc++; }
    uint64_t sz = C[c].size < 20000000ull ? C[c].size : 20000000ull;
    init_match(&M[nm++], c, col++, 200, 0.5, sz);
  }
  /* AddDoubleIndirect, predictor.cpp:166-178 */
  static const unsigned PI[11][4] = {{1,8,1,8},{2,8,1,8},{1,8,2,8},{2,8,2,8},{1,8,3,8},{3,8,1,8},{4,6,4,8},
      {5,5,5,5},{1,8,4,8},{1,8,5,6},{6,4,6,4}};
  for (int i = 0; i < 11; ++i) {
    add_indirect(&C[nc], PI[i][0], PI[i][1], PI[i][2], PI[i][3]);
    init_indirect(o, &M[nm++], nc, col++, 400, 0, &g);
    nc++;
  }
  /* AddMixers contexts, predictor.cpp:199-328 */
  add_hash(&C[nc++], 2, 4); /* 41 */
  add_hash(&C[nc++], 3, 2); /* 42 */
  int (*map)[256] = o->interval_maps;
  for (int i = 0; i < 256; ++i)
    map[0][i] = (i < 1) + (i < 32) + (i < 64) + (i < 128) + (i < 255) + (i < 142) + (i < 138) + (i < 140) +
                (i < 137) + (i < 97);
  for (int i = 0; i < 256; ++i)
    map[1][i] = (i < 41) + (i < 92) + (i < 124) + (i < 58) + (i < 11) + (i < 46) + (i < 36) + (i < 47) + (i < 64) +
                (i < 4) + (i < 61) + (i < 97) + (i < 125) + (i < 45) + (i < 48);
  for (int i = 0; i < 256; ++i)
    map[2][i] = (i >= 'a' && i <= 'z') || (i >= 'A' && i <= 'Z') || (i >= '0' && i <= '9') || i >= 0x80;
  /* the two literal 256-entry tables of predictor.cpp:255-272 ("wrt_2b") and :285-302 ("wrt_3b") are
   * reference DATA (oracle/ref_tables.h) */
  for (int i = 0; i < 256; ++i) { map[3][i] = REF_INTERVAL_WRT2B[i]; map[4][i] = REF_INTERVAL_WRT3B[i]; }
  add_interval(&C[nc++], map[0], 8);  /* 43 interval1 */
  add_interval(&C[nc++], map[1], 8);  /* 44 interval2 */
  add_interval(&C[nc++], map[2], 7);  /* 45 interval3 */
  add_interval(&C[nc++], o->interval_maps[3], 10); /* 46 interval4 */
  add_interval(&C[nc++], o->interval_maps[3], 15); /* 47 interval5 */
  add_interval(&C[nc++], o->interval_maps[3], 7);  /* 48 interval8 */
  add_interval(&C[nc++], o->interval_maps[4], 9);  /* 49 interval6 */
  add_interval_hash(&C[nc++], o->interval_maps[4], 8, 7, 2); /* 50 interval7 */
  add_interval(&C[nc++], o->interval_maps[4], 7);  /* 51 interval9 */
  add_combined(&C[nc++], 1, 0); /* 52 */
  add_combined(&C[nc++], 2, 1); /* 53 */
  if (nc != N_CTX || nm != N_MODELS || col != 2076) { fprintf(stderr, "orc_ctx_create: layout error\n"); abort(); }
  return o;
}

void orc_ctx_destroy(orc_ctx* o) {
  if (!o) return;
  for (int i = 0; i < N_CTX; ++i) free(o->ctx[i].hashes);
  for (int i = 0; i < N_MODELS; ++i) { free(o->m[i].pred); free(o->m[i].cnt); free(o->m[i].chk); free(o->m[i].map); }
  free(o->history); free(o->shared_map); free(o->bc_active); free(o->bc_distance); free(o->br_stats);
  free(o);
}

/* ---- context updates ---- */
static void update_context(orc_ctx* o, ctx_t* c) {
  const unsigned byte = o->bit_context;
  switch (c->type) {
    case C_BRACKET: { /* bracket-context.cpp:11-34, distance_limit 256; the stack_limit test at :24 compares
                       * brackets_.size() (=4) with 15 and never fires */
      if (o->bc_n) {
        if (bracket_close(o->bc_active[o->bc_n - 1]) == (int)byte || o->bc_distance[o->bc_n - 1] >= 256 - 1) o->bc_n--;
        else ++o->bc_distance[o->bc_n - 1];
      }
      if (bracket_close(byte) >= 0) {
        if (o->bc_n == o->bc_cap) {
          o->bc_cap *= 2;
          o->bc_active = realloc(o->bc_active, o->bc_cap * 4);
          o->bc_distance = realloc(o->bc_distance, o->bc_cap * 4);
        }
        o->bc_active[o->bc_n] = byte; o->bc_distance[o->bc_n] = 0; o->bc_n++;
      }
      c->context = o->bc_n ? 256 * (o->bc_active[o->bc_n - 1] + 1) + o->bc_distance[o->bc_n - 1] : 0;
      break;
    }
    case C_SPARSE: { /* sparse.cpp:17-22; factors_ are unsigned int, the product is 64-bit */
      static const unsigned F[6] = {1, 256, 29 * 31, 29 * 31 * 37, 29 * 31 * 37 * 41, 29 * 31 * 37 * 41 * 43};
      c->context = o->words[c->orders[0]];
      for (int i = 1; i < c->n_orders; ++i) c->context += F[i] * o->words[c->orders[i]];
      break;
    }
    case C_HASH: /* context-hash.cpp:9-11 */
      c->context = (c->context * (1 << c->hash_size) + byte) % c->size;
      break;
    case C_INDIRECT: /* indirect-hash.cpp:13-17 */
      c->hashes[c->context1] = (c->context * (1 << c->hash_size) + byte) % c->size;
      c->context1 = (c->context1 * (1 << c->hash_size1) + byte) % c->size1;
      c->context = c->hashes[c->context1];
      break;
    case C_INTERVAL: /* interval.cpp:17-19 */
      c->context = c->mask & ((c->context << c->shift) + c->map[byte]);
      break;
    case C_INTERVALHASH: /* interval-hash.cpp:18-21 */
      c->interval = c->mask & ((c->interval << c->shift) + c->map[byte]);
      c->context = (c->context * (1 << c->hash_size) + c->interval) % c->size;
      break;
    case C_COMBINED: /* combined-context.cpp:13-15 */
      c->context = (o->recent[c->r2] << c->cshift) + o->recent[c->r1];
      break;
  }
}

static void update_contexts(orc_ctx* o, int bit) { /* context-manager.cpp:69-94 */
  o->bit_context += o->bit_context + bit;
  o->long_bit_context = o->bit_context;
  if (o->bit_context >= 256) {
    o->bit_context -= 256;
    o->long_bit_context = 1;
    o->longest_match = 0;
    if (o->bit_context == '\n') o->line_break = 0;
    else if (o->line_break < 99) ++o->line_break;
    /* UpdateHistory :23-27 */
    o->history[o->history_pos] = o->bit_context;
    if (++o->history_pos == HISTORY_SIZE) o->history_pos = 0;
    /* UpdateWords :29-48 */
    unsigned char c = o->bit_context;
    if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c >= 0x80) o->words[7] = o->words[7] * 997 * 16 + c;
    else o->words[7] = 0;
    if (c >= 'A' && c <= 'Z') c += 'a' - 'A';
    if ((c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == 8 || c == 6 || c >= 0x80) {
      o->words[0] = o->words[0] * 997 * 16 + c;
      o->words[0] &= 0xfffffff;
      o->words[1] = o->words[1] * 263 * 32 + c;
    } else {
      for (int i = 6; i >= 2; --i) o->words[i] = o->words[i - 1];
      o->words[1] = 0;
    }
    /* UpdateRecentBytes :50-55 */
    for (int i = 7; i >= 1; --i) o->recent[i] = o->recent[i - 1];
    o->recent[0] = o->bit_context;
    /* UpdateWRTContext :57-67 */
    if (o->bit_context < 0x80) o->wrt_state = 0;
    else {
      if (o->wrt_state == 0) o->wrt_context = 0;
      o->wrt_state = 1;
      o->wrt_context <<= 8;
      o->wrt_context += o->bit_context;
      if (o->wrt_context > 0xFFEFCF) o->wrt_context = 0;
    }
    for (int i = 0; i < N_CTX; ++i) update_context(o, &o->ctx[i]);
  }
  for (int i = 0; i < N_BITCTX; ++i) { /* bit-context.cpp:11-13 */
    uint64_t bc = BITCTX_SRC[i] >= 0 ? o->ctx[BITCTX_SRC[i]].context : o->recent[1];
    o->bitctx[i] = (bc << 8) + o->long_bit_context;
  }
}

/* ---- Bracket model ---- */
static void bracket_set(orc_ctx* o, float p, int sym) { /* bracket.cpp:27-28 */
  for (int i = 0; i < 256; ++i) o->br_probs[i] = (1 - p) / 255;
  o->br_probs[sym] = p;
}
static void bracket_byte_update(orc_ctx* o) { /* bracket.cpp:13-60 */
  const unsigned byte = o->bit_context;
  for (int i = 0; i < 256; ++i) o->br_probs[i] = 1. / 256;
  int top = o->br_n ? (int)o->br_active[o->br_n - 1] : -1;
  if (o->br_n == 0 || (bracket_close6(byte) >= 0 && !(top == (int)byte && bracket_close6(byte) == (int)byte))) {
    if (bracket_close6(byte) >= 0) {
      o->br_active[o->br_n] = byte; o->br_distance[o->br_n] = 0; o->br_n++;
      if (o->br_n > 10) { /* stack_limit_ = 10 */
        memmove(o->br_active, o->br_active + 1, (o->br_n - 1) * 4);
        memmove(o->br_distance, o->br_distance + 1, (o->br_n - 1) * 4);
        o->br_n--;
      }
      float p = (1. * o->br_stats[byte][0][0]) / o->br_stats[byte][0][1];
      bracket_set(o, p, bracket_close6(byte));
    }
  } else {
    unsigned active = o->br_active[o->br_n - 1], distance = o->br_distance[o->br_n - 1];
    ++o->br_stats[active][distance][1];
    if (bracket_close6(active) == (int)byte) ++o->br_stats[active][distance][0];
    if (o->br_stats[active][distance][1] > 100000) { /* stats_limit_ */
      o->br_stats[active][distance][0] /= 2;
      o->br_stats[active][distance][1] /= 2;
    }
    if (bracket_close6(active) == (int)byte || distance >= 200 - 1) { /* distance_limit_ = 200 */
      o->br_n--;
      if (o->br_n) {
        int a = o->br_active[o->br_n - 1], d = o->br_distance[o->br_n - 1];
        float p = (1. * o->br_stats[a][d][0]) / o->br_stats[a][d][1];
        bracket_set(o, p, bracket_close6(a));
      }
    } else {
      ++o->br_distance[o->br_n - 1];
      ++distance;
      float p = (1. * o->br_stats[active][distance][0]) / o->br_stats[active][distance][1];
      bracket_set(o, p, bracket_close6(active));
    }
  }
  /* ByteModel::ByteUpdate, byte-model.cpp:39-45 */
  o->br_top = 255; o->br_bot = 0;
  for (int i = 0; i < 256; ++i) if (!o->vocab[i]) o->br_probs[i] = 0;
}
static float bytemodel_predict(const float* probs, int bot, int top, int* ex) { /* byte-model.cpp:8-24 */
  int mid = bot + ((top - bot) / 2);
  float num = 0.0f;
  for (int i = mid + 1; i <= top; ++i) num = num + probs[i];
  float denom = num;
  for (int i = bot; i <= mid; ++i) denom = denom + probs[i];
  *ex = bot;
  float mx = probs[bot];
  for (int i = bot + 1; i <= top; i++) if (probs[i] > mx) { mx = probs[i]; *ex = i; }
  if (denom == 0) return 0.5;
  return num / denom;
}

/* ---- one Predict() of the 54 models + readout of the selectors Mixer::Mix sees ---- */
void orc_ctx_predict(orc_ctx* o, float* probs54, uint64_t* sel47) {
  for (int i = 0; i < N_MODELS; ++i) {
    model_t* m = &o->m[i];
    const uint64_t bctx = m->type == M_BRACKET ? 0 : o->ctx[m->ctx].context;
    switch (m->type) {
      case M_BRACKET: m->out = bytemodel_predict(o->br_probs, o->br_bot, o->br_top, &o->br_ex); break;
      case M_DIRECT: m->out = m->pred[bctx * 256 + o->bit_context]; break;          /* direct.cpp:15-18 */
      case M_DIRECTHASH: m->out = m->pred[m->index * 256 + o->bit_context]; break;  /* direct-hash.cpp:16-19 */
      case M_INDIRECT:                                                              /* indirect.cpp:16-20 */
        m->map_index += o->bit_context;
        m->out = m->ipred[o->shared_map[m->map_index]];
        break;
      case M_MATCH:                                                                 /* match.cpp:17-21 */
        if (m->cur_byte & m->bit_pos) m->out = m->mpred[m->match_length];
        else m->out = 1 - m->mpred[m->match_length];
        break;
    }
    probs54[i] = m->out;
  }
  if (sel47) {
    const uint64_t* b = o->bitctx; const ctx_t* C = o->ctx;
    const uint64_t s[47] = { /* predictor.cpp:199-356, SURVEY.md Appendix A.2 */
      b[0], b[0], b[1], b[1], b[2], b[3], o->recent[2], o->recent[3], o->zero_context, o->line_break,
      o->longest_match, o->wrt_context, 0 /* auxiliary_context_: derived in the mixing network */,
      C[43].context, C[44].context, C[45].context, b[4], C[46].context, C[47].context, b[5], C[49].context,
      C[50].context, b[6], b[7], C[52].context, C[53].context,
      o->zero_context, o->zero_context, o->long_bit_context, o->long_bit_context, o->long_bit_context,
      o->recent[0], o->recent[1], o->recent[2], o->longest_match, o->wrt_context, C[43].context, C[44].context,
      C[45].context, C[46].context, C[47].context, C[49].context, C[50].context, b[5], b[4], b[6],
      o->zero_context};
    memcpy(sel47, s, sizeof s);
  }
}

static void direct_perceive(model_t* m, uint64_t row, unsigned bc, int bit) { /* direct.cpp:20-28, direct-hash.cpp:21-29 */
  float divisor = m->divisor;
  uint8_t* cnt = &m->cnt[row * 256 + bc];
  float* p = &m->pred[row * 256 + bc];
  if (*cnt < m->limit) {
    ++*cnt;
    divisor = 1.0 / (*cnt + m->delta);
  }
  *p += (bit - *p) * divisor;
}

void orc_ctx_perceive(orc_ctx* o, int bit) {
  /* Predictor::Perceive, predictor.cpp:421-425 */
  for (int i = 0; i < N_MODELS; ++i) {
    model_t* m = &o->m[i];
    switch (m->type) {
      case M_BRACKET: { /* ByteModel::Perceive, byte-model.cpp:30-37 */
        int mid = o->br_bot + ((o->br_top - o->br_bot) / 2);
        if (bit) o->br_bot = mid + 1; else o->br_top = mid;
        break;
      }
      case M_DIRECT: direct_perceive(m, o->ctx[m->ctx].context, o->bit_context, bit); break;
      case M_DIRECTHASH: direct_perceive(m, m->index, o->bit_context, bit); break;
      case M_INDIRECT: { /* indirect.cpp:22-27 */
        int state = o->shared_map[m->map_index];
        m->ipred[state] += (bit - m->ipred[state]) * m->divisor;
        o->shared_map[m->map_index] = m->run_map ? o->runmap[state * 2 + bit] : o->nonstat[state][bit];
        m->map_index -= o->bit_context;
        break;
      }
      case M_MATCH: { /* match.cpp:23-46 */
        int match = 0;
        if (bit == ((m->cur_byte & m->bit_pos) != 0)) match = 1;
        m->bit_pos /= 2;
        float divisor = m->divisor;
        if (m->mcnt[m->match_length] < m->limit) {
          ++m->mcnt[m->match_length];
          divisor = 1.0 / (m->mcnt[m->match_length] + m->delta);
        }
        m->mpred[m->match_length] += (match - m->mpred[m->match_length]) * divisor;
        if (match) { if (m->match_length < 255) ++m->match_length; }
        else m->match_length = 0;
        if (o->bit_context >= 128) {
          m->map[o->ctx[m->ctx].context % m->map_size] = (uint32_t)m->m_history_pos;
          ++m->m_history_pos;
        }
        break;
      }
    }
  }
  const int byte_update = o->bit_context >= 128; /* predictor.cpp:439-440 */
  update_contexts(o, bit);
  if (byte_update) { /* predictor.cpp:443-446 */
    for (int i = 0; i < N_MODELS; ++i) {
      model_t* m = &o->m[i];
      const uint64_t bctx = m->type == M_BRACKET ? 0 : o->ctx[m->ctx].context;
      switch (m->type) {
        case M_BRACKET: bracket_byte_update(o); break;
        case M_DIRECT: break;
        case M_DIRECTHASH: /* direct-hash.cpp:31-48 */
          m->index = bctx % m->size;
          for (int k = 0; k < 20; ++k) {
            if (m->chk[m->index] == 0) { m->chk[m->index] = bctx; break; }
            if (m->chk[m->index] == bctx) break;
            if (k == 19) {
              for (int j = 0; j < 256; ++j) { m->pred[m->index * 256 + j] = 0.5; m->cnt[m->index * 256 + j] = 0; }
              m->chk[m->index] = bctx;
              break;
            }
            ++m->index;
            if (m->index == m->size) m->index = 0;
          }
          break;
        case M_INDIRECT: /* indirect.cpp:29-31 */
          m->map_index = (257 * bctx + m->map_offset) % (SHARED_MAP_SIZE - 257);
          break;
        case M_MATCH: { /* match.cpp:48-60 */
          if (m->match_length < 8) m->cur_match = m->map[bctx % m->map_size];
          else ++m->cur_match;
          m->cur_match %= HISTORY_SIZE;
          m->cur_byte = o->history[m->cur_match];
          m->bit_pos = 128;
          uint64_t mc = m->match_length / 32;
          if (mc > o->longest_match) o->longest_match = mc;
          break;
        }
      }
    }
    o->bit_context = 1; /* predictor.cpp:468 */
  }
}

/* ---- state readout for the parity tests (same layout as oracle/ref_harness.cpp:ref_get_manager) ---- */
void orc_ctx_get_manager(const orc_ctx* o, uint64_t* regs25, uint64_t* ctx54, uint64_t* bitctx8) {
  int n = 0;
  regs25[n++] = o->bit_context; regs25[n++] = o->long_bit_context; regs25[n++] = o->zero_context;
  regs25[n++] = o->history_pos; regs25[n++] = o->line_break; regs25[n++] = o->longest_match;
  regs25[n++] = 0; /* auxiliary_context_ lives in the mixing network */
  regs25[n++] = o->wrt_context; regs25[n++] = o->wrt_state;
  for (int i = 0; i < 8; ++i) regs25[n++] = o->recent[i];
  for (int i = 0; i < 8; ++i) regs25[n++] = o->words[i];
  for (int i = 0; i < N_CTX; ++i) ctx54[i] = o->ctx[i].context;
  for (int i = 0; i < N_BITCTX; ++i) bitctx8[i] = o->bitctx[i];
}
const float* orc_ctx_bracket_probs(const orc_ctx* o) { return o->br_probs; }
uint64_t orc_ctx_indirect_offset(const orc_ctx* o, int model) { return o->m[model].map_offset; }
int orc_ctx_model_column(const orc_ctx* o, int model) { return o->m[model].col; }

/* Batch driver for the tests: walks Predict()/Perceive() over n bytes (MSB first).
 * probs [8n][54] f32, sel [8n][47] u64 (either may be NULL). */
void orc_ctx_run(orc_ctx* o, const uint8_t* bytes, size_t n, float* probs, uint64_t* sel) {
  float p[N_MODELS];
  size_t t = 0;
  for (size_t i = 0; i < n; ++i)
    for (int j = 7; j >= 0; --j, ++t) {
      orc_ctx_predict(o, probs ? probs + t * N_MODELS : p, sel ? sel + t * 47 : NULL);
      orc_ctx_perceive(o, (bytes[i] >> j) & 1);
    }
}

/* state injection: the twin of ref_debug_set_history (oracle/ref_harness.cpp) */
void orc_ctx_set_history(orc_ctx* o, uint64_t pos, const uint8_t* tail, uint64_t n) {
  for (uint64_t i = 0; i < n; ++i) o->history[(pos - n + i) % HISTORY_SIZE] = tail[i];
  o->history_pos = pos % HISTORY_SIZE;
  for (int k = 0; k < N_MODELS; ++k)
    if (o->m[k].type == M_MATCH) o->m[k].m_history_pos = pos;
}
