/* oracle/paq8_core.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the numeric building blocks of the vendored paq8 model that SURVEY.md 8(a') lists for the paq8 /
 * fxcm device stages: the two-layer int16 mixer (reference src/models/paq8.cpp:513-598 with dot_product / train
 * :403-432), APM1 (:600-621), StateMap (:623-645), StateMap32 (:645-690) and APM (:691-712). Data tables come from
 * oracle/paq8_tables.h (dumped from the reference build). Pinned against the reference's own classes, compiled from
 * paq8.cpp by oracle/ref_paq8core.cpp, in tests/test_oracle_paq8core.py and against tests/golden/paq8core_vectors.npz.
 *
 * Representation choices (values identical): the mixer's lazily created weight rows (unordered_map keyed by the
 * selector, rows filled with init_w on first use, :530-537) are one dense zero-cost array [m][N] pre-filled with
 * init_w; wrap-around of 32-bit sums is written with unsigned arithmetic. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "paq8_tables.h"

static int p8_squash(int d) {  /* Squash::operator() :348-352 */
  if (d > 2047) return 4095;
  if (d < -2047) return 0;
  return P8_SQUASH[d + 2048];
}
static int p8_stretch(int p) { return P8_STRETCH[p]; }
int orc_p8_squash(int d) { return p8_squash(d); }
int orc_p8_stretch(int p) { return p8_stretch(p); }

static int16_t sat16(int v) { return (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v); }

/* dot_product, SSE2 form (:403-413): per pair of terms a 32-bit sum of two 16x16 products (pmaddwd: wraps, only
 * reachable with both factors -32768), arithmetic >> 8, accumulated modulo 2^32 -- order-free. */
static int p8_dot(const int16_t* t, const int16_t* w, int n) {
  uint32_t sum = 0;
  for (int i = 0; i + 1 < n; i += 2) {
    const uint32_t pair = (uint32_t)((int32_t)t[i] * w[i]) + (uint32_t)((int32_t)t[i + 1] * w[i + 1]);
    sum += (uint32_t)((int32_t)pair >> 8);
  }
  return (int32_t)sum;
}

/* train, SSE2 form (:415-430): w += ((sat(2 t) * err >> 16) + 1) >> 1 with 16-bit saturating adds; err is the low 16
 * bits of e (always in range for the callers). n is a multiple of 8. */
static void p8_train(const int16_t* t, int16_t* w, int n, int e) {
  if (!e) return;
  const int16_t err = (int16_t)e;
  for (int i = 0; i < n; ++i) {
    int v = sat16(2 * (int)t[i]);
    v = (v * (int)err) >> 16;          /* pmulhw: high half of the signed product */
    v = sat16(v + 1) >> 1;             /* paddsw one, psraw 1 */
    w[i] = sat16(v + (int)w[i]);
  }
}

typedef struct OrcP8Mixer {
  int N, S, init_w, M;     /* inputs (padded to 8), selectable sets per bit, initial weight, total selector range */
  int16_t* tx;             /* [N] */
  int16_t* wx;             /* [M][N] */
  int* cxt;                /* [S] */
  int* pr;                 /* [S] */
  int ncxt, base, nx;
  struct OrcP8Mixer* mp;   /* second layer (S inputs, 1 set, init 0x7fff) when S > 1 */
  float* exp_out;          /* where add() exports squash(x) / 4095 (AddPrediction, :504-507,542-545) */
  int* exp_n;
} OrcP8Mixer;

OrcP8Mixer* orc_p8_mixer_new(int n, int m, int s, int w) {
  OrcP8Mixer* x = (OrcP8Mixer*)calloc(1, sizeof *x);
  x->N = (n + 7) & -8; x->S = s; x->init_w = w; x->M = m;
  x->tx = (int16_t*)calloc(x->N, 2);
  x->wx = (int16_t*)malloc((size_t)m * x->N * 2);
  for (size_t i = 0; i < (size_t)m * x->N; ++i) x->wx[i] = (int16_t)w;
  x->cxt = (int*)calloc(s, sizeof(int));
  x->pr = (int*)malloc(s * sizeof(int));
  for (int i = 0; i < s; ++i) x->pr[i] = 2048;  /* :586-587 */
  if (s > 1) x->mp = orc_p8_mixer_new(s, 1, 1, 0x7fff);  /* :588-590 */
  return x;
}
void orc_p8_mixer_free(OrcP8Mixer* x) {
  if (!x) return;
  orc_p8_mixer_free(x->mp);
  free(x->tx); free(x->wx); free(x->cxt); free(x->pr); free(x);
}
static void p8_mixer_update(OrcP8Mixer* x, int y) {  /* :528-541 */
  for (int i = 0; i < x->ncxt; ++i) {
    const int err = ((y << 12) - x->pr[i]) * 7;
    p8_train(x->tx, x->wx + (size_t)x->cxt[i] * x->N, x->nx, err);
  }
  x->nx = x->base = x->ncxt = 0;
}
static void p8_mixer_add(OrcP8Mixer* x, int v) {  /* :543-546 */
  if (x->exp_out) x->exp_out[(*x->exp_n)++] = (float)p8_squash(v) * (float)(1.0 / 4095);  /* conversion_factor :502 */
  x->tx[x->nx++] = (int16_t)v;
}
static void p8_mixer_set(OrcP8Mixer* x, int cx, int range) {  /* :548-551 */
  x->cxt[x->ncxt++] = x->base + cx;
  x->base += range;
}
static int p8_mixer_p(OrcP8Mixer* x, int y) {  /* :553-581 */
  while (x->nx & 7) x->tx[x->nx++] = 0;
  if (x->mp) {
    p8_mixer_update(x->mp, y);
    for (int i = 0; i < x->ncxt; ++i) {
      const int d = p8_dot(x->tx, x->wx + (size_t)x->cxt[i] * x->N, x->nx);
      x->pr[i] = p8_squash((int32_t)((uint32_t)d * 9u) >> 9);
      p8_mixer_add(x->mp, p8_stretch(x->pr[i]));
    }
    p8_mixer_set(x->mp, 0, 1);
    return p8_mixer_p(x->mp, y);
  }
  const int z = p8_dot(x->tx, x->wx, x->nx);
  x->base = p8_squash((int32_t)((uint32_t)z * 16u) >> 13);
  return x->pr[0] = p8_squash(z >> 9);
}
/* One coded bit as contextModel2 drives the mixer (:8136-8205): update with the previous bit, the add() calls, the
 * set() calls, p(). exported/nexp as in refp8_mixer_step. */
int orc_p8_mixer_step(OrcP8Mixer* x, int y_prev, const int16_t* in, int nx, const int* cx, const int* range, int ncx,
                      float* exported, int* nexp) {
  *nexp = 0;
  x->exp_out = exported; x->exp_n = nexp;
  if (x->mp) { x->mp->exp_out = exported; x->mp->exp_n = nexp; }
  p8_mixer_update(x, y_prev);
  for (int i = 0; i < nx; ++i) p8_mixer_add(x, in[i]);
  for (int i = 0; i < ncx; ++i) p8_mixer_set(x, cx[i], range[i]);
  return p8_mixer_p(x, y_prev);
}

/* ---- APM1 (:600-621) ---- */
typedef struct { int index, N; uint16_t* t; } OrcP8Apm1;
OrcP8Apm1* orc_p8_apm1_new(int n) {
  OrcP8Apm1* a = (OrcP8Apm1*)calloc(1, sizeof *a);
  a->N = n;
  a->t = (uint16_t*)malloc((size_t)n * 33 * 2);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < 33; ++j) a->t[i * 33 + j] = i == 0 ? (uint16_t)(p8_squash((j - 16) * 128) * 16) : a->t[j];
  return a;
}
int orc_p8_apm1_p(OrcP8Apm1* a, int y, int pr, int cxt, int rate) {
  pr = p8_stretch(pr);
  const int g = (y << 16) + (y << rate) - y - y;
  a->t[a->index] += (g - a->t[a->index]) >> rate;
  a->t[a->index + 1] += (g - a->t[a->index + 1]) >> rate;
  const int w = pr & 127;
  a->index = ((pr + 2048) >> 7) + cxt * 33;
  return (a->t[a->index] * (128 - w) + a->t[a->index + 1] * w) >> 11;
}

/* ---- StateMap (:623-645) ---- */
typedef struct { int cxt; uint16_t t[256]; } OrcP8StateMap;
OrcP8StateMap* orc_p8_statemap_new(void) {
  OrcP8StateMap* s = (OrcP8StateMap*)calloc(1, sizeof *s);
  for (int i = 0; i < 256; ++i) {
    int n0 = P8_STATE[4 * i + 2], n1 = P8_STATE[4 * i + 3];
    if (n0 == 0) n1 *= 64;
    if (n1 == 0) n0 *= 64;
    s->t[i] = (uint16_t)(65536 * (n1 + 1) / (n0 + n1 + 2));
  }
  return s;
}
int orc_p8_statemap_p(OrcP8StateMap* s, int y, int cx) {
  s->t[s->cxt] += ((y << 16) - s->t[s->cxt] + 128) >> 8;
  return s->t[s->cxt = cx] >> 4;
}

/* ---- StateMap32 (:645-690) and APM (:691-712) ---- */
typedef struct { int N, cxt; uint32_t* t; } OrcP8StateMap32;
static void p8_sm32_update(OrcP8StateMap32* s, int y, int limit) {
  uint32_t p0 = s->t[s->cxt];
  const int n = p0 & 1023;
  const int pr = p0 >> 10;
  if (n < limit) ++p0;
  else p0 = (p0 & 0xfffffc00u) | (uint32_t)limit;
  const int target = y << 22;
  const int delta = ((target - pr) >> 3) * (16384 / (n + n + 3));  /* dt[n], :8243-8245 */
  p0 += (uint32_t)delta & 0xfffffc00u;
  s->t[s->cxt] = p0;
}
OrcP8StateMap32* orc_p8_statemap32_new(int n) {
  OrcP8StateMap32* s = (OrcP8StateMap32*)calloc(1, sizeof *s);
  s->N = n;
  s->t = (uint32_t*)malloc((size_t)n * 4);
  if (n == 256) {
    for (int i = 0; i < n; ++i) {
      uint32_t n0 = P8_STATE[4 * i + 2], n1 = P8_STATE[4 * i + 3];
      if (n0 == 0) n1 *= 64;
      if (n1 == 0) n0 *= 64;
      s->t[i] = ((n1 << 16) / (n0 + n1 + 1)) << 16;
    }
  } else {
    for (int i = 0; i < n; ++i) s->t[i] = 1u << 31;
  }
  return s;
}
int orc_p8_statemap32_p(OrcP8StateMap32* s, int y, int cx, int limit) {
  p8_sm32_update(s, y, limit);
  return s->t[s->cxt = cx] >> 20;
}
OrcP8StateMap32* orc_p8_apm_new(int n) {
  OrcP8StateMap32* s = orc_p8_statemap32_new(n * 24);
  for (int i = 0; i < s->N; ++i) {
    const int p = ((i % 24 * 2 + 1) * 4096) / 48 - 2048;
    s->t[i] = ((uint32_t)p8_squash(p) << 20) + 6;
  }
  return s;
}
int orc_p8_apm_p(OrcP8StateMap32* s, int y, int pr, int cx, int limit) {
  p8_sm32_update(s, y, limit);
  pr = (p8_stretch(pr) + 2048) * 23;
  const int wt = pr & 0xfff;
  cx = cx * 24 + (pr >> 12);
  s->cxt = cx + (wt >> 11);
  return (int)(((s->t[cx] >> 13) * (uint32_t)(4096 - wt) + (s->t[cx + 1] >> 13) * (uint32_t)wt) >> 19);
}
void orc_p8_free(void* p) { free(p); }
