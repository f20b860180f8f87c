/* oracle/sse_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the final SSE/APM stage (Eugene Shelwien's SSE as wired in
 * reference src/mixer/sse.cpp).  Integer core, float in/out.  Pinned against the
 * unmodified reference by tests/test_oracle_vs_ref.py.
 */
#include "cmix_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

enum { SCALElog = 15, SCALE = 1 << 15, hSCALE = SCALE / 2, mSCALE = SCALE - 1 };

/* volumes: sse.cpp:197-200 */
#define SM6_VOL (3 * 128 * 256 * 256)
#define MIX1_VOL (4 * 256 * 8 * 79)
#define SM7_VOL (3 * 32 * 256 * 255)
#define MIX2_VOL (3 * 2 * 256 * 256)

/* tuned constants: sse.cpp:180-195 */
enum {
  f0C = 10240, f1C = 7935, f2C = 9592, sm6wrB = 106, sm6mw = 0, sm6C1 = 8092,
  x1W0 = 7649, x1wr = 6202, f3C = 8200, f4C = 7677, sm7wrB = 127, sm7mw = 8192,
  sm7C1 = 8202, x2W0 = 2561, x2wr = 8320
};

static uint16_t g_st[SCALE], g_sq[SCALE];
static int g_tables_ready = 0;

/* sse.cpp:80-135 (Init_ST_SQ and its helpers; note the local log2/exp2). */
static double l2(double a) { return 1.44269504088896340736 * log(a); }
static double e2(double a) { return exp(a / 1.44269504088896340736); }
static double st(double p) { return l2((1 - p) / p); }
static double sq(double p) { return 1.0 / (1.0 + e2(p)); }

static void init_tables(void) {
  if (g_tables_ready) return;
  const double st_coef = (hSCALE - 1) / l2(SCALE - 1);
  const double sq_coef = 1.0 / st_coef;
  unsigned i, s, x, y;
  memset(g_st, 0, sizeof g_st);
  memset(g_sq, 0, sizeof g_sq);
  for (i = 1; i < SCALE; i++) {
    unsigned v = (unsigned)(sq((double)((int)i - hSCALE) * sq_coef) * SCALE);
    g_sq[i] = (uint16_t)v;
  }
  x = 0;
  g_st[x] = 0;
  for (i = 1; i < SCALE; i++) {
    s = (unsigned)(st((double)i / SCALE) * st_coef + hSCALE);
    g_st[i] = (uint16_t)s;
    if ((uint16_t)s != g_st[x]) {
      y = i - 1;
      g_sq[g_st[x]] = (uint16_t)((x + y + 1) / 2);
      x = i;
    }
  }
  g_tables_ready = 1;
}

void orc_sse_tables(uint16_t* t_st, uint16_t* t_sq) {
  init_tables();
  memcpy(t_st, g_st, sizeof g_st);
  memcpy(t_sq, g_sq, sizeof g_sq);
}

/* sse.cpp:138-143 */
static unsigned extrap(int p1, int C) {
  p1 = (((p1 - hSCALE) * C) >> 13) + hSCALE;
  if (p1 < 1) p1 = 1;
  if (p1 > mSCALE) p1 = mSCALE;
  return (unsigned)p1;
}

static int rdiv(int x, int a, int d) { return x >= 0 ? (x + a) >> d : -((-x + a) >> d); }

typedef struct { int P, sw; uint16_t* C1; } updstr;

/* SSEi<7>::SSE_Pred, sse.cpp:37-51 */
static int sse_pred(uint16_t* P, int iP, updstr* X) {
  int freq = (6 * iP) >> SCALElog;
  X->sw = (6 * iP) & mSCALE;
  X->C1 = &P[freq];
  int f = (((SCALE - X->sw) * X->C1[0] + X->sw * X->C1[1]) >> SCALElog) - 8192;
  if (f <= 0) f = 1;
  if (f >= SCALE) f = mSCALE;
  X->P = f;
  return f;
}

/* SSEi<7>::SSE_Update, sse.cpp:53-62 */
static void sse_update(int c, int wr0, updstr* X) {
  X->P = (X->P * (SCALE - wr0)) >> SCALElog;
  if (c == 0) X->P += wr0;
  int dC = X->C1[0] - X->C1[1];
  int sw_dC = (X->sw * dC + mSCALE) >> SCALElog;
  X->C1[0] = (uint16_t)(X->P + sw_dC + 8192);
  X->C1[1] = (uint16_t)(X->P - (dC - sw_dC) + 8192);
}

/* Mixer::Mixup, sse.cpp:166-170 (argument order as at the call sites :262,:268) */
static int mixup(int w, int s1, int s0) {
  int x = s1 + rdiv((w - hSCALE) * (s0 - s1), 1 << (SCALElog - 1), SCALElog);
  return (x > 0) ? ((x < SCALE) ? x : SCALE - 1) : 1;
}

/* Mixer::Update, sse.cpp:172-178 */
static void mix_update(int* w, int y, int p0, int p1, int wq, int pm) {
  int py = SCALE - (y << SCALElog);
  int e = py - pm;
  int d = rdiv(e * (p0 - p1), 1 << (SCALElog - 1), SCALElog);
  d = rdiv(d * wq, 1 << (SCALElog - 1), SCALElog);
  *w += d;
}

/* sse.cpp:154 M_mx1mask0: partial byte j (leading-1 form) quantised to 79 classes */
static int mx1mask(int j) {
  if (j < 32) return j ? j - 1 : 0;
  if (j < 64) return 31 + (j - 32) / 2;
  if (j < 128) return 47 + (j - 64) / 4;
  return 63 + (j - 128) / 8;
}
/* sse.cpp:158 M_sm7mask0 */
static int sm7mask(int j) { return j ? j - 1 : 0; }

struct orc_sse {
  uint16_t (*s6)[7];
  uint16_t (*s7)[7];
  int* x1;
  int* x2;
  updstr su6, su7;
  int sm6x, mix1, sm7x, mix2;
  unsigned mix1_s0, mix1_s1, mix1_p, mix2_s0, mix2_s1, mix2_p;
  unsigned j, pc, ffl;
};

/* M_T::M_Init, sse.cpp:216-228 with SSEi::Init :27-33 and Mixer::Init :161-163 */
orc_sse* orc_sse_create(void) {
  init_tables();
  orc_sse* s = (orc_sse*)calloc(1, sizeof *s);
  s->s6 = malloc((size_t)SM6_VOL * 14);
  s->s7 = malloc((size_t)SM7_VOL * 14);
  s->x1 = malloc((size_t)MIX1_VOL * 4);
  s->x2 = malloc((size_t)MIX2_VOL * 4);
  {
    int SCw = (SCALE - sm6mw) / 6, INC = sm6mw / 2 + 8192;
    for (size_t i = 0; i < SM6_VOL; i++)
      for (int k = 0, p = INC; k < 7; k++, p += SCw) s->s6[i][k] = (uint16_t)p;
  }
  {
    int SCw = (SCALE - sm7mw) / 6, INC = sm7mw / 2 + 8192;
    for (size_t i = 0; i < SM7_VOL; i++)
      for (int k = 0, p = INC; k < 7; k++, p += SCw) s->s7[i][k] = (uint16_t)p;
  }
  for (size_t i = 0; i < MIX1_VOL; i++) s->x1[i] = x1W0 + hSCALE;
  for (size_t i = 0; i < MIX2_VOL; i++) s->x2[i] = x2W0 + hSCALE;
  s->j = 1;
  s->pc = 0;
  s->ffl = 0;
  return s;
}

void orc_sse_destroy(orc_sse* s) {
  if (!s) return;
  free(s->s6);
  free(s->s7);
  free(s->x1);
  free(s->x2);
  free(s);
}

/* M_T1::M_Estimate, sse.cpp:243-289 */
static unsigned estimate(orc_sse* s, unsigned p) {
  unsigned j = s->j, pc = s->pc, ffl = s->ffl, prq = p >> 11;
  int a = (prq > 0) + (prq > 14);
  int b = (prq > 0) + (prq > 7) + (prq > 14);
  s->sm7x = ((((a << 5) + (int)(ffl & 31)) << 8) + (int)(pc & 255)) * 255 + sm7mask((int)j);
  s->mix2 = ((((a << 1) + (int)(ffl & 1)) << 8) + (int)(pc & 255)) * 256 + (int)j;
  s->sm6x = ((((a << 7) + (int)(ffl & 127)) << 8) + (int)(pc & 255)) * 256 + (int)j;
  s->mix1 = ((((b << 8) + (int)(ffl & 255)) << 3) + (int)((pc >> 5) & 7)) * 79 + mx1mask((int)j);

  unsigned p0 = p;
  unsigned p1 = (unsigned)sse_pred(s->s6[s->sm6x], g_sq[extrap(g_st[p0], f0C)], &s->su6);
  unsigned s0 = extrap(g_st[p0], f1C);
  unsigned s1 = extrap(g_st[p1], f2C);
  s->mix1_s0 = s0;
  s->mix1_s1 = s1;
  unsigned s2 = (unsigned)mixup(s->x1[s->mix1], (int)s0, (int)s1);
  s2 = extrap((int)s2, sm6C1);
  s->mix1_p = g_sq[s2];

  unsigned p2 = (unsigned)sse_pred(s->s7[s->sm7x], g_sq[extrap(g_st[p0], f3C)], &s->su7);
  unsigned s4 = extrap(g_st[p2], f4C);
  s->mix2_s0 = s2;
  s->mix2_s1 = s4;
  unsigned s5 = (unsigned)mixup(s->x2[s->mix2], (int)s2, (int)s4);
  s5 = extrap((int)s5, sm7C1);
  s->mix2_p = g_sq[s5];
  return s->mix2_p;
}

/* SSE::Predict, sse.cpp:320-324 */
float orc_sse_predict(orc_sse* s, float input) {
  int discrete = (int)(1 + (1 - input) * 32766);
  int est = (int)estimate(s, (unsigned)discrete);
  return (float)(1 - ((est - 1) / 32766.0));
}

/* M_T1::M_Update, sse.cpp:291-306 */
void orc_sse_perceive(orc_sse* s, int bit) {
  sse_update(bit, sm6wrB, &s->su6);
  mix_update(&s->x1[s->mix1], bit, (int)s->mix1_s0, (int)s->mix1_s1, x1wr, (int)s->mix1_p);
  sse_update(bit, sm7wrB, &s->su7);
  mix_update(&s->x2[s->mix2], bit, (int)s->mix2_s0, (int)s->mix2_s1, x2wr, (int)s->mix2_p);
  s->j += s->j + (unsigned)bit;
  if (s->j >= 256) {
    s->ffl = (uint8_t)(s->ffl * 2 + (s->pc >= 0x40));
    s->pc = (uint8_t)s->j;
    s->j = 1;
  }
}
