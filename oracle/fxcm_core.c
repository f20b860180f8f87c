/* oracle/fxcm_core.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the numeric building blocks of the vendored fxcm model (reference src/models/fxcmv1.cpp): the
 * squash / stretch / ilog tables and derived tables (:151-229, :4846-4875), Mixer1 with its SSE2 dot product and
 * training step (:472-660), StateMap (:672-705), StateMap1 (:707-736), APM (:1622-1643), RunContextMap (:756-829),
 * SmallStationaryContextMap (:831-863) and DirectStateMap (:1646-1683). All integer. Pinned against the reference's
 * own structs (oracle/ref_fxcmcore.cpp -> oracle/_ref/libcmixreffxcm.so) in tests/test_oracle_fxcmcore.py; the data
 * tables are dumped from the live reference build by scripts/gen_fxcm_tables.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fxcm_core.h"
#include "fxcm_tables.h"

int fx_squash(int d) { return d < -2047 ? 1 : d > 2047 ? 4095 : FX_SQUASH[d + 2047]; }
int fx_stretch(int p) { return FX_STRETCH[p]; }
int fx_ilog(int x) { return FX_ILOG[x & 255]; }
int fx_clp(int z) { return z < -2047 ? -2047 : z > 2047 ? 2047 : z; }
int fx_sc(int p) { return p > 0 ? p >> 7 : (p + 127) >> 7; }  /* :906-909 */
int fx_dt(int i) { return i == 1023 ? 1 : 4096 / (i + 2); }     /* :4848-4851 */
const uint8_t* fx_sta(int which) {
  static const uint8_t* const tabs[6] = {FX_STA1, FX_STA2, FX_STA4, FX_STA5, FX_STA6, FX_STA7};
  return tabs[which];
}
int fx_pre1(int state) {  /* pre2(STA7) :1684-1690 */
  const uint32_t n0 = FX_STA7[state * 4 + 2] * 3u + 1, n1 = FX_STA7[state * 4 + 3] * 3u + 1;
  return (int16_t)fx_clp(fx_stretch((int)((n1 << 12) / (n0 + n1)))) >> 2;
}
int orc_fx_squash(int d) { return fx_squash(d); }
int orc_fx_stretch(int p) { return fx_stretch(p); }
void orc_fx_tables(int16_t* squash4095, int16_t* stretch4096, uint8_t* ilog256, int32_t* dt1024, uint8_t* sta6x1024, int16_t* pre1_256,
                   int16_t* st2_p1_4096, int16_t* st2_p2_4096) {
  memcpy(squash4095, FX_SQUASH, sizeof FX_SQUASH);
  memcpy(stretch4096, FX_STRETCH, sizeof FX_STRETCH);
  memcpy(ilog256, FX_ILOG, 256);
  for (int i = 0; i < 1024; i++) dt1024[i] = fx_dt(i);
  for (int k = 0; k < 6; k++) memcpy(sta6x1024 + 1024 * k, fx_sta(k), 1024);
  for (int i = 0; i < 256; i++) pre1_256[i] = (int16_t)fx_pre1(i);
  for (int i = 0; i < 4096; i++) { st2_p1_4096[i] = (int16_t)fx_clp(fx_sc(12 * (i - 2048))); st2_p2_4096[i] = (int16_t)fx_clp(fx_sc(14 * (i - 2048))); }
}

/* ---- the model's input vector: add() stores the stretch-domain value and exports squash(value) / 4095 to cmix
 * (Inputs::add :197-202, AddPrediction :98-101). Some callers step the export index back so that the next add()
 * overwrites the slot (fx_unexport). ---- */
void fx_add(FxSink* s, int p) {
  s->n[s->ncount++] = (int16_t)p;
  s->exported[s->pidx++] = (float)fx_squash(p) * (float)(1.0 / 4095);
}
void fx_unexport(FxSink* s) { s->pidx--; }
static int sink_drain(FxSink* s, int16_t* out, float* exported, int* nexported) {
  const int n = s->ncount;
  memcpy(out, s->n, (size_t)n * 2);
  if (exported) { memcpy(exported, s->exported, (size_t)s->pidx * 4); *nexported = s->pidx; }
  s->ncount = s->pidx = 0;
  return n;
}

/* ---- Mixer1 ---- */
static int sat16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
/* dot_product, SSE2 form (:522-541): per pair of terms a 32-bit sum of two 16x16 products (pmaddwd), arithmetic >> 8,
 * accumulated modulo 2^32 in four lanes and added up -- order-free. */
static int fx_dot(const int16_t* t, const int16_t* w, int n) {
  uint32_t sum = 0;
  for (int i = 0; i + 1 < n; i += 2) {
    const uint32_t pair = (uint32_t)((int32_t)t[i] * w[i]) + (uint32_t)((int32_t)t[i + 1] * w[i + 1]);
    sum += (uint32_t)((int32_t)pair >> 8);
  }
  return (int32_t)sum;
}
/* train, SSE2 form (:543-557): w += ((sat(2 t) * err >> 16) + 1) >> 1 with 16-bit saturating adds */
static void fx_train(const int16_t* t, int16_t* w, int n, int e) {
  if (!e) return;
  const int16_t err = (int16_t)e;
  for (int i = 0; i < n; ++i) {
    int v = sat16(2 * (int)t[i]);
    v = (v * (int)err) >> 16;
    v = sat16(v + 1) >> 1;
    w[i] = (int16_t)sat16(v + (int)w[i]);
  }
}
FxMixer* fx_mixer_new(int n, int m, int shift, int elim, int uperr) {
  FxMixer* x = (FxMixer*)calloc(1, sizeof *x);
  x->N = n; x->M = m; x->shift1 = shift; x->elim = elim; x->uperr = uperr; x->pr = 2048;
  x->tx = (int16_t*)calloc((size_t)n, 2);
  x->wx = (int16_t*)malloc((size_t)n * m * 2);
  for (size_t i = 0; i < (size_t)n * m; ++i) x->wx[i] = 129;  /* setTxWx :653 */
  return x;
}
void fx_mixer_update(FxMixer* x, int y) {  /* :625-633 */
  int err = ((y << 12) - x->pr) * x->uperr / 4;
  if (err > 32767) err = 32767;
  if (err < -32768) err = -32768;
  if (err >= -x->elim && err <= x->elim) err = 0;
  x->err = err;
  fx_train(x->tx, x->wx + (size_t)x->cxt * x->N, x->N, err);
}
int fx_mixer_p(FxMixer* x) {  /* :636-640 */
  const int dp = (int32_t)((uint32_t)fx_dot(x->tx, x->wx + (size_t)x->cxt * x->N, x->N) * (uint32_t)x->shift1) >> 11;
  return x->pr = fx_squash(dp);
}
int fx_mixer_p1(FxMixer* x) {  /* :641-651: the clamped stretch-domain value goes on to the final mixers */
  int dp = (int32_t)((uint32_t)fx_dot(x->tx, x->wx + (size_t)x->cxt * x->N, x->N) * (uint32_t)x->shift1) >> 11;
  dp = fx_clp(dp);
  x->pr = fx_squash(dp);
  return dp;
}
FxMixer* orc_fx_mixer_new(int n, int m, int shift, int elim, int uperr) { return fx_mixer_new(n, m, shift, elim, uperr); }
int orc_fx_mixer_step(FxMixer* x, int y, const int16_t* in, int cxt, int elim, int use_p1, int* pr_out) {
  x->elim = elim;
  fx_mixer_update(x, y);
  memcpy(x->tx, in, (size_t)x->N * 2);
  x->cxt = cxt;
  const int r = use_p1 ? fx_mixer_p1(x) : fx_mixer_p(x);
  *pr_out = x->pr;
  return r;
}

/* ---- StateMap: bit-history state -> probability, fixed rate 1/8192 of the 32-bit value (:672-705) ---- */
void fx_statemap_init(FxStateMap* s, int n, const uint8_t* nn) {
  s->N = n; s->cxt = 0; s->pr = 2048;
  s->t = (uint32_t*)calloc((size_t)n, 4);
  for (int i = 0; i < n; ++i) {
    const uint32_t n0 = nn[(i & 255) * 4 + 2] * 3u + 1, n1 = nn[(i & 255) * 4 + 3] * 3u + 1;
    s->t[i] = ((n1 << 20) / (n0 + n1)) << 12;
  }
}
int fx_statemap_set(FxStateMap* s, int y, int c) {
  uint32_t* p = &s->t[s->cxt];
  *p += (uint32_t)(y << 19) - (*p >> 13);
  return s->pr = (int)(s->t[s->cxt = c] >> 20);
}
FxStateMap* orc_fx_statemap_new(int n, int which_sta) { FxStateMap* s = (FxStateMap*)calloc(1, sizeof *s); fx_statemap_init(s, n, fx_sta(which_sta)); return s; }
int orc_fx_statemap_set(FxStateMap* s, int y, int c) { return fx_statemap_set(s, y, c); }

/* ---- StateMap1: direct context -> probability with a count-driven rate (:707-736) ---- */
void fx_statemap1_init(FxStateMap1* s, int n, int limit) {
  s->N = n; s->cxt = 0; s->pr = 2048; s->mask = n - 1; s->limit = limit;
  s->t = (uint32_t*)malloc((size_t)n * 4);
  for (int i = 0; i < n; ++i) s->t[i] = 1u << 31;
}
int fx_statemap1_set(FxStateMap1* s, int y, int c) {
  uint32_t* p = &s->t[s->cxt];
  uint32_t p0 = *p;
  const int n = (int)(p0 & 1023);
  const uint32_t pr1 = p0 >> 12;
  p0 += (uint32_t)(n < s->limit);
  p0 += (((uint32_t)(y << 20) - pr1) * (uint32_t)fx_dt(n) + 512) & 0xfffffc00u;
  *p = p0;
  return s->pr = (int)(s->t[s->cxt = (c & s->mask)] >> 20);
}
FxStateMap1* orc_fx_statemap1_new(int n, int limit) { FxStateMap1* s = (FxStateMap1*)calloc(1, sizeof *s); fx_statemap1_init(s, n, limit); return s; }
int orc_fx_statemap1_set(FxStateMap1* s, int y, int c) { return fx_statemap1_set(s, y, c); }

/* ---- APM: 33-bin interpolated refinement (:1622-1643) ---- */
FxApm* fx_apm_new(int contexts) {
  FxApm* a = (FxApm*)calloc(1, sizeof *a);
  a->t = (uint16_t*)malloc((size_t)contexts * 33 * 2);
  for (int j = 0; j < 33; ++j) a->t[j] = (uint16_t)(fx_squash((j - 16) * 128) * 16);
  for (int i = 33; i < contexts * 33; ++i) a->t[i] = a->t[i - 33];
  return a;
}
int fx_apm_p(FxApm* a, int pr, int cxt, int rate, int y) {
  pr = fx_stretch(pr);
  const int g = (y << 16) + (y << rate) - y * 2;
  a->t[a->index] += (g - a->t[a->index]) >> rate;
  a->t[a->index + 1] += (g - a->t[a->index + 1]) >> rate;
  const int w = pr & 127;
  a->index = ((pr + 2048) >> 7) + cxt * 33;
  return (a->t[a->index] * (128 - w) + a->t[a->index + 1] * w) >> 11;
}
FxApm* orc_fx_apm_new(int contexts) { return fx_apm_new(contexts); }
int orc_fx_apm_p(FxApm* a, int pr, int cxt, int rate, int y) { return fx_apm_p(a, pr, cxt, rate, y); }

/* ---- RunContextMap: context -> (last byte, run count) in 4-byte slots, 4-way probe with move-to-front (:756-829) ---- */
static uint32_t rcm_find(FxRcm* r, uint32_t i) {  /* offset of byte 1 of the element */
  enum { B = 4, M = 4 };
  uint8_t* t = r->t;
  const uint16_t chk = (uint16_t)(((i >> 16) ^ i) & 0xffff);
  i = (i * M) & r->n;
  uint32_t p = 0;
  int j;
  for (j = 0; j < M; ++j) {
    p = (i + (uint32_t)j) * B;
    uint16_t cur;
    memcpy(&cur, t + p, 2);
    if (t[p + 2] == 0) { memcpy(t + p, &chk, 2); break; }
    if (cur == chk) break;
  }
  if (j == 0) return p + 1;
  uint8_t tmp[B];
  if (j == M) {
    --j;
    memset(tmp, 0, B);
    memcpy(tmp, &chk, 2);
    if (t[(i + (uint32_t)j) * B + 2] > t[(i + (uint32_t)j - 1) * B + 2]) --j;
  } else memcpy(tmp, t + p, B);
  memmove(t + (i + 1) * B, t + i * B, (size_t)j * B);
  memcpy(t + i * B, tmp, B);
  return i * B + 1;
}
void fx_rcm_init(FxRcm* r, int m, int rcm_ml) {
  r->t = (uint8_t*)calloc((size_t)m + 64, 1);
  r->n = (uint32_t)(m / 4 - 1);
  r->cp = 1;
  for (int k = 0; k < 256; k++) {
    int c = fx_ilog(k) * 8;
    if ((k & 1) == 0) c = c * rcm_ml / 4;
    r->rc[k + 256] = (int16_t)fx_clp(c);
    r->rc[k] = (int16_t)fx_clp(-c);
  }
}
void fx_rcm_set(FxRcm* r, uint32_t cx, int c1) {
  uint8_t* cp = r->t + r->cp;
  if (cp[0] == 0) { cp[0] = 2; cp[1] = (uint8_t)c1; }
  else if (cp[1] != c1) { cp[0] = 1; cp[1] = (uint8_t)c1; }
  else if (cp[0] < 254) cp[0] = (uint8_t)(cp[0] + 2);
  r->cp = rcm_find(r, cx) + 1;
}
int fx_rcm_p(const FxRcm* r, int bpos, int c0) {
  const uint8_t* cp = r->t + r->cp;
  const int bposshift = 7 - bpos, c0shift_bpos = (c0 << 1) ^ (256 >> bposshift);  /* update1 :4778-4779 */
  const int b = c0shift_bpos ^ (cp[1] >> bposshift);
  return b <= 1 ? r->rc[b * 256 + cp[0]] : 0;
}
int fx_rcm_mix(FxRcm* r, FxSink* s, int bpos, int c0) { fx_add(s, fx_rcm_p(r, bpos, c0)); return r->t[r->cp] != 0; }
FxRcm* orc_fx_rcm_new(int m, int ml) { FxRcm* r = (FxRcm*)calloc(1, sizeof *r); fx_rcm_init(r, m, ml); return r; }
void orc_fx_rcm_set(FxRcm* r, uint32_t cx, int c1) { fx_rcm_set(r, cx, c1); }
int orc_fx_rcm_mix(FxRcm* r, int y, int bpos, int c0, int16_t* out) {
  (void)y;
  FxSink s = {{0}, 0, {0}, 0};
  const int ret = fx_rcm_mix(r, &s, bpos, c0);
  sink_drain(&s, out, NULL, NULL);
  return ret;
}

/* ---- SmallStationaryContextMap (:831-863) ---- */
void fx_sscm_init(FxSscm* m, int bits_of_context, int input_bits) {
  m->Context = 0; m->Mask = (1 << bits_of_context) - 1; m->Stride = (1 << input_bits) - 1; m->bCount = 0; m->bTotal = input_bits; m->B = 0;
  m->N = (int)((1ull << bits_of_context) * ((1ull << input_bits) - 1));
  m->Data = (uint16_t*)malloc((size_t)m->N * 2);
  for (int i = 0; i < m->N; ++i) m->Data[i] = 0x7FFF;
  m->cp = 0;
}
void fx_sscm_set(FxSscm* m, uint32_t ctx) { m->Context = (int)(ctx & (uint32_t)m->Mask) * m->Stride; m->bCount = m->B = 0; }
void fx_sscm_mix(FxSscm* m, FxSink* s, int y, int r) {
  const int rate = r + 7;
  uint16_t* cp = &m->Data[m->cp];
  *cp += ((y << 16) - (*cp) + (1 << (rate - 1))) >> rate;
  m->B += (y && m->B > 0);
  m->cp = m->Context + m->B;
  const int Prediction = m->Data[m->cp] >> 4;
  fx_add(s, fx_stretch(Prediction) / 4);
  fx_add(s, (Prediction - 2048) / 8);
  fx_unexport(s);
  m->bCount++; m->B += m->B + 1;
  if (m->bCount == m->bTotal) m->bCount = m->B = 0;
}
FxSscm* orc_fx_sscm_new(int bits_of_context, int input_bits) { FxSscm* m = (FxSscm*)calloc(1, sizeof *m); fx_sscm_init(m, bits_of_context, input_bits); return m; }
void orc_fx_sscm_set(FxSscm* m, uint32_t ctx) { fx_sscm_set(m, ctx); }
int orc_fx_sscm_mix(FxSscm* m, int y, int rate, int16_t* out, float* exported, int* nexported) {
  FxSink s = {{0}, 0, {0}, 0};
  fx_sscm_mix(m, &s, y, rate);
  return sink_drain(&s, out, exported, nexported);
}

/* ---- DirectStateMap: c direct contexts, each a bit-history state fed to its own StateMap (:1646-1683) ---- */
void fx_dsm_init(FxDsm* d, int m, int c, const uint8_t* nn) {
  d->nn = nn; d->mask = (1u << m) - 1; d->index = 0; d->count = c;
  d->cxt = (int*)calloc((size_t)c, sizeof(int));
  d->CxtState = (uint8_t*)calloc((size_t)d->mask + 1, 1);
  d->sm = (FxStateMap*)calloc((size_t)c, sizeof(FxStateMap));
  for (int i = 0; i < c; i++) fx_statemap_init(&d->sm[i], 256, nn);
}
void fx_dsm_set(FxDsm* d, FxSink* s, uint32_t cx, int y) {
  uint8_t* st = &d->CxtState[d->cxt[d->index]];
  *st = d->nn[*st * 4 + y];
  d->cxt[d->index] = (int)(cx & d->mask);
  const int state = d->CxtState[d->cxt[d->index]];
  fx_statemap_set(&d->sm[d->index], y, state);
  fx_add(s, fx_stretch(d->sm[d->index].pr) >> 2);
  fx_unexport(s);
  fx_add(s, fx_pre1(state));
  fx_unexport(s);
  d->index++;
}
FxDsm* orc_fx_dsm_new(int m, int c, int which_sta) { FxDsm* d = (FxDsm*)calloc(1, sizeof *d); fx_dsm_init(d, m, c, fx_sta(which_sta)); return d; }
int orc_fx_dsm_step(FxDsm* d, int y, const uint32_t* cx, int n, int16_t* out, float* exported, int* nexported) {
  FxSink s = {{0}, 0, {0}, 0};
  for (int i = 0; i < n; i++) fx_dsm_set(d, &s, cx[i], y);
  d->index = 0;
  return sink_drain(&s, out, exported, nexported);
}
