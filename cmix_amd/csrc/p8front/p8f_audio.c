/* p8front/p8f_audio.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's audio models (reference src/models/paq8.cpp:5510-5865), switched on by audioModel when it has read a RIFF /
 * WAVE header with uncompressed PCM (:5810-5865; the detector itself lives in p8f_front.c):
 *   audio8bModel :5552-5657   8-bit samples, mono / stereo: eight recursive-least-squares predictors per channel (OLS<double, int8_t>
 *                             :1364-1466, 28..128 taps over a geometric sample history, the other channel mixed in for stereo) and three
 *                             fixed extrapolations; each prediction is the context of three SmallStationaryContextMaps (33 maps)
 *   wavModel :5659-5804       16-bit samples: one 48-tap (mono: 48 own samples; stereo: 36 + 12 of the other channel) least-squares
 *                             predictor per channel, its covariance updated per sample and refactored every sample, in DOUBLE storage
 *                             with LONG DOUBLE accumulators (x87 80-bit on the x86-64 hosts this runs on, as in the reference build);
 *                             11 hashed contexts of a ContextMap and seven SmallStationaryContextMaps(8, 8) on residual classes
 * Both are followed by recordModel (:5861), whose maps are the generic ones (p8f_record.c). Everything here is a function of the byte
 * stream; the tables the predictions feed are recorded, the device learns. Sums run in the reference's index order, products are rounded
 * before they are added (-ffp-contract=off).
 * Parity: tests/test_p8stage_host.py (stage vs per-step hashes of the unmodified reference on WAV streams). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct CM1 CM1;
CM1* p8f_cm_new(uint64_t size_bytes, int count);
int p8f_cm_step(CM1* c, int y1, int bp, int c0, int c1, const uint64_t* ctx, int nset, int16_t* out, int* nout);
typedef struct DMap DMap;
DMap* p8f_dmap_new(int kind, int bits_of_context, int bits_per_context, int rate);
void p8f_dmap_set_direct(DMap* m, uint32_t ctx);
int p8f_dmap_mix(DMap* m, int y, int a, int mul, int div, int16_t* out);

#define PHI64 0x9E3779B97F4A7C15ull
static const uint64_t MUL[8] = {PHI64, 0x993DDEFFB1462949ull, 0xE9C91DC159AB0D2Dull, 0x83D6A14F1B0CED73ull,
                                0xA14F1B0CED5A841Full, 0xC0E51314A614F4EFull, 0xDA9CC2600AE45A27ull, 0x826797AA04A65737ull};
static uint64_t hashn(int n, const int64_t* x) {  /* hash(x0 .. x(n-1)) :742-773; int arguments widen with their sign */
  uint64_t h = 0;
  for (int i = 0; i < n; ++i) h += ((uint64_t)x[i] + 1) * MUL[i];
  return h;
}
#define H2(a, b) hashn(2, (const int64_t[]){(int64_t)(a), (int64_t)(b)})
#define H3(a, b, c) hashn(3, (const int64_t[]){(int64_t)(a), (int64_t)(b), (int64_t)(c)})
#define H4(a, b, c, d) hashn(4, (const int64_t[]){(int64_t)(a), (int64_t)(b), (int64_t)(c), (int64_t)(d)})
static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static unsigned ilog2u(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }
static unsigned popcnt(unsigned v) { return (unsigned)__builtin_popcount(v); }

/* the byte history as the audio models read it (:5513-5542); mode = the WAV kind (bit 0: stereo, bit 1: 16-bit, bit 2: the other sign
 * convention), S = the stereo offset wavModel sets (:5682) */
typedef struct { const uint8_t* hist; uint32_t bmask; int pos, mode, S; } Hist;
#define HB(h, i) ((int)(h)->hist[((uint32_t)(h)->pos - (uint32_t)(i)) & (h)->bmask])
static int s2(const Hist* h, int i) { return (int)(short)(HB(h, i) + 256 * HB(h, i - 1)); }
static int t2(const Hist* h, int i) { return (int)(short)(HB(h, i - 1) + 256 * HB(h, i)); }
static int X1(const Hist* h, int i) {
  switch (h->mode) {
    case 0: return HB(h, i) - 128;
    case 1: return HB(h, i << 1) - 128;
    case 2: return s2(h, i << 1);
    case 3: return s2(h, i << 2);
    case 4: return (HB(h, i) ^ 128) - 128;
    case 5: return (HB(h, i << 1) ^ 128) - 128;
    case 6: return t2(h, i << 1);
    case 7: return t2(h, i << 2);
    default: return 0;
  }
}
static int X2(const Hist* h, int i) {
  switch (h->mode) {
    case 0: return HB(h, i + h->S) - 128;
    case 1: return HB(h, (i << 1) - 1) - 128;
    case 2: return s2(h, (i + h->S) << 1);
    case 3: return s2(h, (i << 2) - 2);
    case 4: return (HB(h, i + h->S) ^ 128) - 128;
    case 5: return (HB(h, (i << 1) - 1) ^ 128) - 128;
    case 6: return t2(h, (i + h->S) << 1);
    case 7: return t2(h, (i << 2) - 2);
    default: return 0;
  }
}
static int sclip8(int v) { return v < -128 ? -128 : v > 127 ? 127 : v; }

/* OLS<double, int8_t> :1364-1466 with the Add() / Predict() pair: the taps are pushed one by one, Update() runs the factorisation every
 * kmax-th sample */
typedef struct { int n, kmax, km, index; double lambda, nu; double *x, *w, *b, *cov, *chol; } AOls;
static void aols_init(AOls* o, int n, int kmax, double lambda) {
  memset(o, 0, sizeof *o);
  o->n = n; o->kmax = kmax; o->lambda = lambda; o->nu = 0.001;
  o->x = (double*)calloc((size_t)n, sizeof(double)); o->w = (double*)calloc((size_t)n, sizeof(double)); o->b = (double*)calloc((size_t)n, sizeof(double));
  o->cov = (double*)calloc((size_t)n * n, sizeof(double)); o->chol = (double*)calloc((size_t)n * n, sizeof(double));
}
#define AC(o, a, i, j) ((o)->a[(size_t)(i) * (size_t)(o)->n + (size_t)(j)])
static int aols_factor(AOls* o) {
  const int n = o->n;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) AC(o, chol, i, j) = AC(o, cov, i, j);
  for (int i = 0; i < n; i++) AC(o, chol, i, i) += o->nu;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < i; j++) {
      double sum = AC(o, chol, i, j);
      for (int k = 0; k < j; k++) sum -= (AC(o, chol, i, k) * AC(o, chol, j, k));
      AC(o, chol, i, j) = sum / AC(o, chol, j, j);
    }
    double sum = AC(o, chol, i, i);
    for (int k = 0; k < i; k++) sum -= (AC(o, chol, i, k) * AC(o, chol, i, k));
    if (sum > 1E-8) AC(o, chol, i, i) = sqrt(sum);
    else return 1;
  }
  return 0;
}
static void aols_solve(AOls* o) {
  const int n = o->n;
  for (int i = 0; i < n; i++) {
    double sum = o->b[i];
    for (int j = 0; j < i; j++) sum -= (AC(o, chol, i, j) * o->w[j]);
    o->w[i] = sum / AC(o, chol, i, i);
  }
  for (int i = n - 1; i >= 0; i--) {
    double sum = o->w[i];
    for (int j = i + 1; j < n; j++) sum -= (AC(o, chol, j, i) * o->w[j]);
    o->w[i] = sum / AC(o, chol, i, i);
  }
}
static void aols_update(AOls* o, int val) {
  const int n = o->n;
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) AC(o, cov, j, i) = o->lambda * AC(o, cov, j, i) + (1.0 - o->lambda) * (o->x[j] * o->x[i]);
  for (int i = 0; i < n; i++) o->b[i] = o->lambda * o->b[i] + (1.0 - o->lambda) * (o->x[i] * ((double)val - 0.0));
  if (++o->km >= o->kmax) {
    if (!aols_factor(o)) aols_solve(o);
    o->km = 0;
  }
}
static void aols_add(AOls* o, double v) { if (o->index < o->n) o->x[o->index++] = v - 0.0; }
static double aols_predict(AOls* o) {
  o->index = 0;
  double sum = 0.;
  for (int i = 0; i < o->n; i++) sum += o->w[i] * o->x[i];
  return sum + 0.0;
}

/* ---------------------------------------------------------------- audio8bModel :5552-5657 */
enum { A_OLS = 8, A_PRD = A_OLS + 3 };
typedef struct Audio8 {
  DMap* map[A_PRD][3];
  AOls ols[A_OLS][2];
  int prd[A_PRD][2][2], residuals[A_PRD][2];
  int stereo, ch, rpos, last_pos;
  uint32_t mask, err_log, mx_ctx;
} Audio8;

Audio8* p8f_audio8_new(void) {
  static const int n[A_OLS] = {128, 90, 90, 90, 90, 90, 28, 28}, kmax[A_OLS] = {24, 30, 31, 32, 33, 34, 4, 3};
  static const double lambda[A_OLS] = {0.9975, 0.9965, 0.996, 0.995, 0.995, 0.9985, 0.98, 0.992};
  Audio8* m = (Audio8*)calloc(1, sizeof *m);
  for (int i = 0; i < A_PRD; i++) for (int j = 0; j < 3; j++) m->map[i][j] = p8f_dmap_new(0, 11, 1, 0);
  for (int i = 0; i < A_OLS; i++) for (int c = 0; c < 2; c++) aols_init(&m->ols[i][c], n[i], kmax[i], lambda[i]);
  return m;
}

/* One step. info: the WAV kind (0 mono, 1 stereo, 4 / 5 the same with the other sign convention); blpos: bytes into the block; record: Stats.Record
 * (in / out); sets[5] / ranges[5]. Returns the number of inputs (66). */
int p8f_audio8_step(Audio8* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int info, int blpos, uint32_t* record, int16_t* out,
                    int* sets, int* ranges) {
  Hist h = {hist, bmask, pos, info, 0};
  const int B = (int8_t)(c0 << (8 - bpos));
  if (bpos == 0) {
    m->rpos = (pos == m->last_pos + 1) ? m->rpos + 1 : 0;
    m->last_pos = pos;
    if (m->rpos == 0) {
      m->stereo = info & 1;
      m->mask = 0;
      *record = ((uint32_t)(m->stereo + 1) << 16) | (*record & 0xFFFF);
    }
    const int stereo = m->stereo;
    m->ch = stereo ? blpos & 1 : 0;
    const int ch = m->ch;
    const int s = (int8_t)((int)((info & 4) > 0 ? HB(&h, 1) ^ 128 : HB(&h, 1)) - 128);
    const int pch = ch ^ stereo;
    int i = 0;
    for (m->err_log = 0; i < A_OLS; i++) {
      aols_update(&m->ols[i][pch], s);
      m->residuals[i][pch] = s - m->prd[i][pch][0];
      const uint32_t ar = (uint32_t)abs(m->residuals[i][pch]);
      m->mask += m->mask + (ar > 4);
      m->err_log += ar * ar;
    }
    for (; i < A_PRD; i++) m->residuals[i][pch] = s - m->prd[i][pch][0];
    m->err_log = (uint32_t)imin(0xF, (int)ilog2u(m->err_log));
    m->mx_ctx = ilog2u((unsigned)imin(0x1F, (int)popcnt(m->mask))) * 2 + (uint32_t)ch;
    int k1 = 90, k2 = k1 - 12 * stereo;
    int j;
    for (j = (i = 1); j <= k1; j++, i += 1 << ((j > 8) + (j > 16) + (j > 64))) aols_add(&m->ols[1][ch], (double)X1(&h, i));
    for (j = (i = 1); j <= k2; j++, i += 1 << ((j > 5) + (j > 10) + (j > 17) + (j > 26) + (j > 37))) aols_add(&m->ols[2][ch], (double)X1(&h, i));
    for (j = (i = 1); j <= k2; j++, i += 1 << ((j > 3) + (j > 7) + (j > 14) + (j > 20) + (j > 33) + (j > 49))) aols_add(&m->ols[3][ch], (double)X1(&h, i));
    for (j = (i = 1); j <= k2; j++, i += 1 + (j > 4) + (j > 8)) aols_add(&m->ols[4][ch], (double)X1(&h, i));
    for (j = (i = 1); j <= k1; j++, i += 2 + ((j > 3) + (j > 9) + (j > 19) + (j > 36) + (j > 61))) aols_add(&m->ols[5][ch], (double)X1(&h, i));
    if (stereo) {
      for (i = 1; i <= k1 - k2; i++) {
        const double v = (double)X2(&h, i);
        aols_add(&m->ols[2][ch], v); aols_add(&m->ols[3][ch], v); aols_add(&m->ols[4][ch], v);
      }
    }
    k1 = 28; k2 = k1 - 6 * stereo;
    for (i = 1; i <= k2; i++) {
      const double v = (double)X1(&h, i);
      aols_add(&m->ols[0][ch], v); aols_add(&m->ols[6][ch], v); aols_add(&m->ols[7][ch], v);
    }
    for (; i <= 96; i++) aols_add(&m->ols[0][ch], (double)X1(&h, i));
    if (stereo) {
      for (i = 1; i <= k1 - k2; i++) {
        const double v = (double)X2(&h, i);
        aols_add(&m->ols[0][ch], v); aols_add(&m->ols[6][ch], v); aols_add(&m->ols[7][ch], v);
      }
      for (; i <= 32; i++) aols_add(&m->ols[0][ch], (double)X2(&h, i));
    } else
      for (; i <= 128; i++) aols_add(&m->ols[0][ch], (double)X1(&h, i));
    for (i = 0; i < A_OLS; i++) {
      m->prd[i][ch][0] = sclip8((int)floor(aols_predict(&m->ols[i][ch])));
      m->prd[i][ch][1] = sclip8(m->prd[i][ch][0] + m->residuals[i][pch]);
    }
    m->prd[i++][ch][0] = sclip8(X1(&h, 1) * 2 - X1(&h, 2));
    m->prd[i++][ch][0] = sclip8(X1(&h, 1) * 3 - X1(&h, 2) * 3 + X1(&h, 3));
    m->prd[i][ch][0] = sclip8(X1(&h, 1) * 4 - X1(&h, 2) * 6 + X1(&h, 3) * 4 - X1(&h, 4));
    for (i = A_OLS; i < A_PRD; i++) m->prd[i][ch][1] = sclip8(m->prd[i][ch][0] + m->residuals[i][pch]);
  }
  const int ch = m->ch;
  int nx = 0;
  for (int i = 0; i < A_PRD; i++) {
    const uint32_t ctx = (uint32_t)((m->prd[i][ch][0] - B) * 8 + bpos);
    p8f_dmap_set_direct(m->map[i][0], ctx);
    p8f_dmap_set_direct(m->map[i][1], ctx);
    p8f_dmap_set_direct(m->map[i][2], (uint32_t)((m->prd[i][ch][1] - B) * 8 + bpos));
    nx += p8f_dmap_mix(m->map[i][0], y, 6, 1, 2 + (i >= A_OLS), out + nx);
    nx += p8f_dmap_mix(m->map[i][1], y, 9, 1, 2 + (i >= A_OLS), out + nx);
    nx += p8f_dmap_mix(m->map[i][2], y, 7, 1, 3, out + nx);
  }
  int ns = 0;
#define SET(v, r) do { sets[ns] = (int)(v); ranges[ns] = (int)(r); ++ns; } while (0)
  SET((m->err_log << 8) | (uint32_t)c0, 4096);
  SET(((uint32_t)(uint8_t)m->mask << 3) | ((uint32_t)ch << 2) | (uint32_t)(bpos >> 1), 2048);
  SET((m->mx_ctx << 7) | (uint32_t)(HB(&h, 1) >> 1), 1280);
  SET((m->err_log << 4) | ((uint32_t)ch << 3) | (uint32_t)bpos, 256);
  SET(m->mx_ctx, 10);
#undef SET
  return nx;
}

/* ---------------------------------------------------------------- wavModel :5659-5804 */
enum { W_N = 49 };
typedef struct Wav16 {
  CM1* cm;
  DMap* scm[7];
  int pr[3][2], n[2], counter[2];
  double F[W_N][W_N][2], L[W_N][W_N];
  int rpos, last_pos;
  int bits, channels, w, ch, col, S, D;
  int z1, z2, z3, z4, z5, z6, z7;
  int cm_set;   /* the ContextMap got its contexts this byte */
} Wav16;

Wav16* p8f_wav16_new(int level) {
  Wav16* m = (Wav16*)calloc(1, sizeof *m);
  for (int i = 0; i < 7; i++) m->scm[i] = p8f_dmap_new(0, 8, 8, 0);   /* scm1..7 are constructed before cm (:5666-5667) */
  m->cm = p8f_cm_new((0x10000ull << level) * 2, 10 + 1);
  return m;
}

/* One step. info: the WAV kind (2 mono, 3 stereo, 6 / 7 big-endian samples); record: Stats.Record (in / out); sets[5] / ranges[5];
 * *cm_active: the ContextMap has contexts this byte (not during a block's first sample): 55 of the step's inputs. Returns the inputs added. */
int p8f_wav16_step(Wav16* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int info, uint32_t* record, int16_t* out, int* sets,
                   int* ranges, int* cm_active) {
  Hist h = {hist, bmask, pos, info, m->S};
  const double a = 0.996, a2 = 1 / a;
  int j, k, l, i = 0;
  long double sum;
  if (!bpos) {
    m->rpos = (pos == m->last_pos + 1) ? m->rpos + 1 : 0;
    m->last_pos = pos;
  }
  if (!bpos && !m->rpos) {
    m->bits = ((info % 4) / 2) * 8 + 8;
    m->channels = info % 2 + 1;
    m->col = 0;
    m->w = m->channels * (m->bits >> 3);
    if (m->channels == 1) { m->S = 48; m->D = 0; } else { m->S = 36; m->D = 12; }
    h.S = m->S;
    for (j = 0; j < m->channels; j++) {
      for (k = 0; k <= m->S + m->D; k++) for (l = 0; l <= m->S + m->D; l++) { m->F[k][l][j] = 0; m->L[k][l] = 0; }
      m->F[1][0][j] = 1;
      m->n[j] = m->counter[j] = m->pr[2][j] = m->pr[1][j] = m->pr[0][j] = 0;
      m->z1 = m->z2 = m->z3 = m->z4 = m->z5 = m->z6 = m->z7 = 0;
    }
  }
  const int S = m->S, D = m->D, bits = m->bits, channels = m->channels, w = m->w;
  if (!bpos) m->cm_set = 0;
  if (!bpos && m->rpos >= w) {
    m->ch = m->rpos % w;
    const int ch = m->ch;
    const int msb = ch % (bits >> 3);
    const int chn = ch / (bits >> 3);
    if (!msb) {
      m->z1 = X1(&h, 1); m->z2 = X1(&h, 2); m->z3 = X1(&h, 3); m->z4 = X1(&h, 4); m->z5 = X1(&h, 5);
      k = X1(&h, 1);
      for (l = 0; l <= imin(S, m->counter[chn] - 1); l++) { m->F[0][l][chn] *= a; m->F[0][l][chn] += X1(&h, l + 1) * k; }
      for (l = 1; l <= imin(D, m->counter[chn]); l++) { m->F[0][l + S][chn] *= a; m->F[0][l + S][chn] += X2(&h, l + 1) * k; }
      if (channels == 2) {
        k = X2(&h, 2);
        for (l = 1; l <= imin(D, m->counter[chn]); l++) { m->F[S + 1][l + S][chn] *= a; m->F[S + 1][l + S][chn] += X2(&h, l + 1) * k; }
        for (l = 1; l <= imin(S, m->counter[chn] - 1); l++) { m->F[l][S + 1][chn] *= a; m->F[l][S + 1][chn] += X1(&h, l + 1) * k; }
        m->z6 = X2(&h, 1) + X1(&h, 1) - X2(&h, 2); m->z7 = X2(&h, 1);
      } else { m->z6 = 2 * X1(&h, 1) - X1(&h, 2); m->z7 = X1(&h, 1); }
      if (++m->n[chn] == 1) {
        if (channels == 1) { for (k = 1; k <= S + D; k++) for (l = k; l <= S + D; l++) m->F[k][l][chn] = (m->F[k - 1][l - 1][chn] - X1(&h, k) * X1(&h, l)) * a2; }
        else for (k = 1; k <= S + D; k++) if (k != S + 1) for (l = k; l <= S + D; l++) if (l != S + 1)
          m->F[k][l][chn] = (m->F[k - 1][l - 1][chn] - (k - 1 <= S ? X1(&h, k) : X2(&h, k - S)) * (l - 1 <= S ? X1(&h, l) : X2(&h, l - S))) * a2;
        for (i = 1; i <= S + D; i++) {
          sum = m->F[i][i][chn];
          for (k = 1; k < i; k++) sum -= m->L[i][k] * m->L[i][k];
          sum = floorl(sum + 0.5);
          sum = 1 / sum;
          if (sum > 0) {
            m->L[i][i] = (double)sqrtl(sum);
            for (j = (i + 1); j <= S + D; j++) {
              sum = m->F[i][j][chn];
              for (k = 1; k < i; k++) sum -= m->L[j][k] * m->L[i][k];
              sum = floorl(sum + 0.5);
              m->L[j][i] = (double)(sum * m->L[i][i]);
            }
          } else break;
        }
        if (i > S + D && m->counter[chn] > S + 1) {
          for (k = 1; k <= S + D; k++) {
            m->F[k][0][chn] = m->F[0][k][chn];
            for (j = 1; j < k; j++) m->F[k][0][chn] -= m->L[k][j] * m->F[j][0][chn];
            m->F[k][0][chn] *= m->L[k][k];
          }
          for (k = S + D; k > 0; k--) {
            for (j = k + 1; j <= S + D; j++) m->F[k][0][chn] -= m->L[j][k] * m->F[j][0][chn];
            m->F[k][0][chn] *= m->L[k][k];
          }
        }
        m->n[chn] = 0;
      }
      sum = 0;
      for (l = 1; l <= S + D; l++) sum += m->F[l][0][chn] * (l <= S ? X1(&h, l) : X2(&h, l - S));
      m->pr[2][chn] = m->pr[1][chn];
      m->pr[1][chn] = m->pr[0][chn];
      m->pr[0][chn] = (int)floorl(sum);
      m->counter[chn]++;
    }
    const int y1 = m->pr[0][chn], y2 = m->pr[1][chn], y3 = m->pr[2][chn];
    int x1 = HB(&h, 1), x2 = HB(&h, 2), x3 = HB(&h, 3);
    if (info == 4 || info == 5) { x1 ^= 128; x2 ^= 128; }
    if (bits == 8) { x1 -= 128; x2 -= 128; }
    const int t = ((bits == 8) || ((!msb) ^ (info < 6)));
    const int z1 = m->z1, z2 = m->z2, z3 = m->z3, z4 = m->z4, z5 = m->z5, z6 = m->z6, z7 = m->z7;
    uint64_t cx[11];
    int n = 0;
    int64_t ii = ch << 4;
    if ((msb) ^ (info < 6)) {
      cx[n++] = H2(++ii, y1 & 0xff);
      cx[n++] = H3(++ii, y1 & 0xff, ((z1 - y2 + z2 - y3) >> 1) & 0xff);
      cx[n++] = H3(++ii, x1, y1 & 0xff);
      cx[n++] = H4(++ii, x1, x2 >> 3, x3);
      if (bits == 8) cx[n++] = H3(++ii, y1 & 0xFE, ilog2u((unsigned)abs((int)(z1 - y2))) * 2 + (z1 > y2));
      else cx[n++] = H2(++ii, (y1 + z1 - y2) & 0xff);
      cx[n++] = H2(++ii, x1);
      cx[n++] = H3(++ii, x1, x2);
      cx[n++] = H2(++ii, z1 & 0xff);
      cx[n++] = H2(++ii, (z1 * 2 - z2) & 0xff);
      cx[n++] = H2(++ii, z6 & 0xff);
      cx[n++] = H3(++ii, y1 & 0xFF, ((z1 - y2 + z2 - y3) / (bits >> 3)) & 0xFF);
    } else {
      cx[n++] = H2(++ii, (y1 - x1 + z1 - y2) >> 8);
      cx[n++] = H2(++ii, (y1 - x1) >> 8);
      cx[n++] = H2(++ii, (y1 - x1 + z1 * 2 - y2 * 2 - z2 + y3) >> 8);
      cx[n++] = H3(++ii, (y1 - x1) >> 8, (z1 - y2 + z2 - y3) >> 9);
      cx[n++] = H2(++ii, z1 >> 12);
      cx[n++] = H2(++ii, x1);
      cx[n++] = H4(++ii, x1 >> 7, x2, x3 >> 7);
      cx[n++] = H2(++ii, z1 >> 8);
      cx[n++] = H2(++ii, (z1 * 2 - z2) >> 8);
      cx[n++] = H2(++ii, y1 >> 8);
      cx[n++] = H2(++ii, (y1 - x1) >> 6);
    }
    int kk = 0;
    p8f_cm_step(m->cm, y, 0, c0, x1, cx, n, out + 14, &kk);   /* (its inputs follow the seven maps' 14) */
    m->cm_set = 1;
    p8f_dmap_set_direct(m->scm[0], (uint32_t)(t * ch));
    p8f_dmap_set_direct(m->scm[1], (uint32_t)((t * ((z1 - x1 + y1) >> 9)) & 0xff));
    p8f_dmap_set_direct(m->scm[2], (uint32_t)((t * ((z1 * 2 - z2 - x1 + y1) >> 8)) & 0xff));
    p8f_dmap_set_direct(m->scm[3], (uint32_t)((t * ((z1 * 3 - z2 * 3 + z3 - x1) >> 7)) & 0xff));
    p8f_dmap_set_direct(m->scm[4], (uint32_t)((t * ((z1 + z7 - x1 + y1 * 2) >> 10)) & 0xff));
    p8f_dmap_set_direct(m->scm[5], (uint32_t)((t * ((z1 * 4 - z2 * 6 + z3 * 4 - z4 - x1) >> 7)) & 0xff));
    p8f_dmap_set_direct(m->scm[6], (uint32_t)((t * ((z1 * 5 - z2 * 10 + z3 * 10 - z4 * 5 + z5 - x1 + y1) >> 9)) & 0xff));
  }
  int nx = 0;
  for (i = 0; i < 7; i++) nx += p8f_dmap_mix(m->scm[i], y, 7, 1, 4, out + nx);   /* scmN.mix(m): rate 7, 1 / 4 */
  if (m->cm_set) {
    int kk = 0;
    if (bpos) p8f_cm_step(m->cm, y, bpos, c0, 0, NULL, 0, out + nx, &kk);
    else kk = 55;
    nx += kk;
  }
  *cm_active = m->cm_set;
  *record = ((uint32_t)w << 16) | (*record & 0xFFFF);
  if (++m->col >= w * 8) m->col = 0;
  const int col = m->col;
  int ns = 0;
#define SET(v, r) do { sets[ns] = (int)(v); ranges[ns] = (int)(r); ++ns; } while (0)
  SET(m->ch + 4 * (int)ilog2u((unsigned)(col & (bits - 1))), 4 * 8);
  SET(col % bits < 8, 2);
  SET(col % bits, bits);
  SET(col, w * 8);
  SET(c0, 256);
#undef SET
  return nx;
}
