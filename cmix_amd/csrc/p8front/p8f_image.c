/* p8front/p8f_image.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's image models: im24bitModel (24/32-bit pixels, below) and im8bitModel (8-bit palette / grayscale, at the end
 * of the file; reference src/models/paq8.cpp:4743-4999).
 *
 * The 24/32-bit image model (reference src/models/paq8.cpp:5001-5353, im24bitModel), switched on by
 * contextModel2 for IMAGE24 / IMAGE32 blocks (:8165-8166) and by imgModel for 24/32-bit BMP / TGA payloads inside other blocks
 * (:5386-5504). Per byte: the pixel neighbourhood out of the byte history, 76 + 58 neighbourhood predictors (contexts of 100
 * StationaryMaps and 59 SmallStationaryContextMaps), six recursive-least-squares predictors per colour plane (OLS<double, U8>
 * :1364-1466, the same f64 order of operations as p8f_lpm.c) and 47 hashed contexts of one ContextMap. Per bit: the maps' contexts
 * with the partial byte folded in, and the 13 weight-set selectors (:5335-5352). Everything that learns is recorded, not held:
 * ContextMap -> the model's family instance, the maps -> the model's lane table (p8_rec.h P8XLayout).
 *
 * Not reproduced: rows whose byte width is not a multiple of the pixel size. The reference then indexes its OLS array with
 * colour = stride + 1 (:5043 with :5226: ols[j][4] of a [6][4] array), i.e. reads past it; the step function returns -1 and the
 * stream is refused (P8F_ERR_IMAGE_PADDING) rather than coded differently.
 * Parity: tests/test_p8stage_host.py (stage vs per-step hashes of the unmodified reference's 1591 values on image streams). */
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
void p8f_dmap_set(DMap* m, uint64_t ctx);
int p8f_dmap_mix(DMap* m, int y, int a, int mul, int div, int16_t* out);
int p8f_ilog(int x);
uint32_t p8f_finalize64(uint64_t h, int bits);

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
#define H5(a, b, c, d, e) hashn(5, (const int64_t[]){(int64_t)(a), (int64_t)(b), (int64_t)(c), (int64_t)(d), (int64_t)(e)})

static int imax(int a, int b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static unsigned ilog2u(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }   /* ilog2 :244-251 */
static uint8_t clip(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }                     /* Clip :4183 */
static uint8_t clamp4(int v, uint8_t a, uint8_t b, uint8_t c, uint8_t d) {                          /* Clamp4 :4186 */
  const int hi = imax(a, imax(b, imax(c, d))), lo = imin(a, imin(b, imin(c, d)));
  return (uint8_t)imin(hi, imax(lo, v));
}
static uint8_t lmdq(uint8_t a, uint8_t b, int limit) {                                              /* LogMeanDiffQt :4190 */
  if (a == b) return 0;
  const int q = (int)ilog2u((unsigned)((a + b) / imax(2, abs(a - b) * 2) + 1));
  return (uint8_t)(((a > b) << 3) | imin(limit, q));
}
#define LMD(a, b) lmdq((uint8_t)(a), (uint8_t)(b), 7)
static uint32_t logqt(uint8_t v, int bits) { return (uint32_t)(0x100 | v) >> imax(0, (int)ilog2u(v) - bits); }   /* LogQt :4193 */

/* OLS<double, U8> :1364-1466 with kmax = 1, nu = 0.001: Update() after every sample, then Factor() and -- if the matrix is
 * positive definite -- Solve(). Sums in the reference's index order, every product rounded before it is added. */
enum { OLS_MAXN = 32 };
typedef struct { int n; double lambda, nu; double x[OLS_MAXN], w[OLS_MAXN], b[OLS_MAXN], cov[OLS_MAXN][OLS_MAXN], chol[OLS_MAXN][OLS_MAXN]; } Ols;
static void ols_init(Ols* o, int n, double lambda) { memset(o, 0, sizeof *o); o->n = n; o->lambda = lambda; o->nu = 0.001; }
static int ols_factor(Ols* o) {
  const int n = o->n;
  for (int i = 0; i < n; i++) for (int j = 0; j < n; j++) o->chol[i][j] = o->cov[i][j];
  for (int i = 0; i < n; i++) o->chol[i][i] += o->nu;
  for (int i = 0; i < n; i++) {
    for (int j = 0; j < i; j++) {
      double sum = o->chol[i][j];
      for (int k = 0; k < j; k++) sum -= (o->chol[i][k] * o->chol[j][k]);
      o->chol[i][j] = sum / o->chol[j][j];
    }
    double sum = o->chol[i][i];
    for (int k = 0; k < i; k++) sum -= (o->chol[i][k] * o->chol[i][k]);
    if (sum > 1E-8) o->chol[i][i] = sqrt(sum);
    else return 1;
  }
  return 0;
}
static void ols_solve(Ols* o) {
  const int n = o->n;
  for (int i = 0; i < n; i++) {
    double sum = o->b[i];
    for (int j = 0; j < i; j++) sum -= (o->chol[i][j] * o->w[j]);
    o->w[i] = sum / o->chol[i][i];
  }
  for (int i = n - 1; i >= 0; i--) {
    double sum = o->w[i];
    for (int j = i + 1; j < n; j++) sum -= (o->chol[j][i] * o->w[j]);
    o->w[i] = sum / o->chol[i][i];
  }
}
static void ols_update(Ols* o, uint8_t val) {
  const int n = o->n;
  for (int j = 0; j < n; j++)
    for (int i = 0; i < n; i++) o->cov[j][i] = o->lambda * o->cov[j][i] + (1.0 - o->lambda) * (o->x[j] * o->x[i]);
  for (int i = 0; i < n; i++) o->b[i] = o->lambda * o->b[i] + (1.0 - o->lambda) * (o->x[i] * ((double)val - 0.0));
  if (!ols_factor(o)) ols_solve(o);
}
static double ols_predict(Ols* o, const uint8_t* v) {   /* Predict(const T** p): the taps become x[] */
  double sum = 0.;
  for (int i = 0; i < o->n; i++) sum += o->w[i] * (o->x[i] = (double)v[i] - 0.0);
  return sum + 0.0;
}

/* ---------------------------------------------------------------- im24bitModel :5001-5353 */
enum { N_MAPS0 = 18, N_MAPS1 = 76, N_OLS = 6, N_MAPS = N_MAPS0 + N_MAPS1 + N_OLS, N_SC = 59, N_CM = 47 };
typedef struct Im24 {
  CM1* cm;
  DMap* map[N_MAPS];
  DMap* sc[N_SC];
  Ols ols[N_OLS][4];
  int color, stride, padding, last_pos, x, line, col;
  int columns[2], column[2], ctx[2];
  uint8_t mctx[N_MAPS1], sctx[N_SC - 1], pols[N_OLS];
  /* the neighbourhood values the per-bit part reads again */
  uint8_t W, WW, N, NN, NW, NE, NNE, NNW, NWW, p1, p2, Wp1, Wp2, Np1, Np2, NWp1, NWp2;
} Im24;

Im24* p8f_im24_new(int level) {
  static const uint8_t map_bits[N_MAPS0][2] = {{8, 8}, {8, 8}, {8, 8}, {2, 8}, {0, 8}, {15, 1}, {15, 1}, {15, 1}, {15, 1}, {15, 1},
                                                {17, 1}, {17, 1}, {17, 1}, {17, 1}, {13, 1}, {13, 1}, {13, 1}, {13, 1}};   /* :5017-5019 */
  static const double lambda[N_OLS] = {0.98, 0.87, 0.9, 0.8, 0.9, 0.7};
  static const int num[N_OLS] = {32, 12, 15, 10, 14, 8};
  Im24* m = (Im24*)calloc(1, sizeof *m);
  m->cm = p8f_cm_new((0x10000ull << level) * 4, N_CM);
  for (int i = 0; i < N_SC; i++) m->sc[i] = i < N_SC - 1 ? p8f_dmap_new(0, 11, 1, 0) : p8f_dmap_new(0, 0, 8, 0);   /* construction order = the reference's: SCMap[] before Map[] */
  for (int i = 0; i < N_MAPS; i++) m->map[i] = i < N_MAPS0 ? p8f_dmap_new(1, map_bits[i][0], map_bits[i][1], 0) : p8f_dmap_new(1, 11, 1, 0);
  for (int j = 0; j < N_OLS; j++) for (int c = 0; c < 4; c++) ols_init(&m->ols[j][c], num[j], lambda[j]);
  m->color = -1; m->stride = 3; m->columns[0] = m->columns[1] = 1;
  return m;
}

/* One step. hist / bmask / pos: the byte history (buf(i) = hist[(pos - i) & bmask]); w: bytes per row; alpha: 1 for 32-bit pixels.
 * out: the step's inputs (returns their number, or -1: a row with padding, see the file comment). sets[13] / ranges[13]: the weight-set
 * selectors in m.set() order. stats[8] (at bpos == 0): W, N, NN, WW, Wp1, Np1, plane, ctx >> 3 (ModelStats.Image :5287-5296). */
int p8f_im24_step(Im24* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int w, int alpha, int16_t* out, int* sets, int* ranges,
                  uint32_t* stats) {
#define BUF(i) ((int)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
  if (bpos == 0) {
    if (m->color < 0 || pos - m->last_pos != 1) {
      m->stride = 3 + alpha;
      m->padding = w % m->stride;
      m->x = m->line = 0;
      m->columns[0] = imax(1, w / imax(1, (int)ilog2u((unsigned)w) * 3));
      m->columns[1] = imax(1, m->columns[0] / imax(1, (int)ilog2u((unsigned)m->columns[0])));
    }
    m->last_pos = pos;
    ++m->x;
    m->x *= m->x < w;
    m->line += (m->x == 0);
    if (m->x + m->padding < w) { ++m->color; m->color *= m->color < m->stride; }
    else m->color = (m->padding > 0) * (m->stride + 1);
    if (m->color > 3) return -1;
    const int s = m->stride, color = m->color;
    m->column[0] = m->x / m->columns[0];
    m->column[1] = m->x / m->columns[1];
    const int WWWWWW = BUF(6 * s), WWWWW = BUF(5 * s), WWWW = BUF(4 * s), WWW = BUF(3 * s), WW = BUF(2 * s), W = BUF(s);
    const int NWWWW = BUF(w + 4 * s), NWWW = BUF(w + 3 * s), NWW = BUF(w + 2 * s), NW = BUF(w + s), N = BUF(w), NE = BUF(w - s), NEE = BUF(w - 2 * s),
              NEEE = BUF(w - 3 * s), NEEEE = BUF(w - 4 * s);
    const int NNWWW = BUF(w * 2 + s * 3), NNWW = BUF((w + s) * 2), NNW = BUF(w * 2 + s), NN = BUF(w * 2), NNE = BUF(w * 2 - s), NNEE = BUF((w - s) * 2),
              NNEEE = BUF(w * 2 - s * 3);
    const int NNNWW = BUF(w * 3 + s * 2), NNNW = BUF(w * 3 + s), NNN = BUF(w * 3), NNNE = BUF(w * 3 - s), NNNEE = BUF(w * 3 - s * 2);
    const int NNNNW = BUF(w * 4 + s), NNNN = BUF(w * 4), NNNNE = BUF(w * 4 - s), NNNNN = BUF(w * 5), NNNNNN = BUF(w * 6);
    const int WWp1 = BUF(s * 2 + 1), Wp1 = BUF(s + 1), p1 = BUF(1), NWp1 = BUF(w + s + 1), Np1 = BUF(w + 1), NEp1 = BUF(w - s + 1), NNp1 = BUF(w * 2 + 1);
    const int WWp2 = BUF(s * 2 + 2), Wp2 = BUF(s + 2), p2 = BUF(2), NWp2 = BUF(w + s + 2), Np2 = BUF(w + 2), NEp2 = BUF(w - s + 2), NNp2 = BUF(w * 2 + 2);
    /* shorthands for bytes the predictors name only by position */
    const int NEEp1 = BUF(w - s * 2 + 1), NEEp2 = BUF(w - s * 2 + 2), NNEp1 = BUF(w * 2 - s + 1), NNEp2 = BUF(w * 2 - s + 2);
    const int NNEEp1 = BUF(w * 2 - s * 2 + 1), NNEEp2 = BUF(w * 2 - s * 2 + 2), NNWp1 = BUF(w * 2 + s + 1), NNWp2 = BUF(w * 2 + s + 2);
    const int NNNp1 = BUF(w * 3 + 1), NNNp2 = BUF(w * 3 + 2), WWWp1 = BUF(s * 3 + 1), WWWp2 = BUF(s * 3 + 2);
    const int NNNEEE = BUF(w * 3 - 3 * s);
    uint8_t* q = m->mctx;
    int j = 0;
    q[j++] = clamp4(N + p1 - Np1, W, NW, N, NE);
    q[j++] = clamp4(N + p2 - Np2, W, NW, N, NE);
    q[j++] = (uint8_t)((W + clamp4(NE * 3 - NNE * 3 + NNNE, W, N, NE, NEE)) / 2);
    q[j++] = clamp4((W + clip(NE * 2 - NNE)) / 2, W, NW, N, NE);
    q[j++] = (uint8_t)((W + NEE) / 2);
    q[j++] = clip((WWW - 4 * WW + 6 * W + clip(NE * 4 - NNE * 6 + NNNE * 4 - NNNNE)) / 4);
    q[j++] = clip((-WWWW + 5 * WWW - 10 * WW + 10 * W + clamp4(NE * 4 - NNE * 6 + NNNE * 4 - NNNNE, N, NE, NEE, NEEE)) / 5);
    q[j++] = clip((-4 * WW + 15 * W + 10 * clip(NE * 3 - NNE * 3 + NNNE) - clip(NEEE * 3 - NNEEE * 3 + NNNEEE)) / 20);
    q[j++] = clip((-3 * WW + 8 * W + clamp4(NEE * 3 - NNEE * 3 + NNNEE, NE, NEE, NEEE, NEEEE)) / 6);
    q[j++] = clip((W + clip(NE * 2 - NNE)) / 2 + p1 - (Wp1 + clip(NEp1 * 2 - NNEp1)) / 2);
    q[j++] = clip((W + clip(NE * 2 - NNE)) / 2 + p2 - (Wp2 + clip(NEp2 * 2 - NNEp2)) / 2);
    q[j++] = clip((-3 * WW + 8 * W + clip(NEE * 2 - NNEE)) / 6 + p1 - (-3 * WWp1 + 8 * Wp1 + clip(NEEp1 * 2 - NNEEp1)) / 6);
    q[j++] = clip((-3 * WW + 8 * W + clip(NEE * 2 - NNEE)) / 6 + p2 - (-3 * WWp2 + 8 * Wp2 + clip(NEEp2 * 2 - NNEEp2)) / 6);
    q[j++] = clip((W + NEE) / 2 + p1 - (Wp1 + NEEp1) / 2);
    q[j++] = clip((W + NEE) / 2 + p2 - (Wp2 + NEEp2) / 2);
    q[j++] = clip((WW + clip(NEE * 2 - NNEE)) / 2 + p1 - (WWp1 + clip(NEEp1 * 2 - NNEEp1)) / 2);
    q[j++] = clip((WW + clip(NEE * 2 - NNEE)) / 2 + p2 - (WWp2 + clip(NEEp2 * 2 - NNEEp2)) / 2);
    q[j++] = clip(WW + NEE - N + p1 - clip(WWp1 + NEEp1 - Np1));
    q[j++] = clip(WW + NEE - N + p2 - clip(WWp2 + NEEp2 - Np2));
    q[j++] = clip(W + N - NW);
    q[j++] = clip(W + N - NW + p1 - clip(Wp1 + Np1 - NWp1));
    q[j++] = clip(W + N - NW + p2 - clip(Wp2 + Np2 - NWp2));
    q[j++] = clip(W + NE - N);
    q[j++] = clip(N + NW - NNW);
    q[j++] = clip(N + NW - NNW + p1 - clip(Np1 + NWp1 - NNWp1));
    q[j++] = clip(N + NW - NNW + p2 - clip(Np2 + NWp2 - NNWp2));
    q[j++] = clip(N + NE - NNE);
    q[j++] = clip(N + NE - NNE + p1 - clip(Np1 + NEp1 - NNEp1));
    q[j++] = clip(N + NE - NNE + p2 - clip(Np2 + NEp2 - NNEp2));
    q[j++] = clip(N + NN - NNN);
    q[j++] = clip(N + NN - NNN + p1 - clip(Np1 + NNp1 - NNNp1));
    q[j++] = clip(N + NN - NNN + p2 - clip(Np2 + NNp2 - NNNp2));
    q[j++] = clip(W + WW - WWW);
    q[j++] = clip(W + WW - WWW + p1 - clip(Wp1 + WWp1 - WWWp1));
    q[j++] = clip(W + WW - WWW + p2 - clip(Wp2 + WWp2 - WWWp2));
    q[j++] = clip(W + NEE - NE);
    q[j++] = clip(W + NEE - NE + p1 - clip(Wp1 + NEEp1 - NEp1));
    q[j++] = clip(W + NEE - NE + p2 - clip(Wp2 + NEEp2 - NEp2));
    q[j++] = clip(NN + p1 - NNp1);
    q[j++] = clip(NN + p2 - NNp2);
    q[j++] = clip(NN + W - NNW);
    q[j++] = clip(NN + W - NNW + p1 - clip(NNp1 + Wp1 - NNWp1));
    q[j++] = clip(NN + W - NNW + p2 - clip(NNp2 + Wp2 - NNWp2));
    q[j++] = clip(NN + NW - NNNW);
    q[j++] = clip(NN + NW - NNNW + p1 - clip(NNp1 + NWp1 - BUF(w * 3 + s + 1)));
    q[j++] = clip(NN + NW - NNNW + p2 - clip(NNp2 + NWp2 - BUF(w * 3 + s + 2)));
    q[j++] = clip(NN + NE - NNNE);
    q[j++] = clip(NN + NE - NNNE + p1 - clip(NNp1 + NEp1 - BUF(w * 3 - s + 1)));
    q[j++] = clip(NN + NE - NNNE + p2 - clip(NNp2 + NEp2 - BUF(w * 3 - s + 2)));
    q[j++] = clip(NN + NNNN - NNNNNN);
    q[j++] = clip(NN + NNNN - NNNNNN + p1 - clip(NNp1 + BUF(w * 4 + 1) - BUF(w * 6 + 1)));
    q[j++] = clip(NN + NNNN - NNNNNN + p2 - clip(NNp2 + BUF(w * 4 + 2) - BUF(w * 6 + 2)));
    q[j++] = clip(WW + p1 - WWp1);
    q[j++] = clip(WW + p2 - WWp2);
    q[j++] = clip(WW + WWWW - WWWWWW);
    q[j++] = clip(WW + WWWW - WWWWWW + p1 - clip(WWp1 + BUF(s * 4 + 1) - BUF(s * 6 + 1)));
    q[j++] = clip(WW + WWWW - WWWWWW + p2 - clip(WWp2 + BUF(s * 4 + 2) - BUF(s * 6 + 2)));
    q[j++] = clip(N * 2 - NN + p1 - clip(Np1 * 2 - NNp1));
    q[j++] = clip(N * 2 - NN + p2 - clip(Np2 * 2 - NNp2));
    q[j++] = clip(W * 2 - WW + p1 - clip(Wp1 * 2 - WWp1));
    q[j++] = clip(W * 2 - WW + p2 - clip(Wp2 * 2 - WWp2));
    q[j++] = clip(N * 3 - NN * 3 + NNN);
    q[j++] = clamp4(N * 3 - NN * 3 + NNN, W, NW, N, NE);
    q[j++] = clamp4(W * 3 - WW * 3 + WWW, W, NW, N, NE);
    q[j++] = clamp4(N * 2 - NN, W, NW, N, NE);
    q[j++] = clip((NNNNN - 6 * NNNN + 15 * NNN - 20 * NN + 15 * N + clamp4(W * 4 - NWW * 6 + NNWWW * 4 - BUF(w * 3 + 4 * s), W, NW, N, NN)) / 6);
    q[j++] = clip((NNNEEE - 4 * NNEE + 6 * NE + clip(W * 4 - NW * 6 + NNW * 4 - NNNW)) / 4);
    q[j++] = clip(((N + 3 * NW) / 4) * 3 - ((NNW + NNWW) / 2) * 3 + (NNNWW * 3 + BUF(w * 3 + 3 * s)) / 4);
    q[j++] = clip((W * 2 + NW) - (WW + 2 * NWW) + NWWW);
    q[j++] = (uint8_t)((clip(W * 2 - NW) + clip(W * 2 - NWW) + N + NE) / 4);
    q[j++] = (uint8_t)NNNNNN;
    q[j++] = (uint8_t)((NEEEE + BUF(w - 6 * s)) / 2);
    q[j++] = (uint8_t)((WWWWWW + WWWW) / 2);
    q[j++] = (uint8_t)(((W + N) * 3 - NW * 2) / 4);
    q[j++] = (uint8_t)N;
    q[j++] = (uint8_t)NN;
    q = m->sctx;
    j = 0;
    q[j++] = (uint8_t)(N + p1 - Np1);
    q[j++] = (uint8_t)(N + p2 - Np2);
    q[j++] = (uint8_t)(W + p1 - Wp1);
    q[j++] = (uint8_t)(W + p2 - Wp2);
    q[j++] = (uint8_t)(NW + p1 - NWp1);
    q[j++] = (uint8_t)(NW + p2 - NWp2);
    q[j++] = (uint8_t)(NE + p1 - NEp1);
    q[j++] = (uint8_t)(NE + p2 - NEp2);
    q[j++] = (uint8_t)(NN + p1 - NNp1);
    q[j++] = (uint8_t)(NN + p2 - NNp2);
    q[j++] = (uint8_t)(WW + p1 - WWp1);
    q[j++] = (uint8_t)(WW + p2 - WWp2);
    q[j++] = (uint8_t)(W + N - NW);
    q[j++] = (uint8_t)(W + N - NW + p1 - Wp1 - Np1 + NWp1);
    q[j++] = (uint8_t)(W + N - NW + p2 - Wp2 - Np2 + NWp2);
    q[j++] = (uint8_t)(W + NE - N);
    q[j++] = (uint8_t)(W + NE - N + p1 - Wp1 - NEp1 + Np1);
    q[j++] = (uint8_t)(W + NE - N + p2 - Wp2 - NEp2 + Np2);
    q[j++] = (uint8_t)(W + NEE - NE);
    q[j++] = (uint8_t)(W + NEE - NE + p1 - Wp1 - NEEp1 + NEp1);
    q[j++] = (uint8_t)(W + NEE - NE + p2 - Wp2 - NEEp2 + NEp2);
    q[j++] = (uint8_t)(N + NN - NNN);
    q[j++] = (uint8_t)(N + NN - NNN + p1 - Np1 - NNp1 + NNNp1);
    q[j++] = (uint8_t)(N + NN - NNN + p2 - Np2 - NNp2 + NNNp2);
    q[j++] = (uint8_t)(N + NE - NNE);
    q[j++] = (uint8_t)(N + NE - NNE + p1 - Np1 - NEp1 + NNEp1);
    q[j++] = (uint8_t)(N + NE - NNE + p2 - Np2 - NEp2 + NNEp2);
    q[j++] = (uint8_t)(N + NW - NNW);
    q[j++] = (uint8_t)(N + NW - NNW + p1 - Np1 - NWp1 + NNWp1);
    q[j++] = (uint8_t)(N + NW - NNW + p2 - Np2 - NWp2 + NNWp2);
    q[j++] = (uint8_t)(NE + NW - NN);
    q[j++] = (uint8_t)(NE + NW - NN + p1 - NEp1 - NWp1 + NNp1);
    q[j++] = (uint8_t)(NE + NW - NN + p2 - NEp2 - NWp2 + NNp2);
    q[j++] = (uint8_t)(NW + W - NWW);
    q[j++] = (uint8_t)(NW + W - NWW + p1 - NWp1 - Wp1 + BUF(w + s * 2 + 1));
    q[j++] = (uint8_t)(NW + W - NWW + p2 - NWp2 - Wp2 + BUF(w + s * 2 + 2));
    q[j++] = (uint8_t)(W * 2 - WW);
    q[j++] = (uint8_t)(W * 2 - WW + p1 - Wp1 * 2 + WWp1);
    q[j++] = (uint8_t)(W * 2 - WW + p2 - Wp2 * 2 + WWp2);
    q[j++] = (uint8_t)(N * 2 - NN);
    q[j++] = (uint8_t)(N * 2 - NN + p1 - Np1 * 2 + NNp1);
    q[j++] = (uint8_t)(N * 2 - NN + p2 - Np2 * 2 + NNp2);
    q[j++] = (uint8_t)(NW * 2 - NNWW);
    q[j++] = (uint8_t)(NW * 2 - NNWW + p1 - NWp1 * 2 + BUF(w * 2 + s * 2 + 1));
    q[j++] = (uint8_t)(NW * 2 - NNWW + p2 - NWp2 * 2 + BUF(w * 2 + s * 2 + 2));
    q[j++] = (uint8_t)(NE * 2 - NNEE);
    q[j++] = (uint8_t)(NE * 2 - NNEE + p1 - NEp1 * 2 + NNEEp1);
    q[j++] = (uint8_t)(NE * 2 - NNEE + p2 - NEp2 * 2 + NNEEp2);
    q[j++] = (uint8_t)(N * 3 - NN * 3 + NNN + p1 - Np1 * 3 + NNp1 * 3 - NNNp1);
    q[j++] = (uint8_t)(N * 3 - NN * 3 + NNN + p2 - Np2 * 3 + NNp2 * 3 - NNNp2);
    q[j++] = (uint8_t)(N * 3 - NN * 3 + NNN);
    q[j++] = (uint8_t)((W + NE * 2 - NNE) / 2);
    q[j++] = (uint8_t)((W + NE * 3 - NNE * 3 + NNNE) / 2);
    q[j++] = (uint8_t)((W + NE * 2 - NNE) / 2 + p1 - (Wp1 + NEp1 * 2 - NNEp1) / 2);
    q[j++] = (uint8_t)((W + NE * 2 - NNE) / 2 + p2 - (Wp2 + NEp2 * 2 - NNEp2) / 2);
    q[j++] = (uint8_t)(NNE + NE - NNNE);
    q[j++] = (uint8_t)(NNE + W - NN);
    q[j++] = (uint8_t)(NNW + W - NNWW);
    /* the six least-squares predictors of this plane; the plane before it learns the byte just coded (:5223-5228) */
    {
      const uint8_t t1[32] = {(uint8_t)WWWWWW, (uint8_t)WWWWW, (uint8_t)WWWW, (uint8_t)WWW, (uint8_t)WW, (uint8_t)W, (uint8_t)NWWWW, (uint8_t)NWWW, (uint8_t)NWW, (uint8_t)NW, (uint8_t)N,
                              (uint8_t)NE, (uint8_t)NEE, (uint8_t)NEEE, (uint8_t)NEEEE, (uint8_t)NNWWW, (uint8_t)NNWW, (uint8_t)NNW, (uint8_t)NN, (uint8_t)NNE, (uint8_t)NNEE, (uint8_t)NNEEE,
                              (uint8_t)NNNWW, (uint8_t)NNNW, (uint8_t)NNN, (uint8_t)NNNE, (uint8_t)NNNEE, (uint8_t)NNNNW, (uint8_t)NNNN, (uint8_t)NNNNE, (uint8_t)NNNNN, (uint8_t)NNNNNN};
      const uint8_t t2[12] = {(uint8_t)WWW, (uint8_t)WW, (uint8_t)W, (uint8_t)NWW, (uint8_t)NW, (uint8_t)N, (uint8_t)NE, (uint8_t)NEE, (uint8_t)NNW, (uint8_t)NN, (uint8_t)NNE, (uint8_t)NNN};
      const uint8_t t3[15] = {(uint8_t)N, (uint8_t)NE, (uint8_t)NEE, (uint8_t)NEEE, (uint8_t)NEEEE, (uint8_t)NN, (uint8_t)NNE, (uint8_t)NNEE, (uint8_t)NNEEE, (uint8_t)NNN, (uint8_t)NNNE,
                              (uint8_t)NNNEE, (uint8_t)NNNN, (uint8_t)NNNNE, (uint8_t)NNNNN};
      const uint8_t t4[10] = {(uint8_t)N, (uint8_t)NE, (uint8_t)NEE, (uint8_t)NEEE, (uint8_t)NN, (uint8_t)NNE, (uint8_t)NNEE, (uint8_t)NNN, (uint8_t)NNNE, (uint8_t)NNNN};
      const uint8_t t5[14] = {(uint8_t)WWWW, (uint8_t)WWW, (uint8_t)WW, (uint8_t)W, (uint8_t)NWWW, (uint8_t)NWW, (uint8_t)NW, (uint8_t)N, (uint8_t)NNWW, (uint8_t)NNW, (uint8_t)NN, (uint8_t)NNNW,
                              (uint8_t)NNN, (uint8_t)NNNN};
      const uint8_t t6[8] = {(uint8_t)WWW, (uint8_t)WW, (uint8_t)W, (uint8_t)NNN, (uint8_t)NN, (uint8_t)N, (uint8_t)p1, (uint8_t)p2};
      const uint8_t* taps[N_OLS] = {t1, t2, t3, t4, t5, t6};
      const int k = color > 0 ? color - 1 : s - 1;
      for (j = 0; j < N_OLS; j++) {
        m->pols[j] = clip((int)floor(ols_predict(&m->ols[j][color], taps[j])));
        ols_update(&m->ols[j][k], (uint8_t)p1);
      }
    }
    int mean = W + NW + N + NE;
    const int var = (W * W + NW * NW + N * N + NE * NE - mean * mean / 4) >> 2;
    mean >>= 2;
    const int logvar = p8f_ilog(var);
    const int plane = imin(color, s - 1);
    m->ctx[0] = (plane << 9) | ((abs(W - N) > 3) << 8) | ((W > N) << 7) | ((W > NW) << 6) | ((abs(N - NW) > 3) << 5) | ((N > NW) << 4) | ((abs(N - NE) > 3) << 3) |
                ((N > NE) << 2) | ((W > WW) << 1) | (N > NN);
    m->ctx[1] = ((LMD(p1, clip(Np1 + NEp1 - NNEp1)) >> 1) << 5) | ((LMD(clip(N + NE - NNE), clip(N + NW - NNW)) >> 1) << 2) | plane;
    uint64_t cx[N_CM];
    int n = 0;
    int64_t i = 0;
    cx[n++] = H3(++i, (N + 1) >> 1, LMD(N, clip(NN * 2 - NNN)));
    cx[n++] = H3(++i, (W + 1) >> 1, LMD(W, clip(WW * 2 - WWW)));
    cx[n++] = H3(++i, clamp4(W + N - NW, W, NW, N, NE), LMD(clip(N + NE - NNE), clip(N + NW - NNW)));
    cx[n++] = H3(++i, (NNN + N + 4) / 8, clip(N * 3 - NN * 3 + NNN) >> 1);
    cx[n++] = H3(++i, (WWW + W + 4) / 8, clip(W * 3 - WW * 3 + WWW) >> 1);
    cx[n++] = H4(++i, color, (W + clip(NE * 3 - NNE * 3 + NNNE)) / 4, LMD(N, (NW + NE) / 2));
    cx[n++] = H3(++i, color, clip((-WWWW + 5 * WWW - 10 * WW + 10 * W + clamp4(NE * 4 - NNE * 6 + NNNE * 4 - NNNNE, N, NE, NEE, NEEE)) / 5) / 4);
    cx[n++] = H3(++i, clip(NEE + N - NNEE), LMD(W, clip(NW + NE - NNE)));
    cx[n++] = H3(++i, clip(NN + W - NNW), LMD(W, clip(NNW + WW - NNWW)));
    cx[n++] = H3(++i, color, p1);
    cx[n++] = H3(++i, color, p2);
    cx[n++] = H4(++i, color, clip(W + N - NW) / 2, clip(W + p1 - Wp1) / 2);
    cx[n++] = H3(++i, clip(N * 2 - NN) / 2, LMD(N, clip(NN * 2 - NNN)));
    cx[n++] = H3(++i, clip(W * 2 - WW) / 2, LMD(W, clip(WW * 2 - WWW)));
    cx[n++] = H2(++i, clamp4(N * 3 - NN * 3 + NNN, W, NW, N, NE) / 2);
    cx[n++] = H2(++i, clamp4(W * 3 - WW * 3 + WWW, W, N, NE, NEE) / 2);
    cx[n++] = H4(++i, color, LMD(W, Wp1), clamp4((p1 * W) / (Wp1 < 1 ? 1 : Wp1), W, N, NE, NEE));
    cx[n++] = H3(++i, color, clamp4(N + p2 - Np2, W, NW, N, NE));
    cx[n++] = H4(++i, color, clip(W + N - NW), m->column[0]);
    cx[n++] = H4(++i, color, clip(N * 2 - NN), LMD(W, clip(NW * 2 - NNW)));
    cx[n++] = H4(++i, color, clip(W * 2 - WW), LMD(N, clip(NW * 2 - NWW)));
    cx[n++] = H3(++i, (W + NEE) / 2, LMD(W, (WW + NE) / 2));
    cx[n++] = H2(++i, clamp4(clip(W * 2 - WW) + clip(N * 2 - NN) - clip(NW * 2 - NNWW), W, NW, N, NE));
    cx[n++] = H4(++i, color, W, p2);
    cx[n++] = H4(++i, N, NN & 0x1F, NNN & 0x1F);
    cx[n++] = H4(++i, W, WW & 0x1F, WWW & 0x1F);
    cx[n++] = H4(++i, color, N, m->column[0]);
    cx[n++] = H4(++i, color, clip(W + NEE - NE), LMD(W, clip(WW + NE - N)));
    cx[n++] = H5(++i, NN, NNNN & 0x1F, NNNNNN & 0x1F, m->column[1]);
    cx[n++] = H5(++i, WW, WWWW & 0x1F, WWWWWW & 0x1F, m->column[1]);
    cx[n++] = H5(++i, NNN, NNNNNN & 0x1F, BUF(w * 9) & 0x1F, m->column[1]);
    cx[n++] = H3(++i, color, m->column[1]);
    cx[n++] = H4(++i, color, W, LMD(W, WW));
    cx[n++] = H4(++i, color, W, p1);
    cx[n++] = H5(++i, color, W / 4, LMD(W, p1), LMD(W, p2));
    cx[n++] = H4(++i, color, N, LMD(N, NN));
    cx[n++] = H4(++i, color, N, p1);
    cx[n++] = H5(++i, color, N / 4, LMD(N, p1), LMD(N, p2));
    cx[n++] = H5(++i, color, (W + N) >> 3, p1 >> 4, p2 >> 4);
    cx[n++] = H4(++i, color, p1 / 2, p2 / 2);
    cx[n++] = H4(++i, color, W, p1 - Wp1);
    cx[n++] = H3(++i, color, W + p1 - Wp1);
    cx[n++] = H4(++i, color, N, p1 - Np1);
    cx[n++] = H3(++i, color, N + p1 - Np1);
    cx[n++] = H3(++i, BUF(w * 3 - s), BUF(w * 3 - 2 * s));
    cx[n++] = H3(++i, BUF(w * 3 + s), BUF(w * 3 + 2 * s));
    cx[n++] = H4(++i, color, mean, logvar >> 4);
    int k = 0;
    p8f_cm_step(m->cm, y, 0, c0, p1, cx, n, out, &k);   /* the 47 contexts of the byte; the family's five inputs each come back below */
    (void)k;
    p8f_dmap_set_direct(m->map[0], (uint32_t)((W & 0xC0) | ((N & 0xC0) >> 2) | ((WW & 0xC0) >> 4) | (NN >> 6)));
    p8f_dmap_set_direct(m->map[1], (uint32_t)((N & 0xC0) | ((NN & 0xC0) >> 2) | ((NE & 0xC0) >> 4) | (NEE >> 6)));
    p8f_dmap_set_direct(m->map[2], (uint32_t)p1);
    p8f_dmap_set_direct(m->map[3], (uint32_t)plane);
    m->W = (uint8_t)W; m->WW = (uint8_t)WW; m->N = (uint8_t)N; m->NN = (uint8_t)NN; m->NW = (uint8_t)NW; m->NE = (uint8_t)NE; m->NNE = (uint8_t)NNE; m->NNW = (uint8_t)NNW;
    m->NWW = (uint8_t)NWW; m->p1 = (uint8_t)p1; m->p2 = (uint8_t)p2; m->Wp1 = (uint8_t)Wp1; m->Wp2 = (uint8_t)Wp2; m->Np1 = (uint8_t)Np1; m->Np2 = (uint8_t)Np2;
    m->NWp1 = (uint8_t)NWp1; m->NWp2 = (uint8_t)NWp2;
    if (stats) {
      stats[0] = (uint32_t)W; stats[1] = (uint32_t)N; stats[2] = (uint32_t)NN; stats[3] = (uint32_t)WW; stats[4] = (uint32_t)Wp1; stats[5] = (uint32_t)Np1;
      stats[6] = (uint32_t)plane; stats[7] = (uint32_t)(m->ctx[0] >> 3);
    }
  }
  const int s = m->stride, plane = imin(m->color, s - 1);
  const int W = m->W, WW = m->WW, N = m->N, NN = m->NN, NW = m->NW, NE = m->NE, NNE = m->NNE, NNW = m->NNW, NWW = m->NWW, p1 = m->p1, p2 = m->p2;
  const int Wp1 = m->Wp1, Wp2 = m->Wp2, Np1 = m->Np1, Np2 = m->Np2, NWp1 = m->NWp1, NWp2 = m->NWp2;
  const int B = (uint8_t)(c0 << (8 - bpos));
  int i = 5;
  p8f_dmap_set_direct(m->map[i++], (uint32_t)((((uint8_t)(clip(W + N - NW) - B)) * 8 + bpos) | (LMD(clip(N + NE - NNE), clip(N + NW - NNW)) << 11)));
  p8f_dmap_set_direct(m->map[i++], (uint32_t)((((uint8_t)(clip(N * 2 - NN) - B)) * 8 + bpos) | (LMD(W, clip(NW * 2 - NNW)) << 11)));
  p8f_dmap_set_direct(m->map[i++], (uint32_t)((((uint8_t)(clip(W * 2 - WW) - B)) * 8 + bpos) | (LMD(N, clip(NW * 2 - NWW)) << 11)));
  p8f_dmap_set_direct(m->map[i++], (uint32_t)((((uint8_t)(clip(W + N - NW) - B)) * 8 + bpos) | (LMD(p1, clip(Wp1 + Np1 - NWp1)) << 11)));
  p8f_dmap_set_direct(m->map[i++], (uint32_t)((((uint8_t)(clip(W + N - NW) - B)) * 8 + bpos) | (LMD(p2, clip(Wp2 + Np2 - NWp2)) << 11)));
  p8f_dmap_set(m->map[i++], H2(W - B, N - B) * 8 + (uint64_t)bpos);
  p8f_dmap_set(m->map[i++], H2(W - B, WW - B) * 8 + (uint64_t)bpos);
  p8f_dmap_set(m->map[i++], H2(N - B, NN - B) * 8 + (uint64_t)bpos);
  p8f_dmap_set(m->map[i++], H2(clip(N + NE - NNE) - B, clip(N + NW - NNW) - B) * 8 + (uint64_t)bpos);
  p8f_dmap_set_direct(m->map[i++], (uint32_t)((plane << 11) | (((uint8_t)(clip(N + p1 - Np1) - B)) * 8 + bpos)));
  p8f_dmap_set_direct(m->map[i++], (uint32_t)((plane << 11) | (((uint8_t)(clip(N + p2 - Np2) - B)) * 8 + bpos)));
  p8f_dmap_set_direct(m->map[i++], (uint32_t)((plane << 11) | (((uint8_t)(clip(W + p1 - Wp1) - B)) * 8 + bpos)));
  p8f_dmap_set_direct(m->map[i++], (uint32_t)((plane << 11) | (((uint8_t)(clip(W + p2 - Wp2) - B)) * 8 + bpos)));
  for (int j = 0; j < N_MAPS1; i++, j++) p8f_dmap_set_direct(m->map[i], (uint32_t)((m->mctx[j] - B) * 8 + bpos));
  for (int j = 0; i < N_MAPS; i++, j++) p8f_dmap_set_direct(m->map[i], (uint32_t)((m->pols[j] - B) * 8 + bpos));
  for (int j = 0; j < N_SC - 1; j++) p8f_dmap_set_direct(m->sc[j], (uint32_t)((m->sctx[j] - B) * 8 + bpos));   /* SmallStationaryContextMap::set masks like set_direct */
  /* the maps' calls, in the reference's order: cm.mix, Map[].mix(m, 1, 3), SCMap[].mix(m, 9, 1, 3) */
  int nx = 0, k = 0;
  if (bpos) p8f_cm_step(m->cm, y, bpos, c0, p1, NULL, 0, out, &k);
  else k = 5 * N_CM;
  nx += k;
  for (int j = 0; j < N_MAPS; j++) nx += p8f_dmap_mix(m->map[j], y, 1023, 1, 3, out + nx);
  for (int j = 0; j < N_SC; j++) nx += p8f_dmap_mix(m->sc[j], y, 9, 1, 3, out + nx);
  if (++m->col >= s * 8) m->col = 0;
  const int x = m->x, line = m->line;
  int ns = 0;
#define SET(v, r) do { sets[ns] = (int)(v); ranges[ns] = (int)(r); ++ns; } while (0)
  SET(0, 1);
  SET(imin(63, m->column[0]) + ((m->ctx[0] >> 3) & 0xC0), 256);
  SET(imin(127, m->column[1]) + ((m->ctx[0] >> 2) & 0x180), 512);
  SET((m->ctx[0] & 0x7FC) | (bpos >> 1), 2048);
  SET(m->col, s * 8);
  SET(x % s, s);
  SET(c0, 256);
  SET((m->ctx[1] << 2) | (bpos >> 1), 1024);
  SET(p8f_finalize64(H5(lmdq((uint8_t)W, (uint8_t)WW, 5), lmdq((uint8_t)N, (uint8_t)NN, 5), lmdq((uint8_t)W, (uint8_t)N, 5), ilog2u((unsigned)W), m->color), 13), 8192);
  SET(p8f_finalize64(H2(m->ctx[0], m->column[0] / 8), 13), 8192);
  SET(p8f_finalize64(H3(logqt((uint8_t)N, 5), lmdq((uint8_t)N, (uint8_t)NN, 3), c0), 13), 8192);
  SET(p8f_finalize64(H3(logqt((uint8_t)W, 5), lmdq((uint8_t)W, (uint8_t)WW, 3), c0), 13), 8192);
  SET(imin(255, (x + line) / 32), 256);
#undef SET
#undef BUF
  return nx;
}

/* ---------------------------------------------------------------- im8bitModel :4743-4999
 * One model, two faces: grayscale images (25 of the ContextMap's 52 contexts, 62 StationaryMaps on neighbourhood predictors and five
 * least-squares predictors) and palette images (all 52 contexts, four of them through IndirectContext<U8> byte histories, and four
 * SmallStationaryContextMaps). The tables are shared (the reference's function statics), the inputs a step produces differ. */
enum { G_MAPS0 = 2, G_MAPS1 = 55, G_OLS = 5, G_MAPS = G_MAPS0 + G_MAPS1 + G_OLS, G_PLT = 4, G_CM = 48 + G_PLT };
typedef struct Im8 {
  CM1* cm;
  DMap* map[G_MAPS];
  DMap* plt[G_PLT];
  Ols ols[G_OLS];
  uint8_t ictx_data[G_PLT][65536];   /* IndirectContext<U8>(16, 8) :1468-1492: the byte that followed a 16-bit context last time */
  uint32_t ictx_at[G_PLT];
  int ctx, last_pos, col, x, line;
  int columns[2], column[2];
  uint8_t mctx[G_MAPS1], pols[G_OLS];
  uint8_t W, WW, N, NN, NW, NE, NNE, NNW, NNWW, NNEE;
} Im8;

Im8* p8f_im8_new(int level) {
  static const double lambda[G_OLS] = {0.996, 0.87, 0.93, 0.8, 0.9};
  static const int num[G_OLS] = {32, 12, 15, 10, 14};
  Im8* m = (Im8*)calloc(1, sizeof *m);
  m->cm = p8f_cm_new((0x10000ull << level) * 4, G_CM);
  for (int i = 0; i < G_MAPS; i++) m->map[i] = i == 0 ? p8f_dmap_new(1, 0, 8, 0) : i == 1 ? p8f_dmap_new(1, 15, 1, 0) : p8f_dmap_new(1, 11, 1, 0);
  for (int i = 0; i < G_PLT; i++) m->plt[i] = p8f_dmap_new(0, 11, 1, 0);
  for (int j = 0; j < G_OLS; j++) ols_init(&m->ols[j], num[j], lambda[j]);
  m->columns[0] = m->columns[1] = 1;
  return m;
}

/* One step; gray: 0 = palette image, else grayscale (the value the caller holds -- imgModel's may be 0x100 | byte, which only matters
 * for the shift in stats[7], :4973). sets[8] / ranges[8]; stats[8] (at bpos == 0): W, N, NN, WW, -, -, -, ctx >> gray. */
int p8f_im8_step(Im8* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int w, int gray, int16_t* out, int* sets, int* ranges, uint32_t* stats) {
#define BUF(i) ((int)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
  if (bpos == 0) {
    if (pos != m->last_pos + 1) {
      m->x = m->line = 0;
      m->columns[0] = imax(1, w / imax(1, (int)ilog2u((unsigned)w) * 2));
      m->columns[1] = imax(1, m->columns[0] / imax(1, (int)ilog2u((unsigned)m->columns[0])));
    } else {
      ++m->x;
      m->x *= m->x < w;
      m->line += (m->x == 0);
    }
    m->last_pos = pos;
    m->column[0] = m->x / m->columns[0];
    m->column[1] = m->x / m->columns[1];
    const int WWWWW = BUF(5), WWWW = BUF(4), WWW = BUF(3), WW = BUF(2), W = BUF(1);
    const int NWWWW = BUF(w + 4), NWWW = BUF(w + 3), NWW = BUF(w + 2), NW = BUF(w + 1), N = BUF(w), NE = BUF(w - 1), NEE = BUF(w - 2), NEEE = BUF(w - 3), NEEEE = BUF(w - 4);
    const int NNWWW = BUF(w * 2 + 3), NNWW = BUF(w * 2 + 2), NNW = BUF(w * 2 + 1), NN = BUF(w * 2), NNE = BUF(w * 2 - 1), NNEE = BUF(w * 2 - 2), NNEEE = BUF(w * 2 - 3);
    const int NNNWW = BUF(w * 3 + 2), NNNW = BUF(w * 3 + 1), NNN = BUF(w * 3), NNNE = BUF(w * 3 - 1), NNNEE = BUF(w * 3 - 2);
    const int NNNNW = BUF(w * 4 + 1), NNNN = BUF(w * 4), NNNNE = BUF(w * 4 - 1), NNNNN = BUF(w * 5), NNNNNN = BUF(w * 6);
    const int WWWWWW = BUF(6);
    uint8_t* q = m->mctx;
    int j = 0;
    q[j++] = clamp4(W + N - NW, W, NW, N, NE);
    q[j++] = clip(W + N - NW);
    q[j++] = clamp4(W + NE - N, W, NW, N, NE);
    q[j++] = clip(W + NE - N);
    q[j++] = clamp4(N + NW - NNW, W, NW, N, NE);
    q[j++] = clip(N + NW - NNW);
    q[j++] = clamp4(N + NE - NNE, W, N, NE, NEE);
    q[j++] = clip(N + NE - NNE);
    q[j++] = (uint8_t)((W + NEE) / 2);
    q[j++] = clip(N * 3 - NN * 3 + NNN);
    q[j++] = clip(W * 3 - WW * 3 + WWW);
    q[j++] = (uint8_t)((W + clip(NE * 3 - NNE * 3 + NNNE)) / 2);
    q[j++] = (uint8_t)((W + clip(NEE * 3 - NNEEE * 3 + BUF(w * 3 - 4))) / 2);
    q[j++] = clip(NN + NNNN - NNNNNN);
    q[j++] = clip(WW + WWWW - WWWWWW);
    q[j++] = clip((NNNNN - 6 * NNNN + 15 * NNN - 20 * NN + 15 * N + clamp4(W * 2 - NWW, W, NW, N, NN)) / 6);
    q[j++] = clip((-3 * WW + 8 * W + clamp4(NEE * 3 - NNEE * 3 + NNNEE, NE, NEE, NEEE, NEEEE)) / 6);
    q[j++] = clip(NN + NW - NNNW);
    q[j++] = clip(NN + NE - NNNE);
    q[j++] = clip((W * 2 + NW) - (WW + 2 * NWW) + NWWW);
    q[j++] = clip(((NW + NWW) / 2) * 3 - NNWWW * 3 + (BUF(w * 3 + 4) + BUF(w * 3 + 5)) / 2);
    q[j++] = clip(NEE + NE - NNEEE);
    q[j++] = clip(NWW + WW - NWWWW);
    q[j++] = clip(((W + NW) * 3 - NWW * 6 + NWWW + NNWWW) / 2);
    q[j++] = clip((NE * 2 + NNE) - (NNEE + NNNEE * 2) + BUF(w * 4 - 3));
    q[j++] = (uint8_t)NNNNNN;
    q[j++] = (uint8_t)((NEEEE + BUF(w - 6)) / 2);
    q[j++] = (uint8_t)((WWWW + WWWWWW) / 2);
    q[j++] = (uint8_t)((W + N + BUF(w - 5) + BUF(w - 7)) / 4);
    q[j++] = clip(NEEE + W - NEE);
    q[j++] = clip(4 * NNN - 3 * NNNN);
    q[j++] = clip(N + NN - NNN);
    q[j++] = clip(W + WW - WWW);
    q[j++] = clip(W + NEE - NE);
    q[j++] = clip(WW + NEE - N);
    q[j++] = (uint8_t)((clip(W * 2 - NW) + clip(W * 2 - NWW) + N + NE) / 4);
    q[j++] = clamp4(N * 2 - NN, W, N, NE, NEE);
    q[j++] = (uint8_t)((N + NNN) / 2);
    q[j++] = clip(NN + W - NNW);
    q[j++] = clip(NWW + N - NNWW);
    q[j++] = clip((4 * WWW - 15 * WW + 20 * W + clip(NEE * 2 - NNEE)) / 10);
    q[j++] = clip((BUF(w * 3 - 3) - 4 * NNEE + 6 * NE + clip(W * 3 - NW * 3 + NNW)) / 4);
    q[j++] = clip((N * 2 + NE) - (NN + 2 * NNE) + NNNE);
    q[j++] = clip((NW * 2 + NNW) - (NNWW + NNNWW * 2) + BUF(w * 4 + 3));
    q[j++] = clip(NNWW + W - NNWWW);
    q[j++] = clip((-NNNN + 5 * NNN - 10 * NN + 10 * N + clip(W * 4 - NWW * 6 + NNWWW * 4 - BUF(w * 3 + 4))) / 5);
    q[j++] = clip(NEE + clip(NEEE * 2 - BUF(w * 2 - 4)) - NEEEE);
    q[j++] = clip(NW + W - NWW);
    q[j++] = clip((N * 2 + NW) - (NN + 2 * NNW) + NNNW);
    q[j++] = clip(NN + clip(NEE * 2 - NNEEE) - NNE);
    q[j++] = clip((-WWWW + 5 * WWW - 10 * WW + 10 * W + clip(NE * 2 - NNE)) / 5);
    q[j++] = clip((-WWWWW + 4 * WWWW - 5 * WWW + 5 * W + clip(NE * 2 - NNE)) / 4);
    q[j++] = clip((WWW - 4 * WW + 6 * W + clip(NE * 3 - NNE * 3 + NNNE)) / 4);
    q[j++] = clip((-NNEE + 3 * NE + clip(W * 4 - NW * 6 + NNW * 4 - NNNW)) / 3);
    q[j++] = (uint8_t)(((W + N) * 3 - NW * 2) / 4);
    {
      /* (the first tap is the function's static WWWWWW, which im8bitModel never assigns -- :4784 starts at WWWWW: always 0) */
      const uint8_t t1[32] = {0, (uint8_t)WWWWW, (uint8_t)WWWW, (uint8_t)WWW, (uint8_t)WW, (uint8_t)W, (uint8_t)NWWWW, (uint8_t)NWWW, (uint8_t)NWW, (uint8_t)NW, (uint8_t)N,
                              (uint8_t)NE, (uint8_t)NEE, (uint8_t)NEEE, (uint8_t)NEEEE, (uint8_t)NNWWW, (uint8_t)NNWW, (uint8_t)NNW, (uint8_t)NN, (uint8_t)NNE, (uint8_t)NNEE, (uint8_t)NNEEE,
                              (uint8_t)NNNWW, (uint8_t)NNNW, (uint8_t)NNN, (uint8_t)NNNE, (uint8_t)NNNEE, (uint8_t)NNNNW, (uint8_t)NNNN, (uint8_t)NNNNE, (uint8_t)NNNNN, (uint8_t)NNNNNN};
      const uint8_t t2[12] = {(uint8_t)WWW, (uint8_t)WW, (uint8_t)W, (uint8_t)NWW, (uint8_t)NW, (uint8_t)N, (uint8_t)NE, (uint8_t)NEE, (uint8_t)NNW, (uint8_t)NN, (uint8_t)NNE, (uint8_t)NNN};
      const uint8_t t3[15] = {(uint8_t)N, (uint8_t)NE, (uint8_t)NEE, (uint8_t)NEEE, (uint8_t)NEEEE, (uint8_t)NN, (uint8_t)NNE, (uint8_t)NNEE, (uint8_t)NNEEE, (uint8_t)NNN, (uint8_t)NNNE,
                              (uint8_t)NNNEE, (uint8_t)NNNN, (uint8_t)NNNNE, (uint8_t)NNNNN};
      const uint8_t t4[10] = {(uint8_t)N, (uint8_t)NE, (uint8_t)NEE, (uint8_t)NEEE, (uint8_t)NN, (uint8_t)NNE, (uint8_t)NNEE, (uint8_t)NNN, (uint8_t)NNNE, (uint8_t)NNNN};
      const uint8_t t5[14] = {(uint8_t)WWWW, (uint8_t)WWW, (uint8_t)WW, (uint8_t)W, (uint8_t)NWWW, (uint8_t)NWW, (uint8_t)NW, (uint8_t)N, (uint8_t)NNWW, (uint8_t)NNW, (uint8_t)NN, (uint8_t)NNNW,
                              (uint8_t)NNN, (uint8_t)NNNN};
      const uint8_t* taps[G_OLS] = {t1, t2, t3, t4, t5};
      for (j = 0; j < G_OLS; j++) {   /* every predictor learns the byte just coded, then predicts (:4868-4871) */
        ols_update(&m->ols[j], (uint8_t)W);
        m->pols[j] = clip((int)floor(ols_predict(&m->ols[j], taps[j])));
      }
    }
    for (j = 0; j < G_PLT; j++) m->ictx_data[j][m->ictx_at[j]] = (uint8_t)W;   /* iCtx[j] += W: an 8-bit cell shifted by 8 keeps only the new byte */
    m->ictx_at[0] = (uint32_t)(W | (NE << 8)) & 0xffff;
    m->ictx_at[1] = (uint32_t)(W | (N << 8)) & 0xffff;
    m->ictx_at[2] = (uint32_t)(W | (WW << 8)) & 0xffff;
    m->ictx_at[3] = (uint32_t)(N | (NN << 8)) & 0xffff;
    uint64_t cx[G_CM];
    int n = 0;
    int64_t i = 0;
    if (!gray) {
      cx[n++] = H2(++i, W);
      cx[n++] = H3(++i, W, m->column[0]);
      cx[n++] = H2(++i, N);
      cx[n++] = H3(++i, N, m->column[0]);
      cx[n++] = H2(++i, NW);
      cx[n++] = H3(++i, NW, m->column[0]);
      cx[n++] = H2(++i, NE);
      cx[n++] = H3(++i, NE, m->column[0]);
      cx[n++] = H2(++i, NWW);
      cx[n++] = H2(++i, NEE);
      cx[n++] = H2(++i, WW);
      cx[n++] = H2(++i, NN);
      cx[n++] = H3(++i, W, N);
      cx[n++] = H3(++i, W, NW);
      cx[n++] = H3(++i, W, NE);
      cx[n++] = H3(++i, W, NEE);
      cx[n++] = H3(++i, W, NWW);
      cx[n++] = H3(++i, N, NW);
      cx[n++] = H3(++i, N, NE);
      cx[n++] = H3(++i, NW, NE);
      cx[n++] = H3(++i, W, WW);
      cx[n++] = H3(++i, N, NN);
      cx[n++] = H3(++i, NW, NNWW);
      cx[n++] = H3(++i, NE, NNEE);
      cx[n++] = H3(++i, NW, NWW);
      cx[n++] = H3(++i, NW, NNW);
      cx[n++] = H3(++i, NE, NEE);
      cx[n++] = H3(++i, NE, NNE);
      cx[n++] = H3(++i, N, NNW);
      cx[n++] = H3(++i, N, NNE);
      cx[n++] = H3(++i, N, NNN);
      cx[n++] = H3(++i, W, WWW);
      cx[n++] = H3(++i, WW, NEE);
      cx[n++] = H3(++i, WW, NN);
      cx[n++] = H3(++i, W, NEEE);
      cx[n++] = H3(++i, W, NEEEE);
      cx[n++] = H4(++i, W, N, NW);
      cx[n++] = H4(++i, N, NN, NNN);
      cx[n++] = H4(++i, W, NE, NEE);
      cx[n++] = H5(++i, W, NW, N, NE);
      cx[n++] = H5(++i, N, NE, NN, NNE);
      cx[n++] = H5(++i, N, NW, NNW, NN);
      cx[n++] = H5(++i, W, WW, NWW, NW);
      cx[n++] = hashn(6, (const int64_t[]){++i, W, NW, N, WW, NWW});
      cx[n++] = H2(++i, m->column[0]);
      cx[n++] = H3(++i, N, m->column[1]);
      cx[n++] = H3(++i, W, m->column[1]);
      cx[n++] = (uint64_t)(++i);   /* cm.set(++i): the bare counter */
      for (j = 0; j < G_PLT; j++) cx[n++] = H2(++i, m->ictx_data[j][m->ictx_at[j]]);
      m->ctx = imin(0x1F, m->x / imin(0x20, m->columns[0]));
    } else {
      cx[n++] = H2(++i, N);
      cx[n++] = H2(++i, W);
      cx[n++] = H2(++i, NW);
      cx[n++] = H2(++i, NE);
      cx[n++] = H3(++i, N, NN);
      cx[n++] = H3(++i, W, WW);
      cx[n++] = H3(++i, NE, NNEE);
      cx[n++] = H3(++i, NW, NNWW);
      cx[n++] = H3(++i, W, NEE);
      cx[n++] = H3(++i, clamp4(W + N - NW, W, NW, N, NE) / 2, LMD(clip(N + NE - NNE), clip(N + NW - NNW)));
      cx[n++] = H4(++i, W / 4, NE / 4, m->column[0]);
      cx[n++] = H3(++i, clip(W * 2 - WW) / 4, clip(N * 2 - NN) / 4);
      cx[n++] = H3(++i, clamp4(N + NE - NNE, W, N, NE, NEE) / 4, m->column[0]);
      cx[n++] = H3(++i, clamp4(N + NW - NNW, W, NW, N, NE) / 4, m->column[0]);
      cx[n++] = H3(++i, (W + NEE) / 4, m->column[0]);
      cx[n++] = H3(++i, clip(W + N - NW), m->column[0]);
      cx[n++] = H3(++i, clamp4(N * 3 - NN * 3 + NNN, W, N, NN, NE), LMD(W, clip(NW * 2 - NNW)));
      cx[n++] = H3(++i, clamp4(W * 3 - WW * 3 + WWW, W, N, NE, NEE), LMD(N, clip(NW * 2 - NWW)));
      cx[n++] = H3(++i, (W + clamp4(NE * 3 - NNE * 3 + NNNE, W, N, NE, NEE)) / 2, LMD(N, (NW + NE) / 2));
      cx[n++] = H3(++i, (N + NNN) / 8, clip(N * 3 - NN * 3 + NNN) / 4);
      cx[n++] = H3(++i, (W + WWW) / 8, clip(W * 3 - WW * 3 + WWW) / 4);
      cx[n++] = H2(++i, clip((-WWWW + 5 * WWW - 10 * WW + 10 * W + clamp4(NE * 4 - NNE * 6 + NNNE * 4 - NNNNE, N, NE, NEE, NEEE)) / 5));
      cx[n++] = H3(++i, clip(N * 2 - NN), LMD(N, clip(NN * 2 - NNN)));
      cx[n++] = H3(++i, clip(W * 2 - WW), LMD(NE, clip(N * 2 - NW)));
      cx[n++] = (uint64_t)(uint32_t)~0xde7ec7edu;   /* cm.set(~0xde7ec7ed): the literal is an unsigned int, its complement widens with zeros */
      m->ctx = imin(0x1F, m->x / imax(1, w / imin(32, m->columns[0]))) | ((((abs(W - N) * 16 > W + N) << 1) | (abs(N - NW) > 8)) << 5) | ((W + N) & 0x180);
    }
    int k = 0;
    p8f_cm_step(m->cm, y, 0, c0, W, cx, n, out, &k);
    m->W = (uint8_t)W; m->WW = (uint8_t)WW; m->N = (uint8_t)N; m->NN = (uint8_t)NN; m->NW = (uint8_t)NW; m->NE = (uint8_t)NE; m->NNE = (uint8_t)NNE; m->NNW = (uint8_t)NNW;
    m->NNWW = (uint8_t)NNWW; m->NNEE = (uint8_t)NNEE;
    if (stats) {
      stats[0] = (uint32_t)W; stats[1] = (uint32_t)N; stats[2] = (uint32_t)NN; stats[3] = (uint32_t)WW; stats[4] = stats[5] = stats[6] = 0;
      stats[7] = (uint32_t)(uint8_t)(m->ctx >> (gray & 31));   /* Stats->Image.ctx = ctx >> gray into a U8; the shift count as x86 takes it */
    }
  }
  const int W = m->W, WW = m->WW, N = m->N, NN = m->NN, NW = m->NW, NE = m->NE, NNE = m->NNE, NNW = m->NNW;
  const int B = (uint8_t)(c0 << (8 - bpos));
  int i = 1;
  p8f_dmap_set_direct(m->map[i++], (uint32_t)((((uint8_t)(clip(W + N - NW) - B)) * 8 + bpos) | (LMD(clip(N + NE - NNE), clip(N + NW - NNW)) << 11)));
  for (int j = 0; j < G_MAPS1; i++, j++) p8f_dmap_set_direct(m->map[i], (uint32_t)((m->mctx[j] - B) * 8 + bpos));
  for (int j = 0; i < G_MAPS; i++, j++) p8f_dmap_set_direct(m->map[i], (uint32_t)((m->pols[j] - B) * 8 + bpos));
  int nx = 0, k = 0;
  const int ncm = gray ? 25 : G_CM;
  if (bpos) p8f_cm_step(m->cm, y, bpos, c0, W, NULL, 0, out, &k);
  else k = 5 * ncm;
  nx += k;
  if (gray) {
    for (int j = 0; j < G_MAPS; j++) nx += p8f_dmap_mix(m->map[j], y, 1023, 1, 4, out + nx);   /* Map[i].mix(m): multiplier 1, divisor 4, limit 1023 */
  } else {
    for (int j = 0; j < G_PLT; j++) {
      p8f_dmap_set_direct(m->plt[j], (uint32_t)((bpos << 8) | m->ictx_data[j][m->ictx_at[j]]));
      nx += p8f_dmap_mix(m->plt[j], y, 7, 1, 4, out + nx);                                        /* pltMap[i].mix(m): rate 7, 1 / 4 */
    }
  }
  m->col = (m->col + 1) & 7;
  int ns = 0;
#define SET(v, r) do { sets[ns] = (int)(v); ranges[ns] = (int)(r); ++ns; } while (0)
  SET(m->ctx, 2048);
  SET(m->col, 8);
  SET((N + W) >> 4, 32);
  SET(c0, 256);
  SET(((abs(W - N) > 4) << 9) | ((abs(N - NE) > 4) << 8) | ((abs(W - NW) > 4) << 7) | ((W > N) << 6) | ((N > NE) << 5) | ((W > NW) << 4) | ((W > WW) << 3) | ((N > NN) << 2) |
          ((NW > m->NNWW) << 1) | (NE > m->NNEE), 1024);
  SET(imin(63, m->column[0]), 64);
  SET(imin(127, m->column[1]), 128);
  SET(imin(255, (m->x + m->line) / 32), 256);
#undef SET
#undef BUF
  (void)NNE; (void)NNW;
  return nx;
}

/* ---------------------------------------------------------------- im1bitModel :4634-4673
 * 1-bit images: the last four pixel rows as shift registers, eleven contexts of neighbouring pixels, a bit history and a StateMap each. */
typedef struct P8fBitMaps P8fBitMaps;
P8fBitMaps* p8f_bitmaps_new(int n, uint32_t cells, int pair);
void p8f_bitmaps_emit(P8fBitMaps* p, int i, int cxt, int first, int16_t* out);
enum { B1_N = 11 };
typedef struct Im1 { P8fBitMaps* maps; uint32_t r0, r1, r2, r3; int cxt[B1_N]; int started; } Im1;
Im1* p8f_im1_new(void) {
  Im1* m = (Im1*)calloc(1, sizeof *m);
  m->maps = p8f_bitmaps_new(B1_N, 0x23000, 6);   /* cxt[6]'s range reaches into cxt[7]'s: one lane for the two */
  return m;
}
/* One step (the model works per bit); w: bytes per row; sets[4] / ranges[4]. Returns the 11 inputs. */
int p8f_im1_step(Im1* m, int y, int bpos, const uint8_t* hist, uint32_t bmask, int pos, int w, int16_t* out, int* sets, int* ranges) {
#define BUF(i) ((uint32_t)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
  m->r0 += m->r0 + (uint32_t)y;
  m->r1 += m->r1 + ((BUF(w - 1) >> (7 - bpos)) & 1);
  m->r2 += m->r2 + ((BUF(w + w - 1) >> (7 - bpos)) & 1);
  m->r3 += m->r3 + ((BUF(w + w + w - 1) >> (7 - bpos)) & 1);
  const uint32_t r0 = m->r0, r1 = m->r1, r2 = m->r2, r3 = m->r3;
  int* c = m->cxt;
  c[0] = (int)((r0 & 0x7) | (r1 >> 4 & 0x38) | (r2 >> 3 & 0xc0));
  c[1] = (int)(0x100 + ((r0 & 1) | (r1 >> 4 & 0x3e) | (r2 >> 2 & 0x40) | (r3 >> 1 & 0x80)));
  c[2] = (int)(0x200 + ((r0 & 1) | (r1 >> 4 & 0x1d) | (r2 >> 1 & 0x60) | (r3 & 0xC0)));
  c[3] = (int)(0x300 + ((uint32_t)y | ((r0 << 1) & 4) | ((r1 >> 1) & 0xF0) | ((r2 >> 3) & 0xA)));
  c[4] = (int)(0x400 + ((r0 >> 4 & 0x2AC) | (r1 & 0xA4) | (r2 & 0x349) | (uint32_t)(!(r3 & 0x14D))));
  c[5] = (int)(0x800 + ((uint32_t)y | ((r1 >> 4) & 0xE) | ((r2 >> 1) & 0x70) | ((r3 << 2) & 0x380)));
  c[6] = (int)(0xC00 + (((r1 & 0x30) ^ (r3 & 0x0c0c)) | (r0 & 3)));
  c[7] = (int)(0x1000 + ((uint32_t)(!(r0 & 0x444)) | (r1 & 0xC0C) | (r2 & 0xAE3) | (r3 & 0x51C)));
  c[8] = (int)(0x2000 + ((r0 & 7) | ((r1 >> 1) & 0x3F8) | ((r2 << 5) & 0xC00)));
  c[9] = (int)(0x3000 + ((r0 & 0x3f) ^ (r1 & 0x3ffe) ^ (r2 << 2 & 0x7f00) ^ (r3 << 5 & 0xf800)));
  c[10] = (int)(0x13000 + ((r0 & 0x3e) ^ (r1 & 0x0c0c) ^ (r2 & 0xc800)));
  for (int i = 0; i < B1_N; i++) p8f_bitmaps_emit(m->maps, i, c[i], !m->started, out + i);
  m->started = 1;
  int ns = 0;
#define SET(v, r) do { sets[ns] = (int)(v); ranges[ns] = (int)(r); ++ns; } while (0)
  SET((r0 & 7) | ((r1 & 0x3E) >> 2) | ((r2 & 0x1C0) << 2), 2048);
  SET((uint32_t)y | ((r1 & 0x1C0) >> 5) | ((r2 & 0x1C0) >> 2) | ((r3 & 0x1C0) << 1), 1024);
  SET(((r1 >> 5) & 0xFE) | (uint32_t)y, 256);
  SET((r0 & 0x3) | ((r1 & 0xF80) >> 5), 128);
#undef SET
#undef BUF
  return B1_N;
}

/* ---------------------------------------------------------------- im4bitModel :4675-4742
 * 4-bit images: two pixels per byte; 14 hashed contexts of the neighbouring nibbles per pixel on a HashTable<16> (the device's, p8stage_dev.h
 * P8L_HT16), a run model, a StateMap32 on the partial nibble. */
typedef struct P8fHt16 P8fHt16;
P8fHt16* p8f_ht16_new(uint32_t table_bytes);
int p8f_ht16_step(P8fHt16* p, const uint64_t* keys, int16_t* out);
typedef struct P8fSm32b P8fSm32b;
P8fSm32b* p8f_sm32b_new(int n);
void p8f_sm32b_emit(P8fSm32b* s, int cx, int16_t* out);
typedef struct Im4 {
  P8fHt16* ht; P8fSm32b* map;
  int WW, W, NWW, NW, N, NE, NEE, NNWW, NNW, NN, NNE, NNEE;
  int col, line, run, prev_color, px;
} Im4;
Im4* p8f_im4_new(int level) {
  Im4* m = (Im4*)calloc(1, sizeof *m);
  m->ht = p8f_ht16_new((uint32_t)((0x10000ull << level) / 2));
  m->map = p8f_sm32b_new(16);
  return m;
}
/* One step; c4: the last four whole bytes; sets[6] / ranges[6]. Returns the 43 inputs. */
int p8f_im4_step(Im4* m, int y, int bpos, int c0, uint32_t c4, const uint8_t* hist, uint32_t bmask, int pos, int w, int16_t* out, int* sets, int* ranges) {
#define BUF(i) ((int)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
  uint64_t keys[14];
  int have = 0;
  if (!bpos || bpos == 4) {
    m->WW = m->W; m->NWW = m->NW; m->NW = m->N; m->N = m->NE; m->NE = m->NEE; m->NNWW = m->NWW; m->NNW = m->NN; m->NN = m->NNE; m->NNE = m->NNEE;
    if (!bpos) { m->W = (int)(c4 & 0xF); m->NEE = BUF(w - 1) >> 4; m->NNEE = BUF(w * 2 - 1) >> 4; }
    else { m->W = c0 & 0xF; m->NEE = BUF(w - 1) & 0xF; m->NNEE = BUF(w * 2 - 1) & 0xF; }
    if (m->W != m->WW || !m->col) { m->prev_color = m->WW; m->run = 0; } else m->run = imin(0xFFF, m->run + 1);
    m->px = 1;
    const int W = m->W, WW = m->WW, N = m->N, NN = m->NN, NW = m->NW, NE = m->NE, NEE = m->NEE, NWW = m->NWW, NNW = m->NNW, NNE = m->NNE, NNEE = m->NNEE, NNWW = m->NNWW;
    const int col = m->col, line = m->line;
    int64_t i = 0;
    keys[0] = H4(i, W, NW, N);
    i++; keys[1] = H3(i, N, imin(0xFFF, col / 8));
    i++; keys[2] = hashn(6, (const int64_t[]){i, W, NW, N, NN, NE});
    i++; keys[3] = H5(i, W, N, NE + NNE * 16, NEE + NNEE * 16);
    i++; keys[4] = H5(i, W, N, NW + NNW * 16, NWW + NNWW * 16);
    i++; keys[5] = H5(i, W, ilog2u((unsigned)(m->run + 1)), m->prev_color, col / imax(1, w / 2));
    i++; keys[6] = H3(i, NE, imin(0x3FF, (col + line) / imax(1, w * 8)));
    i++; keys[7] = H3(i, NW, (col - line) / imax(1, w * 8));
    i++; keys[8] = H4(i, WW * 16 + W, NN * 16 + N, NNWW * 16 + NW);
    i++; keys[9] = H3(i, N, NN);
    i++; keys[10] = H3(i, W, WW);
    i++; keys[11] = H3(i, W, NE);
    i++; keys[12] = H4(i, WW, NN, NEE);
    keys[13] = (uint64_t)(int64_t)-1;   /* cp[13] = t[-1] */
    have = 1;
    ++m->col;
    m->col *= m->col < w * 2;
    m->line += (!m->col);
  } else m->px += m->px + y;
  int nx = p8f_ht16_step(m->ht, have ? keys : NULL, out);
  p8f_sm32b_emit(m->map, m->px, out + nx); nx++;
  int ns = 0;
#define SET(v, r) do { sets[ns] = (int)(v); ranges[ns] = (int)(r); ++ns; } while (0)
  SET(m->W * 16 + m->px, 256);
  SET(imin(31, m->col / imax(1, w / 16)) + m->N * 32, 512);
  SET((bpos & 3) + 4 * m->W + 64 * imin(7, (int)ilog2u((unsigned)(m->run + 1))), 512);
  SET(m->W + m->NE * 16 + (bpos & 3) * 256, 1024);
  SET(m->px, 16);
  SET(0, 1);
#undef SET
#undef BUF
  return nx;
}
