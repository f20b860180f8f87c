/* p8front/p8f_lpm.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's linearPredictionModel (reference src/models/paq8.cpp:4476-4502): three recursive
 * least-squares predictors OLS<double, U8> (:1364-1466; 32 taps, covariance decay 0.995, Cholesky factor + solve every
 * 4th byte) over strides of the last 64 bytes, two fixed linear extrapolations, each turned into two mixer inputs by a
 * SmallStationaryContextMap(11, 1) (:891-933). The one piece of paq8's text path that computes in double precision:
 * every sum below runs in the reference's index order, products rounded before they are added (no contraction: built
 * with -ffp-contract=off), sqrt and division are IEEE. Parity: tests/test_p8stage_host.py (stage vs columns 434..2024 of reference traces). */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct DMap DMap;
DMap* p8f_dmap_new(int kind, int bits_of_context, int bits_per_context, int rate);
void p8f_dmap_set_direct(DMap* m, uint32_t ctx);
int p8f_dmap_mix(DMap* m, int y, int a, int mul, int div, int16_t* out);

enum { N = 32, KMAX = 4, NOLS = 3, NPRD = 5 };
typedef struct {
  int km, index;
  double lambda, nu;
  double x[N], w[N], b[N], cov[N][N], chol[N][N];
} Ols;
typedef struct { Ols ols[NOLS]; DMap* map[NPRD]; uint8_t prd[NPRD]; } Lpm;   /* map: SmallStationaryContextMap(11, 1), on the device */

static int ols_factor(Ols* o) {  /* :1372-1395 */
  for (int i = 0; i < N; i++) for (int j = 0; j < N; j++) o->chol[i][j] = o->cov[i][j];
  for (int i = 0; i < N; i++) o->chol[i][i] += o->nu;
  for (int i = 0; i < N; i++) {
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
static void ols_solve(Ols* o) {  /* :1397-1410 */
  for (int i = 0; i < N; i++) {
    double sum = o->b[i];
    for (int j = 0; j < i; j++) sum -= (o->chol[i][j] * o->w[j]);
    o->w[i] = sum / o->chol[i][i];
  }
  for (int i = N - 1; i >= 0; i--) {
    double sum = o->w[i];
    for (int j = i + 1; j < N; j++) sum -= (o->chol[j][i] * o->w[j]);
    o->w[i] = sum / o->chol[i][i];
  }
}
static void ols_update(Ols* o, uint8_t val) {  /* :1452-1464 */
  for (int j = 0; j < N; j++)
    for (int i = 0; i < N; i++) o->cov[j][i] = o->lambda * o->cov[j][i] + (1.0 - o->lambda) * (o->x[j] * o->x[i]);
  for (int i = 0; i < N; i++) o->b[i] = o->lambda * o->b[i] + (1.0 - o->lambda) * (o->x[i] * ((double)val - 0.0));
  if (++o->km >= KMAX) {
    if (!ols_factor(o)) ols_solve(o);
    o->km = 0;
  }
}
static void ols_add(Ols* o, uint8_t v) { if (o->index < N) o->x[o->index++] = (double)v - 0.0; }
static double ols_predict(Ols* o) {
  o->index = 0;
  double sum = 0.;
  for (int i = 0; i < N; i++) sum += o->w[i] * o->x[i];
  return sum + 0.0;
}
static uint8_t clip(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

Lpm* p8f_lpm_new(void) {
  Lpm* m = (Lpm*)calloc(1, sizeof *m);
  for (int i = 0; i < NOLS; ++i) { m->ols[i].lambda = 0.995; m->ols[i].nu = 0.001; }
  for (int i = 0; i < NPRD; ++i) m->map[i] = p8f_dmap_new(0, 11, 1, 0);
  return m;
}
/* last[i-1] = buf(i), i = 1 .. 64 */
int p8f_lpm_step(Lpm* m, int y, int bpos, int c0, const uint8_t* last, int16_t* out) {
  if (bpos == 0) {
    const uint8_t W = last[0], WW = last[1], WWW = last[2];
    for (int i = 0; i < NOLS; i++) ols_update(&m->ols[i], W);
    for (int i = 1; i <= 32; i++) {
      ols_add(&m->ols[0], last[i - 1]);
      ols_add(&m->ols[1], last[i * 2 - 2]);
      ols_add(&m->ols[2], last[i * 2 - 1]);
    }
    int i = 0;
    for (; i < NOLS; i++) m->prd[i] = clip((int)floor(ols_predict(&m->ols[i])));
    m->prd[i++] = clip(W * 2 - WW);
    m->prd[i] = clip(W * 3 - WW * 3 + WWW);
  }
  const uint8_t B = (uint8_t)(c0 << (8 - bpos));
  int n = 0;
  for (int i = 0; i < NPRD; i++) {
    p8f_dmap_set_direct(m->map[i], (uint32_t)((m->prd[i] - B) * 8 + bpos));   /* :4498 */
    n += p8f_dmap_mix(m->map[i], y, 6, 1, 2, out + n);                          /* mix(m, 6, 1, 2) :909-919 */
  }
  return n;
}
