/* oracle/fxcm_maps.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of fxcm's three hashed context maps (reference src/models/fxcmv1.cpp): ContextMap (:971-1174, 64-byte
 * buckets of 7 slots), ContextMap1 (:1176-1379, 32-byte buckets of 3) and ContextMap2 (:1408-1612, 128-byte buckets of
 * 14, table twice the size argument). They differ in bucket geometry only, so one parametrised body serves all
 * three: a bucket is {A 16-bit checksums, one byte holding the last two slots used, A x 7 bit-history states};
 * slot 0..6 of a 7-byte group are the histories after 0, 1 and 2 more bits; a new bucket is looked up at bits 0, 2
 * and 5 of each byte; bytes 3..4 of the byte-boundary group double as a run model (count, last byte); histories for
 * bits 2-7 are only created once a context has been seen twice. Per context and bit the map emits 5 inputs (6 with
 * the st2 table): StateMap probability through two stretch tables, two state-derived terms, a confidence constant,
 * the run prediction. Pinned against the reference's own structs in tests/test_oracle_fxcmcore.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "fxcm_core.h"

#define FX_MAXCXT 8
#define NONE 0xffffffffu
typedef struct {
  int A, B;                     /* slots per bucket, bytes per bucket */
  int C, cn, result, cms, cms3, cms4, kep, skip2;
  uint8_t* t; uint32_t tmask;
  uint32_t cp[FX_MAXCXT], cp0[FX_MAXCXT], runp[FX_MAXCXT], cxt[FX_MAXCXT];   /* byte offsets into t (cp: NONE = null pointer) */
  FxStateMap sm[FX_MAXCXT];
  uint16_t cxtMask;
  const uint8_t* nn;
  int16_t rc1[512], st1[4096], st2[4096], st32[256], st8[256];
} FxCm;

/* E::get / E1::get (:933-957, :1381-1406): slot with this checksum, else replace the lowest-priority slot that is not one
 * of the last two used. Returns the offset of the slot's 7 states inside the bucket. */
static uint32_t bucket_get(uint8_t* b, int A, uint16_t ch, int keep) {
#define CHK(i) (*(uint16_t*)(b + 2 * (i)))
#define BH(i) ((uint32_t)(2 * A + 1 + 7 * (i)))
  const uint8_t last = b[2 * A];
  if (CHK(last & 15) == ch) return BH(last & 15);
  int lowest = 0xffff, bi = 0;
  for (int i = 0; i < A; ++i) {
    if (CHK(i) == ch) { b[2 * A] = (uint8_t)(last << 4 | i); return BH(i); }
    const int pri = b[BH(i)];
    if (pri < lowest && (last & 15) != i && (last >> 4) != i) { lowest = pri; bi = i; }
  }
  b[2 * A] = (uint8_t)(last << 4 | bi | keep);
  CHK(bi) = ch;
  memset(b + BH(bi), 0, 7);
  return BH(bi);
#undef CHK
#undef BH
}
static uint32_t state_byte_location(int bpos, int c0) {  /* getStateByteLocation :959-964 */
  const uint32_t smask = (0x31031010u >> (bpos << 2)) & 0x0F;
  return smask + ((uint32_t)c0 & smask);
}
static int pre(const uint8_t* nn, int state) {
  const uint32_t n0 = nn[state * 4 + 2] * 3u + 1, n1 = nn[state * 4 + 3] * 3u + 1;
  return (int)((n1 << 12) / (n0 + n1));
}

/* kind 0 / 1 / 2 = ContextMap / ContextMap1 / ContextMap2; the other arguments are Init's (:1005, :1209, :1441);
 * which_st2: 0 = st2_p0 (never filled: zeros), 1 = st2_p1, 2 = st2_p2 (:4857-4860). */
FxCm* orc_fx_cm_new(int kind, uint32_t m, int c, int s3, int which_sta, int cs4, int k, int u, int which_st2) {
  FxCm* x = (FxCm*)calloc(1, sizeof *x);
  x->A = kind == 0 ? 7 : kind == 1 ? 3 : 14;
  x->B = kind == 0 ? 64 : kind == 1 ? 32 : 128;
  x->C = c & 255;
  if (kind == 2) { const int m2 = (int)(m * 2); x->tmask = (uint32_t)((m2 >> 7) - 1); x->t = (uint8_t*)calloc((size_t)(m2 >> 7) + 128, 128); }
  else { x->tmask = (m >> 6) - 1; x->t = (uint8_t*)calloc((size_t)(m >> 6) + 64, (size_t)x->B); }
  x->cxtMask = (uint16_t)(((1 < x->C) - 1) * 2);  /* sic: a comparison, not a shift (:1009) */
  x->kep = k; x->nn = fx_sta(which_sta);
  const int cmul = (c >> 8) & 255;
  x->cms = (c >> 16) & 255; x->cms4 = cs4; x->cms3 = s3; x->skip2 = u;
  for (int i = 0; i < x->C; i++) fx_statemap_init(&x->sm[i], 256, x->nn);
  const uint32_t first = (uint32_t)(2 * x->A + 1);  /* &t[0].bh[0][0] */
  for (int i = 0; i < x->C; ++i) { x->cp0[i] = x->cp[i] = first; x->runp[i] = first + 3; }
  for (int rc = 0; rc < 256; rc++) {
    int v = fx_ilog(rc);
    v = v << (2 + (~rc & 1));
    if ((rc & 1) == 0) v = v * cmul / 4;
    x->rc1[rc + 256] = (int16_t)fx_clp(v);
    x->rc1[rc] = (int16_t)fx_clp(-v);
  }
  for (int i = 0; i < 4096; i++) {
    x->st1[i] = (int16_t)fx_clp(fx_sc(x->cms * fx_stretch(i)));
    x->st2[i] = which_st2 == 0 ? 0 : (int16_t)fx_clp(fx_sc((which_st2 == 1 ? 12 : 14) * (i - 2048)));
  }
  for (int s = 0; s < 256; s++) {
    const int n0 = -!x->nn[s * 4 + 2], n1 = -!x->nn[s * 4 + 3];
    int r = 0, sp0 = 0;
    if (n1 - n0 == 1) { sp0 = 0; r = 1; }
    if (n1 - n0 == -1) { sp0 = 4095; r = 1; }
    if (r) {
      x->st8[s] = (int16_t)fx_clp(fx_sc(x->cms4 * (pre(x->nn, s) - sp0)));
      x->st32[s] = (int16_t)fx_clp(fx_sc(x->cms3 * fx_stretch(pre(x->nn, s))));
      if (s < 8) x->st32[s] = 0;
    }
  }
  return x;
}
void fx_cm_set(FxCm* x, uint32_t cx) {  /* :1057-1065 */
  const uint32_t i = (uint32_t)x->cn++;
  cx = cx * 987654323u + i;
  cx = cx << 16 | cx >> 16;
  x->cxt[i] = cx * 123456791u + i;
  x->cxtMask = (uint16_t)(x->cxtMask * 2);
}
uint32_t fx_cm_context(const FxCm* x, int i) { return x->cxt[i]; }
int fx_cm_skipmask(const FxCm* x) { return x->cxtMask; }
void fx_cm_skip(FxCm* x) { x->cn++; x->cxtMask = (uint16_t)((x->cxtMask + 1) * 2); }  /* sets() :1066-1070 */

static void skipped(const FxCm* x, FxSink* s) {  /* mix4 :1099-1107 */
  fx_add(s, 0);
  if (x->skip2 == 1) fx_add(s, 0);
  fx_add(s, 0); fx_add(s, 0);
  fx_add(s, 64); fx_unexport(s);
  fx_add(s, 0);
}
int fx_cm_mix(FxCm* x, FxSink* s, int y, int bpos, int c0, int c1) {  /* mix1 / mix :1110-1173 */
  uint8_t* t = x->t;
  x->result = 0;
  for (int i = 0; i < x->cn; ++i) {
    if ((x->cxtMask >> (x->cn - i)) & 1) { skipped(x, s); continue; }
    if (x->cp[i] != NONE) t[x->cp[i]] = x->nn[t[x->cp[i]] * 4 + y];
    int state = 0;
    if (bpos > 1 && t[x->runp[i]] == 0) x->cp[i] = NONE;
    else {
      const uint16_t chk = (uint16_t)((x->cxt[i] >> 16) ^ (uint32_t)i);
#define BUCKET(ctx) ((size_t)((ctx) & x->tmask) * (size_t)x->B)
      if (bpos == 2 || bpos == 5) { const size_t b = BUCKET(x->cxt[i] + (uint32_t)c0); x->cp0[i] = x->cp[i] = (uint32_t)(b + bucket_get(t + b, x->A, chk, x->kep)); }
      else if (bpos) x->cp[i] = x->cp0[i] + state_byte_location(bpos, c0);
      else {
        size_t b = BUCKET(x->cxt[i] + (uint32_t)c0);
        x->cp0[i] = x->cp[i] = (uint32_t)(b + bucket_get(t + b, x->A, chk, x->kep));
        if (t[x->cp0[i] + 3] == 2) {  /* second visit: create the histories for bits 2-7 of the byte seen the first time */
          const int c = t[x->cp0[i] + 4] + 256;
          b = BUCKET(x->cxt[i] + (uint32_t)(c >> 6));
          uint8_t* p = t + b + bucket_get(t + b, x->A, chk, x->kep);
          p[0] = (uint8_t)(1 + ((c >> 5) & 1));
          p[1 + ((c >> 5) & 1)] = (uint8_t)(1 + ((c >> 4) & 1));
          p[3 + ((c >> 4) & 3)] = (uint8_t)(1 + ((c >> 3) & 1));
          b = BUCKET(x->cxt[i] + (uint32_t)(c >> 3));
          p = t + b + bucket_get(t + b, x->A, chk, x->kep);
          p[0] = (uint8_t)(1 + ((c >> 2) & 1));
          p[1 + ((c >> 2) & 1)] = (uint8_t)(1 + ((c >> 1) & 1));
          p[3 + ((c >> 1) & 3)] = (uint8_t)(1 + (c & 1));
          t[x->cp0[i] + 6] = 0;
        }
        uint8_t* run = t + x->runp[i];  /* run count of the previous context */
        if (run[0] == 0) { run[0] = 2; run[1] = (uint8_t)c1; }
        else if (run[1] != c1) { run[0] = 1; run[1] = (uint8_t)c1; }
        else if (run[0] < 254) run[0] = (uint8_t)(run[0] + 2);
        x->runp[i] = x->cp0[i] + 3;
      }
#undef BUCKET
      state = t[x->cp[i]];
    }
    if (state == 0) {  /* mix3 :1077-1097 */
      fx_add(s, 0);
      if (x->skip2 == 1) fx_add(s, 0);
      fx_add(s, 0); fx_add(s, 0);
      fx_add(s, 64); fx_unexport(s);
    } else {
      const int p1 = fx_statemap_set(&x->sm[i], y, state);
      fx_add(s, x->st1[p1]);
      if (x->skip2 == 1) fx_add(s, x->st2[p1]);
      fx_add(s, x->st8[state]);
      fx_add(s, x->st32[state]);
      fx_add(s, 0); fx_unexport(s);
      x->result++;
    }
    const uint8_t* run = t + x->runp[i];
    const int bposshift = 7 - bpos, c0shift_bpos = (c0 << 1) ^ (256 >> bposshift);
    const int b = c0shift_bpos ^ (run[1] >> bposshift);
    fx_add(s, b <= 1 ? x->rc1[run[0] + b * 256] : 0);
  }
  if (bpos == 7) { x->cn = 0; x->cxtMask = 0; }
  return x->result;
}
int orc_fx_cm_step(FxCm* x, int y, int bpos, int c0, uint32_t c4, const uint32_t* cx, const uint8_t* skip, int n, int16_t* out, float* exported,
                   int* nexported, int* ninputs) {
  FxSink s;
  s.ncount = s.pidx = 0;
  if (bpos == 0)
    for (int i = 0; i < n; ++i) { if (skip[i]) fx_cm_skip(x); else fx_cm_set(x, cx[i]); }
  const int r = fx_cm_mix(x, &s, y, bpos, c0, (int)(c4 & 255));
  memcpy(out, s.n, (size_t)s.ncount * 2);
  memcpy(exported, s.exported, (size_t)s.pidx * 4);
  *nexported = s.pidx; *ninputs = s.ncount;
  return r;
}
