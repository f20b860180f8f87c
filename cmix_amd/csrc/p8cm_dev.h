// p8cm_dev.h -- paq8's older ContextMap (reference src/models/paq8.cpp:1010-1145: the same 64-byte bucket as ContextMap2,
// found through E::get :1038-1047; a u16 StateMap per context :623-645; five mixer inputs per context and bit) for the
// whole FAMILY of instances a predictor holds -- wordModel's (61 contexts), sparseModel's (42), sparseModel1's (31),
// indirectModel's (15), nestModel's (12), recordModel's, distanceModel's, XMLModel's: about 190 contexts, 950 of paq8's
// 1552 mixer inputs. The family is one unit because of ONE awkward global: a bit history reaching state >= 204 decays
// with probability drawn from the process-wide rnd() (:152-165, :1075), and the draws are consumed in the order the
// reference walks models and contexts -- a data-dependent, global order.
//
// Execution model: one workgroup, one lane per context of the family (instances in the reference's calling order);
// a bit is four barrier steps:
//   a  read-only: the <= 5 buckets the context touches this bit (as in p8cm2_dev.h) and whether its state update draws
//   b  overlap check inside each instance; every lane counts the draws of the lanes before it (its draw's rank)
//   b2 lane 0 produces as many values of the generator as there are draws this bit
//   c  every lane runs its context with "its" draw -- or, if ANY instance has an overlap this bit (a draw decision may
//      then depend on an earlier context's write), lane 0 walks the whole family in the reference's order
// Registers in LDS for the chunk. Single source: tests/host/p8cm_emul.cpp runs it on the host against the oracle.
#ifndef CMX_P8CM_DEV_H
#define CMX_P8CM_DEV_H
#include <stdint.h>

#include "p8cm2_dev.h"   // P8_HD, bucket geometry, p8d_bucket_find

enum { P8CM_MAXI = 16, P8CM_MAXS = 256 };

struct P8CmInst { uint8_t* table; uint32_t mask; int first, count; };   // slots first .. first + count - 1
struct P8CmRegs { uint32_t cp[P8CM_MAXS], cp0[P8CM_MAXS], runp[P8CM_MAXS]; int sm_cxt[P8CM_MAXS]; };
struct P8Rnd { uint32_t table[64]; int i; };
struct P8CmDev {
  P8CmInst inst[P8CM_MAXI]; int ninst, nslots, slot_parallel;
  uint8_t slot_inst[P8CM_MAXS];
  const uint8_t* nex; const int16_t* stretch; const uint8_t* ilog;
  uint16_t* sm;                          // [nslots][256] StateMap cells
  P8CmRegs regs;
  P8Rnd rnd;
  int last_y, c1;                        // carried between chunks: the last coded bit, the last whole byte
  // placement of the outputs (a stand-alone family: row = 5 * nslots inputs, context s at 5 s; inside the paq8 stage:
  // the 1552-vector, contexts where contextModel2's walk puts them) and the one context whose VALUE is device state:
  // sparseModel's hash(2, order) (reference :4513), looked up from the order-N map's return value of the byte
  int row_stride, order_slot;
  int16_t slot_off[P8CM_MAXS];
  uint32_t order_ctx[16]; uint16_t order_chk[16];
};
struct P8CmShared {
  P8CmRegs r;
  int32_t touched[P8CM_MAXS][5];
  uint8_t draws[P8CM_MAXS];
  uint16_t rank[P8CM_MAXS];
  uint32_t draw_val[P8CM_MAXS];
  int conflict, ndraws;
  P8Rnd rnd;
  // the slots [act_lo, act_hi) are the ones whose ContextMap is called this byte with a context set (an image / audio model's family:
  // im8bitModel sets 25 or 52 contexts, wavModel's map is silent during a block's first sample; ContextMap::mix loops over cn :1072);
  // act_hi == 0: all
  int act_lo, act_hi;
};
P8_HD int p8d_cm_lo(const P8CmShared* sh) { return sh->act_hi > 0 ? sh->act_lo : 0; }
P8_HD int p8d_cm_hi(const P8CmDev* d, const P8CmShared* sh) { return sh->act_hi > 0 ? sh->act_hi : d->nslots; }
struct P8CmBit { int y, bp, c0, c1, order; const uint32_t* ctx; const uint16_t* chk; int16_t* out; };


P8_HD uint32_t p8d_cm_ctxof(const P8CmDev* d, const P8CmBit& u, int s) { return s == d->order_slot ? d->order_ctx[u.order] : u.ctx[s]; }
P8_HD uint16_t p8d_cm_chkof(const P8CmDev* d, const P8CmBit& u, int s) { return s == d->order_slot ? d->order_chk[u.order] : u.chk[s]; }
P8_HD uint32_t p8d_rnd_next(P8Rnd* g) {   // Random::operator() :158-161
  const uint32_t i = (uint32_t)g->i + 1u;   // (the reference's `int i` wraps after 2^31 draws; only i mod 64 matters: unsigned here, no overflow to reason about)
  g->i = (int)i;
  return g->table[i & 63] = g->table[(i - 24) & 63] ^ g->table[(i - 55) & 63];
}
// one context, one bit: ContextMap::mix1's loop body (:1072-1145). rv: the generator value this context consumes if its
// update draws (parallel path), or NULL to draw from the generator itself (serial walk).
P8_HD void p8d_cm_ctx(P8CmDev* d, P8CmShared* sh, const P8CmBit& u, int s, const uint32_t* rv) {
  const P8CmInst* x = &d->inst[d->slot_inst[s]];
  uint8_t* T = x->table;
  P8CmRegs* r = &sh->r;
  const int bp = u.bp, c0 = u.c0, c1 = u.c1, y1 = u.y;
  if (r->cp[s] != P8_NIL) {
    int ns = d->nex[4 * T[r->cp[s]] + y1];
    if (ns >= 204) {
      const uint32_t v = rv ? *rv : p8d_rnd_next(&sh->rnd);
      if ((uint32_t)(v << ((452 - ns) >> 3))) ns -= 4;
    }
    T[r->cp[s]] = (uint8_t)ns;
  }
  if (bp > 1 && T[r->runp[s]] == 0) r->cp[s] = P8_NIL;
  else if (bp == 1 || bp == 3 || bp == 6) r->cp[s] = r->cp0[s] + 1 + (uint32_t)(c0 & 1);
  else if (bp == 4 || bp == 7) r->cp[s] = r->cp0[s] + 3 + (uint32_t)(c0 & 3);
  else if (bp == 2 || bp == 5) r->cp0[s] = r->cp[s] = p8d_bucket_find(T, (p8d_cm_ctxof(d, u, s) + (uint32_t)c0) & x->mask, p8d_cm_chkof(d, u, s));
  else {
    const uint16_t checksum = p8d_cm_chkof(d, u, s);
    const uint32_t cx = p8d_cm_ctxof(d, u, s);
    r->cp0[s] = r->cp[s] = p8d_bucket_find(T, (cx + (uint32_t)c0) & x->mask, checksum);
    uint8_t* s0 = T + r->cp0[s];
    if (s0[3] == 2) {
      const int cc = s0[4] + 256;
      uint8_t* p = T + p8d_bucket_find(T, (cx + (uint32_t)(cc >> 6)) & x->mask, checksum);
      p[0] = (uint8_t)(1 + ((cc >> 5) & 1));
      p[1 + ((cc >> 5) & 1)] = (uint8_t)(1 + ((cc >> 4) & 1));
      p[3 + ((cc >> 4) & 3)] = (uint8_t)(1 + ((cc >> 3) & 1));
      p = T + p8d_bucket_find(T, (cx + (uint32_t)(cc >> 3)) & x->mask, checksum);
      p[0] = (uint8_t)(1 + ((cc >> 2) & 1));
      p[1 + ((cc >> 2) & 1)] = (uint8_t)(1 + ((cc >> 1) & 1));
      p[3 + ((cc >> 1) & 3)] = (uint8_t)(1 + (cc & 1));
      s0[6] = 0;
    }
    uint8_t* rp = T + r->runp[s];  // run count of the previous context
    if (rp[0] == 0) { rp[0] = 2; rp[1] = (uint8_t)c1; }
    else if (rp[1] != c1) { rp[0] = 1; rp[1] = (uint8_t)c1; }
    else if (rp[0] < 254) rp[0] = (uint8_t)(rp[0] + 2);
    else if (rp[0] == 255) rp[0] = 128;
    r->runp[s] = r->cp0[s] + 3;
  }
  int16_t* o = u.out + d->slot_off[s];
  const uint8_t* rp = T + r->runp[s];
  const int rc = rp[0];
  if ((rp[1] + 256) >> (8 - bp) == c0) {
    const int b = ((rp[1] >> (7 - bp)) & 1) * 2 - 1;
    o[0] = (int16_t)(b * (d->ilog[rc + 1] << (2 + (~rc & 1))));
  } else o[0] = 0;
  const int st8 = r->cp[s] != P8_NIL ? T[r->cp[s]] : 0;
  uint16_t* smt = d->sm + (size_t)s * 256;   // StateMap::p :636-641
  smt[r->sm_cxt[s]] = (uint16_t)(smt[r->sm_cxt[s]] + (((y1 << 16) - smt[r->sm_cxt[s]] + 128) >> 8));
  r->sm_cxt[s] = st8;
  const int p1 = smt[st8] >> 4;
  const int st = (d->stretch[p1] + (1 << 1)) >> 2;
  o[1] = (int16_t)st;
  o[2] = (int16_t)((p1 - 2047 + (1 << 2)) >> 3);
  const int n0 = -!d->nex[4 * st8 + 2], n1 = -!d->nex[4 * st8 + 3];
  const int dn = n1 - n0;
  o[3] = (int16_t)(st * (dn < 0 ? -dn : dn));
  const int p0 = 4095 - p1;
  o[4] = (int16_t)(((p1 & n0) - (p0 & n1) + (1 << 3)) >> 4);
}
P8_HD void p8d_cm_touch(P8CmDev* d, P8CmShared* sh, const P8CmBit& u, int s) {   // step a
  const P8CmInst* x = &d->inst[d->slot_inst[s]];
  const uint8_t* T = x->table;
  const P8CmRegs* r = &sh->r;
  int32_t* L = sh->touched[s];
  for (int j = 0; j < 5; j++) L[j] = -1;
  if (s == p8d_cm_lo(sh)) sh->conflict = 0;
  sh->draws[s] = 0;
  if (r->cp[s] != P8_NIL) {
    L[0] = (int32_t)(r->cp[s] >> 6);
    sh->draws[s] = d->nex[4 * T[r->cp[s]] + u.y] >= 204;
  }
  L[1] = (int32_t)(r->runp[s] >> 6);
  if (u.bp > 1 && T[r->runp[s]] == 0) return;
  if (u.bp == 0 || u.bp == 2 || u.bp == 5) {
    const uint32_t nb = (p8d_cm_ctxof(d, u, s) + (uint32_t)u.c0) & x->mask;
    L[2] = (int32_t)nb;
    if (u.bp == 0) {
      const uint8_t* p = T + (size_t)nb * P8_B_SIZE;
      const uint16_t* cs = (const uint16_t*)p;
      const int mru = p[P8_B_MRU];
      int slot = -1;
      if (cs[mru & 15] == p8d_cm_chkof(d, u, s)) slot = mru & 15;
      else for (int j = 0; j < 7; ++j) if (cs[j] == p8d_cm_chkof(d, u, s)) { slot = j; break; }
      if (slot >= 0 && p[P8_B_STATE + 7 * slot + 3] == 2) {
        const int cc = p[P8_B_STATE + 7 * slot + 4] + 256;
        L[3] = (int32_t)((p8d_cm_ctxof(d, u, s) + (uint32_t)(cc >> 6)) & x->mask);
        L[4] = (int32_t)((p8d_cm_ctxof(d, u, s) + (uint32_t)(cc >> 3)) & x->mask);
      }
    }
  }
}
P8_HD void p8d_cm_check(P8CmDev* d, P8CmShared* sh, int s) {   // step b
  const P8CmInst* x = &d->inst[d->slot_inst[s]];
  const int32_t* L = sh->touched[s];
  int hit = 0;
  const int lo = p8d_cm_lo(sh), nall = p8d_cm_hi(d, sh);
  for (int o = x->first > lo ? x->first : lo; o < x->first + x->count && o < nall && !hit; o++) {
    if (o == s) continue;
    const int32_t* O = sh->touched[o];
    for (int a = 0; a < 5 && !hit; a++)
      if (L[a] >= 0)
        for (int b = 0; b < 5; b++) if (L[a] == O[b]) { hit = 1; break; }
  }
  if (hit) sh->conflict = 1;
  int rank = 0;
  for (int o = lo; o < s; o++) rank += sh->draws[o];
  sh->rank[s] = (uint16_t)rank;
  if (s == nall - 1) sh->ndraws = rank + sh->draws[s];
}
P8_HD void p8d_cm_draw(P8CmDev* d, P8CmShared* sh, int s) {   // step b2
  if (s != p8d_cm_lo(sh) || sh->conflict || !d->slot_parallel) return;
  for (int k = 0; k < sh->ndraws; k++) sh->draw_val[k] = p8d_rnd_next(&sh->rnd);
}
P8_HD void p8d_cm_run(P8CmDev* d, P8CmShared* sh, const P8CmBit& u, int s) {   // step c
  if (!sh->conflict && d->slot_parallel) p8d_cm_ctx(d, sh, u, s, &sh->draw_val[sh->rank[s]]);
  else if (s == p8d_cm_lo(sh)) { const int nall = p8d_cm_hi(d, sh); for (int j = s; j < nall; j++) p8d_cm_ctx(d, sh, u, j, nullptr); }
}
// uniform values of step t of a chunk. order: the order-N map's return values per step (NULL: no order context)
P8_HD P8CmBit p8d_cm_bit(const P8CmDev* d, const uint32_t* ctx, const uint16_t* chk, const uint8_t* bits_in, int16_t* out, const uint8_t* order, int t, int* last_y, int* c1) {
  P8CmBit u;
  const int bp = t & 7, nslots = d->nslots;
  int c0 = 1;
  for (int j = 0; j < bp; j++) c0 = c0 * 2 + bits_in[t - bp + j];
  u.y = *last_y; u.bp = bp; u.c0 = c0; u.c1 = *c1;
  u.order = order ? order[t - bp] : 0;
  u.ctx = ctx + (size_t)(t >> 3) * (size_t)nslots;
  u.chk = chk + (size_t)(t >> 3) * (size_t)nslots;
  u.out = out + (size_t)t * (size_t)d->row_stride;
  *last_y = bits_in[t];
  if (bp == 7) *c1 = (c0 * 2 + bits_in[t]) & 0xff;
  return u;
}
#endif
