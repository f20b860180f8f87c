// p8stage_dev.h -- device bodies of the paq8 stage that are not one of the table families (p8cm_dev.h, p8cm2_dev.h,
// p8dmc_dev.h): (1) the SMALL LANES -- every small learner of contextModel2 and its sub-models, one lane each, driven by
// the op words of the host front end (p8_rec.h): SmallStationaryContextMap, StationaryMap, IndirectMap (reference
// src/models/paq8.cpp:891-1008), StateMap32 read-outs (:645-690), picModel's maps (:3844-3864), plus the inputs the host
// computes itself; (2) the tail of Predictor::update (:8281-8358): selector completion, the APM / APM1 chains, the
// exported values. Single source: tests/host/p8stage_emul.cpp runs these bodies on the host.
#ifndef CMX_P8STAGE_DEV_H
#define CMX_P8STAGE_DEV_H
#include <stdint.h>

#include "p8_rec.h"
#include "p8cm2_dev.h"   // P8_HD

// ---------------------------------------------------------------- small lanes
struct P8JpgDev;
struct P8LaneDev {
  P8Lane q;
  P8JpgDev* jpg;             // P8L_JPG: the worker's state (below)
  uint32_t mask, stride;     // context mask and cells per context (dmaps)
  uint8_t* c8; uint16_t* c16; uint32_t* c32;   // the lane's cells, by kind
  uint32_t* sm;              // IND: StateMap32(256); PIC: unused
  uint16_t* sm16;            // PIC: u16 StateMap(256)
};
struct P8LaneRegs { uint32_t cp, context, B, bcount; int sm_cxt; };
struct P8LanesDev {
  int nlanes;
  P8LaneDev lane[P8_NLANE];
  P8LaneRegs regs[P8_NLANE];   // home between chunks
  const uint8_t* nex; const int16_t* stretch; const uint8_t* ilog;
  int last_y;
};

P8_HD int p8s_sm32(uint32_t* t, int* cxt, int y, int cx, int limit) {   // StateMap32::p :660-672
  uint32_t p0 = t[*cxt];
  const int n = p0 & 1023, pr = p0 >> 10;
  if (n < limit) ++p0; else p0 = (p0 & 0xfffffc00u) | (uint32_t)limit;
  p0 += ((uint32_t)(((y << 22) - pr) >> 3) * (uint32_t)(16384 / (n + n + 3))) & 0xfffffc00u;   // the product wraps (as the reference's compiled code does)
  t[*cxt] = p0;
  *cxt = cx;
  return (int)(t[cx] >> 20);
}

// the small maps of one image model (p8_rec.h P8XLayout): the same lanes, a table of their own, op words from the chunk's xops rows
struct P8XLanesDev {
  int nlanes, model;
  P8LaneDev lane[P8_XL_NLANE];
  P8LaneRegs regs[P8_XL_NLANE];   // home between chunks
  const uint8_t* nex; const int16_t* stretch; const int16_t* squash;
};

// one lane, one step. x: the step's 1552-vector; order: the order-N map's return value of this step; lim_off: the first input
// position that is NOT this table's at this step (a step of an image model ends the generic layout at the common prefix: a generic
// lane behind it neither runs nor writes; P8_NX otherwise).
struct P8LaneTabs { const uint8_t* nex; const int16_t* stretch; const uint8_t* ilog; };
P8_HD void p8s_lane_step_t(const P8LaneDev* L, const P8LaneTabs* d, P8LaneRegs* r, uint32_t op, int y, int order, int16_t* x, int lim_off) {
  const int kind = L->q.kind;
  if (L->q.off >= lim_off) return;
  int16_t* o = x + L->q.off;
  if (kind == P8L_DIRECT) { if (op & P8OP_MIX) o[0] = (int16_t)(op & 0xffffu); return; }
  if (kind == P8L_NONE || kind == P8L_RCM) return;   // (RCM: p8s_lane_rcm, which needs the step's partial byte)
  if (!(op & P8OP_MIX)) { for (int j = 0; j < L->q.nout; j++) o[j] = 0; return; }
  if (kind == P8L_SM32) {
    const int p = p8s_sm32(L->c32, &r->sm_cxt, y, (int)(op & P8OP_CTX), L->q.limit);
    o[0] = (op & P8OP_ZERO) ? (int16_t)0 : (int16_t)((d->stretch[p] + (L->q.a ? 0 : 1)) >> 1);   // (a: im4bitModel's stretch(p) >> 1, :4735)
    return;
  }
  if (kind == P8L_PIC) {   // t[old] = nex(t[old], y); stretch(sm.p(t[new]))
    uint8_t* t = L->c8;
    int s = d->nex[4 * t[r->cp] + y];
    if ((op & P8OP_ZERO) && L->q.a == 0) { const int extra = L->q.limit ? L->q.limit : 2; for (int k = 0; k < extra; k++) s = d->nex[4 * s + y]; }   // the model's first call: all its maps sit on cell 0 of their shared array
    t[r->cp] = (uint8_t)s;
    r->cp = op & P8OP_CTX;
    const int st = t[r->cp];
    uint16_t* m = L->sm16;
    m[r->sm_cxt] = (uint16_t)(m[r->sm_cxt] + (((y << 16) - m[r->sm_cxt] + 128) >> 8));
    r->sm_cxt = st;
    o[0] = d->stretch[m[st] >> 4];
    return;
  }
  // the three direct-lookup maps: set() :900-907 / :944-951 / :985-992, then mix()
  if (op & P8OP_SET) {
    r->context = (op & P8OP_ORDER) ? ((uint32_t)order & L->mask) * L->stride : (op & P8OP_CTX);
    r->B = r->bcount = 0;
  }
  int pred;
  if (kind == P8L_SSCM) {
    uint16_t* c = L->c16;
    const int a = L->q.a;
    c[r->cp] = (uint16_t)(c[r->cp] + (((y << 16) - c[r->cp] + (1 << (a - 1))) >> a));
    r->B += (uint32_t)(y && r->B > 0);
    r->cp = r->context + r->B;
    pred = c[r->cp] >> 4;
  } else if (kind == P8L_STAT) {
    uint32_t* c = L->c32;
    const uint32_t v = c[r->cp];
    const int lim = L->q.limit;
    const uint32_t count = (uint32_t)lim < (v & 0x3FF) + 1 ? (uint32_t)lim : (v & 0x3FF) + 1;
    int p = (int)(v >> 10), err = (y << 22) - p;
    err = (int)((uint32_t)(err / 8) * (uint32_t)(16384 / (int)(count + count + 3))) / 1024;   // the product wraps
    p = p + err; p = p < 0 ? 0 : p > 0x3FFFFF ? 0x3FFFFF : p;
    c[r->cp] = ((uint32_t)p << 10) | count;
    r->B += (uint32_t)(y && r->B > 0);
    r->cp = r->context + r->B;
    pred = (int)(c[r->cp] >> 20);
  } else {   // P8L_IND
    uint8_t* c = L->c8;
    c[r->cp] = d->nex[4 * c[r->cp] + y];
    r->B += (uint32_t)(y && r->B > 0);
    r->cp = r->context + r->B;
    pred = p8s_sm32(L->sm, &r->sm_cxt, y, c[r->cp], L->q.limit);
  }
  const int mul = L->q.mul, div = L->q.div;
  o[0] = (int16_t)((d->stretch[pred] * mul) / div);
  o[1] = (int16_t)(((pred - 2048) * mul) / (div * 2));
  r->bcount++; r->B += r->B + 1;
  if (r->bcount == L->q.bits_per_ctx) r->bcount = r->B = 0;
}
// md: the step's model (0: generic); a lane of the generic table runs in a model's step only if the model calls it (P8Lane.modes)
// P8L_HT16 (im4bitModel :4675-4742): HashTable<16>::operator[] :841-856 on the lane's table; c32: [0..13] the contexts' byte offsets, [16..29] their
// StateMap contexts, [31] started; sm16: 14 StateMaps. ops: the lane's op word and the 14 hash words behind it. bpos: the step's bit position.
P8_HD uint32_t p8s_ht16_find(uint8_t* p, uint32_t i, uint8_t chk) {
  enum { B = 16 };
  if (p[i] == chk) return i + 1;
  if (p[i ^ B] == chk) return (i ^ B) + 1;
  if (p[i ^ (B * 2)] == chk) return (i ^ (B * 2)) + 1;
  if (p[i + 1] > p[(i + 1) ^ B] || p[i + 1] > p[(i + 1) ^ (B * 2)]) i ^= B;
  if (p[i + 1] > p[(i + 1) ^ B ^ (B * 2)]) i ^= B ^ (B * 2);
  for (int k = 0; k < B; k++) p[i + k] = 0;
  p[i] = chk;
  return i + 1;
}
P8_HD void p8s_lane_ht16(const P8LaneDev* L, const P8LaneTabs* d, const uint32_t* ops, int y, int bpos, int16_t* x) {
  enum { S = 14 };
  if (!(ops[0] & P8OP_MIX)) return;
  uint32_t* st = L->c32;
  uint8_t* T = L->c8;
  if (!st[31]) { for (int i = 0; i < S; i++) st[i] = 1; st[31] = 1; }   // cp[i] = t[263 * i] + 1: every such key is item 0 with checksum 0
  for (int i = 0; i < S; i++) T[st[i]] = d->nex[4 * T[st[i]] + y];
  if (bpos == 0 || bpos == 4) {
    for (int i = 0; i < S; i++) { const uint32_t wd = ops[1 + i]; st[i] = p8s_ht16_find(T, (wd & 0x3FFFFFu) << 4, (uint8_t)(wd >> 22)); }
  } else {
    const uint32_t j = (uint32_t)(y + 1) << (bpos & 3);
    for (int i = 0; i < S; i++) st[i] += j;
  }
  int16_t* o = x + L->q.off;
  for (int i = 0; i < S; i++) {
    const int s = T[st[i]];
    const int n0 = -!d->nex[4 * s + 2], n1 = -!d->nex[4 * s + 3];
    uint16_t* m = L->sm16 + 256 * i;
    uint32_t* cx = &st[16 + i];
    m[*cx] = (uint16_t)(m[*cx] + (((y << 16) - m[*cx] + 128) >> 8));
    *cx = (uint32_t)s;
    const int p1 = m[s] >> 4;
    const int sv = d->stretch[p1] >> 1;
    const int dn = n1 - n0;
    o[3 * i] = (int16_t)sv;
    o[3 * i + 1] = (int16_t)((p1 - 2047) >> 2);
    o[3 * i + 2] = (int16_t)(sv * (dn < 0 ? -dn : dn));
  }
}
// ---- P8L_JPG: jpegModel's learning half (:6482-6596) on one lane --------------------------------------------------------------------------
// paq8's Mixer arithmetic in scalar form (dot_product :407-413 / :478-484, train :419-430 / :486-494; the wave-parallel form is p8stage.hip's)
P8_HD int p8s_sat16(int v) { return v > 32767 ? 32767 : v < -32768 ? -32768 : v; }
P8_HD int p8s_dot16(const int16_t* t, const int16_t* w, int n) {
  uint32_t sum = 0;
  for (int i = 0; i + 1 < n; i += 2) {
    const uint32_t pair = (uint32_t)((int32_t)t[i] * w[i]) + (uint32_t)((int32_t)t[i + 1] * w[i + 1]);
    sum += (uint32_t)((int32_t)pair >> 8);
  }
  return (int32_t)sum;
}
P8_HD void p8s_train16(const int16_t* t, int16_t* w, int n, int e) {
  if (!e) return;
  const int16_t err = (int16_t)e;
  for (int i = 0; i < n; ++i) {
    int v = p8s_sat16(2 * (int)t[i]);
    v = (v * (int)err) >> 16;
    v = p8s_sat16(v + 1) >> 1;
    w[i] = (int16_t)p8s_sat16(v + (int)w[i]);
  }
}
struct P8JpgDev {
  uint8_t* t; uint32_t mask;            // BH<9> t(MEM()): items of 9 bytes (u16 checksum, 7 bit histories), 8-item neighbourhoods
  uint32_t cp[32]; int smc[32]; int started;
  uint16_t* sm;                         // [32][256] StateMaps
  int16_t* w1;                          // m1's weights [2050][40] (Mixer m1(N + 1, 2050, 3): rows start at 0), created lazily in the reference = all zero here
  int16_t w2[8];                        // its second layer's one row (0x7fff)
  int16_t tx[40], tx2[8];               // the inputs of its last p()
  int nx, row[3], pr[3], pr2, live;     // live: a p() has happened (update() has something to learn from)
  uint32_t* a1; uint32_t* a2; int a1c, a2c;   // APM a1(0x8000), a2(0x20000)
};
// BH<B>::operator[] :788-813: the item of checksum chk in the neighbourhood starting at item i, moved to the front; returns its byte offset + 1
template <int B>
P8_HD uint32_t p8s_bh_get(uint8_t* t, uint32_t i, uint16_t chk) {
  enum { M = 8 };
  uint32_t p = 0;
  int j;
  for (j = 0; j < M; ++j) {
    p = (i + (uint32_t)j) * B;
    const uint16_t cur = (uint16_t)(t[p] | (t[p + 1] << 8));
    if (t[p + 2] == 0) { t[p] = (uint8_t)chk; t[p + 1] = (uint8_t)(chk >> 8); break; }
    if (cur == chk) break;
  }
  if (j == 0) return p + 1;
  uint8_t tmp[B];
  if (j == M) {
    --j;
    for (int k = 0; k < B; k++) tmp[k] = 0;
    tmp[0] = (uint8_t)chk; tmp[1] = (uint8_t)(chk >> 8);
    if (t[(i + (uint32_t)j) * B + 2] > t[(i + (uint32_t)j - 1) * B + 2]) --j;
  } else for (int k = 0; k < B; k++) tmp[k] = t[p + k];
  for (int k = j * B - 1; k >= 0; k--) t[(i + 1) * B + (uint32_t)k] = t[i * B + (uint32_t)k];   // memmove(&t[(i + 1) * B], &t[i * B], j * B)
  for (int k = 0; k < B; k++) t[i * B + (uint32_t)k] = tmp[k];
  return i * B + 1;
}
// P8L_RCM: RunContextMap::set (at a byte's first step) + mix :866-888. c0 / bpos: the partial byte and bit position of the step; op2: the next lane's word
P8_HD void p8s_lane_rcm(const P8LaneDev* L, const P8LaneTabs* d, P8LaneRegs* r, uint32_t op, uint32_t op2, int bpos, int c0, int16_t* x) {
  if (!(op & P8OP_MIX)) { x[L->q.off] = 0; return; }
  uint8_t* T = L->c8;
  if (op & P8OP_SET) {
    const int c1 = (int)(op & 0xffu);
    uint8_t* cp = T + r->cp;
    if (cp[0] == 0 || cp[1] != c1) { cp[0] = 1; cp[1] = (uint8_t)c1; }
    else if (cp[0] < 255) ++cp[0];
    r->cp = p8s_bh_get<4>(T, op2 & L->mask, (uint16_t)((op >> 8) & 0xffffu)) + 1;
  }
  const uint8_t* cp = T + r->cp;
  x[L->q.off] = (int16_t)(((cp[1] + 256) >> (8 - bpos)) == c0 ? (((cp[1] >> (7 - bpos)) & 1) * 2 - 1) * d->ilog[cp[0] + 1] * 8 : 0);
}
// one coded step. ops: the worker's op word, then the 64 + 5 raw words; x: the step's vector (inputs at off .. off + 69, exported-only values behind)
P8_HD void p8s_lane_jpg(const P8LaneDev* L, const P8LaneTabs* d, const int16_t* squash, const uint32_t* ops, int y, int16_t* x) {
  enum { N = 32 };
  if (!(ops[0] & P8OP_MIX)) return;
  P8JpgDev* J = L->jpg;
  const int hbcount = (int)(ops[0] & 3u), hcl = (int)((ops[0] >> 2) & 1u);
  uint8_t* T = J->t;
  if (J->started) for (int i = 0; i < N; ++i) T[J->cp[i]] = d->nex[4 * T[J->cp[i]] + y];   // if (cp[N-1]) *cp[i] = nex(*cp[i], y)
  if (J->live) {   // m1.update() :529-541: every selected set learns the bit from its own output
    for (int i = 0; i < 3; ++i) p8s_train16(J->tx, J->w1 + (size_t)J->row[i] * 40, J->nx, ((y << 12) - J->pr[i]) * 7);
  }
  int16_t* o = x + L->q.off;
  int nx = 0;
  J->tx[nx++] = 128;                    // m1.add(128)
  o[70] = 128;
  for (int i = 0; i < N; ++i) {
    if (hbcount == 0) J->cp[i] = p8s_bh_get<9>(T, ops[2 + 2 * i] & J->mask, (uint16_t)ops[1 + 2 * i]) + 1;
    else J->cp[i] += hbcount == 1 ? (uint32_t)(1 + hcl * 3) : (uint32_t)(1 + hcl);
    uint16_t* m = J->sm + 256 * i;
    m[J->smc[i]] = (uint16_t)(m[J->smc[i]] + (((y << 16) - m[J->smc[i]] + 128) >> 8));   // sm[i].p(*cp[i])
    J->smc[i] = T[J->cp[i]];
    const int p = m[J->smc[i]] >> 4;
    const int sp = d->stretch[p];
    o[2 * i] = (int16_t)((p - 2048) >> 2);
    J->tx[nx++] = (int16_t)sp;
    o[2 * i + 1] = (int16_t)sp;
  }
  J->started = 1;
  // m1.p() :553-572: pad, the second layer learns, the three sets, the second layer
  while (nx & 7) J->tx[nx++] = 0;
  J->nx = nx;
  if (J->live) p8s_train16(J->tx2, J->w2, 8, ((y << 12) - J->pr2) * 7);   // mp->update()
  for (int i = 0; i < 8; ++i) J->tx2[i] = 0;
  for (int i = 0; i < 3; ++i) {
    J->row[i] = (int)ops[65 + i];
    const int dp = p8s_dot16(J->tx, J->w1 + (size_t)J->row[i] * 40, nx);
    const int dv = (int32_t)((uint32_t)dp * 9u) >> 9;
    J->pr[i] = dv > 2047 ? 4095 : dv < -2047 ? 0 : squash[dv + 2048];
    J->tx2[i] = d->stretch[J->pr[i]];
    o[71 + i] = J->tx2[i];
  }
  {
    const int z = p8s_dot16(J->tx2, J->w2, 8);
    const int zz = z >> 9;
    J->pr2 = zz > 2047 ? 4095 : zz < -2047 ? 0 : squash[zz + 2048];
  }
  J->live = 1;
  int pr = J->pr2;
  o[64] = d->stretch[pr]; o[65] = (int16_t)(pr - 2048);
  {   // a1.p(pr, ctx, 1023), a2.p(pr, ctx, 1023): APM::p :699-711
    uint32_t* tb[2] = {J->a1, J->a2};
    int* cxs[2] = {&J->a1c, &J->a2c};
    for (int k = 0; k < 2; ++k) {
      uint32_t* t = tb[k];
      int* cxt = cxs[k];
      uint32_t p0 = t[*cxt];
      const int n = p0 & 1023, q = p0 >> 10;
      if (n < 1023) ++p0; else p0 = (p0 & 0xfffffc00u) | 1023u;
      p0 += ((uint32_t)(((y << 22) - q) >> 3) * (uint32_t)(16384 / (n + n + 3))) & 0xfffffc00u;
      t[*cxt] = p0;
      int s = (d->stretch[pr] + 2048) * 23;
      const int wt = s & 0xfff;
      const int cx = (int)ops[68 + k] * 24 + (s >> 12);
      *cxt = cx + (wt >> 11);
      pr = (int)(((t[cx] >> 13) * (uint32_t)(4096 - wt) + (t[cx + 1] >> 13) * (uint32_t)wt) >> 19);
      o[66 + 2 * k] = d->stretch[pr]; o[67 + 2 * k] = (int16_t)(pr - 2048);
    }
  }
}
// P8L_PIC2: the second context's registers live in the fields a bit-history lane does not use (context = its cell, B = its StateMap context)
P8_HD void p8s_lane_pic2(const P8LaneDev* L, const P8LaneTabs* d, P8LaneRegs* r, uint32_t op, uint32_t op2, int y, int16_t* x) {
  if (!(op & P8OP_MIX)) return;
  uint8_t* t = L->c8;
  uint16_t* m1 = L->sm16; uint16_t* m2 = L->sm16 + 256;
  t[r->cp] = d->nex[4 * t[r->cp] + y];
  t[r->context] = d->nex[4 * t[r->context] + y];
  r->cp = op & P8OP_CTX; r->context = op2 & P8OP_CTX;
  const int s1 = t[r->cp], s2 = t[r->context];
  m1[r->sm_cxt] = (uint16_t)(m1[r->sm_cxt] + (((y << 16) - m1[r->sm_cxt] + 128) >> 8));
  r->sm_cxt = s1;
  m2[r->B] = (uint16_t)(m2[r->B] + (((y << 16) - m2[r->B] + 128) >> 8));
  r->B = (uint32_t)s2;
  int16_t* o = x + L->q.off;
  o[0] = d->stretch[m1[s1] >> 4];
  o[1] = d->stretch[m2[s2] >> 4];
}
P8_HD void p8s_lane_step(const P8LanesDev* d, P8LaneRegs* r, int l, uint32_t op, int y, int order, int16_t* x, int md = 0) {
  const P8LaneTabs tb = {d->nex, d->stretch, d->ilog};
  if (md && !((d->lane[l].q.modes >> md) & 1u)) return;
  p8s_lane_step_t(&d->lane[l], &tb, r, op, y, order, x, P8_NX);
}
// a lane of the GENERIC table at one step: ops = its op word (the next lane's follows), c0 / bpos = the step's partial byte / bit position
P8_HD void p8s_glane_step(const P8LanesDev* d, P8LaneRegs* r, int l, const uint32_t* ops, int y, int order, int bpos, int c0, int16_t* x, int md) {
  if (d->lane[l].q.kind == P8L_RCM) {
    if (md && !((d->lane[l].q.modes >> md) & 1u)) return;
    const P8LaneTabs tb = {d->nex, d->stretch, d->ilog};
    p8s_lane_rcm(&d->lane[l], &tb, r, ops[0], ops[1], bpos, c0, x);
  } else p8s_lane_step(d, r, l, ops[0], y, order, x, md);
}

// ---------------------------------------------------------------- tail of Predictor::update
struct P8TailDev {
  uint32_t* apm[4]; int apm_cxt[4];      // TEXT: four APM (StateMap32 of 0x10000 * 24 cells) :691-712
  uint16_t* apm1[3]; int apm1_idx[3];    // TEXT: three APM1 (0x10000 * 33 cells) :600-621
  uint16_t* gen[7]; int gen_idx[7];      // other blocks: seven APM1 (0x2000 and 6 x 0x10000 contexts)
  uint32_t* col_apm[4]; int col_cxt[4];  // IMAGE24 / IMAGE32 (Image.Color :8222-8225): four APM (0x1000, 3 x 0x10000 contexts) ...
  uint16_t* col_apm1[2]; int col_idx[2]; // ... and two APM1 (0x10000)
  uint32_t* pal_apm[4]; int pal_cxt[4];  // IMAGE8 (Image.Palette :8226-8229): the same shapes
  uint16_t* pal_apm1[2]; int pal_idx[2];
  uint32_t* gray_apm[3]; int gray_cxt[3];   // IMAGE8GRAY (Image.Gray :8230-8232): three APM (0x1000, 2 x 0x10000)
  uint64_t misses;
  int pr;                                 // the last final prediction (12 bits)
  const int16_t* stretch; const int16_t* squash;   // squash: index d + 2048
  float out[P8_NOUT];                     // PAQ8::Predict()'s vector, as the reference keeps it between bits
};

P8_HD int p8s_squash(const int16_t* t, int d) { return d > 2047 ? 4095 : d < -2047 ? 0 : t[d + 2048]; }
P8_HD int p8s_apm(uint32_t* t, int* cxt, const int16_t* stretch, int y, int pr, int cx, int limit) {   // APM::p :699-711
  uint32_t p0 = t[*cxt];
  const int n = p0 & 1023, q = p0 >> 10;
  if (n < limit) ++p0; else p0 = (p0 & 0xfffffc00u) | (uint32_t)limit;
  p0 += ((uint32_t)(((y << 22) - q) >> 3) * (uint32_t)(16384 / (n + n + 3))) & 0xfffffc00u;
  t[*cxt] = p0;
  pr = (stretch[pr] + 2048) * 23;
  const int wt = pr & 0xfff;
  cx = cx * 24 + (pr >> 12);
  *cxt = cx + (wt >> 11);
  return (int)(((t[cx] >> 13) * (uint32_t)(4096 - wt) + (t[cx + 1] >> 13) * (uint32_t)wt) >> 19);
}
P8_HD int p8s_apm1(uint16_t* t, int* index, const int16_t* stretch, int y, int pr, int cxt, int rate) {   // APM1::pp :609-620
  pr = stretch[pr];
  const int g = (y << 16) + (y << rate) - y - y;
  t[*index] = (uint16_t)(t[*index] + ((g - t[*index]) >> rate));
  t[*index + 1] = (uint16_t)(t[*index + 1] + ((g - t[*index + 1]) >> rate));
  const int w = pr & 127;
  *index = ((pr + 2048) >> 7) + cxt * 33;
  return (t[*index] * (128 - w) + t[*index + 1] * w) >> 11;
}
// selectors: the device terms (p8_rec.h)
P8_HD int p8s_sel(int i, int host, int order, int last_pr) {
  const int o3 = order > 3 ? order - 3 : 0, o5 = order > 5 ? order - 5 : 0;
  if (i == P8_SEL_ORDER3) return host + (o3 << 3);
  if (i == P8_SEL_ORDER5_A || i == P8_SEL_ORDER5_B || i == P8_SEL_ORDER5_C) return host + o5 * 256;
  if (i == P8_SEL_LASTPR) return host + last_pr / 16;
  return host;
}
// the chain after the mixer (:8281-8358) in three lane-parallel phases; y = the bit coded before this step, pr0 = the
// mixer's output. res[0..3]: first group's outputs, res[4..6]: second group's. Each lane owns one table.
//   phase A  lanes 0..3:  TEXT: the four APMs on pr0           other: gen[0..3] on pr0
//   phase B  lanes 0..2:  TEXT: APM1[0] on the average, APM1[1,2] on res[0]     other: gen[4..6] on res[0]
//   phase C  one lane:    the exported values and the final prediction
P8_HD void p8s_tail_a(P8TailDev* d, const P8ApmRec* a, int y, int pr0, int j, int* res) {
  const int16_t* st = d->stretch;
  if (a->text == P8_APM_TEXT) {
    const int cx = j == 0 ? (a->c[0] | (int)((d->misses & 0xF) << 4)) : j == 1 ? a->c[1 + (int)(d->misses & 3)] : a->c[3 + j];
    res[j] = p8s_apm(d->apm[j], &d->apm_cxt[j], st, y, pr0, cx, a->limit);
  } else {
    const int cx = j == 0 ? (a->c[0] | (int)(d->misses & 7)) : a->c[j];
    res[j] = p8s_apm1(d->gen[j], &d->gen_idx[j], st, y, pr0, cx, 7);
  }
}
P8_HD void p8s_tail_b(P8TailDev* d, const P8ApmRec* a, int y, int pr0, int j, int* res) {
  const int16_t* st = d->stretch;
  const int avg = (pr0 + res[1] + res[2] + res[3] + 2) >> 2;
  if (a->text == P8_APM_TEXT) res[4 + j] = p8s_apm1(d->apm1[j], &d->apm1_idx[j], st, y, j == 0 ? avg : res[0], a->c[7 + j], j == 0 ? 7 : 6);
  else res[4 + j] = p8s_apm1(d->gen[4 + j], &d->gen_idx[4 + j], st, y, res[0], j == 0 ? a->c[4] : a->c[1 + j], 7);
}
// writes the 10 or 11 exported stage values at o[] and returns the final prediction
P8_HD int p8s_tail_c(const P8ApmRec* a, int pr0, const int* res, float* o) {
  const float cf = (float)(1.0 / 4095);
  int k = 0;
  o[k++] = (float)pr0 * cf;
  const int avg = (pr0 + res[1] + res[2] + res[3] + 2) >> 2;
  for (int j = 0; j < 4; j++) o[k++] = (float)res[j] * cf;
  if (a->text == P8_APM_TEXT) o[k++] = (float)avg * cf;   // the general path does not export this one (:8345)
  for (int j = 4; j < 7; j++) o[k++] = (float)res[j] * cf;
  int pr = (res[0] + res[4] + res[5] + res[6] + 2) >> 2;
  o[k++] = (float)pr * cf;
  pr = (pr + avg + 1) >> 1;
  o[k++] = (float)pr * cf;
  return pr;
}
// Image.Color (:8299-8314), the chain behind the mixer for IMAGE24 / IMAGE32 steps: writes the 10 exported values at o[] and returns
// the final prediction. A serial chain (one lane): image steps are rare and their mixer is a plain one (p8stage.hip).
P8_HD int p8s_tail_color(P8TailDev* d, const P8ApmRec* a, int y, int pr0, float* o) {
  const int16_t* st = d->stretch;
  const float cf = (float)(1.0 / 4095);
  const int lim = a->limit;
  int pr = p8s_apm(d->col_apm[0], &d->col_cxt[0], st, y, pr0, a->c[0] | (int)(d->misses & 0xF), lim);
  int pr1 = p8s_apm(d->col_apm[1], &d->col_cxt[1], st, y, pr0, a->c[1], lim);
  int pr2 = p8s_apm(d->col_apm[2], &d->col_cxt[2], st, y, pr0, a->c[2], lim);
  const int pr3 = p8s_apm(d->col_apm[3], &d->col_cxt[3], st, y, pr0, a->c[3], lim);
  o[0] = (float)pr0 * cf; o[1] = (float)pr * cf; o[2] = (float)pr1 * cf; o[3] = (float)pr2 * cf; o[4] = (float)pr3 * cf;
  const int avg = (pr0 + pr1 + pr2 + pr3 + 2) >> 2;
  o[5] = (float)avg * cf;
  pr1 = p8s_apm1(d->col_apm1[0], &d->col_idx[0], st, y, pr, a->c[4], 7);
  pr2 = p8s_apm1(d->col_apm1[1], &d->col_idx[1], st, y, pr, a->c[5], 7);
  o[6] = (float)pr1 * cf; o[7] = (float)pr2 * cf;
  pr = (pr * 2 + pr1 * 3 + pr2 * 3 + 4) >> 3;
  o[8] = (float)pr * cf;
  pr = (pr + avg + 1) >> 1;
  o[9] = (float)pr * cf;
  return pr;
}
// Image.Palette (:8325-8340): 10 exported values
P8_HD int p8s_tail_palette(P8TailDev* d, const P8ApmRec* a, int y, int pr0, float* o) {
  const int16_t* st = d->stretch;
  const float cf = (float)(1.0 / 4095);
  const int lim = a->limit;
  int pr = p8s_apm(d->pal_apm[0], &d->pal_cxt[0], st, y, pr0, a->c[0] | (int)(d->misses & 0xF), lim);
  int pr1 = p8s_apm(d->pal_apm[1], &d->pal_cxt[1], st, y, pr0, a->c[1], lim);
  int pr2 = p8s_apm(d->pal_apm[2], &d->pal_cxt[2], st, y, pr0, a->c[2], lim);
  const int pr3 = p8s_apm(d->pal_apm[3], &d->pal_cxt[3], st, y, pr0, a->c[3], lim);
  o[0] = (float)pr0 * cf; o[1] = (float)pr * cf; o[2] = (float)pr1 * cf; o[3] = (float)pr2 * cf; o[4] = (float)pr3 * cf;
  const int avg = (pr0 + pr1 + pr2 + pr3 + 2) >> 2;
  o[5] = (float)avg * cf;
  pr1 = p8s_apm1(d->pal_apm1[0], &d->pal_idx[0], st, y, avg, a->c[4], 5);
  pr2 = p8s_apm1(d->pal_apm1[1], &d->pal_idx[1], st, y, pr, a->c[5], 6);
  o[6] = (float)pr1 * cf; o[7] = (float)pr2 * cf;
  pr = (pr * 2 + pr1 + pr2 + 2) >> 2;
  o[8] = (float)pr * cf;
  pr = (pr + avg + 1) >> 1;
  o[9] = (float)pr * cf;
  return pr;
}
// Image.Gray (:8315-8324): 6 exported values; the second APM refines the first one's output
P8_HD int p8s_tail_gray(P8TailDev* d, const P8ApmRec* a, int y, int pr0, float* o) {
  const int16_t* st = d->stretch;
  const float cf = (float)(1.0 / 4095);
  const int lim = a->limit;
  int pr = p8s_apm(d->gray_apm[0], &d->gray_cxt[0], st, y, pr0, a->c[0] | (int)(d->misses & 0xF), lim);
  const int pr1 = p8s_apm(d->gray_apm[1], &d->gray_cxt[1], st, y, pr, a->c[1], lim);
  const int pr2 = p8s_apm(d->gray_apm[2], &d->gray_cxt[2], st, y, pr0, a->c[2], lim);
  o[0] = (float)pr0 * cf; o[1] = (float)pr * cf; o[2] = (float)pr1 * cf; o[3] = (float)pr2 * cf;
  const int avg = (2 * pr0 + pr1 + pr2 + 2) >> 2;
  o[4] = (float)avg * cf;
  pr = (pr + avg + 1) >> 1;
  o[5] = (float)pr * cf;
  return pr;
}
// the chain of an image model's step by its kind (P8ApmRec.text)
P8_HD int p8s_tail_image(P8TailDev* d, const P8ApmRec* a, int y, int pr0, float* o) {
  if (a->text == P8_APM_GENERIC || a->text == P8_APM_TEXT) {   // a model's step inside an ordinary block, or inside a TEXT block (the text chain, :8281-8296): the chain of the block's type, one table after the other
    int res[8];
    for (int j = 0; j < 4; j++) p8s_tail_a(d, a, y, pr0, j, res);
    for (int j = 0; j < 3; j++) p8s_tail_b(d, a, y, pr0, j, res);
    return p8s_tail_c(a, pr0, res, o);
  }
  return a->text == P8_APM_COLOR ? p8s_tail_color(d, a, y, pr0, o) : a->text == P8_APM_GRAY ? p8s_tail_gray(d, a, y, pr0, o) : p8s_tail_palette(d, a, y, pr0, o);
}
#endif
