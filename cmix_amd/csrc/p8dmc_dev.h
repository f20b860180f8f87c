// p8dmc_dev.h -- paq8's dynamic Markov coding forest (reference src/models/paq8.cpp:7637-7822: ten dmcModel state graphs
// of 12-byte nodes grown by cloning, fixed-point counts, a bit-history nibble pair per node read through a
// StateMap32(256); dmcForest mixes them into 6 inputs and resets eight of them when full). Inputs: the coded bits only,
// so this block needs no front end. One workgroup of 64: step 1 = one lane per model (update + st(): a handful of
// dependent node accesses each, the ten chains side by side instead of one after the other), step 2 = lane 0 combines
// the six inputs and flags models due for a reset, step 3 = all lanes rebuild a flagged model's 256 byte trees.
// Single source: tests/host/p8dmc_emul.cpp runs it on the host against the oracle.
#ifndef CMX_P8DMC_DEV_H
#define CMX_P8DMC_DEV_H
#include <stdint.h>

#include "p8cm2_dev.h"   // P8_HD

enum { P8DMC_N = 10, P8DMC_THREADS = 64, P8DMC_BASE = 255 * 256 };
struct P8DmcNode { uint16_t c0, c1; uint32_t nx0, nx1; };   // upper 28 bits: successor; lower 4 + 4: bit-history state
struct P8DmcModel { P8DmcNode* t; uint32_t size, top, curr, threshold, threshold_fine, extra, th_start; uint32_t* sm; int sm_cxt; };
struct P8DmcDev { P8DmcModel m[P8DMC_N]; const uint8_t* nex; const int16_t* stretch; int last_y; uint32_t bits_done; };   // last coded bit, bits so far (chunks need not be whole bytes)
struct P8DmcShared { int st[P8DMC_N]; int reset[P8DMC_N]; };

P8_HD uint8_t p8d_nd_state(const P8DmcNode* n) { return (uint8_t)(((n->nx0 & 0xf) << 4) | (n->nx1 & 0xf)); }
P8_HD void p8d_nd_set_state(P8DmcNode* n, uint8_t s) { n->nx0 = (n->nx0 & 0xfffffff0u) | (uint32_t)(s >> 4); n->nx1 = (n->nx1 & 0xfffffff0u) | (uint32_t)(s & 0xf); }
P8_HD uint32_t p8d_inc_counter(uint32_t x, uint32_t inc) { return (((x << 6) - x) >> 6) + (inc << 10); }
// node q of a freshly reset graph (resetstategraph :7664-7684): 256 byte trees of 255 nodes in heap order
P8_HD P8DmcNode p8d_dmc_fresh(uint32_t q, uint32_t th_start) {
  const uint32_t i = q % 255;
  P8DmcNode n;
  if (i < 127) { n.nx0 = (q + i + 1) << 4; n.nx1 = (q + i + 2) << 4; }
  else { const uint32_t root = (i - 127) * 2 * 255; n.nx0 = root << 4; n.nx1 = (root + 255) << 4; }
  n.c0 = n.c1 = th_start < 1024 ? 2048 : 512;
  return n;
}
P8_HD void p8d_dmc_model(P8DmcDev* d, P8DmcShared* sh, int k, int y) {   // dmcModel::update + st() :7687-7751
  P8DmcModel* M = &d->m[k];
  P8DmcNode* t = M->t;
  P8DmcNode* cur = &t[M->curr];
  uint32_t c0 = cur->c0, c1 = cur->c1;
  const uint32_t n = y == 0 ? c0 : c1;
  cur->c0 = (uint16_t)p8d_inc_counter(c0, (uint32_t)(1 - y));
  cur->c1 = (uint16_t)p8d_inc_counter(c1, (uint32_t)y);
  p8d_nd_set_state(cur, d->nex[4 * p8d_nd_state(cur) + y]);
  if (n > M->threshold) {
    const uint32_t next = y == 0 ? cur->nx0 >> 4 : cur->nx1 >> 4;
    P8DmcNode* nx = &t[next];
    c0 = nx->c0; c1 = nx->c1;
    const uint32_t nn = c0 + c1;
    if (nn > n + M->threshold) {
      if (M->top != M->size) {
        const uint32_t c0_top = (uint32_t)((uint64_t)c0 * n / nn), c1_top = (uint32_t)((uint64_t)c1 * n / nn);
        P8DmcNode* tp = &t[M->top];
        tp->c0 = (uint16_t)c0_top; tp->c1 = (uint16_t)c1_top;
        nx->c0 = (uint16_t)(c0 - c0_top); nx->c1 = (uint16_t)(c1 - c1_top);
        tp->nx0 = (nx->nx0 >> 4) << 4; tp->nx1 = (nx->nx1 >> 4) << 4;
        p8d_nd_set_state(tp, p8d_nd_state(nx));
        if (y == 0) cur->nx0 = (cur->nx0 & 0xf) | (M->top << 4); else cur->nx1 = (cur->nx1 & 0xf) | (M->top << 4);
        ++M->top;
        if (M->threshold < 8 * 1024) M->threshold = (++M->threshold_fine) >> 11;
      } else M->extra += nn >> 10;
    }
  }
  M->curr = y == 0 ? t[M->curr].nx0 >> 4 : t[M->curr].nx1 >> 4;
  const P8DmcNode* c = &t[M->curr];
  const uint32_t n0 = c->c0 + 1u, n1 = c->c1 + 1u;
  const int pr1 = (int)((n1 << 12) / (n0 + n1));
  uint32_t p0 = M->sm[M->sm_cxt];                       // StateMap32::p, limit 256
  const int cnt = p0 & 1023, pr = p0 >> 10;
  if (cnt < 256) ++p0; else p0 = (p0 & 0xfffffc00u) | 256u;
  p0 += (uint32_t)((((y << 22) - pr) >> 3) * (16384 / (cnt + cnt + 3))) & 0xfffffc00u;
  M->sm[M->sm_cxt] = p0;
  M->sm_cxt = p8d_nd_state(c);
  const int pr2 = (int)(M->sm[M->sm_cxt] >> 20);
  sh->st[k] = d->stretch[pr1] + d->stretch[pr2];
}
P8_HD void p8d_dmc_step1(P8DmcDev* d, P8DmcShared* sh, int tid, int y) { if (tid < P8DMC_N) p8d_dmc_model(d, sh, tid, y); }
P8_HD void p8d_dmc_step2(P8DmcDev* d, P8DmcShared* sh, int tid, int bpos, int16_t* out) {   // dmcForest::mix :7796-7815
  if (tid != 0) return;
  out[0] = (int16_t)(sh->st[9] >> 3);
  out[1] = (int16_t)(sh->st[8] >> 3);
  for (int j = 0; j < 4; j++) out[2 + j] = (int16_t)((sh->st[7 - 2 * j] + sh->st[6 - 2 * j]) >> 4);
  for (int k = 0; k < P8DMC_N; k++) sh->reset[k] = bpos == 0 && k < 8 && (d->m[k].extra >> 7) > d->m[k].size;
}
P8_HD void p8d_dmc_step3(P8DmcDev* d, P8DmcShared* sh, int tid) {
  for (int k = 0; k < 8; k++) {
    if (!sh->reset[k]) continue;
    P8DmcModel* M = &d->m[k];
    for (uint32_t q = (uint32_t)tid; q < P8DMC_BASE; q += P8DMC_THREADS) M->t[q] = p8d_dmc_fresh(q, M->th_start);
    if (tid == 0) { M->top = P8DMC_BASE; M->curr = M->extra = 0; M->threshold = M->th_start; M->threshold_fine = M->th_start << 11; }
  }
}
#endif
