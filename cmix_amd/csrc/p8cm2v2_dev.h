// p8cm2v2_dev.h -- ContextMap2 (reference src/models/paq8.cpp:1164-1358) for the paq8 stage, second design, on top of the
// first (p8cm2_dev.h, kept as the stand-alone block and as the exact serial fallback): one lane per context, the 7
// state bytes of the context's current bucket slot and its 4 byte-history bytes cached in LDS with write-through, the
// bucket of a lookup bit (positions 0, 2, 5) fetched by all lanes before the one barrier of the bit, overlap detection
// through an LDS hash set at lookup bits only. A bit with an overlap -- or while two contexts sit on one slot -- runs
// on lane 0 with the first design's code on the table itself (exact by construction), then the lanes reload.
// ContextMap2 draws no random numbers, so instances are independent. Single source: tests/host/p8stage_emul.cpp.
#ifndef CMX_P8CM2V2_DEV_H
#define CMX_P8CM2V2_DEV_H
#include <stdint.h>

#include "p8cm2_dev.h"

enum { P8C2_HASH = 1024 };
#ifdef __HIPCC__
#define P8C2_CAS(p, c, v) atomicCAS((p), (c), (v))
#else
static inline uint32_t p8c2_cas_host(uint32_t* p, uint32_t c, uint32_t v) { const uint32_t o = *p; if (o == c) *p = v; return o; }
#define P8C2_CAS(p, c, v) p8c2_cas_host((p), (c), (v))
#endif

struct P8Cm2V2Shared {
  P8Cm2Shared base;                 // regs (bit_state, bit_state0, byte_hist, StateMap contexts, has_history), nz[]
  uint8_t slot[P8CM2_MAXC][8];      // cached: the 7 bytes at bit_state0
  uint8_t hist[P8CM2_MAXC][4];      // cached: the 4 bytes at byte_hist
  uint8_t bk[P8CM2_MAXC][64];       // the bucket about to be searched
  uint8_t nex[1024];
  int16_t stretch[4096];
  uint8_t ilog[260];
  uint32_t hash[2][P8C2_HASH];
  uint32_t conf[2];                 // an overlap in this lookup bit
  uint32_t shared;                  // two contexts on one slot (lasts until they look up again)
};
struct P8Cm2Tmp { int look; uint32_t nb; };

P8_HD void p8c2_load(const P8Cm2Dev* d, P8Cm2V2Shared* sh, int tid, int nthreads) {
  for (int i = tid; i < 1024; i += nthreads) sh->nex[i] = d->nex[i];
  for (int i = tid; i < 4096; i += nthreads) sh->stretch[i] = d->stretch[i];
  for (int i = tid; i < 257; i += nthreads) sh->ilog[i] = d->ilog[i];
  { uint32_t* p = &sh->hash[0][0]; for (int i = tid; i < 2 * P8C2_HASH; i += nthreads) p[i] = 0; }
  if (tid == 0) { sh->base.r = d->regs; sh->conf[0] = sh->conf[1] = 0; sh->shared = 0; }
}
P8_HD void p8c2_reload(const P8Cm2Dev* d, P8Cm2V2Shared* sh, int i) {
  const uint8_t* T = d->table;
  const P8Cm2Regs* r = &sh->base.r;
  for (int k = 0; k < 7; k++) sh->slot[i][k] = T[r->bit_state0[i] + k];
  for (int k = 0; k < 4; k++) sh->hist[i][k] = T[r->byte_hist[i] + k];
}
P8_HD void p8c2_insert(P8Cm2V2Shared* sh, int lk, uint32_t bucket) {
  const uint32_t key = bucket + 1;
  uint32_t h = (key * 2654435761u) >> 22;
  uint32_t* tab = sh->hash[lk & 1];
  for (;;) {
    const uint32_t old = P8C2_CAS(&tab[h], 0u, key);
    if (old == 0) return;
    if (old == key) { sh->conf[lk & 1] = 1; return; }
    h = (h + 1) & (P8C2_HASH - 1);
  }
}
// phase 1 of a lookup bit: touched buckets into the hash set, the bucket about to be searched into LDS
P8_HD void p8c2_phase1(const P8Cm2Dev* d, P8Cm2V2Shared* sh, const P8Cm2Bit& u, int lk, int i, P8Cm2Tmp* t) {
  const P8Cm2Regs* r = &sh->base.r;
  t->look = 0; t->nb = 0;
  uint32_t L[5]; int n = 0;
  if (r->bit_state[i] != P8_NIL) L[n++] = r->bit_state[i] >> 6;
  L[n++] = r->byte_hist[i] >> 6;
  if (!(u.bpos > 1 && sh->hist[i][0] == 0)) {
    t->look = 1;
    t->nb = (u.ctx[i] + u.bits) & d->mask;
    L[n++] = t->nb;
    const uint8_t* g = d->table + (size_t)t->nb * 64;
    uint8_t* b = sh->bk[i];
#ifdef __HIPCC__
    const uint4* g4 = reinterpret_cast<const uint4*>(g);
    uint4* b4 = reinterpret_cast<uint4*>(b);
    const uint4 v0 = g4[0], v1 = g4[1], v2 = g4[2], v3 = g4[3];
    b4[0] = v0; b4[1] = v1; b4[2] = v2; b4[3] = v3;
#else
    for (int j = 0; j < 64; j++) b[j] = g[j];
#endif
    if (u.bpos == 0) {
      const uint16_t* cs = (const uint16_t*)b;
      const int mru = b[P8_B_MRU];
      int slot = -1;
      if (cs[mru & 15] == u.chk[i]) slot = mru & 15;
      else for (int j = 0; j < 7; ++j) if (cs[j] == u.chk[i]) { slot = j; break; }
      if (slot >= 0 && b[P8_B_STATE + 7 * slot + 3] == 2) {
        const int cc = b[P8_B_STATE + 7 * slot + 4] + 256;
        L[n++] = (u.ctx[i] + (uint32_t)(cc >> 6)) & d->mask;
        L[n++] = (u.ctx[i] + (uint32_t)(cc >> 3)) & d->mask;
      }
    }
  }
  for (int a = 0; a < n; a++) {
    int dup = 0;
    for (int c = 0; c < a; c++) dup |= L[c] == L[a];
    if (!dup) p8c2_insert(sh, lk, L[a]);
  }
}
P8_HD void p8c2_clear_next(P8Cm2V2Shared* sh, int lk, int tid, int nthreads) {
  uint32_t* tab = sh->hash[(lk + 1) & 1];
  for (int i = tid; i < P8C2_HASH; i += nthreads) tab[i] = 0;
  if (tid == 0) sh->conf[(lk + 1) & 1] = 0;
}
P8_HD int p8c2_find_staged(uint8_t* T, uint32_t nb, uint8_t* b, uint16_t checksum) {   // Bucket::Find on the staged copy; header changes go to the table
  uint16_t* cs = (uint16_t*)b;
  uint8_t* g = T + (size_t)nb * 64;
  const int mru = b[P8_B_MRU];
  if (cs[mru & 15] == checksum) return mru & 15;
  int worst = 0xFFFF, index = 0;
  for (int i = 0; i < 7; ++i) {
    if (cs[i] == checksum) { b[P8_B_MRU] = (uint8_t)(mru << 4 | i); g[P8_B_MRU] = b[P8_B_MRU]; return i; }
    if (b[P8_B_STATE + 7 * i] < worst && (mru & 15) != i && mru >> 4 != i) { worst = b[P8_B_STATE + 7 * i]; index = i; }
  }
  b[P8_B_MRU] = (uint8_t)(0xF0 | index); g[P8_B_MRU] = b[P8_B_MRU];
  cs[index] = checksum; ((uint16_t*)g)[index] = checksum;
  for (int k = 0; k < 7; k++) { b[P8_B_STATE + 7 * index + k] = 0; g[P8_B_STATE + 7 * index + k] = 0; }
  return index;
}

// lane-parallel: ContextMap2::Update + mix for context i on the cached bytes (:1209-1266, :1321-1357)
P8_HD void p8c2_run(P8Cm2Dev* d, P8Cm2V2Shared* sh, const P8Cm2Bit& u, int i, const P8Cm2Tmp* t) {
  uint8_t* T = d->table;
  P8Cm2Regs* r = &sh->base.r;
  uint8_t* sl = sh->slot[i];
  uint8_t* hs = sh->hist[i];
  const int bp = u.bpos;
  // ---- Update ----
  if (r->bit_state[i] != P8_NIL) {
    const uint32_t o = r->bit_state[i] - r->bit_state0[i];
    const uint8_t ns = sh->nex[4 * sl[o] + u.y];
    sl[o] = ns; T[r->bit_state[i]] = ns;
    if (r->bit_state[i] - r->byte_hist[i] < 4u) hs[r->bit_state[i] - r->byte_hist[i]] = ns;
  }
  if (bp > 1 && hs[0] == 0) r->bit_state[i] = P8_NIL;
  else if (bp == 1 || bp == 3 || bp == 6) r->bit_state[i] = r->bit_state0[i] + 1 + (uint32_t)u.y;
  else if (bp == 4 || bp == 7) r->bit_state[i] = r->bit_state0[i] + 3 + (u.bits & 3);
  else {
    const uint16_t chk = u.chk[i];
    const uint32_t ctx = u.ctx[i], nb = t->nb;
    uint8_t* b = sh->bk[i];
    if (r->bit_state[i] != P8_NIL && (r->bit_state[i] >> 6) == nb) b[r->bit_state[i] & 63] = sl[r->bit_state[i] - r->bit_state0[i]];   // own store above
    const int idx = p8c2_find_staged(T, nb, b, chk);
    const uint32_t ns0 = nb * 64 + P8_B_STATE + 7 * (uint32_t)idx;
    const uint32_t old_bh = r->byte_hist[i];
    r->bit_state[i] = r->bit_state0[i] = ns0;
    for (int k = 0; k < 7; k++) sl[k] = b[P8_B_STATE + 7 * idx + k];
    if (bp == 0) {
      int refresh = 0;
      if (sl[3] == 2) {   // pending bit histories for bits 2-7
        const int cc = sl[4] + 256;
        uint32_t p = p8d_bucket_find(T, (ctx + (uint32_t)(cc >> 6)) & d->mask, chk);
        T[p] = (uint8_t)(1 + ((cc >> 5) & 1));
        T[p + 1 + ((cc >> 5) & 1)] = (uint8_t)(1 + ((cc >> 4) & 1));
        T[p + 3 + ((cc >> 4) & 3)] = (uint8_t)(1 + ((cc >> 3) & 1));
        p = p8d_bucket_find(T, (ctx + (uint32_t)(cc >> 3)) & d->mask, chk);
        T[p] = (uint8_t)(1 + ((cc >> 2) & 1));
        T[p + 1 + ((cc >> 2) & 1)] = (uint8_t)(1 + ((cc >> 1) & 1));
        T[p + 3 + ((cc >> 1) & 3)] = (uint8_t)(1 + (cc & 1));
        T[ns0 + 6] = 0; sl[6] = 0;
        refresh = 1;
      }
      // byte history of the PREVIOUS context
      uint8_t h0 = hs[0], h1 = hs[1], h2 = hs[2], h3;
      if ((old_bh >> 6) == nb) { h0 = b[old_bh & 63]; h1 = b[(old_bh + 1) & 63]; h2 = b[(old_bh + 2) & 63]; }   // the search may have replaced the slot that holds them
      if (refresh) { h0 = T[old_bh]; h1 = T[old_bh + 1]; h2 = T[old_bh + 2]; }
      h3 = h2; h2 = h1;
      if (h0 == 0) { h0 = 2; h1 = u.last_byte; }
      else if (h1 != u.last_byte) { h0 = 1; h1 = u.last_byte; }
      else if (h0 < 254) h0 = (uint8_t)(h0 + 2);
      else if (h0 == 255) h0 = 128;
      T[old_bh] = h0; T[old_bh + 1] = h1; T[old_bh + 2] = h2; T[old_bh + 3] = h3;
      const uint8_t hv[4] = {h0, h1, h2, h3};
      for (int k = 0; k < 4; k++) if (old_bh + (uint32_t)k - ns0 < 7u) sl[old_bh + (uint32_t)k - ns0] = hv[k];   // the same context again
      if (refresh) for (int k = 0; k < 7; k++) sl[k] = T[ns0 + k];
      r->byte_hist[i] = ns0 + 3;
      for (int k = 0; k < 4; k++) hs[k] = sl[3 + k];
      r->has_history[i] = sl[0] > 15;
    }
  }
  // ---- mix ----
  int16_t* o = u.out + d->out_off + 7 * i;
  int state = r->bit_state[i] != P8_NIL ? sl[r->bit_state[i] - r->bit_state0[i]] : 0;
  sh->base.nz[i] = (uint8_t)(state > 0);
  int p1 = p8d_sm32(d->m8 + (size_t)i * P8_M8, &r->m8_cxt[i], u.y, state);
  int n0 = sh->nex[4 * state + 2], n1 = sh->nex[4 * state + 3], k = n1 + 1;
  k = (k * 64) / (k + n0 + 1);
  n0 = -!n0; n1 = -!n1;
  int v = 0;
  if ((uint32_t)((hs[1] + 256) >> (8 - bp)) == u.bits) {
    const int run = hs[0];
    const int sign = ((hs[1] >> (7 - bp)) & 1) * 2 - 1;
    v = sign * (sh->ilog[run + 1] << (3 - (run & 1)));
  } else if (bp > 0 && (hs[0] & 1) > 0) {
    if ((uint32_t)((hs[2] + 256) >> (8 - bp)) == u.bits) v = (((hs[2] >> (7 - bp)) & 1) * 2 - 1) * 128;
    else if (r->has_history[i] && (uint32_t)((hs[3] + 256) >> (8 - bp)) == u.bits) v = (((hs[3] >> (7 - bp)) & 1) * 2 - 1) * 128;
  }
  o[0] = (int16_t)v;
  if (r->has_history[i]) {
    state = (hs[1] >> (7 - bp)) & 1;
    state |= ((hs[2] >> (7 - bp)) & 1) * 2;
    state |= ((hs[3] >> (7 - bp)) & 1) * 4;
  } else state = 8;
  const int st = sh->stretch[p1] >> 2;
  o[1] = (int16_t)st;
  o[2] = (int16_t)((p1 - 2047) >> 3);
  p1 >>= 4;
  const int p0 = 255 - p1;
  const int dn = n1 - n0;
  o[3] = (int16_t)(st * (dn < 0 ? -dn : dn));
  o[4] = (int16_t)((p1 & n0) - (p0 & n1));
  o[5] = (int16_t)(sh->stretch[p8d_sm32(d->m12 + (size_t)i * P8_M12, &r->m12_cxt[i], u.y, (state << 9) | (bp << 6) | k)] >> 2);
  o[6] = (int16_t)(sh->stretch[p8d_sm32(d->m6 + (size_t)i * P8_M6, &r->m6_cxt[i], u.y, (state << 3) | bp)] >> 2);
}

// an overlap, or shared slots: lane 0 runs the first design's code on the table (exact), in the reference's order
P8_HD void p8c2_walk(P8Cm2Dev* d, P8Cm2V2Shared* sh, const P8Cm2Bit& u, int look) {
  for (int j = 0; j < d->C; j++) p8d_update(d, &sh->base, u, j);
  for (int j = 0; j < d->C; j++) p8d_mix(d, &sh->base, u, j);
  if (look) {   // slots change hands at lookup bits only
    const P8Cm2Regs* r = &sh->base.r;
    int share = 0;
    for (int a = 0; a < d->C && !share; a++) {
      const uint32_t cur_a = r->bit_state[a] != P8_NIL ? r->bit_state0[a] : 0xFFFFFFF0u, run_a = r->byte_hist[a] - 3;
      for (int b = a + 1; b < d->C; b++) {
        const uint32_t cur_b = r->bit_state[b] != P8_NIL ? r->bit_state0[b] : 0xFFFFFFF1u, run_b = r->byte_hist[b] - 3;
        if (cur_a == cur_b || cur_a == run_b || run_a == cur_b) { share = 1; break; }
      }
    }
    sh->shared = (uint32_t)share;
  }
}
#endif
