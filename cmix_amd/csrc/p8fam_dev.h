// p8fam_dev.h -- the ContextMap family of the paq8 stage, second design (the first, p8cm_dev.h, is kept as the stand-alone
// building block): the same 204 contexts of 16 instances with the shared rnd() stream (reference src/models/paq8.cpp:
// 1010-1145, :152-165), one lane per context, but
//   * the 7 state bytes of a context's current bucket slot, its run bytes and its StateMap live in LDS (160 KB per CU:
//     204 x 512 B of StateMaps + tables + staging); HBM/L2 is touched only at the three bucket lookups per byte
//     (bit positions 0, 2, 5), whose 64-byte buckets are fetched by all lanes at once before a barrier; every
//     modification is also stored through to the table at once (fire and forget), so the table is always current;
//   * overlap detection (two contexts of one instance touching the same bucket in the same bit, the only way contexts
//     interact) runs at lookup bits only, through an LDS hash set with compare-and-swap: O(1) per lane;
//   * the rnd() draws: a bit history reaching state >= 204 decays with a value of the ONE process-wide generator,
//     consumed in the reference's walk order. Lanes publish a draw bit, ranks are prefix popcounts, the values come from
//     an LDS ring that always holds the next 256 values of the sequence;
//   * a bit with an overlap runs in ROUNDS over the instance order: runs of overlap-free instances lane-parallel, an
//     instance with an overlap walked by its first lane on the table itself in the reference's order -- exact whatever
//     the overlap does to replacement decisions or draws -- then its lanes reload their cached bytes.
// One barrier per bit on the common path. Single source: tests/host/p8stage_emul.cpp runs these steps on the host.
#ifndef CMX_P8FAM_DEV_H
#define CMX_P8FAM_DEV_H
#include <stdint.h>
#include <string.h>

#include "p8cm_dev.h"

enum { P8F_HASH = 2048, P8F_RV = 512, P8F_NIL = 0xFF, P8F_LOOK = 256, P8F_KMAX = 16 };
#define P8F_NOKEY 0xFFFFFFFFu

#ifdef __HIPCC__
#define P8F_CAS(p, c, v) atomicCAS((p), (c), (v))
#define P8F_OR(p, v) atomicOr((p), (v))
#define P8F_POPC(x) __popc(x)
#define P8F_INC(p) atomicAdd((p), 1u)
#else
static inline uint32_t p8f_cas_host(uint32_t* p, uint32_t c, uint32_t v) { const uint32_t o = *p; if (o == c) *p = v; return o; }
#define P8F_CAS(p, c, v) p8f_cas_host((p), (c), (v))
#define P8F_OR(p, v) (*(p) |= (v))
#define P8F_POPC(x) __builtin_popcount(x)
static inline uint32_t p8f_inc_host(uint32_t* p) { return (*p)++; }
#define P8F_INC(p) p8f_inc_host((p))
#endif

// per-context state between chunks (HBM); during a chunk the copy in P8FamShared is the live one
struct P8FamHome {
  uint32_t cp0[P8CM_MAXS], runp[P8CM_MAXS];          // byte offsets into the instance's table: slot base, run bytes
  uint8_t cpo[P8CM_MAXS], rc[P8CM_MAXS], rb[P8CM_MAXS], smc[P8CM_MAXS];   // cp - cp0 (P8F_NIL: none), run count / byte, StateMap context
  alignas(8) uint8_t slot[P8CM_MAXS][8];              // the slot's 7 state bytes
};
struct P8FamShared {
  P8FamHome r;
  uint8_t bk[P8CM_MAXS][64];        // the bucket each lane is about to search (fetched before the barrier of a lookup bit)
  uint8_t nex[1024];
  int16_t stretch[4096];
  uint8_t ilog[260];
  uint32_t hash[2][P8F_HASH];       // touched (instance, bucket) pairs of a lookup bit
  uint32_t db[3][8];                // draw bits of the bit, one per slot
  uint8_t conflict[2][P8CM_MAXI + 8];
  uint32_t anyconf[2];
  uint8_t shared[P8CM_MAXI + 8];   // the instance has two contexts on one bucket slot (established by an overlap; lasts until they look up again)
  uint32_t anyshared;
  uint32_t rv[P8F_RV];              // rv[idx & 511] = value number idx of the generator, idx in (i - 64, i + 256]
  uint32_t walk_cnt;
  uint8_t wfull[P8CM_MAXI + 8];     // the instance was walked whole this bit: its lanes reload
  // a lookup bit's overlap, narrowed down (p8f_miniwalk): the contexts whose keys another context of the instance also holds
  uint8_t ink[P8CM_MAXS];           // 1: such a context, 2: already walked this bit
  uint32_t kcount;                  // how many have registered their keys
  uint32_t kkeys[P8F_KMAX][5];
  uint8_t kslot[P8F_KMAX], knk[P8F_KMAX];
  uint16_t sm[1];                   // [nslots][256] u16 StateMaps follow (dynamic LDS)
};
// per lane, per bit scratch: registers on the device, an array on the host
struct P8FamTmp { int ns, draw, look; uint32_t nb; uint32_t L[5]; int nk; uint32_t cx; uint16_t ck;
                  uint8_t* tab; uint32_t mask; int inst, off; };   // tab / mask / inst / off: the context's table, bucket mask, instance and output offset (p8f_lane: once per chunk -- P8CmDev is global memory)   // cx / ck: the context's hash and checksum of this byte (filled in by the caller: p8f_ctx / p8f_chk, read once per byte)   // L / nk: the buckets the context touches at a lookup bit
struct P8FamUni { int y, bp, c0, c1, order, lk; uint32_t rnd_i; const uint32_t* ctx; const uint16_t* chk; int16_t* out; int t; };

P8_HD uint16_t* p8f_smrow(P8FamShared* sh, int s) { return sh->sm + (size_t)s * 256; }
P8_HD uint32_t p8f_ctx(const P8CmDev* d, const P8FamUni& u, int s) { return s == d->order_slot ? d->order_ctx[u.order] : u.ctx[s]; }
P8_HD uint16_t p8f_chk(const P8CmDev* d, const P8FamUni& u, int s) { return s == d->order_slot ? d->order_chk[u.order] : u.chk[s]; }

// ---- chunk start / end -------------------------------------------------------------------------------------------
P8_HD void p8f_load(const P8CmDev* d, const P8FamHome* home, const uint16_t* sm_home, P8FamShared* sh, int tid, int nthreads) {
  const int S = d->nslots;
  for (int s = tid; s < S; s += nthreads) {
    sh->r.cp0[s] = home->cp0[s]; sh->r.runp[s] = home->runp[s]; sh->r.cpo[s] = home->cpo[s]; sh->r.rc[s] = home->rc[s]; sh->r.rb[s] = home->rb[s];
    sh->r.smc[s] = home->smc[s];
    for (int j = 0; j < 8; j++) sh->r.slot[s][j] = home->slot[s][j];
  }
  for (int i = tid; i < S * 256; i += nthreads) sh->sm[i] = sm_home[i];
  for (int i = tid; i < 1024; i += nthreads) sh->nex[i] = d->nex[i];
  for (int i = tid; i < 4096; i += nthreads) sh->stretch[i] = d->stretch[i];
  for (int i = tid; i < 257; i += nthreads) sh->ilog[i] = d->ilog[i];
  { uint32_t* p = &sh->hash[0][0]; for (int i = tid; i < 2 * P8F_HASH; i += nthreads) p[i] = 0; }
  { uint32_t* p = &sh->db[0][0]; for (int i = tid; i < 24; i += nthreads) p[i] = 0; }
  { uint8_t* p = &sh->conflict[0][0]; for (int i = tid; i < 2 * (P8CM_MAXI + 8); i += nthreads) p[i] = 0; }
  if (tid == 0) {
    sh->anyconf[0] = sh->anyconf[1] = 0; sh->walk_cnt = 0; sh->kcount = 0; sh->anyshared = 0;
    for (int k = 0; k < P8CM_MAXI + 8; k++) sh->shared[k] = 0;
    const int i0 = d->rnd.i;   // V(idx) for idx in (i0 - 64, i0] is the generator's table; then the next 256
    for (int k = 0; k < 64; k++) { const uint32_t idx = (uint32_t)i0 - (uint32_t)k; sh->rv[idx & (P8F_RV - 1)] = d->rnd.table[idx & 63]; }
    // (the counter is the reference's `int i` (paq8.cpp:154,163), which wraps; only its value mod 64 matters to the generator. Every index below is taken
    // modulo 2^32 and every range is walked by COUNT: a `<=` between two such indices fails in the step in which the counter passes 2^32 -- after
    // 4.29 G draws, 8.0 MB into enwik-like text -- and leaves the ring with stale values for good: round 5's 8 MiB finding)
    for (uint32_t k = 1; k <= (uint32_t)P8F_LOOK; k++) {
      const uint32_t idx = (uint32_t)i0 + k;
      sh->rv[idx & (P8F_RV - 1)] = sh->rv[(idx - 24) & (P8F_RV - 1)] ^ sh->rv[(idx - 55) & (P8F_RV - 1)];
    }
  }
}
P8_HD void p8f_store(P8CmDev* d, P8FamHome* home, uint16_t* sm_home, const P8FamShared* sh, uint32_t rnd_i, int tid, int nthreads) {
  const int S = d->nslots;
  for (int s = tid; s < S; s += nthreads) {
    home->cp0[s] = sh->r.cp0[s]; home->runp[s] = sh->r.runp[s]; home->cpo[s] = sh->r.cpo[s]; home->rc[s] = sh->r.rc[s]; home->rb[s] = sh->r.rb[s];
    home->smc[s] = sh->r.smc[s];
    for (int j = 0; j < 8; j++) home->slot[s][j] = sh->r.slot[s][j];
  }
  for (int i = tid; i < S * 256; i += nthreads) sm_home[i] = sh->sm[i];
  if (tid == 0) {
    for (int k = 0; k < 64; k++) { const uint32_t idx = rnd_i - (uint32_t)k; d->rnd.table[idx & 63] = sh->rv[idx & (P8F_RV - 1)]; }
    d->rnd.i = (int)rnd_i;
  }
}
// keep the ring 256 values ahead of i: values (old_i + 256, new_i + 256], 24 at a time (a value depends on those 24 and 55
// back, so 24 consecutive new ones are independent of each other). One call = one group of 24; callers loop over the
// groups with all 24 lanes per group (one wavefront in lockstep on the device): P8F_REFILL below. Indices are modulo 2^32 (the
// generator's counter wraps, p8f_load); `left` = how many values from `base` on are still due.
P8_HD void p8f_refill_group(P8FamShared* sh, uint32_t base, uint32_t left, int lane24) {
  const uint32_t idx = base + (uint32_t)lane24;
  if ((uint32_t)lane24 < left) sh->rv[idx & (P8F_RV - 1)] = sh->rv[(idx - 24) & (P8F_RV - 1)] ^ sh->rv[(idx - 55) & (P8F_RV - 1)];
}
// the groups of one step: the counter went from prev_i to rnd_i
#define P8F_REFILL(sh, prev_i, rnd_i, lane24) \
  for (uint32_t p8f_k = 0, p8f_n = (uint32_t)(rnd_i) - (uint32_t)(prev_i); p8f_k < p8f_n; p8f_k += 24) \
    p8f_refill_group((sh), (uint32_t)(prev_i) + (uint32_t)P8F_LOOK + 1u + p8f_k, p8f_n - p8f_k, (lane24))

P8_HD void p8f_lane(const P8CmDev* d, int s, P8FamTmp* t) {
  t->inst = d->slot_inst[s];
  t->tab = d->inst[t->inst].table; t->mask = d->inst[t->inst].mask;
  t->off = d->slot_off[s];
}
// ---- phase 1 (every bit): the state update's outcome and whether it draws; at a lookup bit also the touched buckets into
//      the hash set and the bucket about to be searched into LDS ----
P8_HD void p8f_insert(P8FamShared* sh, int lk, int inst, uint32_t bucket) {
  const uint32_t key = ((uint32_t)(inst + 1) << 26) | bucket;
  uint32_t h = (key * 2654435761u) >> 21;
  uint32_t* tab = sh->hash[lk & 1];
  for (;;) {
    const uint32_t old = P8F_CAS(&tab[h], 0u, key);
    if (old == 0) return;
    if ((old & 0x7fffffffu) == key) {   // another context of the instance holds it: the entry is marked (bit 31) for p8f_marked
      sh->conflict[lk & 1][inst] = 1; sh->anyconf[lk & 1] = 1;
      P8F_OR(&tab[h], 0x80000000u);
      return;
    }
    h = (h + 1) & (P8F_HASH - 1);
  }
}
P8_HD void p8f_phase1(const P8CmDev* d, P8FamShared* sh, const P8FamUni& u, int s, P8FamTmp* t) {
  P8FamHome* r = &sh->r;
  const int inst = t->inst;
  const uint32_t xmask = t->mask;
  t->ns = 0; t->draw = 0; t->look = 0; t->nb = 0; t->nk = 0;
  if (r->cpo[s] != P8F_NIL) {
    t->ns = sh->nex[4 * r->slot[s][r->cpo[s]] + u.y];
    t->draw = t->ns >= 204;
  }
  if (t->draw) P8F_OR(&sh->db[u.t % 3][s >> 5], 1u << (s & 31));
  if (s < 8) sh->db[(u.t + 1) % 3][s] = 0;
  const int bp = u.bp;
  if (!(bp == 0 || bp == 2 || bp == 5)) return;
  // touched buckets: the old slot's, the run bytes', the one about to be searched, and at a byte boundary the two a second visit creates histories in.
  // Fixed positions (P8F_NOKEY = unused), every loop over them with constant bounds: a list filled through a running index is an array in
  // scratch memory on the device, and every access to it a global-memory round trip.
  uint32_t K0 = P8F_NOKEY, K1, K2 = P8F_NOKEY, K3 = P8F_NOKEY, K4 = P8F_NOKEY;
  if (r->cpo[s] != P8F_NIL) K0 = r->cp0[s] >> 6;
  K1 = r->runp[s] >> 6;
  if (!(bp > 1 && r->rc[s] == 0)) {
    t->look = 1;
    t->nb = (t->cx + (uint32_t)u.c0) & xmask;
    K2 = t->nb;
    const uint8_t* g = t->tab + (size_t)t->nb * 64;
    uint8_t* b = sh->bk[s];
#ifdef __HIPCC__
    const uint4* g4 = reinterpret_cast<const uint4*>(g);
    uint4* b4 = reinterpret_cast<uint4*>(b);
    const uint4 v0 = g4[0], v1 = g4[1], v2 = g4[2], v3 = g4[3];
    b4[0] = v0; b4[1] = v1; b4[2] = v2; b4[3] = v3;
#else
    for (int j = 0; j < 64; j++) b[j] = g[j];
#endif
    if (bp == 0) {
      const uint16_t* cs = (const uint16_t*)b;
      const uint16_t chk = t->ck;
      const int mru = b[P8_B_MRU];
      int slot = -1;
      {   // (independent loads first)
        uint16_t cj[7];
        const uint16_t cm = cs[mru & 15];
#ifdef __HIPCC__
#pragma unroll
#endif
        for (int j = 0; j < 7; ++j) cj[j] = cs[j];
#ifdef __HIPCC__
#pragma unroll
#endif
        for (int j = 6; j >= 0; --j) if (cj[j] == chk) slot = j;
        if (cm == chk) slot = mru & 15;
      }
      if (slot >= 0 && b[P8_B_STATE + 7 * slot + 3] == 2) {
        const int cc = b[P8_B_STATE + 7 * slot + 4] + 256;
        K3 = (t->cx + (uint32_t)(cc >> 6)) & xmask;
        K4 = (t->cx + (uint32_t)(cc >> 3)) & xmask;
      }
    }
  }
  if (K1 == K0) K1 = P8F_NOKEY;                                   // a context's own repeats are dropped
  if (K2 == K0 || K2 == K1) K2 = P8F_NOKEY;
  if (K3 == K0 || K3 == K1 || K3 == K2) K3 = P8F_NOKEY;
  if (K4 == K0 || K4 == K1 || K4 == K2 || K4 == K3) K4 = P8F_NOKEY;
  if (K0 != P8F_NOKEY) p8f_insert(sh, u.lk, inst, K0);
  if (K1 != P8F_NOKEY) p8f_insert(sh, u.lk, inst, K1);
  if (K2 != P8F_NOKEY) p8f_insert(sh, u.lk, inst, K2);
  if (K3 != P8F_NOKEY) p8f_insert(sh, u.lk, inst, K3);
  if (K4 != P8F_NOKEY) p8f_insert(sh, u.lk, inst, K4);
  t->nk = 5;
  t->L[0] = K0; t->L[1] = K1; t->L[2] = K2; t->L[3] = K3; t->L[4] = K4;
}
// after the barrier of a lookup bit: does another context of the instance hold one of this context's keys?
P8_HD int p8f_marked(const P8FamShared* sh, int lk, int inst, const P8FamTmp* t) {
  const uint32_t* tab = sh->hash[lk & 1];
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int a = 0; a < 5; a++) {
    if (a >= t->nk || t->L[a] == P8F_NOKEY) continue;
    const uint32_t key = ((uint32_t)(inst + 1) << 26) | t->L[a];
    uint32_t h = (key * 2654435761u) >> 21;
    for (int guard = 0; guard < P8F_HASH; guard++) {
      const uint32_t v = tab[h];
      if ((v & 0x7fffffffu) == key) { if (v >> 31) return 1; break; }
      if (v == 0) break;
      h = (h + 1) & (P8F_HASH - 1);
    }
  }
  return 0;
}
// housekeeping of a lookup bit's run phase: the OTHER parity's hash set and flags are cleared for the next lookup bit
P8_HD void p8f_clear_next(P8FamShared* sh, int lk, int tid, int nthreads) {
  uint32_t* tab = sh->hash[(lk + 1) & 1];
  for (int i = tid; i < P8F_HASH; i += nthreads) tab[i] = 0;
  if (tid < P8CM_MAXI + 8) sh->conflict[(lk + 1) & 1][tid] = 0;
  if (tid == 0) sh->anyconf[(lk + 1) & 1] = 0;
}

// draws among slots [a, b) of this bit
P8_HD int p8f_count(const P8FamShared* sh, int t, int a, int b) {
  const uint32_t* w = sh->db[t % 3];
  int n = 0;
  for (int k = a >> 5; k <= (b - 1) >> 5 && a < b; k++) {
    uint32_t m = w[k];
    if (k == a >> 5) m &= ~0u << (a & 31);
    if (k == (b - 1) >> 5 && ((b & 31) != 0)) m &= (1u << (b & 31)) - 1u;
    n += P8F_POPC(m);
  }
  return n;
}

// the five inputs of a context (ContextMap::mix1's tail :1119-1143) from the cached bytes; the StateMap learns in LDS
P8_HD void p8f_outputs_v(P8FamShared* sh, const P8FamUni& u, int s, int off, int rc, int rb, int st8) {   // off: the context's place in the row; rc / rb: run count / byte, st8: the state in force (0: none)
  P8FamHome* r = &sh->r;
  int16_t* o = u.out + off;
  const int bp = u.bp, c0 = u.c0;
  int o0 = 0;
  if ((rb + 256) >> (8 - bp) == c0) {
    const int b = ((rb >> (7 - bp)) & 1) * 2 - 1;
    o0 = b * (sh->ilog[rc + 1] << (2 + (~rc & 1)));
  }
  uint16_t* smt = p8f_smrow(sh, s);
  const int sc = r->smc[s];
  const int n0 = -!sh->nex[4 * st8 + 2], n1 = -!sh->nex[4 * st8 + 3];
  const int old = smt[sc];
  const int upd = (uint16_t)(old + (((u.y << 16) - old + 128) >> 8));
  smt[sc] = (uint16_t)upd;
  r->smc[s] = (uint8_t)st8;
  const int p1 = (st8 == sc ? upd : (int)smt[st8]) >> 4;
  const int st = (sh->stretch[p1] + (1 << 1)) >> 2;
  const int dn = n1 - n0;
  const int p0 = 4095 - p1;
  o[0] = (int16_t)o0;
  o[1] = (int16_t)st;
  o[2] = (int16_t)((p1 - 2047 + (1 << 2)) >> 3);
  o[3] = (int16_t)(st * (dn < 0 ? -dn : dn));
  o[4] = (int16_t)(((p1 & n0) - (p0 & n1) + (1 << 3)) >> 4);
}
P8_HD void p8f_outputs(const P8CmDev* d, P8FamShared* sh, const P8FamUni& u, int s) {
  const P8FamHome* r = &sh->r;
  p8f_outputs_v(sh, u, s, d->slot_off[s], r->rc[s], r->rb[s], r->cpo[s] != P8F_NIL ? r->slot[s][r->cpo[s]] : 0);
}

// a store to the table that also keeps the lane's cached bytes right when the address falls inside them
P8_HD void p8f_wr(uint8_t* T, P8FamHome* r, int s, uint32_t addr, uint8_t v) {
  T[addr] = v;
  if (addr - r->cp0[s] < 7u) r->slot[s][addr - r->cp0[s]] = v;
  if (addr == r->runp[s]) r->rc[s] = v;
  if (addr == r->runp[s] + 1) r->rb[s] = v;
}
// Bucket::Find on the staged copy; header changes go to the table. Returns the slot index. The seven checksums and priorities are read
// before the first comparison: loads that do not depend on each other cost one LDS latency, not one per slot looked at.
P8_HD int p8f_find_staged(uint8_t* T, uint32_t nb, uint8_t* b, uint16_t checksum) {
  uint16_t* cs = (uint16_t*)b;
  uint8_t* g = T + (size_t)nb * 64;
  const int mru = b[P8_B_MRU], m0 = mru & 15, m1 = mru >> 4;
  const uint16_t cm = cs[m0];
  uint16_t c[7]; uint8_t pr[7];
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int i = 0; i < 7; ++i) { c[i] = cs[i]; pr[i] = b[P8_B_STATE + 7 * i]; }
  if (cm == checksum) return m0;
  int found = -1, worst = 0xFFFF, index = 0;
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int i = 0; i < 7; ++i) {
    if (found < 0) {
      if (c[i] == checksum) found = i;
      else if (pr[i] < worst && m0 != i && m1 != i) { worst = pr[i]; index = i; }
    }
  }
  if (found >= 0) { b[P8_B_MRU] = (uint8_t)(mru << 4 | found); g[P8_B_MRU] = b[P8_B_MRU]; return found; }
  b[P8_B_MRU] = (uint8_t)(0xF0 | index); g[P8_B_MRU] = b[P8_B_MRU];
  cs[index] = checksum; ((uint16_t*)g)[index] = checksum;
  for (int k = 0; k < 7; k++) { b[P8_B_STATE + 7 * index + k] = 0; g[P8_B_STATE + 7 * index + k] = 0; }
  return index;
}
// a bucket of the table -> a lane's staging area (vector loads: one round trip)
P8_HD void p8f_stage_bucket(uint8_t* b, const uint8_t* g) {
#ifdef __HIPCC__
  const uint4* g4 = reinterpret_cast<const uint4*>(g);
  uint4* b4 = reinterpret_cast<uint4*>(b);
  const uint4 v0 = g4[0], v1 = g4[1], v2 = g4[2], v3 = g4[3];
  b4[0] = v0; b4[1] = v1; b4[2] = v2; b4[3] = v3;
#else
  for (int j = 0; j < 64; j++) b[j] = g[j];
#endif
}

// ---- run phase, lane-parallel: ContextMap::mix1's loop body for context s (:1072-1145) on the cached bytes.
//      rank: number of draws of this bit before this context. ----
// The context's registers (slot base, run bytes' address, position in the slot, run count / byte, the slot's seven bytes) are read from LDS once
// at the top -- loads that do not depend on each other --, live in locals (the slot's bytes packed in one 64-bit value) and are written back once at
// the end: as byte arrays in LDS every access was a round trip the compiler had to wait for before the next one (any byte store may alias).
P8_HD int p8f_sv_get(uint64_t sv, int k) { return (int)((sv >> (8 * k)) & 0xff); }
P8_HD uint64_t p8f_sv_set(uint64_t sv, int k, int v) { return (sv & ~((uint64_t)0xff << (8 * k))) | ((uint64_t)(v & 0xff) << (8 * k)); }
P8_HD void p8f_run(const P8CmDev* d, P8FamShared* sh, const P8FamUni& u, int s, const P8FamTmp* t, int rank) {
  P8FamHome* r = &sh->r;
  uint8_t* T = t->tab;
  const uint32_t xmask = t->mask;
  const int bp = u.bp, c0 = u.c0;
  uint32_t cp0 = r->cp0[s], runp = r->runp[s];
  int cpo = r->cpo[s], rc = r->rc[s], rb = r->rb[s];
  uint64_t sv;
  memcpy(&sv, __builtin_assume_aligned(r->slot[s], 8), 8);
  if (cpo != P8F_NIL) {
    int ns = t->ns;
    if (t->draw) {
      const uint32_t v = sh->rv[(u.rnd_i + 1u + (uint32_t)rank) & (P8F_RV - 1)];
      if ((uint32_t)(v << ((452 - ns) >> 3))) ns -= 4;
    }
    sv = p8f_sv_set(sv, cpo, ns);
    T[cp0 + (uint32_t)cpo] = (uint8_t)ns;
    if (cp0 + (uint32_t)cpo == runp) rc = ns & 0xff;           // the state byte doubles as this context's run count / byte
    if (cp0 + (uint32_t)cpo == runp + 1) rb = ns & 0xff;
  }
  if (bp > 1 && rc == 0) cpo = P8F_NIL;
  else if (bp == 1 || bp == 3 || bp == 6) cpo = 1 + (c0 & 1);
  else if (bp == 4 || bp == 7) cpo = 3 + (c0 & 3);
  else {
    const uint16_t checksum = t->ck;
    const uint32_t cx = t->cx, nb = t->nb;
    uint8_t* b = sh->bk[s];
    // the staged bucket predates this lane's own state store above: same bucket -> same byte in the copy
    if (cpo != P8F_NIL && ((cp0 + (uint32_t)cpo) >> 6) == nb) b[(cp0 + (uint32_t)cpo) & 63] = (uint8_t)p8f_sv_get(sv, cpo);
    const int idx = p8f_find_staged(T, nb, b, checksum);
    const uint32_t ncp0 = nb * 64 + P8_B_STATE + 7 * (uint32_t)idx;
    const uint32_t old_runp = runp;
    cp0 = ncp0; cpo = 0;
    {
      const uint8_t* q = b + P8_B_STATE + 7 * idx;
      const uint32_t q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3], q4 = q[4], q5 = q[5], q6 = q[6];
      sv = (uint64_t)(q0 | q1 << 8 | q2 << 16 | q3 << 24) | ((uint64_t)(q4 | q5 << 8 | q6 << 16) << 32);
    }
    if (bp == 0) {
      int refresh = 0;
      // run count of the PREVIOUS context (:1107-1112)
      int prc = rc, prb = rb;
      if ((old_runp >> 6) == nb) { prc = b[old_runp & 63]; prb = b[(old_runp + 1) & 63]; }   // the search above may have replaced the very slot that holds them
      if (p8f_sv_get(sv, 3) == 2) {   // second visit: create the bit histories of bits 2-7 from the one byte seen (:1096-1106)
        // the two buckets go through the lane's staging area (done with: the slot is in sv, the run bytes in prc / prb): one vector
        // fetch and a search on LDS each, instead of a walk over table bytes one dependent load at a time
        const int cc = p8f_sv_get(sv, 4) + 256;
        uint32_t held = nb;
        for (int v = 0; v < 2; v++) {
          const uint32_t nb2 = (cx + (uint32_t)(v ? cc >> 3 : cc >> 6)) & xmask;
          if (nb2 != held) { p8f_stage_bucket(b, T + (size_t)nb2 * 64); held = nb2; }
          const uint32_t o = (uint32_t)(P8_B_STATE + 7 * p8f_find_staged(T, nb2, b, checksum));
          uint8_t* g2 = T + (size_t)nb2 * 64;
          const int cs3 = v ? cc : cc >> 3;   // v = 0: bits 5, 4, 3 of cc;  v = 1: bits 2, 1, 0
          const uint32_t i1 = o + 1 + ((cs3 >> 2) & 1), i2 = o + 3 + ((cs3 >> 1) & 3);
          const uint8_t v0 = (uint8_t)(1 + ((cs3 >> 2) & 1)), v1 = (uint8_t)(1 + ((cs3 >> 1) & 1)), v2 = (uint8_t)(1 + (cs3 & 1));
          b[o] = v0; g2[o] = v0;
          b[i1] = v1; g2[i1] = v1;
          b[i2] = v2; g2[i2] = v2;
        }
        T[ncp0 + 6] = 0; sv = p8f_sv_set(sv, 6, 0);
        refresh = 1;   // those stores may have landed on the old run bytes (same checksum in a coinciding bucket): read them back
      }
      if (refresh) { prc = T[old_runp]; prb = T[old_runp + 1]; }
      const int c1 = u.c1;
      if (prc == 0) { prc = 2; prb = c1; }
      else if (prb != c1) { prc = 1; prb = c1; }
      else if (prc < 254) prc += 2;
      else if (prc == 255) prc = 128;
      T[old_runp] = (uint8_t)prc; T[old_runp + 1] = (uint8_t)prb;
      if (old_runp - ncp0 < 7u) sv = p8f_sv_set(sv, (int)(old_runp - ncp0), prc);           // the same context again: its run bytes are in the new slot
      if (old_runp + 1 - ncp0 < 7u) sv = p8f_sv_set(sv, (int)(old_runp + 1 - ncp0), prb);
      if (refresh) {
        const uint32_t q0 = T[ncp0], q1 = T[ncp0 + 1], q2 = T[ncp0 + 2], q3 = T[ncp0 + 3], q4 = T[ncp0 + 4], q5 = T[ncp0 + 5], q6 = T[ncp0 + 6];
        sv = (uint64_t)(q0 | q1 << 8 | q2 << 16 | q3 << 24) | ((uint64_t)(q4 | q5 << 8 | q6 << 16) << 32);
      }
      runp = ncp0 + 3;
      rc = p8f_sv_get(sv, 3); rb = p8f_sv_get(sv, 4);
    }
  }
  r->cp0[s] = cp0; r->runp[s] = runp; r->cpo[s] = (uint8_t)cpo; r->rc[s] = (uint8_t)rc; r->rb[s] = (uint8_t)rb;
  memcpy(__builtin_assume_aligned(r->slot[s], 8), &sv, 8);
  p8f_outputs_v(sh, u, s, t->off, rc, rb, cpo != P8F_NIL ? p8f_sv_get(sv, cpo) : 0);
}

// Bucket::Find for the walks: the bucket as it is in the table NOW -> the context's staging area (one vector fetch), the search on the copy
// (p8f_find_staged: changes go to both). Returns the byte offset of the slot's first state byte, like p8d_bucket_find.
P8_HD uint32_t p8f_find_fetch(uint8_t* T, uint32_t nb, uint8_t* b, uint16_t checksum) {
  p8f_stage_bucket(b, T + (size_t)nb * 64);
  return nb * 64 + P8_B_STATE + 7 * (uint32_t)p8f_find_staged(T, nb, b, checksum);
}
// ---- an instance with an overlap: its first lane walks it on the table, in the reference's order. One context of the walk = p8f_walk_a
//      (its state update with the draw, the bucket search) + p8f_walk_b (second visit, run bytes, registers, outputs); rank: draws of this
//      bit before the context. ----
struct P8FamWalk { uint32_t cp, cp0, runp, cx, nb2[2]; uint16_t checksum; int need2, cc, drew; };
P8_HD void p8f_walk_a(const P8CmDev* d, P8FamShared* sh, const P8FamUni& u, int inst, int s, int rank, P8FamWalk* w) {
  P8FamHome* r = &sh->r;
  const P8CmInst* x = &d->inst[inst];
  uint8_t* T = x->table;
  const int bp = u.bp, c0 = u.c0;
  uint32_t cp = r->cpo[s] != P8F_NIL ? r->cp0[s] + r->cpo[s] : P8_NIL;
  uint32_t cp0 = r->cp0[s];
  const uint32_t runp = r->runp[s];
  w->drew = 0; w->need2 = 0; w->cc = 0; w->cx = 0; w->checksum = 0; w->nb2[0] = w->nb2[1] = 0;
  if (cp != P8_NIL) {
    int ns = sh->nex[4 * T[cp] + u.y];
    if (ns >= 204) {
      const uint32_t v = sh->rv[(u.rnd_i + 1u + (uint32_t)rank) & (P8F_RV - 1)];
      w->drew = 1;
      if ((uint32_t)(v << ((452 - ns) >> 3))) ns -= 4;
    }
    T[cp] = (uint8_t)ns;
  }
  if (bp > 1 && T[runp] == 0) cp = P8_NIL;
  else if (bp == 1 || bp == 3 || bp == 6) cp = cp0 + 1 + (uint32_t)(c0 & 1);
  else if (bp == 4 || bp == 7) cp = cp0 + 3 + (uint32_t)(c0 & 3);
  else if (bp == 2 || bp == 5) cp0 = cp = p8f_find_fetch(T, (p8f_ctx(d, u, s) + (uint32_t)c0) & x->mask, sh->bk[s], p8f_chk(d, u, s));
  else {
    w->checksum = p8f_chk(d, u, s);
    w->cx = p8f_ctx(d, u, s);
    cp0 = cp = p8f_find_fetch(T, (w->cx + (uint32_t)c0) & x->mask, sh->bk[s], w->checksum);
    const uint8_t* s0 = sh->bk[s] + (cp0 & 63);   // (the fetched copy is current: every change of the search went to both)
    if (s0[3] == 2) {
      w->need2 = 1;
      w->cc = s0[4] + 256;
      w->nb2[0] = (w->cx + (uint32_t)(w->cc >> 6)) & x->mask;
      w->nb2[1] = (w->cx + (uint32_t)(w->cc >> 3)) & x->mask;
    }
  }
  w->cp = cp; w->cp0 = cp0; w->runp = runp;
}
P8_HD void p8f_walk_b(const P8CmDev* d, P8FamShared* sh, const P8FamUni& u, int inst, int s, const P8FamWalk* w) {
  P8FamHome* r = &sh->r;
  const P8CmInst* x = &d->inst[inst];
  uint8_t* T = x->table;
  const int c1 = u.c1;
  const uint32_t cp = w->cp, cp0 = w->cp0;
  uint32_t runp = w->runp;
  if (u.bp == 0) {
    if (w->need2) {
      const int cc = w->cc;
      uint8_t* p = T + p8f_find_fetch(T, w->nb2[0], sh->bk[s], w->checksum);
      p[0] = (uint8_t)(1 + ((cc >> 5) & 1));
      p[1 + ((cc >> 5) & 1)] = (uint8_t)(1 + ((cc >> 4) & 1));
      p[3 + ((cc >> 4) & 3)] = (uint8_t)(1 + ((cc >> 3) & 1));
      p = T + p8f_find_fetch(T, w->nb2[1], sh->bk[s], w->checksum);
      p[0] = (uint8_t)(1 + ((cc >> 2) & 1));
      p[1 + ((cc >> 2) & 1)] = (uint8_t)(1 + ((cc >> 1) & 1));
      p[3 + ((cc >> 1) & 3)] = (uint8_t)(1 + (cc & 1));
      T[cp0 + 6] = 0;
    }
    uint8_t* rp = T + runp;
    if (rp[0] == 0) { rp[0] = 2; rp[1] = (uint8_t)c1; }
    else if (rp[1] != c1) { rp[0] = 1; rp[1] = (uint8_t)c1; }
    else if (rp[0] < 254) rp[0] = (uint8_t)(rp[0] + 2);
    else if (rp[0] == 255) rp[0] = 128;
    runp = cp0 + 3;
  }
  r->cp0[s] = cp0; r->runp[s] = runp; r->cpo[s] = cp == P8_NIL ? (uint8_t)P8F_NIL : (uint8_t)(cp - cp0);
  // outputs need this context's bytes as they are NOW (a later context of the walk may change them again)
  for (int k = 0; k < 7; k++) r->slot[s][k] = T[cp0 + k];
  r->rc[s] = T[runp]; r->rb[s] = T[runp + 1];
  p8f_outputs(d, sh, u, s);
}
// the whole instance. draws0: draws of this bit before the instance; returns its own. Afterwards every lane of the instance reloads (p8f_reload).
P8_HD int p8f_walk(const P8CmDev* d, P8FamShared* sh, const P8FamUni& u, int inst, int draws0) {
  const P8CmInst* x = &d->inst[inst];
  int cnt = 0;
  for (int s = x->first; s < x->first + x->count; s++) {
    P8FamWalk w;
    p8f_walk_a(d, sh, u, inst, s, draws0 + cnt, &w);
    cnt += w.drew;
    p8f_walk_b(d, sh, u, inst, s, &w);
  }
  return cnt;
}
// ---- the same, narrowed down (lookup bits). Contexts interact through shared buckets only, and phase 1 has listed every bucket a context
//      touches: a context none of whose keys another context of the instance also holds (sh->ink[s] == 0) is independent of all the others and
//      runs on its own lane afterwards; only the ones that do share a key (ink == 1; their keys in kkeys) are walked here, in order. What makes
//      this exact:
//        * ranks: a context's rank is the number of draws before it. Phase 1's draw bits (sh->db) are exact for independent contexts; a walked
//          context's bit is corrected here as soon as its real state byte has been seen, before any later context's rank is taken from db;
//        * a walked context may find, after an earlier one changed its bucket, a second visit that phase 1 did not list (buckets outside its
//          keys, possibly an independent context's). Then the reference's order matters for everything not yet processed: the walk falls back
//          to the whole instance -- the unprocessed contexts below it first (independent of all that was done so far), then the rest of this
//          context, then everything above it -- and *full tells the lanes to reload instead of running.
//      base: draws of this bit before the instance (db is exact below it). whole: walk every context (sh->db is corrected all the same).
//      force: test switch, every second visit counts as unlisted. Returns the instance's draws.
P8_HD int p8f_miniwalk(const P8CmDev* d, P8FamShared* sh, const P8FamUni& u, int inst, int base, int whole, int force, uint32_t* full) {
  const P8CmInst* x = &d->inst[inst];
  const int first = x->first, end = x->first + x->count;
  uint32_t* db = sh->db[u.t % 3];
  *full = 0;
  const int nK = (int)sh->kcount;
  // one context of the walk; its draw bit is made exact before anybody above it looks at db
#define P8F_WALK_A(q, w_)                                                                          \
  {                                                                                                \
    p8f_walk_a(d, sh, u, inst, (q), base + p8f_count(sh, u.t, first, (q)), &(w_));                 \
    const uint32_t bit_ = 1u << ((q) & 31);                                                        \
    if ((((db[(q) >> 5] & bit_) != 0) ? 1 : 0) != (w_).drew) db[(q) >> 5] ^= bit_;                 \
  }
  if (whole || nK > P8F_KMAX) {   // asked for (no key list: not a lookup bit, or contexts sitting on one slot), or more than the key list holds: the whole instance
    *full = 1;
    for (int q = first; q < end; q++) { P8FamWalk w; P8F_WALK_A(q, w) p8f_walk_b(d, sh, u, inst, q, &w); sh->ink[q] = 2; }
  } else {
    for (int s = first; s < end; s++) {
      if (sh->ink[s] != 1) continue;
      P8FamWalk w;
      P8F_WALK_A(s, w)
      int listed = 1;
      if (w.need2) {
        int j = 0;
        while (j < nK && sh->kslot[j] != s) j++;
        int in0 = 0, in1 = 0;
        if (j < nK) for (int a = 0; a < sh->knk[j]; a++) { in0 |= sh->kkeys[j][a] == w.nb2[0]; in1 |= sh->kkeys[j][a] == w.nb2[1]; }
        listed = in0 && in1 && !force;
      }
      if (!listed) {
        *full = 1;
        for (int q = first; q < s; q++) if (sh->ink[q] != 2) { P8FamWalk v; P8F_WALK_A(q, v) p8f_walk_b(d, sh, u, inst, q, &v); sh->ink[q] = 2; }
        p8f_walk_b(d, sh, u, inst, s, &w); sh->ink[s] = 2;
        for (int q = s + 1; q < end; q++) { P8FamWalk v; P8F_WALK_A(q, v) p8f_walk_b(d, sh, u, inst, q, &v); sh->ink[q] = 2; }
        break;
      }
      p8f_walk_b(d, sh, u, inst, s, &w); sh->ink[s] = 2;
    }
  }
#undef P8F_WALK_A
  sh->kcount = 0;
  return p8f_count(sh, u.t, first, end);
}
// a lane of an instance that is about to be mini-walked: is this context one of those that share a key? If so its keys go on the list.
P8_HD void p8f_register(P8FamShared* sh, const P8FamUni& u, int inst, int s, const P8FamTmp* t) {
  const int mine = p8f_marked(sh, u.lk, inst, t);
  sh->ink[s] = (uint8_t)mine;
  if (mine) {
    const uint32_t j = P8F_INC(&sh->kcount);
    if (j < P8F_KMAX) { sh->kslot[j] = (uint8_t)s; sh->knk[j] = (uint8_t)t->nk; for (int a = 0; a < 5; a++) sh->kkeys[j][a] = t->L[a]; }
  }
}
// After a walk (the only way two contexts can come to sit on one slot: the later one's search replaced or found the
// slot the earlier one had just taken): do two contexts of the instance share a slot -- as state bytes, or one's state
// bytes under the other's run bytes? While they do, the instance is walked every bit (p8f_walk), lookup bit or not.
// walked_only: after a narrowed walk only the walked contexts (sh->ink == 2) can have come to share a slot -- sharing takes a common bucket, i.e. a
// common key, and the contexts with a common key are the walked ones.
P8_HD int p8f_shares(const P8CmDev* d, const P8FamShared* sh, int inst, int walked_only) {
  const P8FamHome* r = &sh->r;
  const P8CmInst* x = &d->inst[inst];
  for (int a = x->first; a < x->first + x->count; a++) {
    if (walked_only && sh->ink[a] != 2) continue;
    const uint32_t cur_a = r->cpo[a] != P8F_NIL ? r->cp0[a] : 0xFFFFFFF0u, run_a = r->runp[a] - 3;
    for (int b = a + 1; b < x->first + x->count; b++) {
      if (walked_only && sh->ink[b] != 2) continue;
      const uint32_t cur_b = r->cpo[b] != P8F_NIL ? r->cp0[b] : 0xFFFFFFF1u, run_b = r->runp[b] - 3;
      if (cur_a == cur_b || cur_a == run_b || run_a == cur_b) return 1;
    }
  }
  return 0;
}
P8_HD void p8f_reload(const P8CmDev* d, P8FamShared* sh, int s) {
  P8FamHome* r = &sh->r;
  const uint8_t* T = d->inst[d->slot_inst[s]].table;
  for (int k = 0; k < 7; k++) r->slot[s][k] = T[r->cp0[s] + k];
  r->rc[s] = T[r->runp[s]]; r->rb[s] = T[r->runp[s] + 1];
}

// uniform values of step t of a chunk, the running state kept by the caller: the partial byte c0 grows by one bit per step and the byte's eight coded bits and its
// order value are read once, at its first step (nine loads that do not depend on each other) -- rebuilding c0 from up to seven already-coded bits and
// re-reading the order value from global memory at EVERY step cost ~2 k clocks of the family kernel's 17 k per bit. Chunks are whole bytes: bits_in[t .. t + 7] exist.
struct P8FamRun { int last_y, c1, lk, c0; uint32_t bits8; int order, nslots, row_stride; };   // nslots / row_stride: d's, read once
// head: the step's uniform values from the run registers (at a byte's first step the order-N map's value of the step, c0 = 1);
// tail: the step's own bit shifted into them. A compressor runs both per step (p8f_uni_inc: the byte's bits are read once per
// byte); a decoder runs the tail of step t - 1 when bit t - 1 arrives, just before the head of step t (cmx_p8s_fam2_kernel<true>).
P8_HD P8FamUni p8f_uni_head(const uint32_t* ctx, const uint16_t* chk, int16_t* out, const uint8_t* order, int t, P8FamRun* st, uint32_t rnd_i) {
  P8FamUni u;
  const int bp = t & 7, nslots = st->nslots;
  if (bp == 0) {
    st->order = order ? order[t] : 0;
    st->c0 = 1;
  }
  u.y = st->last_y; u.bp = bp; u.c0 = st->c0; u.c1 = st->c1; u.t = t; u.rnd_i = rnd_i;
  u.order = st->order;
  u.ctx = ctx + (size_t)(t >> 3) * (size_t)nslots;
  u.chk = chk + (size_t)(t >> 3) * (size_t)nslots;
  u.out = out + (size_t)t * (size_t)st->row_stride;
  if (bp == 0 || bp == 2 || bp == 5) ++st->lk;
  u.lk = st->lk;
  return u;
}
P8_HD void p8f_uni_tail(P8FamRun* st, int bp, int bit) {
  st->last_y = bit;
  if (bp == 7) st->c1 = (st->c0 * 2 + bit) & 0xff;
  st->c0 = st->c0 * 2 + bit;
}
P8_HD P8FamUni p8f_uni_inc(const P8CmDev* d, const uint32_t* ctx, const uint16_t* chk, const uint8_t* bits_in, int16_t* out, const uint8_t* order, int t, P8FamRun* st,
                           uint32_t rnd_i) {
  const int bp = t & 7;
  (void)d;
  if (bp == 0) {
    const uint32_t b0 = bits_in[t], b1 = bits_in[t + 1], b2 = bits_in[t + 2], b3 = bits_in[t + 3], b4 = bits_in[t + 4], b5 = bits_in[t + 5], b6 = bits_in[t + 6], b7 = bits_in[t + 7];
    st->bits8 = (b0 & 1) | (b1 & 1) << 1 | (b2 & 1) << 2 | (b3 & 1) << 3 | (b4 & 1) << 4 | (b5 & 1) << 5 | (b6 & 1) << 6 | (b7 & 1) << 7;
  }
  const P8FamUni u = p8f_uni_head(ctx, chk, out, order, t, st, rnd_i);
  p8f_uni_tail(st, bp, (int)((st->bits8 >> bp) & 1u));
  return u;
}
#endif
