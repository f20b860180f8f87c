// p8cm2_dev.h -- paq8's ContextMap2 (reference src/models/paq8.cpp:1164-1358: 64-byte buckets of 7 slots found through
// Bucket::Find :1173-1189, bit histories stepped by the nex() state table, byte-history / run statistics in the slot of
// the byte boundary, three StateMap32 per context :645-690) for one instance of C contexts over a chunk of known bits:
// the table kernel behind contextModel2's order-N map (10 contexts, 2 GB), TextModel's map (33, 2 GB) and exeModel's
// (20, 256 MB) -- 441 of paq8's 1552 mixer inputs. Seven inputs per context and bit.
//
// Execution model, as in fxcm_dev.h: one workgroup per instance, one lane per context; a bit is three barrier steps:
//   a  read-only: each lane lists the <= 5 buckets its context touches this bit (bit-history byte's, byte history's,
//      at bits 0/2/5 the bucket about to be searched, at bit 0 the two buckets whose pending histories get created)
//   b  lanes compare lists; any overlap makes the whole instance walk serially this bit (update pass over all contexts,
//      then the mix pass, exactly the reference's order) on lane 0
//   c  otherwise each lane runs update + mix of its own context
// The per-context registers live in LDS for the chunk. Single source: tests/host/p8cm2_emul.cpp runs these functions
// on the host (loop over lanes per step, shuffled) against the oracle's restatement.
#ifndef CMX_P8CM2_DEV_H
#define CMX_P8CM2_DEV_H
#include <stdint.h>

#ifdef __HIPCC__
#define P8_HD __host__ __device__ inline
#else
#define P8_HD inline
#endif

#define P8_NIL 0xFFFFFFFFu
enum { P8CM2_MAXC = 64, P8_B_MRU = 14, P8_B_STATE = 15, P8_B_SIZE = 64, P8_M6 = 72, P8_M8 = 256, P8_M12 = 4608 };

struct P8Cm2Regs {                      // per-context registers (ContextMap2's member arrays)
  uint32_t bit_state[P8CM2_MAXC], bit_state0[P8CM2_MAXC], byte_hist[P8CM2_MAXC];
  int m6_cxt[P8CM2_MAXC], m8_cxt[P8CM2_MAXC], m12_cxt[P8CM2_MAXC];
  uint8_t has_history[P8CM2_MAXC];
};
struct P8Cm2Dev {
  uint8_t* table; uint32_t mask; int C, slot_parallel;
  const uint8_t* nex;                   // nex(s, k) = nex[4 s + k]: next state on 0 / on 1, n0, n1 (:131-341)
  const int16_t* stretch;               // stretch(p), p in [0, 4095]
  const uint8_t* ilog;                  // ilog(x), x in [0, 256]
  uint32_t *m6, *m8, *m12;              // StateMap32 cells: [C][72], [C][256], [C][4608]
  P8Cm2Regs regs;                       // home between chunks
  uint32_t bits; int last_y;            // c0-style partial byte (:1196-1200) and the last coded bit, carried between chunks
  int row_stride, out_off;              // output placement: stand-alone 7 C per row at 0; in the paq8 stage the 1552-vector
};
struct P8Cm2Shared { P8Cm2Regs r; int32_t touched[P8CM2_MAXC][5]; int conflict; uint8_t nz[P8CM2_MAXC]; };   // nz: state > 0 (mix()'s return value counts them)
struct P8Cm2Bit { int y, bpos; uint32_t bits; uint8_t last_byte; const uint32_t* ctx; const uint16_t* chk; int16_t* out; };

P8_HD int p8d_sm32(uint32_t* t, int* cxt, int y, int cx) {   // StateMap32::p with limit 1023 (:660-672)
  uint32_t p0 = t[*cxt];
  const int n = p0 & 1023, pr = p0 >> 10;
  if (n < 1023) ++p0; else p0 = (p0 & 0xfffffc00u) | 1023u;
  p0 += ((uint32_t)(((y << 22) - pr) >> 3) * (uint32_t)(16384 / (n + n + 3))) & 0xfffffc00u;   // the product wraps (as the reference's compiled code does)
  t[*cxt] = p0;
  *cxt = cx;
  return (int)(t[cx] >> 20);
}
P8_HD uint32_t p8d_bucket_find(uint8_t* table, uint32_t b, uint16_t checksum) {   // Bucket::Find -> byte offset of BitState[slot][0]
  uint8_t* p = table + (size_t)b * P8_B_SIZE;
  uint16_t* cs = (uint16_t*)p;
  uint8_t* mru = p + P8_B_MRU;
  if (cs[*mru & 15] == checksum) return b * P8_B_SIZE + P8_B_STATE + 7 * (*mru & 15);
  int worst = 0xFFFF, index = 0;
  for (int i = 0; i < 7; ++i) {
    if (cs[i] == checksum) { *mru = (uint8_t)(*mru << 4 | i); return b * P8_B_SIZE + P8_B_STATE + 7 * i; }
    if (p[P8_B_STATE + 7 * i] < worst && (*mru & 15) != i && *mru >> 4 != i) { worst = p[P8_B_STATE + 7 * i]; index = i; }
  }
  *mru = (uint8_t)(0xF0 | index);
  cs[index] = checksum;
  for (int k = 0; k < 7; k++) p[P8_B_STATE + 7 * index + k] = 0;
  return b * P8_B_SIZE + P8_B_STATE + 7 * index;
}
P8_HD void p8d_update(P8Cm2Dev* d, P8Cm2Shared* sh, const P8Cm2Bit& u, int i) {   // ContextMap2::Update, one context (:1209-1266)
  uint8_t* T = d->table;
  P8Cm2Regs* r = &sh->r;
  if (r->bit_state[i] != P8_NIL) T[r->bit_state[i]] = d->nex[4 * T[r->bit_state[i]] + u.y];
  if (u.bpos > 1 && T[r->byte_hist[i]] == 0) { r->bit_state[i] = P8_NIL; return; }
  if (u.bpos == 0) {
    const uint16_t chk = u.chk[i];
    const uint32_t ctx = u.ctx[i];
    r->bit_state[i] = r->bit_state0[i] = p8d_bucket_find(T, (ctx + u.bits) & d->mask, chk);
    uint8_t* s0 = T + r->bit_state0[i];
    if (s0[3] == 2) {  // pending bit histories for bits 2-7
      const int cc = s0[4] + 256;
      uint8_t* p = T + p8d_bucket_find(T, (ctx + (uint32_t)(cc >> 6)) & d->mask, chk);
      p[0] = (uint8_t)(1 + ((cc >> 5) & 1));
      p[1 + ((cc >> 5) & 1)] = (uint8_t)(1 + ((cc >> 4) & 1));
      p[3 + ((cc >> 4) & 3)] = (uint8_t)(1 + ((cc >> 3) & 1));
      p = T + p8d_bucket_find(T, (ctx + (uint32_t)(cc >> 3)) & d->mask, chk);
      p[0] = (uint8_t)(1 + ((cc >> 2) & 1));
      p[1 + ((cc >> 2) & 1)] = (uint8_t)(1 + ((cc >> 1) & 1));
      p[3 + ((cc >> 1) & 3)] = (uint8_t)(1 + (cc & 1));
      s0[6] = 0;
    }
    uint8_t* bh = T + r->byte_hist[i];  // byte history of the PREVIOUS context
    bh[3] = bh[2];
    bh[2] = bh[1];
    if (bh[0] == 0) { bh[0] = 2; bh[1] = u.last_byte; }
    else if (bh[1] != u.last_byte) { bh[0] = 1; bh[1] = u.last_byte; }
    else if (bh[0] < 254) bh[0] = (uint8_t)(bh[0] + 2);
    else if (bh[0] == 255) bh[0] = 128;
    r->byte_hist[i] = r->bit_state0[i] + 3;
    r->has_history[i] = T[r->bit_state0[i]] > 15;
  } else if (u.bpos == 2 || u.bpos == 5) {
    r->bit_state[i] = r->bit_state0[i] = p8d_bucket_find(T, (u.ctx[i] + u.bits) & d->mask, u.chk[i]);
  } else if (u.bpos == 1 || u.bpos == 3 || u.bpos == 6) r->bit_state[i] = r->bit_state0[i] + 1 + (uint32_t)u.y;
  else r->bit_state[i] = r->bit_state0[i] + 3 + (u.bits & 3);   // 4, 7
}
P8_HD void p8d_mix(P8Cm2Dev* d, P8Cm2Shared* sh, const P8Cm2Bit& u, int i) {   // ContextMap2::mix, one context (:1321-1357)
  const uint8_t* T = d->table;
  P8Cm2Regs* r = &sh->r;
  int16_t* o = u.out + d->out_off + 7 * i;
  int state = r->bit_state[i] != P8_NIL ? T[r->bit_state[i]] : 0;
  sh->nz[i] = (uint8_t)(state > 0);
  int p1 = p8d_sm32(d->m8 + (size_t)i * P8_M8, &r->m8_cxt[i], u.y, state);
  int n0 = d->nex[4 * state + 2], n1 = d->nex[4 * state + 3], k = n1 + 1;
  k = (k * 64) / (k + n0 + 1);
  n0 = -!n0; n1 = -!n1;
  const uint8_t* bh = T + r->byte_hist[i];
  const int bp = u.bpos;
  int v = 0;
  if ((uint32_t)((bh[1] + 256) >> (8 - bp)) == u.bits) {
    const int run = bh[0];
    const int sign = ((bh[1] >> (7 - bp)) & 1) * 2 - 1;
    v = sign * (d->ilog[run + 1] << (3 - (run & 1)));
  } else if (bp > 0 && (bh[0] & 1) > 0) {
    if ((uint32_t)((bh[2] + 256) >> (8 - bp)) == u.bits) v = (((bh[2] >> (7 - bp)) & 1) * 2 - 1) * 128;
    else if (r->has_history[i] && (uint32_t)((bh[3] + 256) >> (8 - bp)) == u.bits) v = (((bh[3] >> (7 - bp)) & 1) * 2 - 1) * 128;
  }
  o[0] = (int16_t)v;
  if (r->has_history[i]) {
    state = (bh[1] >> (7 - bp)) & 1;
    state |= ((bh[2] >> (7 - bp)) & 1) * 2;
    state |= ((bh[3] >> (7 - bp)) & 1) * 4;
  } else state = 8;
  const int st = d->stretch[p1] >> 2;
  o[1] = (int16_t)st;
  o[2] = (int16_t)((p1 - 2047) >> 3);
  p1 >>= 4;
  const int p0 = 255 - p1;
  const int dn = n1 - n0;
  o[3] = (int16_t)(st * (dn < 0 ? -dn : dn));
  o[4] = (int16_t)((p1 & n0) - (p0 & n1));
  o[5] = (int16_t)(d->stretch[p8d_sm32(d->m12 + (size_t)i * P8_M12, &r->m12_cxt[i], u.y, (state << 9) | (bp << 6) | k)] >> 2);
  o[6] = (int16_t)(d->stretch[p8d_sm32(d->m6 + (size_t)i * P8_M6, &r->m6_cxt[i], u.y, (state << 3) | bp)] >> 2);
}
// step a: the buckets context i touches this bit (read-only)
P8_HD void p8d_touch(P8Cm2Dev* d, P8Cm2Shared* sh, const P8Cm2Bit& u, int i) {
  const uint8_t* T = d->table;
  const P8Cm2Regs* r = &sh->r;
  int32_t* L = sh->touched[i];
  for (int j = 0; j < 5; j++) L[j] = -1;
  if (i == 0) sh->conflict = 0;
  if (r->bit_state[i] != P8_NIL) L[0] = (int32_t)(r->bit_state[i] >> 6);
  L[1] = (int32_t)(r->byte_hist[i] >> 6);
  if (u.bpos > 1 && T[r->byte_hist[i]] == 0) return;
  if (u.bpos == 0 || u.bpos == 2 || u.bpos == 5) {
    const uint32_t nb = (u.ctx[i] + u.bits) & d->mask;
    L[2] = (int32_t)nb;
    if (u.bpos == 0) {
      const uint8_t* p = T + (size_t)nb * P8_B_SIZE;
      const uint16_t* cs = (const uint16_t*)p;
      const int mru = p[P8_B_MRU];
      int slot = -1;
      if (cs[mru & 15] == u.chk[i]) slot = mru & 15;
      else for (int j = 0; j < 7; ++j) if (cs[j] == u.chk[i]) { slot = j; break; }
      if (slot >= 0 && p[P8_B_STATE + 7 * slot + 3] == 2) {
        const int cc = p[P8_B_STATE + 7 * slot + 4] + 256;
        L[3] = (int32_t)((u.ctx[i] + (uint32_t)(cc >> 6)) & d->mask);
        L[4] = (int32_t)((u.ctx[i] + (uint32_t)(cc >> 3)) & d->mask);
      }
    }
  }
}
P8_HD void p8d_conflict(P8Cm2Dev* d, P8Cm2Shared* sh, int i) {   // step b
  const int32_t* L = sh->touched[i];
  for (int o = 0; o < d->C; o++) {
    if (o == i) continue;
    const int32_t* O = sh->touched[o];
    for (int a = 0; a < 5; a++)
      if (L[a] >= 0)
        for (int b = 0; b < 5; b++) if (L[a] == O[b]) { sh->conflict = 1; return; }
  }
}
P8_HD void p8d_run(P8Cm2Dev* d, P8Cm2Shared* sh, const P8Cm2Bit& u, int i) {   // step c
  if (!sh->conflict && d->slot_parallel) { p8d_update(d, sh, u, i); p8d_mix(d, sh, u, i); }
  else if (i == 0) {
    for (int j = 0; j < d->C; j++) p8d_update(d, sh, u, j);
    for (int j = 0; j < d->C; j++) p8d_mix(d, sh, u, j);
  }
}
// uniform values of step t of a chunk: y = the bit coded before it, bits / last_byte as ContextMap2::Update leaves them
// y: the bit before step t (a compressor reads it from the chunk's bits, a decoder from its box: cmx_late.h)
P8_HD P8Cm2Bit p8d_bit_y(const P8Cm2Dev* d, const uint32_t* ctx, const uint16_t* chk, int y, int16_t* out, int t, uint32_t* run_bits) {
  const int C = d->C;
  P8Cm2Bit u;
  u.y = y;
  u.bpos = t & 7;
  *run_bits += *run_bits + (uint32_t)u.y;
  u.last_byte = (uint8_t)(*run_bits & 0xFF);
  if (u.bpos == 0) *run_bits = 1;
  u.bits = *run_bits;
  u.ctx = ctx + (size_t)(t >> 3) * (size_t)C;
  u.chk = chk + (size_t)(t >> 3) * (size_t)C;
  u.out = out + (size_t)t * (size_t)d->row_stride;
  return u;
}
P8_HD P8Cm2Bit p8d_bit(const P8Cm2Dev* d, const uint32_t* ctx, const uint16_t* chk, const uint8_t* bits_in, int16_t* out, int t, uint32_t* run_bits, int* last_y) {
  const P8Cm2Bit u = p8d_bit_y(d, ctx, chk, *last_y, out, t, run_bits);
  *last_y = bits_in[t];
  return u;
}
#endif
