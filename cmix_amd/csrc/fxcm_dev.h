// fxcm_dev.h -- the per-bit numeric core of the fxcm stage (reference src/models/fxcmv1.cpp, vendored by cmix v21):
// 31 hashed context maps (ContextMap :971-1174, ContextMap1 :1176-1379, ContextMap2 :1408-1612 with E::get :933-957 /
// E1::get :1381-1406), their StateMaps (:672-705), 7 SmallStationaryContextMaps (:831-863), RunContextMap (:756-829),
// MatchModel2 (:3420-3676) with 3 StateMap1 (:707-736), SparseMatchModel (:1742-1829), ten 512-input + two 16-input
// int16 Mixer1 (:472-660), the mixer selectors (:4616-4737), update1's failure history, six chained APMs (:1622-1643)
// and the final blend (:4758-4833). All integer; what leaves the stage are the 431 values FXCM::Predict() hands to
// cmix per bit (layer-0 columns 3..433).
//
// Execution model: ONE workgroup of FX_THREADS threads per stream walks the bits of a chunk. A bit is five phases
// separated by workgroup barriers; inside a phase no thread reads what another thread of the same phase writes:
//   1  wave 0: one lane per *unit* (31 context maps, 7 SSCMs, the match lane, the run map, the LSTM input, mixers
//      10/11 training, 6 APM cell updates) -- a map lane walks its contexts in order because the contexts of one map
//      share buckets (replacement and the last-used byte are order dependent); waves 1..3: training of the ten
//      512-weight rows on the PREVIOUS input vector (the input vector is double buffered in LDS)
//   2  thread 0: dead-zone adaptation, failure history, the ten selectors
//   3  all threads: one int16 pair of each of the ten dot products (pmaddwd + >>8 per pair is order free)
//   4  160 threads: 16-to-1 partial sums          5  thread 0: final mixers, APM chain, exports
// The same functions compile for the host (tests/host/fxcm_emul.cpp runs the phases with a loop over thread ids, in
// shuffled order, to check the kernel body against the oracle without a GPU); FX_HD is empty there.
#ifndef CMX_FXCM_DEV_H
#define CMX_FXCM_DEV_H
#include <stdint.h>

#include "fxcm_rec.h"

#ifdef __HIPCC__
#define FX_HD __host__ __device__ inline
#else
#define FX_HD inline
#endif

#define FX_NONE 0xffffffffu
#define FX_NOKEY 0xffffffffu
#ifdef __HIPCC__
#define FX_CAS(p, c, v) atomicCAS((p), (c), (v))
#else
static inline uint32_t fx_cas_host(uint32_t* p, uint32_t c, uint32_t v) { const uint32_t o = *p; if (o == c) *p = v; return o; }
#define FX_CAS(p, c, v) fx_cas_host((p), (c), (v))
#endif
enum { FX_THREADS = 256, FX_BMASK = 0xffffff,
       FX_U_SSCM = 81, FX_U_MATCH = 88, FX_U_RCM = 89, FX_U_LSTM = 90, FX_U_MIX10 = 91, FX_U_MIX11 = 92, FX_U_APM = 93, FX_UNITS = 99, FX_TRAIN0 = 128,
       FX_M_WGS = 2, FX_M_WAVES = 16,   // role M of the device stage: workgroups x 8 wavefronts, each wavefront owns whole maps
       FX_TAB_RC1 = 0, FX_TAB_ST1 = 512, FX_TAB_ST2 = 512 + 4096, FX_TAB_ST32 = 512 + 8192, FX_TAB_ST8 = 512 + 8192 + 256, FX_TAB_LEN = 512 + 8192 + 512 };

// Pointer members of the stream's device structures. The kernels work on an LDS COPY of FxDev, so every table pointer is itself loaded from memory and
// the compiler cannot know where it points: each access through it became a FLAT instruction, which waits for the LDS counter AND for every outstanding
// global load and store of the wavefront (444 of them in cmx_fxcm_roles_kernel in round 4). These wrappers keep the 8-byte layout the host fills in
// and, on the device, hand the pointer out through an address-space cast: FxGP = device-global memory, FxLP = the workgroup's LDS (the tables the
// kernel copies there at its start). Host code (fxcm_build.h, tests/host) sees plain pointers; `.p` is the raw value.
template <class T> struct FxGP {
  T* p;
  FX_HD FxGP& operator=(T* q) { p = q; return *this; }
  FX_HD operator T*() const {
#if defined(__HIP_DEVICE_COMPILE__)
    __attribute__((address_space(1))) T* q = (__attribute__((address_space(1))) T*)p;
    asm("" : "+v"(q));   // (opaque: a bare cast there and back is folded away and the access stays flat)
    return (T*)q;
#else
    return p;
#endif
  }
};
template <class T> struct FxLP {
  T* p;
  FX_HD FxLP& operator=(T* q) { p = q; return *this; }
  FX_HD operator T*() const {
#if defined(__HIP_DEVICE_COMPILE__)
    __attribute__((address_space(3))) T* q = (__attribute__((address_space(3))) T*)p;
    asm("" : "+v"(q));
    return (T*)q;
#else
    return p;
#endif
  }
};

struct FxMapDev {
  FxGP<uint8_t> t; uint32_t tmask;     // buckets
  int A, B, C, kep, u;                 // slots per bucket, bytes per bucket, contexts, keep flag, st2 input on/off
  int slot_base, tx_off, exp_off;      // first slot in FxByteRec::cx; first input / first exported value (normal layout)
  FxLP<const uint8_t> nn;              // state table: next state on 0 / on 1, n0, n1 (on the device: the kernel's LDS copy, FxDev::sta)
  FxGP<const int16_t> tab;             // rc1[512] st1[4096] st2[4096] st32[256] st8[256]
  FxGP<uint32_t> sm;                   // C StateMaps of 256 cells
  uint32_t cp[8], cp0[8], runp[8], cxt[8];
  int sm_cxt[8];
};
struct FxMatchInfo { uint32_t length, index, lengthBak, indexBak; uint8_t expectedByte, delta; };
struct FxMtf { int root, index, prev[4], next[4]; };

struct FxDev {                         // everything a stream owns on the device (global memory)
  FxMapDev maps[FX_NMAPS];
  uint8_t slot_map[FX_NSLOTS], slot_idx[FX_NSLOTS];   // slot -> map (mixing order), context index within the map
  int slot_parallel;                   // 1: one lane per context slot with a per-map serial fallback; 0: one lane per map
  uint8_t mw_slot[FX_M_WAVES + 1], mw_map[FX_M_WAVES + 1];   // role M's wavefront w owns slots [mw_slot[w], mw_slot[w + 1]) = maps [mw_map[w], mw_map[w + 1])
  FxLP<const int16_t> squash, stretch; // squash[d + 2047], stretch[p]   (on the device: LDS copies)
  FxLP<const uint8_t> wrt;             // byte -> 2-bit class [256], 3-bit class [256] of cmix's WRT-swapped alphabet
  const uint8_t* sta[6];               // the six state tables the maps' nn pointers refer to (the kernel keeps copies in LDS)
  FxGP<uint16_t> sscm_data[FX_NSSCM]; int sscm_mask[FX_NSSCM], sscm_ctx[FX_NSSCM], sscm_B[FX_NSSCM], sscm_bcount[FX_NSSCM], sscm_cp[FX_NSSCM];
  FxGP<uint32_t> sm1_t[3]; int sm1_mask[3], sm1_cxt[3];
  FxGP<uint8_t> rcm_t; uint32_t rcm_n, rcm_cp; int16_t rcm_rc[512];
  FxMatchInfo cand[4]; uint32_t nActive; FxGP<uint32_t> mhash; uint32_t mhashmask;
  FxGP<uint32_t> sp_table; FxMtf sp_list; uint32_t sp_hashes[4], sp_hashIndex, sp_length, sp_index; uint8_t sp_expectedByte, sp_valid;
  FxGP<uint8_t> buffer; int pos;
  FxGP<int16_t> wx[12]; int mx_M[12], mx_shift[12], mx_uperr[12];
  FxGP<uint16_t> apm_t[6];
  // scalars carried between chunks (copied to / from FxShared by thread 0)
  int mx_elim[12], mx_cxt[12], mx_pr[12], apm_index[6];
  int blpos, lastbyte, pr, parity, have_rec;
  uint32_t fails, failz, failcount;
  int16_t tx[2][FX_TX], in2[16];
  float pending[FX_OUTPUTS];
  FxByteRec rec;                       // the record of the last completed byte (or the start-up values)
};

struct FxShared {                      // LDS on the device (~100 KB: dynamic shared memory)
  int16_t tx[2][FX_TX], in2[16];
  int32_t part[FX_NMIX1][FX_THREADS], part2[FX_NMIX1][16];
  int mx_elim[12], mx_cxt[12], mx_pr[12], apm_index[6];
  int pr, parity;
  uint32_t fails, failz, failcount;
  int slot_res[FX_NSLOTS], isMatch;
  uint32_t m_ctx[3];                   // MatchModel2's StateMap1 contexts of the bit (0: none)
  // cached bytes of every context (write-through: the table is always current): the 7 state bytes at cp0, the 2 run bytes at
  // runp, and -- at bits 0 / 2 / 5 -- the bucket about to be searched, fetched by the touch step before the barrier
  uint8_t mslot[FX_NSLOTS][8], mrun[FX_NSLOTS][2];
  alignas(16) uint8_t mbk[FX_NSLOTS][128];
  uint8_t mlook[FX_NSLOTS];            // the touch step staged a bucket for this context
  uint32_t ohash[2][1024];             // (map, bucket) pairs the contexts touch this bit: a compare-and-swap hash set, two parities
  int mconf[2][FX_NMAPS];              // two contexts of the map touch the same bucket this bit: the map runs serially
  // per-slot registers of the context maps and their StateMaps, resident for the chunk (home: FxMapDev / FxMapDev::sm)
  uint32_t mcp[FX_NMAPS][8], mcp0[FX_NMAPS][8], mrunp[FX_NMAPS][8], mcxt[FX_NMAPS][8];
  int msmc[FX_NMAPS][8];
  uint32_t sm[FX_NSLOTS][256];
};

struct FxBit {                         // uniform per-bit values every thread derives from the byte stream
  int q;                               // the update's number within the stream's chunk (parity of the overlap hash set)
  int y, bpos, c0, lastbyte, blpos, rate, sscmrate, boundary, lstmpr, lstmex, normal;
  const FxByteRec* rec;                // the record in force for this bit
  float* orow;                         // where the 431 exported values of this bit go
};

FX_HD int fxd_squash(const FxDev* d, int v) { return v < -2047 ? 1 : v > 2047 ? 4095 : d->squash[v + 2047]; }
FX_HD int fxd_clp(int z) { return z < -2047 ? -2047 : z > 2047 ? 2047 : z; }
FX_HD int fxd_min(int a, int b) { return a < b ? a : b; }
FX_HD int fxd_max(int a, int b) { return a < b ? b : a; }
FX_HD int fxd_dt(int i) { return i == 1023 ? 1 : 4096 / (i + 2); }
FX_HD int fxd_sat16(int v) { return v < -32768 ? -32768 : v > 32767 ? 32767 : v; }
FX_HD uint32_t fxd_hash3(uint32_t a, uint32_t b, uint32_t c) {  // hash :2276-2279
  const uint32_t h = a * 110002499u + b * 30005491u + c * 50004239u;
  return h ^ h >> 9 ^ a >> 3 ^ b >> 3 ^ c >> 4;
}
// Inputs::add (:197-202) exports squash(value) / 4095 to cmix (AddPrediction :98-101)
FX_HD float fxd_export(const FxDev* d, int v) { return (float)fxd_squash(d, v) * (float)(1.0 / 4095); }

// exported-value layout after the maps (normal) or without them (the stream's first byte, when no map holds a context)
struct FxLayout { int tx_rcm, tx_lstm, exp_rcm, exp_mix; };
FX_HD FxLayout fxd_layout(const FxDev* d, int normal) {
  FxLayout l;
  const int tx0 = 2 * FX_NSSCM + 7 + 2, ex0 = FX_NSSCM + 7 + 2;
  if (normal) {
    const FxMapDev* m = &d->maps[FX_NMAPS - 1];
    l.tx_rcm = m->tx_off + m->C * (5 + m->u); l.exp_rcm = m->exp_off + m->C * (4 + m->u);
  } else { l.tx_rcm = tx0; l.exp_rcm = ex0; }
  l.tx_lstm = l.tx_rcm + 1;
  l.exp_mix = l.exp_rcm + 2;           // after the run map's value and squash(64)
  return l;
}

// ---------------------------------------------------------------- context maps
// E::get / E1::get: slot with this checksum, else replace the lowest-priority slot that is not one of the last two used
FX_HD uint32_t fxd_bucket_get(uint8_t* b, int A, uint16_t ch, int keep) {
  uint16_t* chk = (uint16_t*)b;
  const uint8_t last = b[2 * A];
  if (chk[last & 15] == ch) return (uint32_t)(2 * A + 1 + 7 * (last & 15));
  int lowest = 0xffff, bi = 0;
  for (int i = 0; i < A; ++i) {
    if (chk[i] == ch) { b[2 * A] = (uint8_t)(last << 4 | i); return (uint32_t)(2 * A + 1 + 7 * i); }
    const int pri = b[2 * A + 1 + 7 * i];
    if (pri < lowest && (last & 15) != i && (last >> 4) != i) { lowest = pri; bi = i; }
  }
  b[2 * A] = (uint8_t)(last << 4 | bi | keep);
  chk[bi] = ch;
  uint8_t* h = b + 2 * A + 1 + 7 * bi;
  for (int k = 0; k < 7; k++) h[k] = 0;
  return (uint32_t)(2 * A + 1 + 7 * bi);
}
// the same search on a staged copy b of the bucket; every change also goes to the table g. All checksums and priorities are read
// before the first comparison (loads that do not depend on each other: one LDS latency, not one per slot looked at).
FX_HD uint32_t fxd_bucket_get_staged(uint8_t* b, uint8_t* g, int A, uint16_t ch, int keep) {
  const uint16_t* chk = (const uint16_t*)b;
  const uint8_t last = b[2 * A];
  const int l0 = last & 15, l1 = last >> 4;
  const uint16_t clast = chk[l0];
  uint16_t c[14]; uint8_t pr[14];
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int i = 0; i < 14; ++i) { const int j = i < A ? i : A - 1; c[i] = chk[j]; pr[i] = b[2 * A + 1 + 7 * j]; }
  if (clast == ch) return (uint32_t)(2 * A + 1 + 7 * l0);
  int found = -1, lowest = 0xffff, bi = 0;
#ifdef __HIPCC__
#pragma unroll
#endif
  for (int i = 0; i < 14; ++i) {
    if (i < A && found < 0) {
      if (c[i] == ch) found = i;
      else if (pr[i] < lowest && l0 != i && l1 != i) { lowest = pr[i]; bi = i; }
    }
  }
  if (found >= 0) { b[2 * A] = (uint8_t)(last << 4 | found); g[2 * A] = b[2 * A]; return (uint32_t)(2 * A + 1 + 7 * found); }
  b[2 * A] = (uint8_t)(last << 4 | bi | keep); g[2 * A] = b[2 * A];
  ((uint16_t*)b)[bi] = ch; ((uint16_t*)g)[bi] = ch;
  for (int k = 0; k < 7; k++) { b[2 * A + 1 + 7 * bi + k] = 0; g[2 * A + 1 + 7 * bi + k] = 0; }
  return (uint32_t)(2 * A + 1 + 7 * bi);
}
// a bucket of the table -> its staged copy (vector loads: one round trip)
FX_HD void fxd_stage_bucket(uint8_t* bk, const uint8_t* g, int B) {
#ifdef __HIPCC__
  // (one branch per bucket size, every vector a named value: an array filled under a run-time bound stays in scratch memory)
  const uint4* g4 = reinterpret_cast<const uint4*>(g);
  uint4* b4 = reinterpret_cast<uint4*>(bk);
  if (B == 128) {
    const uint4 v0 = g4[0], v1 = g4[1], v2 = g4[2], v3 = g4[3], v4 = g4[4], v5 = g4[5], v6 = g4[6], v7 = g4[7];
    b4[0] = v0; b4[1] = v1; b4[2] = v2; b4[3] = v3; b4[4] = v4; b4[5] = v5; b4[6] = v6; b4[7] = v7;
  } else if (B == 64) {
    const uint4 v0 = g4[0], v1 = g4[1], v2 = g4[2], v3 = g4[3];
    b4[0] = v0; b4[1] = v1; b4[2] = v2; b4[3] = v3;
  } else {
    const uint4 v0 = g4[0], v1 = g4[1];
    b4[0] = v0; b4[1] = v1;
  }
#else
  for (int q = 0; q < B; q++) bk[q] = g[q];
#endif
}
FX_HD void fxd_map_reload(FxDev* d, FxShared* sh, int k, int i) {   // the cached bytes of context i of map k from the table
  const FxMapDev* x = &d->maps[k];
  const int s = x->slot_base + i;
  for (int q = 0; q < 7; q++) sh->mslot[s][q] = x->t[sh->mcp0[k][i] + q];
  sh->mrun[s][0] = x->t[sh->mrunp[k][i]]; sh->mrun[s][1] = x->t[sh->mrunp[k][i] + 1];
}
// one context of one map for one bit on the cached bytes: same values as fxd_map_ctx below (which stays as the serial walk
// for maps with an overlap), HBM touched only through the staged bucket, write-through stores and the output tables
FX_HD void fxd_map_ctx_fast(FxDev* d, FxShared* sh, const FxBit& u, int k, int i) {
  const FxMapDev* x = &d->maps[k];
  uint32_t *cp = sh->mcp[k], *cp0 = sh->mcp0[k], *runp = sh->mrunp[k], *cxt = sh->mcxt[k];
  int* sm_cxt = sh->msmc[k];
  int16_t* tx = sh->tx[sh->parity ^ 1] + x->tx_off + i * (5 + x->u);
  float* ex = u.orow + x->exp_off + i * (4 + x->u);
  uint8_t* t = x->t;
  const int16_t* tab = x->tab;
  const int y = u.y, bpos = u.bpos, c0 = u.c0, c1 = u.lastbyte;
  const int s = x->slot_base + i;
  uint8_t* sl = sh->mslot[s];
  uint8_t* rn = sh->mrun[s];
  int result = 0, n = 0, e = 0;
#define ADD(v) do { const int v_ = (v); tx[n++] = (int16_t)v_; ex[e++] = fxd_export(d, v_); } while (0)
#define ADDQ(v) do { tx[n++] = (int16_t)(v); } while (0)
  if ((u.rec->skip[s >> 5] >> (s & 31)) & 1) {
    ADD(0); if (x->u) ADD(0); ADD(0); ADD(0); ADDQ(64); ADD(0);
  } else {
    if (cp[i] != FX_NONE) {
      const uint8_t ns = x->nn[sl[cp[i] - cp0[i]] * 4 + y];
      sl[cp[i] - cp0[i]] = ns; t[cp[i]] = ns;
      if (cp[i] - runp[i] < 2u) rn[cp[i] - runp[i]] = ns;
    }
    int state = 0;
    if (bpos > 1 && rn[0] == 0) cp[i] = FX_NONE;
    else {
      const uint16_t chk = (uint16_t)((cxt[i] >> 16) ^ (uint32_t)i);
      if (bpos && bpos != 2 && bpos != 5) {
        const uint32_t smask = (0x31031010u >> (bpos << 2)) & 0x0F;
        cp[i] = cp0[i] + smask + ((uint32_t)c0 & smask);
      } else {
        const uint32_t nb = (cxt[i] + (uint32_t)c0) & x->tmask;
        const size_t b = (size_t)nb * (size_t)x->B;
        uint8_t* bk = sh->mbk[s];
        if (cp[i] != FX_NONE && cp[i] - (uint32_t)b < (uint32_t)x->B) bk[cp[i] - (uint32_t)b] = sl[cp[i] - cp0[i]];   // own store above, same bucket
        const uint32_t off = fxd_bucket_get_staged(bk, t + b, x->A, chk, x->kep);
        const uint32_t old_runp = runp[i];
        cp0[i] = cp[i] = (uint32_t)b + off;
        for (int q = 0; q < 7; q++) sl[q] = bk[off + q];
        if (bpos == 0) {
          int refresh = 0;
          int r0 = rn[0], r1 = rn[1];   // run count of the previous context
          if (old_runp - (uint32_t)b < (uint32_t)x->B) { r0 = bk[old_runp - (uint32_t)b]; r1 = bk[old_runp + 1 - (uint32_t)b]; }   // the search may have replaced the slot that holds them
          if (sl[3] == 2) {  // second visit: create the histories for bits 2-7 of the byte seen the first time
            // the two buckets go through the staging area of the lookup above (done with: its slot is in sl, the run bytes in r0 / r1):
            // one vector fetch and a search on LDS each, instead of a walk over table bytes one dependent load at a time
            const int c = sl[4] + 256;
            uint32_t held = nb;
            for (int v = 0; v < 2; v++) {
              const int cc = v ? c >> 3 : c >> 6, sh3 = v ? 0 : 3;
              const uint32_t nb2 = (cxt[i] + (uint32_t)cc) & x->tmask;
              uint8_t* g2 = t + (size_t)nb2 * (size_t)x->B;
              if (nb2 != held) { fxd_stage_bucket(bk, g2, x->B); held = nb2; }
              const uint32_t o2 = fxd_bucket_get_staged(bk, g2, x->A, chk, x->kep);
              const int cs = c >> sh3;   // v = 0: bits 5, 4, 3 of c;  v = 1: bits 2, 1, 0
              const uint32_t i1 = o2 + 1 + ((cs >> 2) & 1), i2 = o2 + 3 + ((cs >> 1) & 3);
              const uint8_t v0 = (uint8_t)(1 + ((cs >> 2) & 1)), v1 = (uint8_t)(1 + ((cs >> 1) & 1)), v2 = (uint8_t)(1 + (cs & 1));
              bk[o2] = v0; g2[o2] = v0;
              bk[i1] = v1; g2[i1] = v1;
              bk[i2] = v2; g2[i2] = v2;
            }
            t[cp0[i] + 6] = 0; sl[6] = 0;
            refresh = 1;
          }
          if (refresh) { r0 = t[old_runp]; r1 = t[old_runp + 1]; }
          if (r0 == 0) { r0 = 2; r1 = c1; }
          else if (r1 != c1) { r0 = 1; r1 = c1; }
          else if (r0 < 254) r0 += 2;
          t[old_runp] = (uint8_t)r0; t[old_runp + 1] = (uint8_t)r1;
          if (old_runp - cp0[i] < 7u) sl[old_runp - cp0[i]] = (uint8_t)r0;
          if (old_runp + 1 - cp0[i] < 7u) sl[old_runp + 1 - cp0[i]] = (uint8_t)r1;
          if (refresh) for (int q = 0; q < 7; q++) sl[q] = t[cp0[i] + q];
          runp[i] = cp0[i] + 3;
          rn[0] = sl[3]; rn[1] = sl[4];
        }
      }
      state = sl[cp[i] - cp0[i]];
    }
    // every table value of the bit is requested before the first output is stored: the tables are reached through generic
    // pointers, so a load that follows an LDS store would have to wait for it, one L2 round trip per value
    const int bposshift = 7 - bpos, c0shift_bpos = (c0 << 1) ^ (256 >> bposshift);
    const int bb = c0shift_bpos ^ (rn[1] >> bposshift);
    int p1 = 0;
    if (state != 0) {
      uint32_t* smt = sh->sm[x->slot_base + i];
      uint32_t* p = &smt[sm_cxt[i]];
      *p += (uint32_t)(y << 19) - (*p >> 13);
      sm_cxt[i] = state;
      p1 = (int)(smt[state] >> 20);
    }
    const int v_st1 = state ? tab[FX_TAB_ST1 + p1] : 0, v_st2 = (state && x->u) ? tab[FX_TAB_ST2 + p1] : 0;
    const int v_st8 = state ? tab[FX_TAB_ST8 + state] : 0, v_st32 = state ? tab[FX_TAB_ST32 + state] : 0;
    const int v_rc = bb <= 1 ? tab[FX_TAB_RC1 + rn[0] + bb * 256] : 0;
    if (state == 0) {
      ADD(0); if (x->u) ADD(0); ADD(0); ADD(0); ADDQ(64);
    } else {
      ADD(v_st1);
      if (x->u) ADD(v_st2);
      ADD(v_st8);
      ADD(v_st32);
      ADDQ(0);
      result++;
    }
    ADD(v_rc);
  }
#undef ADD
#undef ADDQ
  sh->slot_res[s] = result;
}

// one context of one map for one bit: ContextMap::mix's loop body (mix1 / mix :1110-1173)
FX_HD void fxd_map_ctx(FxDev* d, FxShared* sh, const FxBit& u, int k, int i) {
  const FxMapDev* x = &d->maps[k];
  uint32_t *cp = sh->mcp[k], *cp0 = sh->mcp0[k], *runp = sh->mrunp[k], *cxt = sh->mcxt[k];
  int* sm_cxt = sh->msmc[k];
  int16_t* tx = sh->tx[sh->parity ^ 1] + x->tx_off + i * (5 + x->u);
  float* ex = u.orow + x->exp_off + i * (4 + x->u);
  uint8_t* t = x->t;
  const int16_t* tab = x->tab;
  const int y = u.y, bpos = u.bpos, c0 = u.c0, c1 = u.lastbyte;
  int result = 0;
  {
    const int s = x->slot_base + i;
    int n = 0, e = 0;
#define ADD(v) do { const int v_ = (v); tx[n++] = (int16_t)v_; ex[e++] = fxd_export(d, v_); } while (0)
#define ADDQ(v) do { tx[n++] = (int16_t)(v); } while (0)
    if ((u.rec->skip[s >> 5] >> (s & 31)) & 1) {   // mix4 :1099-1107
      ADD(0); if (x->u) ADD(0); ADD(0); ADD(0); ADDQ(64); ADD(0);
    } else {
      if (cp[i] != FX_NONE) t[cp[i]] = x->nn[t[cp[i]] * 4 + y];
      int state = 0;
      if (bpos > 1 && t[runp[i]] == 0) cp[i] = FX_NONE;
      else {
        const uint16_t chk = (uint16_t)((cxt[i] >> 16) ^ (uint32_t)i);
        if (bpos == 2 || bpos == 5) {
          const size_t b = (size_t)((cxt[i] + (uint32_t)c0) & x->tmask) * (size_t)x->B;
          cp0[i] = cp[i] = (uint32_t)(b + fxd_bucket_get(t + b, x->A, chk, x->kep));
        } else if (bpos) {
          const uint32_t smask = (0x31031010u >> (bpos << 2)) & 0x0F;   // getStateByteLocation :959-964
          cp[i] = cp0[i] + smask + ((uint32_t)c0 & smask);
        } else {
          size_t b = (size_t)((cxt[i] + (uint32_t)c0) & x->tmask) * (size_t)x->B;
          cp0[i] = cp[i] = (uint32_t)(b + fxd_bucket_get(t + b, x->A, chk, x->kep));
          if (t[cp0[i] + 3] == 2) {  // second visit: create the histories for bits 2-7 of the byte seen the first time
            const int c = t[cp0[i] + 4] + 256;
            b = (size_t)((cxt[i] + (uint32_t)(c >> 6)) & x->tmask) * (size_t)x->B;
            uint8_t* p = t + b + fxd_bucket_get(t + b, x->A, chk, x->kep);
            p[0] = (uint8_t)(1 + ((c >> 5) & 1));
            p[1 + ((c >> 5) & 1)] = (uint8_t)(1 + ((c >> 4) & 1));
            p[3 + ((c >> 4) & 3)] = (uint8_t)(1 + ((c >> 3) & 1));
            b = (size_t)((cxt[i] + (uint32_t)(c >> 3)) & x->tmask) * (size_t)x->B;
            p = t + b + fxd_bucket_get(t + b, x->A, chk, x->kep);
            p[0] = (uint8_t)(1 + ((c >> 2) & 1));
            p[1 + ((c >> 2) & 1)] = (uint8_t)(1 + ((c >> 1) & 1));
            p[3 + ((c >> 1) & 3)] = (uint8_t)(1 + (c & 1));
            t[cp0[i] + 6] = 0;
          }
          uint8_t* run = t + runp[i];  // run count of the previous context
          if (run[0] == 0) { run[0] = 2; run[1] = (uint8_t)c1; }
          else if (run[1] != c1) { run[0] = 1; run[1] = (uint8_t)c1; }
          else if (run[0] < 254) run[0] = (uint8_t)(run[0] + 2);
          runp[i] = cp0[i] + 3;
        }
        state = t[cp[i]];
      }
      // (the table values are requested together, before the first output is stored: see fxd_map_ctx_fast)
      const uint8_t* run = t + runp[i];
      const int run0 = run[0], run1 = run[1];
      const int bposshift = 7 - bpos, c0shift_bpos = (c0 << 1) ^ (256 >> bposshift);
      const int b = c0shift_bpos ^ (run1 >> bposshift);
      int p1 = 0;
      if (state != 0) {
        uint32_t* smt = sh->sm[x->slot_base + i];   // StateMap::set :693-700
        uint32_t* p = &smt[sm_cxt[i]];
        *p += (uint32_t)(y << 19) - (*p >> 13);
        sm_cxt[i] = state;
        p1 = (int)(smt[state] >> 20);
      }
      const int v_st1 = state ? tab[FX_TAB_ST1 + p1] : 0, v_st2 = (state && x->u) ? tab[FX_TAB_ST2 + p1] : 0;
      const int v_st8 = state ? tab[FX_TAB_ST8 + state] : 0, v_st32 = state ? tab[FX_TAB_ST32 + state] : 0;
      const int v_rc = b <= 1 ? tab[FX_TAB_RC1 + run0 + b * 256] : 0;
      if (state == 0) {  // mix3 :1077-1097
        ADD(0); if (x->u) ADD(0); ADD(0); ADD(0); ADDQ(64);
      } else {
        ADD(v_st1);
        if (x->u) ADD(v_st2);
        ADD(v_st8);
        ADD(v_st32);
        ADDQ(0);
        result++;
      }
      ADD(v_rc);
    }
    sh->slot_res[s] = result;
  }
}


// Which buckets will context i of map k touch in this bit? Read-only. Serial order inside a map only matters when two of
// its contexts touch the same bucket (checksum replacement, the last-used byte, shared state bytes), so: every context
// lane puts its buckets into an LDS hash set (fxd_map_touch: a pair that is already there = an overlap), and a map with an
// overlap is walked by its first lane in context order while all other maps run one lane per context (fxd_map_run).
// The list: the bucket of the state byte being updated, the bucket holding the run bytes, at bits 0 / 2 / 5 the bucket
// about to be looked up, and at bit 0 the two buckets a second visit creates histories in -- known from a read-only
// look at the slot the lookup will return, which is exact unless an earlier context writes that bucket first, i.e.
// unless there is an overlap.
FX_HD void fxd_map_touch(FxDev* d, FxShared* sh, const FxBit& u, int s, uint32_t* tab, uint32_t hmask) {   // tab / hmask: the bit's hash set (cleared)
  const int k = d->slot_map[s], i = d->slot_idx[s];
  const FxMapDev* x = &d->maps[k];
  const int par = u.q & 1;
  // (fixed positions, FX_NOKEY = unused, loops with constant bounds: a list filled through a running index is an array in scratch memory on
  // the device, every access a global-memory round trip)
  uint32_t K0 = FX_NOKEY, K1 = FX_NOKEY, K2 = FX_NOKEY, K3 = FX_NOKEY, K4 = FX_NOKEY;
  if (!u.normal) return;
  if (u.boundary) sh->mcxt[k][i] = u.rec->cx[s];
  if ((u.rec->skip[s >> 5] >> (s & 31)) & 1) return;
  const int sh_b = x->B == 32 ? 5 : x->B == 64 ? 6 : 7;
  const uint32_t cp = sh->mcp[k][i], runp = sh->mrunp[k][i], cxt = sh->mcxt[k][i];
  const int bpos = u.bpos;
  const int lookbit = bpos == 0 || bpos == 2 || bpos == 5;
  if (lookbit) {   // searches read and rewrite bucket headers: contexts interact through a shared BUCKET
    if (cp != FX_NONE) K0 = cp >> sh_b;
    K1 = runp >> sh_b;
  } else {         // between searches a context only touches its own slot's bytes: they interact through a shared SLOT only
    const uint32_t first = (uint32_t)(2 * x->A + 1), bm = (uint32_t)x->B - 1, cur0 = sh->mcp0[k][i], run0 = runp - 3;
    if (cp != FX_NONE) K0 = ((cur0 >> sh_b) << 4) | (((cur0 & bm) - first) / 7);
    K1 = ((run0 >> sh_b) << 4) | (((run0 & bm) - first) / 7);
  }
  sh->mlook[s] = 0;
  if (!(bpos > 1 && sh->mrun[s][0] == 0) && lookbit) {
    const uint32_t nb = (cxt + (uint32_t)u.c0) & x->tmask;
    K2 = nb;
    sh->mlook[s] = 1;
    fxd_stage_bucket(sh->mbk[s], x->t + (size_t)nb * (size_t)x->B, x->B);   // the bucket about to be searched -> LDS
    if (bpos == 0) {
      const uint8_t* b = sh->mbk[s];
      const uint16_t* chk = (const uint16_t*)b;
      const uint16_t ch = (uint16_t)((cxt >> 16) ^ (uint32_t)i);
      const int A = x->A, last = b[2 * A];
      int slot = -1;
      {   // every checksum is read before the first comparison (independent loads)
        uint16_t cj[14];
        const uint16_t clast = chk[last & 15];
#ifdef __HIPCC__
#pragma unroll
#endif
        for (int j = 0; j < 14; ++j) cj[j] = chk[j < A ? j : A - 1];
#ifdef __HIPCC__
#pragma unroll
#endif
        for (int j = 13; j >= 0; --j) if (j < A && cj[j] == ch) slot = j;
        if (clast == ch) slot = last & 15;
      }
      if (slot >= 0 && b[2 * A + 1 + 7 * slot + 3] == 2) {
        const int c = b[2 * A + 1 + 7 * slot + 4] + 256;
        K3 = (cxt + (uint32_t)(c >> 6)) & x->tmask;
        K4 = (cxt + (uint32_t)(c >> 3)) & x->tmask;
      }
    }
  }
  // into the hash set: a key another context of the map has put there = an overlap (this lane's own repeats are dropped first)
  if (K1 == K0) K1 = FX_NOKEY;
  if (K2 == K0 || K2 == K1) K2 = FX_NOKEY;
  if (K3 == K0 || K3 == K1 || K3 == K2) K3 = FX_NOKEY;
  if (K4 == K0 || K4 == K1 || K4 == K2 || K4 == K3) K4 = FX_NOKEY;
#define FX_PUT(K)                                                                   \
  if ((K) != FX_NOKEY) {                                                            \
    const uint32_t key = ((uint32_t)(k + 1) << 26) | (K);                           \
    uint32_t h = ((key * 2654435761u) >> 22) & hmask;                               \
    for (;;) {                                                                      \
      const uint32_t old = FX_CAS(&tab[h], 0u, key);                                \
      if (old == 0) break;                                                          \
      if (old == key) { sh->mconf[par][k] = 1; break; }                             \
      h = (h + 1) & hmask;                                                          \
    }                                                                               \
  }
  FX_PUT(K0) FX_PUT(K1) FX_PUT(K2) FX_PUT(K3) FX_PUT(K4)
#undef FX_PUT
}
// the other parity's hash set and flags are cleared for the next bit (any phase after the maps have run)
FX_HD void fxd_map_clear_next(FxShared* sh, const FxBit& u, int tid) {
  uint32_t* tab = sh->ohash[(u.q + 1) & 1];
  for (int i = tid; i < 1024; i += FX_THREADS) tab[i] = 0;
  if (tid < FX_NMAPS) sh->mconf[(u.q + 1) & 1][tid] = 0;
}
FX_HD void fxd_map_run(FxDev* d, FxShared* sh, const FxBit& u, int s) {
  const int k = d->slot_map[s], i = d->slot_idx[s];
  if (!u.normal) { sh->slot_res[s] = 0; return; }
  if (!sh->mconf[u.q & 1][k] && d->slot_parallel) fxd_map_ctx_fast(d, sh, u, k, i);
  else if (i == 0) {   // overlap: the map's first lane walks it on the table in the reference's order, then the cached bytes are read back
    for (int j = 0; j < d->maps[k].C; j++) fxd_map_ctx(d, sh, u, k, j);
    for (int j = 0; j < d->maps[k].C; j++) fxd_map_reload(d, sh, k, j);
  }
}

// ---------------------------------------------------------------- SmallStationaryContextMap :831-863
FX_HD void fxd_sscm_unit(FxDev* d, FxShared* sh, const FxBit& u, int j) {
  int16_t* tx = sh->tx[sh->parity ^ 1] + 2 * j;
  if (u.boundary) { d->sscm_ctx[j] = (int)(u.rec->sscm[j] & (uint32_t)d->sscm_mask[j]) * 255; d->sscm_bcount[j] = d->sscm_B[j] = 0; }
  const int rate = u.sscmrate + 7;
  uint16_t* data = d->sscm_data[j];
  uint16_t* cp = &data[d->sscm_cp[j]];
  *cp = (uint16_t)(*cp + (((u.y << 16) - (*cp) + (1 << (rate - 1))) >> rate));
  int B = d->sscm_B[j];
  B += (u.y && B > 0);
  d->sscm_cp[j] = d->sscm_ctx[j] + B;
  const int pred = data[d->sscm_cp[j]] >> 4;
  const int v = d->stretch[pred] / 4;
  tx[0] = (int16_t)v; u.orow[j] = fxd_export(d, v);
  tx[1] = (int16_t)((pred - 2048) / 8);
  d->sscm_bcount[j]++; B += B + 1;
  if (d->sscm_bcount[j] == 8) d->sscm_bcount[j] = B = 0;
  d->sscm_B[j] = B;
}

// ---------------------------------------------------------------- RunContextMap :756-829
FX_HD uint32_t fxd_rcm_find(FxDev* d, uint32_t i) {  // offset of byte 1 of the element; 4-byte elements, 4-way probe, move to front
  uint32_t* t = (uint32_t*)(uint8_t*)d->rcm_t;
  const uint32_t chk = ((i >> 16) ^ i) & 0xffff;
  i = (i * 4) & d->rcm_n;
  int j;
  uint32_t found = 0;
  for (j = 0; j < 4; ++j) {
    const uint32_t el = t[i + (uint32_t)j];
    if (((el >> 16) & 0xff) == 0) { found = (el & 0xffff0000u) | chk; t[i + (uint32_t)j] = found; break; }
    if ((el & 0xffff) == chk) { found = el; break; }
  }
  if (j == 0) return i * 4 + 1;
  if (j == 4) {
    --j;
    found = chk;
    if (((t[i + 3] >> 16) & 0xff) > ((t[i + 2] >> 16) & 0xff)) --j;
  }
  for (int k = j; k > 0; --k) t[i + (uint32_t)k] = t[i + (uint32_t)k - 1];
  t[i] = found;
  return i * 4 + 1;
}
FX_HD void fxd_rcm_unit(FxDev* d, FxShared* sh, const FxBit& u) {
  const FxLayout l = fxd_layout(d, u.normal);
  uint8_t* t = d->rcm_t;
  if (u.boundary) {   // set :788-795 with the parser's c1
    uint8_t* cp = t + d->rcm_cp;
    const int c1 = u.rec->pc1;
    if (cp[0] == 0) { cp[0] = 2; cp[1] = (uint8_t)c1; }
    else if (cp[1] != c1) { cp[0] = 1; cp[1] = (uint8_t)c1; }
    else if (cp[0] < 254) cp[0] = (uint8_t)(cp[0] + 2);
    d->rcm_cp = fxd_rcm_find(d, u.rec->rcm_cx) + 1;
  }
  const uint8_t* cp = t + d->rcm_cp;
  const int bposshift = 7 - u.bpos, c0shift_bpos = (u.c0 << 1) ^ (256 >> bposshift);
  const int b = c0shift_bpos ^ (cp[1] >> bposshift);
  const int v = b <= 1 ? d->rcm_rc[b * 256 + cp[0]] : 0;
  sh->tx[sh->parity ^ 1][l.tx_rcm] = (int16_t)v;
  u.orow[l.exp_rcm] = fxd_export(d, v);
  u.orow[l.exp_rcm + 1] = fxd_export(d, 64);   // the constant input's export (:4612)
}

// ---------------------------------------------------------------- the match lane: byte history, MatchModel2, SparseMatchModel
#define FXB(i) ((int)d->buffer[((uint32_t)d->pos - (uint32_t)(i)) & FX_BMASK])
#define FXBR(i) ((int)d->buffer[(uint32_t)(i) & FX_BMASK])
FX_HD int fxd_statemap1(FxDev* d, int j, int y, int c) {
  uint32_t* t = d->sm1_t[j];
  uint32_t p0 = t[d->sm1_cxt[j]];
  const int n = (int)(p0 & 1023);
  const uint32_t pr1 = p0 >> 12;
  p0 += (uint32_t)(n < 1023);
  p0 += (((uint32_t)(y << 20) - pr1) * (uint32_t)fxd_dt(n) + 512) & 0xfffffc00u;
  t[d->sm1_cxt[j]] = p0;
  d->sm1_cxt[j] = c & d->sm1_mask[j];
  return (int)(t[d->sm1_cxt[j]] >> 20);
}
FX_HD uint32_t fxd_mi_prio(const FxMatchInfo* c) {
  return (uint32_t)(c->length != 0) << 31 | (uint32_t)c->delta << 30 | (c->delta ? (c->lengthBak >> 1) : (c->length >> 1)) << 24 | (c->index & 0x00ffffff);
}
// part A: the candidates (:3447-3566) and the three StateMap1 contexts; part B (fxd_match2_sm): one StateMap1 read-out each
FX_HD void fxd_match2(FxDev* d, FxShared* sh, const FxBit& u, int16_t* tx, float* ex, int* isMatch) {
  enum { MAXLEN = 62, MINLEN_RM = 3, LEN1 = 5, LEN2 = 7, LEN3 = 9 };
  const int pc1 = u.rec->pc1;
  const uint32_t n = (uint32_t)fxd_max((int)d->nActive, 1);
  for (uint32_t i = 0; i < n; i++) {
    FxMatchInfo* c = &d->cand[i];
    if (c->length != 0) {   // MatchInfo::update :3447-3490
      const int expectedBit = (c->expectedByte >> ((8 - u.bpos) & 7)) & 1;
      if (u.y != expectedBit) {
        if (c->length != 0 && c->lengthBak != 0) { c->lengthBak = 0; c->indexBak = 0; }
        else { c->lengthBak = c->length; c->indexBak = c->index; c->delta = 1; }
        c->length = 0;
      }
    }
    if (u.bpos == 0) {
      if (c->length == 0 && !c->delta && c->lengthBak != 0) {   // one byte after the mismatch: try to pick the match up again
        c->indexBak++;
        if (c->lengthBak < MAXLEN) c->lengthBak++;
        if (FXBR(c->indexBak) == pc1) { c->length = c->lengthBak; c->index = c->indexBak; }
        else c->lengthBak = c->indexBak = 0;
      }
      if (c->length != 0) {
        c->index++;
        if (c->length < MAXLEN) c->length++;
        if (c->length != 0 && c->lengthBak != 0 && c->length - c->lengthBak >= MINLEN_RM) c->lengthBak = c->indexBak = 0;
      }
      c->delta = 0;
    }
    if (d->nActive != 0 && c->length == 0 && !c->delta && c->lengthBak == 0) {
      d->nActive--;
      if (d->nActive == i) break;
      for (uint32_t k = i; k < d->nActive; k++) d->cand[k] = d->cand[k + 1];
      i--;
    }
  }
  if (u.bpos == 0) {
    const uint32_t lens[4] = {LEN3, LEN2, LEN1, LEN1};
    for (int k = 0; k < 4; k++) {
      uint32_t* slot = d->mhash + 4 * (size_t)(u.rec->mh[k] & d->mhashmask);
      if (d->nActive < 4) {   // AddCandidates :3540-3566
        const uint32_t LEN = lens[k];
        for (uint32_t i = 0; d->nActive < 4 && i < 4; i++) {
          const uint32_t matchpos = slot[i];
          if (matchpos == 0) break;
          int ok = 1;
          for (int length = 1; length <= (int)LEN; length++) if (FXB(length) != FXBR(matchpos - (uint32_t)length)) { ok = 0; break; }
          if (!ok) continue;
          int same = 0;
          for (uint32_t q = 0; q < d->nActive; q++) if (d->cand[q].index == matchpos) { same = 1; break; }
          if (!same) {
            FxMatchInfo* c = &d->cand[d->nActive++];
            c->length = LEN - LEN1 + 1; c->index = matchpos; c->lengthBak = c->indexBak = 0; c->expectedByte = 0; c->delta = 0;
          }
        }
      }
      slot[3] = slot[2]; slot[2] = slot[1]; slot[1] = slot[0]; slot[0] = (uint32_t)d->pos;
    }
    for (uint32_t i = 0; i < d->nActive; i++) d->cand[i].expectedByte = (uint8_t)FXBR(d->cand[i].index);
  }
  uint32_t ctx[3] = {0, 0, 0};
  int best = 0;
  for (uint32_t i = 1; i < d->nActive; i++) if (fxd_mi_prio(&d->cand[i]) > fxd_mi_prio(&d->cand[best])) best = (int)i;
  const uint32_t length = d->cand[best].length;
  const uint32_t expectedByte = d->cand[best].expectedByte;
  const int delta = d->cand[best].delta;
  const int expectedBit = length != 0 ? (int)(expectedByte >> (7 - u.bpos)) & 1 : 0;
  int v = 0;
  if (length != 0) {
    const uint32_t dense = length <= 16 ? length - 1 : 12 + (length >> 2);
    ctx[0] = (dense << 4) | ((uint32_t)expectedBit << 3) | (uint32_t)u.bpos;
    ctx[1] = (expectedByte << 11) | ((uint32_t)u.bpos << 8) | (uint32_t)pc1;
    v = (2 * expectedBit - 1) * (int)(length << 5);
  }
  tx[0] = (int16_t)v; ex[0] = fxd_export(d, v);
  if (delta) ctx[2] = (expectedByte << 8) | (uint32_t)u.c0;
  for (int i = 0; i < 3; i++) sh->m_ctx[i] = ctx[i];
  *isMatch = (int)length;
}
FX_HD void fxd_match2_sm(FxDev* d, FxShared* sh, const FxBit& u, int i) {   // i = 0..2 (:3620-3640)
  int16_t* tx = sh->tx[sh->parity ^ 1] + 2 * FX_NSSCM;
  float* ex = u.orow + FX_NSSCM;
  int a = 0, b = 0;
  if (sh->m_ctx[i] != 0) {
    const int p1 = fxd_statemap1(d, i, u.y, (int)sh->m_ctx[i]);
    a = d->stretch[p1] >> 2; b = (p1 - 2048) >> 3;
  }
  tx[1 + 2 * i] = (int16_t)a; ex[1 + 2 * i] = fxd_export(d, a);
  tx[2 + 2 * i] = (int16_t)b; ex[2 + 2 * i] = fxd_export(d, b);
}
FX_HD void fxd_mtf_front(FxMtf* l, int i) {  // MTFList::MoveToFront :1715-1731
  if ((l->index = i) == l->root) return;
  const int p = l->prev[i], n = l->next[i];
  if (p >= 0) l->next[p] = l->next[i];
  if (n >= 0) l->prev[n] = l->prev[i];
  l->prev[l->root] = i;
  l->next[i] = l->root;
  l->root = i;
  l->prev[l->root] = -1;
}
FX_HD void fxd_sparse(FxDev* d, const FxBit& u, int16_t* tx, float* ex) {
  const uint32_t strides[4] = {1, 1, 2, 1}, minlens[4] = {3, 4, 6, 5};
  const uint32_t tmask = 1024 * 1024 - 1;
  if (u.bpos == 0) {   // update :1760-1805
    for (uint32_t i = 0; i < 4; i++) {
      uint32_t h = (i + 1) * 191;
      for (uint32_t j = 0, k = 1; j < minlens[i]; j++, k += strides[i]) h = h * 191 + ((uint32_t)FXB(k) << i);
      d->sp_hashes[i] = h & tmask;
    }
    if (d->sp_length) {
      d->sp_index++;
      if (d->sp_length < 64) d->sp_length++;
    } else {
      FxMtf* l = &d->sp_list;
      for (int i = l->index = l->root; i >= 0; i = (l->index >= 0 ? (l->index = l->next[l->index]) : l->index)) {
        d->sp_index = d->sp_table[d->sp_hashes[i]];
        if (d->sp_index > 0) {
          uint32_t offset = 1;
          while (d->sp_length < minlens[i] && (FXB(offset) ^ FXBR(d->sp_index - offset)) == 0) { d->sp_length++; offset += strides[i]; }
          if (d->sp_length >= minlens[i]) {
            d->sp_length -= minlens[i] - 1;
            d->sp_hashIndex = (uint32_t)i;
            fxd_mtf_front(l, i);
            break;
          }
        }
        d->sp_length = d->sp_index = 0;
      }
    }
    for (uint32_t i = 0; i < 4; i++) d->sp_table[d->sp_hashes[i]] = (uint32_t)d->pos;
    d->sp_expectedByte = (uint8_t)FXBR(d->sp_index);
    d->sp_valid = d->sp_length > 1;
  }
  const uint8_t B = (uint8_t)(u.c0 << (8 - u.bpos));
  if (d->sp_length > 0 && ((d->sp_expectedByte ^ B) >> (8 - u.bpos)) != 0) d->sp_length = 0;
  int a = 0, b = 0;
  if (d->sp_valid && d->sp_length > 1) {
    const int expectedBit = (d->sp_expectedByte >> (7 - u.bpos)) & 1, sign = 2 * expectedBit - 1;
    const int l1 = (int)d->sp_length - 1, l2 = (int)d->sp_length - 2;
    a = sign * ((l1 < 32 ? l1 : 32) << 5);
    b = sign * (1 << (l2 < 3 ? l2 : 3)) * (l1 < 8 ? l1 : 8) << 4;
  }
  tx[0] = (int16_t)a; ex[0] = fxd_export(d, a);
  tx[1] = (int16_t)b; ex[1] = fxd_export(d, b);
}
FX_HD void fxd_match_unit(FxDev* d, FxShared* sh, const FxBit& u) {
  if (u.boundary) { d->buffer[(uint32_t)d->pos & FX_BMASK] = (uint8_t)u.lastbyte; d->pos++; }   // :3806-3807
  int16_t* tx = sh->tx[sh->parity ^ 1] + 2 * FX_NSSCM;
  float* ex = u.orow + FX_NSSCM;
  fxd_match2(d, sh, u, tx, ex, &sh->isMatch);
  fxd_sparse(d, u, tx + 7, ex + 7);
}
#undef FXB
#undef FXBR

// ---------------------------------------------------------------- Mixer1 :472-660
FX_HD int fxd_mixer_err(const FxShared* sh, const FxDev* d, const FxBit& u, int k) {
  int elim = sh->mx_elim[k];
  if (u.boundary && k < FX_NMIX1) {   // update1 :4765-4771: the dead zone follows the recent failures
    if ((sh->fails & 255) == 0) elim = fxd_max(256, elim + 1);
    else elim = fxd_max(0, fxd_min(16, elim - 1));
  }
  int err = ((u.y << 12) - sh->mx_pr[k]) * d->mx_uperr[k] / 4;
  if (err > 32767) err = 32767;
  if (err < -32768) err = -32768;
  if (err >= -elim && err <= elim) err = 0;
  return err;
}
FX_HD int16_t fxd_train1(int t, int w, int err) {   // train, SSE2 form :543-557
  int v = fxd_sat16(2 * t);
  v = (v * err) >> 16;
  v = fxd_sat16(v + 1) >> 1;
  return (int16_t)fxd_sat16(v + w);
}
FX_HD void fxd_train_small(FxDev* d, FxShared* sh, const FxBit& u, int k) {   // mixers 10 / 11: 16 inputs
  const int err = fxd_mixer_err(sh, d, u, k);
  if (!err) return;
  int16_t* w = d->wx[k] + (size_t)sh->mx_cxt[k] * 16;
  for (int i = 0; i < 16; i++) w[i] = fxd_train1(sh->in2[i], w[i], (int16_t)err);
}
// trainer thread j of nj: pairs j, j + nj, ... of the 10 x 256 pairs of the first-layer rows, on the previous inputs
FX_HD void fxd_train_rows(FxDev* d, FxShared* sh, const FxBit& u, int j, int nj) {
  const int16_t* tx = sh->tx[sh->parity];
  for (int p = j; p < FX_NMIX1 * (FX_TX / 2); p += nj) {
    const int k = p >> 8, i = (p & 255) * 2;
    const int err = fxd_mixer_err(sh, d, u, k);
    if (!err) continue;
    int16_t* w = d->wx[k] + (size_t)sh->mx_cxt[k] * FX_TX + i;
    const int16_t w0 = fxd_train1(tx[i], w[0], (int16_t)err), w1 = fxd_train1(tx[i + 1], w[1], (int16_t)err);
    w[0] = w0; w[1] = w1;
  }
}
FX_HD int fxd_dot16(const int16_t* t, const int16_t* w) {
  uint32_t sum = 0;
  for (int i = 0; i < 16; i += 2) {
    const uint32_t pair = (uint32_t)((int32_t)t[i] * w[i]) + (uint32_t)((int32_t)t[i + 1] * w[i + 1]);
    sum += (uint32_t)((int32_t)pair >> 8);
  }
  return (int32_t)sum;
}

// ---------------------------------------------------------------- APM :1622-1643, split: cell update (phase 1) and lookup (phase 5)
FX_HD void fxd_apm_update(FxDev* d, FxShared* sh, const FxBit& u, int j) {
  const int rate = j == 0 ? 3 : j == 1 ? u.rate + 1 : u.rate;
  uint16_t* t = d->apm_t[j] + sh->apm_index[j];
  const int g = (u.y << 16) + (u.y << rate) - u.y * 2;
  t[0] = (uint16_t)(t[0] + ((g - t[0]) >> rate));
  t[1] = (uint16_t)(t[1] + ((g - t[1]) >> rate));
}
FX_HD int fxd_apm_p(FxDev* d, FxShared* sh, int j, int pr, int cxt) {
  pr = d->stretch[pr];
  const int w = pr & 127;
  const int idx = ((pr + 2048) >> 7) + cxt * 33;
  sh->apm_index[j] = idx;
  const uint16_t* t = d->apm_t[j] + idx;
  return (t[0] * (128 - w) + t[1] * w) >> 11;
}

// ---------------------------------------------------------------- the phases
// phase 1 in three barrier-separated steps: a = bucket lists + everything that is not a context map, b = overlap check, c = the maps
FX_HD void fxd_phase1a(FxDev* d, FxShared* sh, const FxBit& u, int tid) {
  if (tid < FX_NSLOTS) fxd_map_touch(d, sh, u, tid, sh->ohash[u.q & 1], 1023u);
  else if (tid < FX_U_MATCH) fxd_sscm_unit(d, sh, u, tid - FX_U_SSCM);
  else if (tid == FX_U_MATCH) { fxd_match_unit(d, sh, u); for (int i = 0; i < 3; i++) fxd_match2_sm(d, sh, u, i); }
  else if (tid == FX_U_RCM) fxd_rcm_unit(d, sh, u);
  else if (tid == FX_U_LSTM) { const FxLayout l = fxd_layout(d, u.normal); sh->tx[sh->parity ^ 1][l.tx_lstm] = d->stretch[u.lstmpr]; }
  else if (tid == FX_U_MIX10 || tid == FX_U_MIX11) fxd_train_small(d, sh, u, 10 + tid - FX_U_MIX10);
  else if (tid < FX_UNITS) fxd_apm_update(d, sh, u, tid - FX_U_APM);
  else if (tid >= FX_TRAIN0) fxd_train_rows(d, sh, u, tid - FX_TRAIN0, FX_THREADS - FX_TRAIN0);
}
FX_HD void fxd_phase1c(FxDev* d, FxShared* sh, const FxBit& u, int tid) { if (tid < FX_NSLOTS) fxd_map_run(d, sh, u, tid); }
FX_HD void fxd_phase2(FxDev* d, FxShared* sh, const FxBit& u, int tid) {
  if (tid != 0) return;
  const int e_l[8] = {1830, 1997, 1973, 1851, 1897, 1690, 1998, 1842};   // :3222
  if (u.boundary) {
    for (int i = 0; i < FX_NMIX1; i++)
      sh->mx_elim[i] = (sh->fails & 255) == 0 ? fxd_max(256, sh->mx_elim[i] + 1) : fxd_max(0, fxd_min(16, sh->mx_elim[i] - 1));
  }
  if (sh->fails & 0x00000080) --sh->failcount;   // update1 :4784-4790
  sh->fails = sh->fails * 2;
  sh->failz = sh->failz * 2;
  int pr = sh->pr;
  if (u.y) pr = 4095 - pr;
  if (pr >= e_l[u.bpos]) { ++sh->fails; ++sh->failcount; }
  if (pr >= 848) ++sh->failz;

  const FxByteRec* r = u.rec;
  const int bpos = u.bpos, c0b = u.c0 << (8 - bpos);
  int ordX = 0, ordW = 0;
  if (u.normal) {   // :4601-4640; maps are in mixing order, see FX_MAPS
    const int b0 = d->maps[0].slot_base;
    int skipped = 0;
    for (int i = 0; i < 3; i++) skipped |= (int)((r->skip[(b0 + i) >> 5] >> ((b0 + i) & 31)) & 1);
    if (skipped) ordX = 2;
    int mr[6];   // ContextMap::mix's return value (contexts with a known state) of cmC2[0..5], cmC2[13], cmC2[14]
    const int which[8] = {0, 1, 2, 3, 4, 5, 21, 23};
    int res8[8];
    for (int q = 0; q < 8; q++) {
      const FxMapDev* x = &d->maps[which[q]];
      int v = 0;
      for (int i = 0; i < x->C; i++) v += sh->slot_res[x->slot_base + i];
      res8[q] = v;
    }
    for (int q = 0; q < 6; q++) mr[q] = res8[q];
    ordX += mr[0];
    if (ordX == 3) ordX = 2;
    ordX += mr[1] + mr[2] + mr[3];
    ordW = mr[4] + mr[5];
    if (ordW > 3) ordW = 3;
    ordW += res8[6] + res8[7];
  }
  const uint32_t s2 = r->s2, s3 = r->s3, s3R = r->s3R, BrFc = r->BrFc, words = r->words, FcIdx = r->FcIdx, isPar = r->isPar;
  const uint32_t isMatch = (uint32_t)sh->isMatch;
  const uint8_t* w2b = d->wrt;
  const uint8_t* w3b = w2b + 256;
  int* cx = sh->mx_cxt;
  int c;
  if (bpos == 0) cx[0] = (int)((s2 & 255) * 8 + (s3 & 7));
  else if (bpos > 3) cx[0] = (int)((((s2 << 2) & 255) + w2b[c0b & 255]) * 8 + BrFc);
  else cx[0] = (int)((s2 & 255) * 8 + BrFc);
  if (bpos) {
    c = c0b;
    if (bpos == 1) c = c + 16 * (int)(words * 2 & 4);
    else if (bpos > 3) c = w2b[c0b & 255] * 64;
    c = fxd_min(bpos, 5) * 256 + (int)(s3R & 7) + (int)FcIdx * 8 + (c & 192);
  } else c = (int)((words & 12) * 16 + (s3R & 7) + BrFc * 8);
  cx[1] = c;
  cx[2] = (int)(((4 * words) & 0xf0) * 4 + (uint32_t)ordX * 256 * 4 + (s2 & 63));
  cx[6] = (int)((s3R & 0xff8) * 4 + ((2 * words) & 0x1c) + (s2 & 3));
  c = c0b;
  cx[3] = bpos * 256 + (int)((((((uint32_t)r->numbers | words) << bpos) & 255) >> bpos) | ((uint32_t)c & 255));
  cx[10] = (int)(((uint32_t)ordX * 8 + (BrFc ? 1u : 0u) * 4 + (s2 & 3)) * 2 + (words & 1));
  if (bpos) {
    if (bpos == 1) c = c + 16 * (int)(s3 & 7);
    else if (bpos == 2) c = c + 16 * (int)(s2 & 3);
    else if (bpos == 3) c = c + 16 * (int)(words & 1);
    else c = bpos + (c & 0xf0);
    if (bpos < 5) c = bpos + (c & 0xf0);
  } else c = 16 * (int)(s2 & 0xf);
  ordX = ordX - 1;
  if (ordX < 0) ordX = 0;
  if (isMatch) ordX = ordX + 1;
  cx[4] = c + ordX * 256 + 8 * (int)isPar;
  cx[5] = (int)(((uint32_t)ordW * 256 + (s2 & 0xf0) + ((s3 & 0x38) >> 2)) * 4 + FcIdx);
  if (bpos > 2) cx[7] = (int)(((s3 & 7) * 8 + w3b[c0b & 255]) * 256 + BrFc * 32 + (words & 7) * 4 + isPar + (isMatch ? 2u : 0u));
  else cx[7] = (int)(((s3 & 63) * 256 + BrFc * 16 + (words & 7) * 2 + isPar) | (isMatch ? 128u : 0u));
  cx[8] = (int)r->deccode;
  cx[9] = (bpos << 8) * 4 + (int)(sh->fails & 3) * 256 + u.lstmex;
  cx[11] = 0;
}
FX_HD void fxd_phase3(FxDev* d, FxShared* sh, const FxBit& u, int tid) {   // dot_product, SSE2 form :522-541: one pair per thread
  fxd_map_clear_next(sh, u, tid);
  const int16_t* tx = sh->tx[sh->parity ^ 1] + 2 * tid;
  const int t0 = tx[0], t1 = tx[1];
  for (int k = 0; k < FX_NMIX1; k++) {
    const int16_t* w = d->wx[k] + (size_t)sh->mx_cxt[k] * FX_TX + 2 * tid;
    const uint32_t pair = (uint32_t)(t0 * (int32_t)w[0]) + (uint32_t)(t1 * (int32_t)w[1]);
    sh->part[k][tid] = (int32_t)pair >> 8;
  }
}
FX_HD void fxd_phase4(FxDev*, FxShared* sh, const FxBit&, int tid) {
  if (tid >= FX_NMIX1 * 16) return;
  const int k = tid >> 4, g = tid & 15;
  uint32_t s = 0;
  for (int j = 0; j < 16; j++) s += (uint32_t)sh->part[k][g * 16 + j];
  sh->part2[k][g] = (int32_t)s;
}
FX_HD void fxd_phase5(FxDev* d, FxShared* sh, const FxBit& u, int tid) {
  if (tid != 0) return;
  const FxLayout l = fxd_layout(d, u.normal);
  float* ex = u.orow + l.exp_mix;
  for (int k = 0; k < FX_NMIX1; k++) {   // p1 :641-651, mxInputs2.add
    uint32_t s = 0;
    for (int g = 0; g < 16; g++) s += (uint32_t)sh->part2[k][g];
    int dp = (int32_t)(s * (uint32_t)d->mx_shift[k]) >> 11;
    dp = fxd_clp(dp);
    sh->mx_pr[k] = fxd_squash(d, dp);
    sh->in2[k] = (int16_t)dp;
    ex[k] = fxd_export(d, dp);
  }
  sh->in2[10] = (int16_t)(d->stretch[u.lstmpr] / 2);
  int dpf[2];
  for (int k = 10; k < 12; k++) {
    int dp = (int32_t)((uint32_t)fxd_dot16(sh->in2, d->wx[k] + (size_t)sh->mx_cxt[k] * 16) * (uint32_t)d->mx_shift[k]) >> 11;
    dp = fxd_clp(dp);
    sh->mx_pr[k] = fxd_squash(d, dp);
    dpf[k - 10] = dp;
  }
  int pr = fxd_squash(d, (dpf[0] * 7 + dpf[1] + 4) >> 3);
  ex += FX_NMIX1;
#define EXPV(v) (*ex++ = (float)(v) * (float)(1.0 / 4095))
  EXPV(pr);
  // update1 :4792-4833: six APMs and the blend
  const uint32_t tri[4] = {0, 4, 3, 7}, trj[4] = {0, 6, 6, 12};
  const FxByteRec* r = u.rec;
  const int c0 = u.c0;
  const uint32_t fails = sh->fails;
  int pt, pu = (fxd_apm_p(d, sh, 0, pr, c0) + 7 * pr + 4) >> 3, pv, pz = (int)sh->failcount + 1;
  pz += (int)tri[(fails >> 5) & 3];
  pz += (int)trj[(fails >> 3) & 3];
  pz += (int)trj[(fails >> 1) & 3];
  if (fails & 1) pz += 8;
  pz = pz / 2;
  pu = fxd_apm_p(d, sh, 3, pu, (int)((((uint32_t)c0 * 2) ^ r->AH1) & 0x3ffff)); EXPV(pu);
  pv = fxd_apm_p(d, sh, 1, pr, (int)((((uint32_t)c0 * 8) ^ fxd_hash3(29, sh->failz & 2047, 0xffffffffu)) & 0xffff)); EXPV(pv);
  if (fails & 255) pv = fxd_apm_p(d, sh, 4, pv, (int)(fxd_hash3((uint32_t)c0, r->s2 & 0xfffc, r->s3R & 0x1ff) & 0x3ffff));
  else pv = fxd_apm_p(d, sh, 4, pv, (int)(fxd_hash3((uint32_t)c0, (r->s2R & 0xfffc) + 0x10000, r->s3R & 0x1ff) & 0x3ffff));
  EXPV(pv);
  pt = fxd_apm_p(d, sh, 2, pr, (int)((((uint32_t)c0 * 32) ^ r->AH2) & 0xffff)); EXPV(pt);
  pz = fxd_apm_p(d, sh, 5, pu, (int)((((uint32_t)c0 * 4) ^ fxd_hash3((uint32_t)fxd_min(9, pz), r->x5 & 0x80ff, 0xffffffffu)) & 0x3ffff)); EXPV(pz);
  if (fails & 255) pr = (pt * 6 + pu + pv * 11 + pz * 14 + 31) >> 5;
  else pr = (pt * 4 + pu * 5 + pv * 12 + pz * 11 + 31) >> 5;
  EXPV(pr);
#undef EXPV
  while (ex < u.orow + FX_OUTPUTS) *ex++ = 0.5f;   // slots no AddPrediction reaches keep the constructor's 0.5 (:94)
  sh->pr = pr;
  sh->parity ^= 1;
}

// scalars between global state and LDS (thread 0, barrier after)
FX_HD void fxd_load_shared(const FxDev* d, FxShared* sh, int tid) {
  for (int i = tid; i < 2 * FX_TX; i += FX_THREADS) (&sh->tx[0][0])[i] = (&d->tx[0][0])[i];
  if (tid < 16) sh->in2[tid] = d->in2[tid];
  if (tid < 12) { sh->mx_elim[tid] = d->mx_elim[tid]; sh->mx_cxt[tid] = d->mx_cxt[tid]; sh->mx_pr[tid] = d->mx_pr[tid]; }
  if (tid < 6) sh->apm_index[tid] = d->apm_index[tid];
  for (int i = tid; i < 2048; i += FX_THREADS) (&sh->ohash[0][0])[i] = 0;
  for (int i = tid; i < 2 * FX_NMAPS; i += FX_THREADS) (&sh->mconf[0][0])[i] = 0;
  if (tid == 0) { sh->pr = d->pr; sh->parity = d->parity; sh->fails = d->fails; sh->failz = d->failz; sh->failcount = d->failcount; sh->isMatch = 0; }
  for (int i = tid; i < FX_NMAPS * 8; i += FX_THREADS) {
    const FxMapDev* x = &d->maps[i >> 3];
    const int j = i & 7;
    sh->mcp[i >> 3][j] = x->cp[j]; sh->mcp0[i >> 3][j] = x->cp0[j]; sh->mrunp[i >> 3][j] = x->runp[j]; sh->mcxt[i >> 3][j] = x->cxt[j]; sh->msmc[i >> 3][j] = x->sm_cxt[j];
  }
  for (int k = 0; k < FX_NMAPS; k++) {
    const FxMapDev* x = &d->maps[k];
    for (int i = tid; i < x->C * 256; i += FX_THREADS) (&sh->sm[x->slot_base][0])[i] = x->sm[i];
  }
  if (tid < FX_NSLOTS) {   // cached bytes: straight from the table (the per-context registers above are this thread's own writes)
    const int k = d->slot_map[tid], i = d->slot_idx[tid];
    const FxMapDev* x = &d->maps[k];
    for (int q = 0; q < 7; q++) sh->mslot[tid][q] = x->t[x->cp0[i] + q];
    sh->mrun[tid][0] = x->t[x->runp[i]]; sh->mrun[tid][1] = x->t[x->runp[i] + 1];
    sh->mlook[tid] = 0;
  }
}
FX_HD void fxd_store_shared(FxDev* d, const FxShared* sh, int tid) {
  for (int i = tid; i < 2 * FX_TX; i += FX_THREADS) (&d->tx[0][0])[i] = (&sh->tx[0][0])[i];
  if (tid < 16) d->in2[tid] = sh->in2[tid];
  if (tid < 12) { d->mx_elim[tid] = sh->mx_elim[tid]; d->mx_cxt[tid] = sh->mx_cxt[tid]; d->mx_pr[tid] = sh->mx_pr[tid]; }
  if (tid < 6) d->apm_index[tid] = sh->apm_index[tid];
  if (tid == 0) { d->pr = sh->pr; d->parity = sh->parity; d->fails = sh->fails; d->failz = sh->failz; d->failcount = sh->failcount; }
  for (int i = tid; i < FX_NMAPS * 8; i += FX_THREADS) {
    FxMapDev* x = &d->maps[i >> 3];
    const int j = i & 7;
    x->cp[j] = sh->mcp[i >> 3][j]; x->cp0[j] = sh->mcp0[i >> 3][j]; x->runp[j] = sh->mrunp[i >> 3][j]; x->cxt[j] = sh->mcxt[i >> 3][j]; x->sm_cxt[j] = sh->msmc[i >> 3][j];
  }
  for (int k = 0; k < FX_NMAPS; k++) {
    FxMapDev* x = &d->maps[k];
    for (int i = tid; i < x->C * 256; i += FX_THREADS) x->sm[i] = (&sh->sm[x->slot_base][0])[i];
  }
}

// The uniform values of the update of bit q (0..8n-1) of a chunk of n bytes. Row q + 1 of the output receives the
// values FXCM::Predict() returns after that update; the last update of the chunk writes to d->pending, which becomes
// row 0 of the next chunk. blpos0 / lastbyte0 / have_rec0: the stream's state when the chunk starts.
FX_HD FxBit fxd_bit(FxDev* d, const uint8_t* bytes, const FxByteRec* recs, const int16_t* lstmpr, const uint8_t* lstmex, float* out, long ostride, int nbits,
                    int q, int blpos0, int lastbyte0, int have_rec0) {
  FxBit u;
  u.q = q;
  const int b = q >> 3, k = q & 7, cur = bytes[b];
  u.y = (cur >> (7 - k)) & 1;
  u.bpos = (k + 1) & 7;
  u.boundary = (k == 7);
  u.c0 = u.boundary ? 1 : ((1 << (k + 1)) | (cur >> (7 - k)));
  u.blpos = blpos0 + b + (u.boundary ? 1 : 0);
  u.lastbyte = u.boundary ? cur : (b > 0 ? bytes[b - 1] : lastbyte0);
  u.sscmrate = (u.blpos > 14 * 256 * 1024);
  u.rate = 6 + (u.blpos > 14 * 256 * 1024) + (u.blpos > 28 * 512 * 1024);
  u.lstmpr = lstmpr[q]; u.lstmex = lstmex[q];
  const int ri = u.boundary ? b : b - 1;     // the record of the last completed byte
  u.normal = ri >= 0 ? 1 : have_rec0;
  u.rec = ri >= 0 ? &recs[ri] : &d->rec;
  u.orow = q + 1 < nbits ? out + (long)(q + 1) * ostride : d->pending;
  return u;
}
#endif
