// fxcm_stage.hip -- the fxcm stage of the pipeline: layer-0 columns 3..433 of the cmix predictor, i.e. the 431 values
// FXCM::Predict() returns per bit (reference src/models/fxcm.cpp, src/models/fxcmv1.cpp; wired at predictor.cpp:98-99,
// :462-468). Two halves:
//   * host:   the text parser (fxcm_parser_host.cpp) turns the chunk's bytes into one FxByteRec per byte on the calling
//             thread, written to page-locked memory and copied to the device on the stage's stream;
//   * device: cmx_fxcm_roles_kernel, THREE persistent workgroups per stream (context maps / units / mixers + APM chain), walks
//             the chunk's bits through the learned tables (fxcm_dev.h: the phases, argued there). State: ~4.4 GB of HBM per stream (3.7 GB context-map
//             buckets, 0.3 GB mixer rows, 60 MB APM cells, 70 MB match / run tables).
// Bound: latency (a map lane walks up to 7 dependent bucket probes per bit); the kernel's HBM traffic is ~0.2 MB per
// input byte algorithmic (SURVEY.md 8d iii + iv). Parity: tests/test_fxcm_stage_host.py runs this kernel's body on the
// host against the oracle; tests/test_zgpu_stage_fxcm.py runs the kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/cmix_amd.h"
#include "fxcm_build.h"
#include "cmx_late.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip
extern "C" int cmx_device_count(void);

typedef short fx_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t fx_pair_train(uint32_t t, uint32_t w, int err) {   // train, SSE2 form (:543-557) in packed 16-bit math
  const fx_s2 tv = __builtin_bit_cast(fx_s2, t), wv = __builtin_bit_cast(fx_s2, w);
  const fx_s2 v = __builtin_elementwise_add_sat(tv, tv);
  fx_s2 r;
  r.x = (short)(__mul24((int)v.x, err) >> 16);
  r.y = (short)(__mul24((int)v.y, err) >> 16);
  const fx_s2 one = {1, 1};
  r = __builtin_elementwise_add_sat(r, one) >> 1;
  r = __builtin_elementwise_add_sat(r, wv);
  return __builtin_bit_cast(uint32_t, r);
}
__device__ __forceinline__ int fx_pair_dot(uint32_t t, uint32_t w) {   // one pmaddwd lane >> 8 (:522-541)
  return __builtin_amdgcn_sdot2(__builtin_bit_cast(fx_s2, t), __builtin_bit_cast(fx_s2, w), 0, false) >> 8;
}
// The trainer lanes' share of fxd_train_rows (same values): trainer j of 128 owns 16-byte groups j, j + 128, ... of the
// 10 x 64 groups of the selected first-layer rows, on the previous inputs -- five vector loads up front, packed math, five
// vector stores, instead of 40 dependent load / store pairs.
__device__ __forceinline__ void fx_train_rows_fast(FxDev* d, FxShared* sh, const FxBit& u, int j) {
  const uint4* tx = reinterpret_cast<const uint4*>(sh->tx[sh->parity]);
  uint4 w[5]; uint4* wp[5]; int err[5];
#pragma unroll
  for (int r = 0; r < 5; r++) {
    const int g = j + 128 * r, k = g >> 6;   // 64 groups of 8 weights per row
    err[r] = (int)(int16_t)fxd_mixer_err(sh, d, u, k);
    wp[r] = reinterpret_cast<uint4*>(d->wx[k] + (size_t)sh->mx_cxt[k] * FX_TX) + (g & 63);
    w[r] = *wp[r];
  }
#pragma unroll
  for (int r = 0; r < 5; r++) {
    if (!err[r]) continue;
    const uint4 t = tx[(j + 128 * r) & 63];
    uint4 v = w[r];
    v.x = fx_pair_train(t.x, v.x, err[r]); v.y = fx_pair_train(t.y, v.y, err[r]);
    v.z = fx_pair_train(t.z, v.z, err[r]); v.w = fx_pair_train(t.w, v.w, err[r]);
    *wp[r] = v;
  }
}

// The step functions are the ones fxd_phase1a .. fxd_phase5 call (tests/host/fxcm_emul.cpp runs those), but the units get wavefronts
// of their own: lanes of one wavefront that take different branches run one branch after the other, so context lanes, SSCMs, the
// match lane, the run map, trainers and APMs sharing wavefronts would cost the SUM of their chains (role assignment: cmx_fxcm_roles_kernel).
enum { FX_DEV_THREADS = 512 };
// a workgroup barrier that orders LDS traffic only (no wait for outstanding global loads / stores, which __syncthreads adds)
__device__ __forceinline__ void fx_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
struct FxApmRows;
__device__ __forceinline__ void fx_apm_prefetch(FxDev* d, const FxShared* sh, const FxBit& u, FxApmRows* A, int j);

// ---- phase 2 on the device, split (same values as fxd_phase2): part A needs nothing from the context maps -- the failure
// history (update1 :4784-4790), the dead zones and the selectors without an order term -- and runs on one lane of wave 4
// UNDER the maps' run phase; part B (after the barrier) sums the maps' return values on 8 lanes and finishes the four
// selectors that carry ordX / ordW (:4601-4640).
__device__ __forceinline__ void fx_fail_next(const FxShared* sh, const FxBit& u, uint32_t* fails, uint32_t* failz, uint32_t* failcount) {
  // e_l[bpos] (:3222) = {1830, 1997, 1973, 1851, 1897, 1690, 1998, 1842} as selects: a table indexed at run time is a load from constant memory
  const int bp_ = u.bpos;
  const int e_lo = bp_ & 2 ? (bp_ & 1 ? 1851 : 1973) : (bp_ & 1 ? 1997 : 1830), e_hi = bp_ & 2 ? (bp_ & 1 ? 1842 : 1998) : (bp_ & 1 ? 1690 : 1897);
  const int e_l_bpos = bp_ & 4 ? e_hi : e_lo;
  uint32_t f = sh->fails, z = sh->failz, c = sh->failcount;
  if (f & 0x00000080) --c;
  f *= 2; z *= 2;
  int pr = sh->pr;
  if (u.y) pr = 4095 - pr;
  if (pr >= e_l_bpos) { ++f; ++c; }
  if (pr >= 848) ++z;
  *fails = f; *failz = z; *failcount = c;
}
__device__ __forceinline__ void fx_phase2a_dev(FxDev* d, FxShared* sh, const FxBit& u, int is_match = -1) {   // is_match >= 0: the value (the caller fetched it from role U's row itself)
  if (u.boundary)
    for (int i = 0; i < FX_NMIX1; i++)
      sh->mx_elim[i] = (sh->fails & 255) == 0 ? fxd_max(256, sh->mx_elim[i] + 1) : fxd_max(0, fxd_min(16, sh->mx_elim[i] - 1));
  uint32_t f, z, c_;
  fx_fail_next(sh, u, &f, &z, &c_);
  sh->fails = f; sh->failz = z; sh->failcount = c_;
  const FxByteRec* r = u.rec;
  const int bpos = u.bpos, c0b = u.c0 << (8 - bpos);
  const uint32_t s2 = r->s2, s3 = r->s3, s3R = r->s3R, BrFc = r->BrFc, words = r->words, FcIdx = r->FcIdx, isPar = r->isPar;
  const uint32_t isMatch = (uint32_t)(is_match >= 0 ? is_match : sh->isMatch);
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
  cx[6] = (int)((s3R & 0xff8) * 4 + ((2 * words) & 0x1c) + (s2 & 3));
  c = c0b;
  cx[3] = bpos * 256 + (int)((((((uint32_t)r->numbers | words) << bpos) & 255) >> bpos) | ((uint32_t)c & 255));
  if (bpos > 2) cx[7] = (int)(((s3 & 7) * 8 + w3b[c0b & 255]) * 256 + BrFc * 32 + (words & 7) * 4 + isPar + (isMatch ? 2u : 0u));
  else cx[7] = (int)(((s3 & 63) * 256 + BrFc * 16 + (words & 7) * 2 + isPar) | (isMatch ? 128u : 0u));
  cx[8] = (int)r->deccode;
  cx[9] = (bpos << 8) * 4 + (int)(f & 3) * 256 + u.lstmex;
  cx[11] = 0;
}
// the eight sums of the maps' return values the selectors use (lanes 0..7)
__device__ __forceinline__ int fx_res8(const FxDev* d, const FxShared* sh, const FxBit& u, int lane) {
  const int which = lane < 6 ? lane : lane == 6 ? 21 : 23;   // (arithmetic, not a table: a table indexed by the lane is a load from constant memory)
  int v = 0;
  if (u.normal) { const FxMapDev* x = &d->maps[which]; for (int i = 0; i < x->C; i++) v += sh->slot_res[x->slot_base + i]; }
  return v;
}
__device__ __forceinline__ void fx_phase2b_tail(FxDev* d, FxShared* sh, const FxBit& u, const int* res8_s) {   // one lane
  const FxByteRec* r = u.rec;
  const int bpos = u.bpos, c0b = u.c0 << (8 - bpos);
  int ordX = 0, ordW = 0;
  if (u.normal) {
    const int b0 = d->maps[0].slot_base;
    int skipped = 0;
    for (int i = 0; i < 3; i++) skipped |= (int)((r->skip[(b0 + i) >> 5] >> ((b0 + i) & 31)) & 1);
    if (skipped) ordX = 2;
    ordX += res8_s[0];
    if (ordX == 3) ordX = 2;
    ordX += res8_s[1] + res8_s[2] + res8_s[3];
    ordW = res8_s[4] + res8_s[5];
    if (ordW > 3) ordW = 3;
    ordW += res8_s[6] + res8_s[7];
  }
  const uint32_t s2 = r->s2, s3 = r->s3, BrFc = r->BrFc, words = r->words, FcIdx = r->FcIdx, isPar = r->isPar;
  const uint32_t isMatch = (uint32_t)sh->isMatch;
  int* cx = sh->mx_cxt;
  cx[2] = (int)(((4 * words) & 0xf0) * 4 + (uint32_t)ordX * 256 * 4 + (s2 & 63));
  cx[10] = (int)(((uint32_t)ordX * 8 + (BrFc ? 1u : 0u) * 4 + (s2 & 3)) * 2 + (words & 1));
  int c = c0b;
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
}

__device__ __forceinline__ void fx_phase2b_dev(FxDev* d, FxShared* sh, const FxBit& u, int lane, int* res8_s) {   // wave 0
  if (lane < 8) res8_s[lane] = fx_res8(d, sh, u, lane);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  if (lane == 0) fx_phase2b_tail(d, sh, u, res8_s);
}

// ---- phase 3 on the device (same values as fxd_phase3): the ten weight pairs are requested together, then multiplied
__device__ __forceinline__ void fx_phase3_dots(FxDev* d, FxShared* sh, const FxBit& u, int tid) {
  const uint32_t t = reinterpret_cast<const uint32_t*>(sh->tx[sh->parity ^ 1])[tid];
  uint32_t w[FX_NMIX1];
#pragma unroll
  for (int k = 0; k < FX_NMIX1; k++) w[k] = reinterpret_cast<const uint32_t*>(d->wx[k] + (size_t)sh->mx_cxt[k] * FX_TX)[tid];
#pragma unroll
  for (int k = 0; k < FX_NMIX1; k++) sh->part[k][tid] = fx_pair_dot(t, w[k]);
}

// ---- phase 5 on the device (same values as fxd_phase5): the ten first-layer finals on ten lanes, the two final mixers on
// two, the six APMs as a three-level tree (0, 1, 2 | 3, 4 | 5) reading the context rows a lane of wave 7 fetched into LDS
// during phase 1a -- every APM context is known when the bit starts (c0, the parser's hashes, the failure history).
struct FxApmRows { uint32_t cx[6]; uint16_t row[6][34]; };
// combo >= 0: the APM contexts of an update whose final probability is not known yet -- it enters only through two comparisons (fx_fail_next: pr >= e_l[bpos],
// pr >= 848; every e_l is above 848), so there are three cases: combo 0 = neither, 1 = only the second, 2 = both. A compressor's role X computes all three for the
// NEXT update on idle lanes while phase 5 of this one runs, and picks one when the probability is there.
__device__ __forceinline__ int fx_fail_combo(const FxShared* sh, const FxBit& u) {
  const int bp_ = u.bpos;
  const int e_lo = bp_ & 2 ? (bp_ & 1 ? 1851 : 1973) : (bp_ & 1 ? 1997 : 1830), e_hi = bp_ & 2 ? (bp_ & 1 ? 1842 : 1998) : (bp_ & 1 ? 1690 : 1897);
  const int e_l_bpos = bp_ & 4 ? e_hi : e_lo;
  int pr = sh->pr;
  if (u.y) pr = 4095 - pr;
  return pr >= e_l_bpos ? 2 : pr >= 848 ? 1 : 0;
}
__device__ __forceinline__ void fx_apm_ctx(const FxShared* sh, const FxBit& u, uint32_t cx[6], int combo = -1) {   // update1 :4792-4826
  uint32_t fails, failz, failcount;
  if (combo < 0) fx_fail_next(sh, u, &fails, &failz, &failcount);
  else {
    uint32_t f = sh->fails, z = sh->failz, c = sh->failcount;
    if (f & 0x00000080) --c;
    f *= 2; z *= 2;
    if (combo == 2) { ++f; ++c; }
    if (combo >= 1) ++z;
    fails = f; failz = z; failcount = c;
  }
  // tri[4] = {0, 4, 3, 7}, trj[4] = {0, 6, 6, 12} as arithmetic (no table in constant memory)
  const FxByteRec* r = u.rec;
  const uint32_t c0 = (uint32_t)u.c0;
  int pz = (int)failcount + 1;
  { const uint32_t a = (fails >> 5) & 3; pz += (int)((a & 1) * 4 + (a >> 1) * 3); }
  { const uint32_t a = (fails >> 3) & 3; pz += (int)(((a & 1) + (a >> 1)) * 6); }
  { const uint32_t a = (fails >> 1) & 3; pz += (int)(((a & 1) + (a >> 1)) * 6); }
  if (fails & 1) pz += 8;
  pz = pz / 2;
  cx[0] = c0;
  cx[3] = ((c0 * 2) ^ r->AH1) & 0x3ffff;
  cx[1] = ((c0 * 8) ^ fxd_hash3(29, failz & 2047, 0xffffffffu)) & 0xffff;
  cx[4] = ((fails & 255) ? fxd_hash3(c0, r->s2 & 0xfffc, r->s3R & 0x1ff) : fxd_hash3(c0, (r->s2R & 0xfffc) + 0x10000, r->s3R & 0x1ff)) & 0x3ffff;
  cx[2] = ((c0 * 32) ^ r->AH2) & 0xffff;
  cx[5] = ((c0 * 4) ^ fxd_hash3((uint32_t)fxd_min(9, pz), r->x5 & 0x80ff, 0xffffffffu)) & 0x3ffff;
}
__device__ __forceinline__ void fx_apm_prefetch(FxDev* d, const FxShared* sh, const FxBit& u, FxApmRows* A, int j) {   // after this lane's fxd_apm_update
  uint32_t cx[6];
  fx_apm_ctx(sh, u, cx);
  const uint32_t cxj = j == 0 ? cx[0] : j == 1 ? cx[1] : j == 2 ? cx[2] : j == 3 ? cx[3] : j == 4 ? cx[4] : cx[5];   // (selects: cx[j] would put the array in scratch memory)
  const uint16_t* p = d->apm_t[j] + (size_t)cxj * 33;
  uint16_t v[33];
#pragma unroll
  for (int q = 0; q < 33; q++) v[q] = p[q];
#pragma unroll
  for (int q = 0; q < 33; q++) A->row[j][q] = v[q];
  A->cx[j] = cxj;
}
// the same in two halves, spread over a whole wavefront: the six row numbers by six lanes (into LDS), then lane l fetches elements l, l + 64, .. of the 6 x 33
// (four loads per lane instead of 33 on six lanes) into registers early in the bit; the commit (registers -> LDS) comes just before phase 5 needs the rows,
// so the wave that fetches does not hold the workgroup's barriers up
struct FxApmRegs { uint16_t v[4]; };
__device__ __forceinline__ void fx_apm_issue(FxDev* d, const uint32_t* apmcx, FxApmRegs* R, int lane) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int idx = lane + 64 * r;
    if (idx < 6 * 33) { const int j = idx / 33, q = idx - 33 * j; R->v[r] = (d->apm_t[j] + (size_t)apmcx[j] * 33)[q]; }
  }
}
__device__ __forceinline__ void fx_apm_commit(FxApmRows* A, const uint32_t* apmcx, const FxApmRegs* R, int lane) {
#pragma unroll
  for (int r = 0; r < 4; r++) {
    const int idx = lane + 64 * r;
    if (idx < 6 * 33) { const int j = idx / 33, q = idx - 33 * j; A->row[j][q] = R->v[r]; }
  }
  if (lane < 6) A->cx[lane] = apmcx[lane];
}
__device__ __forceinline__ int fx_apm_row_p(const FxDev* d, FxShared* sh, const FxApmRows* A, int j, int pr) {   // fxd_apm_p on the fetched row
  pr = d->stretch[pr];
  const int w = pr & 127, i = (pr + 2048) >> 7;
  sh->apm_index[j] = i + (int)A->cx[j] * 33;
  return (A->row[j][i] * (128 - w) + A->row[j][i + 1] * w) >> 11;
}
// msw / mscx != nullptr (round 5): the selected rows of the final mixers 10 / 11 in LDS (16 weights each) -- trained there at the top of the next bit
__device__ __forceinline__ void fx_phase5_dev(FxDev* d, FxShared* sh, const FxBit& u, const FxApmRows* A, int lane, int* scr, int16_t* msw = nullptr, int* mscx = nullptr) {   // wave 0; scr: 16 ints of LDS
  const FxLayout l = fxd_layout(d, u.normal);
  float* ex = u.orow + l.exp_mix;
  if (lane < FX_NMIX1) {   // p1 :641-651, mxInputs2.add
    const int k = lane;
    uint32_t s = 0;
    for (int g = 0; g < 16; g++) s += (uint32_t)sh->part2[k][g];
    int dp = (int32_t)(s * (uint32_t)d->mx_shift[k]) >> 11;
    dp = fxd_clp(dp);
    sh->mx_pr[k] = fxd_squash(d, dp);
    sh->in2[k] = (int16_t)dp;
    ex[k] = fxd_export(d, dp);
  }
  if (lane == 10) sh->in2[10] = (int16_t)(d->stretch[u.lstmpr] / 2);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  if (msw && lane == 0) *(volatile int*)&scr[12] = u.q + 1;   // mx_pr[0..9] of this bit are in place (the next update's mixer errors are computed from them, fx_roles_body)
  if (lane < 2) {
    const int k = 10 + lane;
    const int16_t* wrow = msw ? msw + 16 * lane : d->wx[k] + (size_t)sh->mx_cxt[k] * 16;
    int dp = (int32_t)((uint32_t)fxd_dot16(sh->in2, wrow) * (uint32_t)d->mx_shift[k]) >> 11;
    dp = fxd_clp(dp);
    sh->mx_pr[k] = fxd_squash(d, dp);
    scr[lane] = dp;
  }
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  if (msw && lane == 0) *(volatile int*)&scr[13] = u.q + 1;   // mx_pr[10..11] as well
  const int pr = fxd_squash(d, (scr[0] * 7 + scr[1] + 4) >> 3);
  // level 1: APM 0, 1, 2 on pr
  if (lane < 3) scr[4 + lane] = fx_apm_row_p(d, sh, A, lane, pr);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  const int pu1 = (scr[4] + 7 * pr + 4) >> 3, pv1 = scr[5], pt = scr[6];
  // level 2: APM 3 on pu, APM 4 on pv
  if (lane == 0) scr[8] = fx_apm_row_p(d, sh, A, 3, pu1);
  if (lane == 1) scr[9] = fx_apm_row_p(d, sh, A, 4, pv1);
  __builtin_amdgcn_s_waitcnt(0xC07F);
  __builtin_amdgcn_wave_barrier();
  if (lane != 0) return;
  const int pu = scr[8], pv = scr[9];
  const int pz = fx_apm_row_p(d, sh, A, 5, pu);
  ex += FX_NMIX1;
#define EXPV(v) (*ex++ = (float)(v) * (float)(1.0 / 4095))
  EXPV(pr); EXPV(pu); EXPV(pv1); EXPV(pv); EXPV(pt); EXPV(pz);
  int fin;
  if (sh->fails & 255) fin = (pt * 6 + pu + pv * 11 + pz * 14 + 31) >> 5;
  else fin = (pt * 4 + pu * 5 + pv * 12 + pz * 11 + 31) >> 5;
  EXPV(fin);
#undef EXPV
  if (!u.normal || !msw) while (ex < u.orow + FX_OUTPUTS) *ex++ = 0.5f;   // slots no AddPrediction reaches keep the constructor's 0.5 (:94); (msw: a compressor's role X fills them once per chunk)
  sh->pr = fin;
  sh->parity ^= 1;
}


// Everything the bit loop reads lives in LDS: FxShared (inputs, StateMaps, per-context registers), a working COPY of the
// stream's FxDev (map descriptors, unit registers, scalars: the step functions read them through `d` many times per bit,
// and from global memory every such read is an L2 round trip the compiler must redo after each store), the squash /
// stretch / state tables, and the record of the byte in force (copied one update ahead). Only the learned tables and the
// per-map output tables stay in HBM / L2.
struct FxLocal {
  FxDev dev;
  int16_t squash[4096], stretch[4096];
  uint8_t wrt[512];
  uint8_t sta[6][1024];
  FxByteRec rec[2];
  FxApmRows apm;
  float ex[2][FX_OUTPUTS + 1];   // the bit's 431 exported values (two parities: the copy-out of bit q runs under bit q + 1): the units write here, one coalesced copy per bit goes to the output row
};
// =====================================================================================================================
// The stage in THREE roles (role M on FX_M_WGS workgroups since round 3). Nothing the context maps, the match models, the SSCMs or the run map
// learn depends on the mixers or on the final probability -- only on the byte stream -- so the bit's work splits into
// three roles that run on three compute units, each with its own LDS state, coupled only by the rows they hand over:
//   role M  (block 0) the 31 context maps: touch -> run; publishes its 5..6 inputs per context, the eight return-value
//           sums the selectors use, and its exported values
//   role U  (block 1) MatchModel2 + SparseMatchModel, the SSCMs, the run map, the LSTM input; publishes 25 inputs, isMatch
//   role X  (block 2) everything that learns from the coded probability: trainers, selectors, dot products, final
//           mixers, APM chain. Takes row q when both M and U have published it.
// M and U run ahead of X as far as the data lets them (they never wait for X); the per-bit cost of the stage is the
// slowest role instead of the sum of the phases. Rows travel through global memory: agent-scope stores, a workgroup-wide
// wait for them, then the role's row counter; X polls the counters (bounded) and loads the row with agent-scope loads.
struct FxXfer {
  unsigned fail; unsigned pad[3];   // sticky: a bounded wait ran out
  unsigned started;                 // zeroed ahead of every launch from here on: workgroups that hold the stream state
  unsigned m_done, u_done;          // rows published by role U (m_done: unused since role M publishes per wavefront)
  unsigned pad2;
  unsigned mw_done[FX_M_WAVES];     // rows published by each of role M's wavefronts
};
enum { FX_ROW_WORDS = 288,          // one row: 1152 bytes
       FX_ROW_RES = 256,            //   [0, 256) role M: its copy of the 512 inputs (its own range is taken); then the 8 sums
       FX_ROW_UTX = 264,            //   role U: inputs 0..23 as 12 words, word 12 = run map | LSTM input
       FX_ROW_MATCH = 280,          //   isMatch
       FX_ROLE_SPIN = 1 << 24 };
struct FxAhead { uint8_t byte[4]; int16_t pr[4][8]; uint8_t ex[4][8]; };   // the stream values of the byte in force and the next one

__device__ __forceinline__ unsigned fx_ld_u(const unsigned* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void fx_st_u(unsigned* p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool fx_wait_ge(unsigned* p, unsigned want, unsigned* fail) {   // one lane
  unsigned it = 0;
  while (fx_ld_u(p) < want)
    if ((++it & 4095u) == 0 && (it > (unsigned)FX_ROLE_SPIN || fx_ld_u(fail))) { fx_st_u(fail, 1u); return false; }
  return true;
}
// fxd_bit from LDS: the chunk's bytes and LSTM hints of the byte in force are staged a byte ahead (FxAhead), the records
// one update ahead (FxLocal::rec)
__device__ __forceinline__ FxBit fx_bit_dev(const FxDev* d, const FxAhead* ah, const FxLocal* loc, int q, int blpos0, int lastbyte0, int have0) {
  FxBit u;
  u.q = q;
  const int b = q >> 3, k = q & 7, cur = ah->byte[b & 3];
  u.y = (cur >> (7 - k)) & 1;
  u.bpos = (k + 1) & 7;
  u.boundary = (k == 7);
  u.c0 = u.boundary ? 1 : ((1 << (k + 1)) | (cur >> (7 - k)));
  u.blpos = blpos0 + b + (u.boundary ? 1 : 0);
  u.lastbyte = u.boundary ? cur : (b > 0 ? ah->byte[(b - 1) & 3] : lastbyte0);
  u.sscmrate = (u.blpos > 14 * 256 * 1024);
  u.rate = 6 + (u.blpos > 14 * 256 * 1024) + (u.blpos > 28 * 512 * 1024);
  u.lstmpr = ah->pr[b & 3][k]; u.lstmex = ah->ex[b & 3][k];
  const int ri = u.boundary ? b : b - 1;
  u.normal = ri >= 0 ? 1 : have0;
  u.rec = ri >= 0 ? &loc->rec[ri & 1] : &d->rec;
  u.orow = nullptr;
  return u;
}
// the helper wave of a role, at the top of update q: next byte's stream values (at bit 0) and the next record (at bit 6)
__device__ __forceinline__ void fx_stage_ahead(FxAhead* ah, FxLocal* loc, const uint8_t* bytes, const FxByteRec* recs, const int16_t* lstmpr, const uint8_t* lstmex,
                                               int n, int q, int lane) {
  const int b = q >> 3, k = q & 7;
  if (k == 0 && b + 1 < n) {
    if (lane < 8) { ah->pr[(b + 1) & 3][lane] = lstmpr[8 * (b + 1) + lane]; ah->ex[(b + 1) & 3][lane] = lstmex[8 * (b + 1) + lane]; }
    else if (lane == 8) ah->byte[(b + 1) & 3] = bytes[b + 1];
  }
  if (k == 6) for (int i = lane; i < (int)(sizeof(FxByteRec) / 4); i += 64) ((uint32_t*)&loc->rec[b & 1])[i] = ((const uint32_t*)&recs[b])[i];
}

// LATE (a decoder, cmx_late.h): the chunk's bytes are not known -- bit q arrives through the box LB (with it the host records of
// the step after it: at a byte's last bit the parser's record of that byte), the LSTM hints of update q through the ByteModel kernel's
// counter LC_BM2, and role X counts the rows it has completed (LC_FX) for the mixing network. `bytes` is unused; recs is host-mapped.
template <bool LATE>
__device__ __forceinline__ void fx_roles_body(FxDev* gd, FxXfer* X, unsigned* rows, const uint8_t* bytes, const FxByteRec* recs,
                                              const int16_t* lstmpr, const uint8_t* lstmex, float* out, long ostride, int n,
                                              unsigned long long* prof, CmxLate LB) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fx_smem[];
  __shared__ int late_go;
  FxShared& sh = *(FxShared*)fx_smem;
  // CMX_FXCM_PROFILE=1: thread 0 of each role accumulates its clocks per phase: prof[16 role + k] (scripts/gpu_fxcm_time.py)
  unsigned long long pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pc0 = __builtin_readcyclecounter();
  __shared__ unsigned long long pbp[8][8];   // role M's clocks by bit position as well (prof[64 + 8 bpos + k])
  if (threadIdx.x < 64) pbp[threadIdx.x >> 3][threadIdx.x & 7] = 0;
#define FX_TICK(k) do { if (prof && threadIdx.x == 0 && blockIdx.x != 1) { const unsigned long long c_ = __builtin_readcyclecounter(); pacc[k] += c_ - pc0; if (blockIdx.x == 0) pbp[u.bpos][k] += c_ - pc0; pc0 = c_; } } while (0)
  FxLocal& loc = *(FxLocal*)(fx_smem + ((sizeof(FxShared) + 15) & ~(size_t)15));
  FxDev* d = &loc.dev;
  __shared__ int res8_s[8], scr_s[16];
  __shared__ FxAhead ah;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int role = (int)blockIdx.x < FX_M_WGS ? 0 : (int)blockIdx.x - FX_M_WGS + 1, mwg = blockIdx.x;   // 0 = M (FX_M_WGS workgroups), 1 = U, 2 = X
  // ---- every role starts from the stream's state (it uses its own part of it) ----
  for (int i = tid; i < (int)(sizeof(FxDev) / 4); i += FX_DEV_THREADS) ((uint32_t*)d)[i] = ((const uint32_t*)gd)[i];
  for (int i = tid; i < 4095; i += FX_DEV_THREADS) loc.squash[i] = gd->squash.p[i];
  for (int i = tid; i < 4096; i += FX_DEV_THREADS) loc.stretch[i] = gd->stretch.p[i];
  for (int i = tid; i < 512; i += FX_DEV_THREADS) loc.wrt[i] = gd->wrt.p[i];
  for (int i = tid; i < 6 * 1024; i += FX_DEV_THREADS) loc.sta[i >> 10][i & 1023] = gd->sta[i >> 10][i & 1023];
  __syncthreads();
  if (tid == 0) { d->squash = loc.squash; d->stretch = loc.stretch; d->wrt = loc.wrt; }
  if (tid < FX_NMAPS) for (int q = 0; q < 6; q++) if (gd->maps[tid].nn.p == gd->sta[q]) d->maps[tid].nn = loc.sta[q];
  const int nbits = 8 * n, blpos0 = d->blpos, lastbyte0 = d->lastbyte, have0 = d->have_rec;
  if (tid < FX_THREADS) fxd_load_shared(d, &sh, tid);
  if (role == 2) for (int i = tid; i < FX_OUTPUTS; i += FX_DEV_THREADS) out[i] = gd->pending[i];   // row 0: what the previous chunk's last update left
  if (!LATE) {
    if (tid < 8) { ah.pr[0][tid] = lstmpr[tid]; ah.ex[0][tid] = lstmex[tid]; }
    if (tid == 8) ah.byte[0] = bytes[0];
  }
  __syncthreads();
  if (role != 2 && tid == 0) sh.parity = 0;   // M and U build the bit's inputs in tx[1]
  // all workgroups hold the state before any of them may write a part of it back (or the pending row)
  if (tid == 0) { __hip_atomic_fetch_add(&X->started, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); fx_wait_ge(&X->started, (unsigned)(FX_M_WGS + 2), &X->fail); }
  if (LATE) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (row 0's stores, every wave's, before the barrier in front of its count)
  __syncthreads();
  const FxLayout ln = fxd_layout(d, 1);   // the normal layout's offsets
  unsigned have_row = 0;
  if (LATE && role == 2 && tid == 0) late_publish(LB, LC_FX, 1u);   // row 0 (above) is in place
  unsigned late_part = 0;   // LATE: the bits of the byte in force decoded so far (kept by the wavefront that stages them)
  // a compressor's role X (round 5): the first-layer rows are trained by the threads that own their words (registers, below); the final mixers' two 16-weight
  // rows in LDS (msw, row numbers mscx); merr = fxd_mixer_err of the update, computed once per bit
  __shared__ int mscx[2], merr[12];
  __shared__ uint32_t apmcx[6], apmspec[3][6];
  __shared__ __attribute__((aligned(16))) int16_t msw[32];
  unsigned long long wtop = 0, wp3 = 0, w7[3] = {0, 0, 0};   // profiling: per-wave clocks of role X's top block / phase 3
  FxApmRegs xapm;
  bool xapm_ok = false;   // loc.apm holds the rows phase 5 of the previous bit read
  uint32_t xw[FX_NMIX1], xt = 0;           // the thread's words of the ten selected rows as the last dot products used them, its pair of the inputs they met
  int xcn[FX_NMIX1];                       // the numbers of those rows in their tables
#pragma unroll
  for (int k = 0; k < FX_NMIX1; k++) { xw[k] = 0; xcn[k] = 0; }
  if (!LATE && role == 2) {
    if (tid < FX_THREADS) {
      xt = reinterpret_cast<const uint32_t*>(sh.tx[sh.parity])[tid];
#pragma unroll
      for (int k = 0; k < FX_NMIX1; k++) { xcn[k] = sh.mx_cxt[k]; xw[k] = reinterpret_cast<const uint32_t*>(d->wx[k] + (size_t)xcn[k] * FX_TX)[tid]; }
    }
    if (tid < 32) msw[tid] = (d->wx[10 + (tid >> 4)] + (size_t)sh.mx_cxt[10 + (tid >> 4)] * 16)[tid & 15];
    if (tid < 2) mscx[tid] = sh.mx_cxt[10 + tid];
    if (tid >= 64 && tid < 76) { const FxBit u0 = fx_bit_dev(d, &ah, &loc, 0, blpos0, lastbyte0, have0); merr[tid - 64] = fxd_mixer_err(&sh, d, u0, tid - 64); }   // the first update's errors
    for (int i = ln.exp_mix + FX_NMIX1 + 7 + tid; i < FX_OUTPUTS; i += FX_DEV_THREADS) { loc.ex[0][i] = 0.5f; loc.ex[1][i] = 0.5f; }   // the slots behind the last export keep 0.5 (fx_phase5_dev)
    if (tid < 16) scr_s[tid] = 0;   // incl. wave 4's two ready flags ([12], [13] = bit + 1): LDS keeps whatever the compute unit's previous kernel left there
    __syncthreads();
  }
  if (role == 0) {
    // ================= role M: the context maps, one free-running wavefront per group of maps =================
    // Nothing a map learns depends on another map or on role X, and the coded bits are the chunk's bytes: every wavefront owns whole maps
    // (FxDev::mw_map; their contexts are its lanes), walks the chunk's bits at its own pace -- touch -> run per bit, synchronised inside the
    // wavefront only -- and publishes its part of row q and its own row counter. A slow bit of one map (a table miss, a second visit, a serial
    // walk after an overlap) delays that wavefront's counter, not the other maps: role X sees the slowest wavefront's AVERAGE, not the sum of
    // every bit's slowest lane. The overlap hash set is per wavefront (128 of the 1024 entries, both parities); the byte records travel
    // through a per-wavefront LDS copy (in FxShared::part, which only role X uses).
    const int mw = 8 * mwg + wave;
    const int s0 = d->mw_slot[mw], s1 = d->mw_slot[mw + 1], k0 = d->mw_map[mw], k1 = d->mw_map[mw + 1];
    const int msl = s0 + lane < s1 ? s0 + lane : FX_NSLOTS;
    if (k0 >= k1) { if (lane == 0) fx_st_u(&X->mw_done[mw], (unsigned)nbits); }
    else {
      const FxMapDev* xa = &d->maps[k0]; const FxMapDev* xb = &d->maps[k1 - 1];
      const int txlo = xa->tx_off, txhi = xb->tx_off + xb->C * (5 + xb->u), exlo = xa->exp_off, exhi = xb->exp_off + xb->C * (4 + xb->u);
      uint32_t* const ht[2] = {sh.ohash[0] + 128 * wave, sh.ohash[1] + 128 * wave};
      FxByteRec* const wrec = reinterpret_cast<FxByteRec*>(&sh.part[0][0]) + 2 * wave;   // [2]: the record in force and the next one
      static_assert(sizeof(FxByteRec) * 16 <= sizeof(sh.part), "per-wavefront record copies do not fit");
      const int which8 = (lane & 7) < 6 ? (lane & 7) : (lane & 7) == 6 ? 21 : 23;
      const bool res_lane = lane < 8 && which8 >= k0 && which8 < k1;
      uint32_t bv = 0;
      int prev_byte = lastbyte0;
      for (int q = 0; q < nbits; q++) {
        const int b = q >> 3, k = q & 7;
        int cur;
        if (LATE) {   // the byte as far as it is known: the decoded bits on top (only those are used), zeros below
          const int y = late_y(LB, q + 1);
          if (y < 0) return;
          late_part = (k == 0 ? 0u : late_part) * 2u + (unsigned)y;
          cur = (int)(late_part << (7 - k));
          if (k == 7) {   // the byte is complete: the parser's record of it is in place (it came with this bit)
            for (int i = lane; i < (int)(sizeof(FxByteRec) / 4); i += 64) ((uint32_t*)&wrec[b & 1])[i] = ((const volatile uint32_t*)&recs[b])[i];
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_wave_barrier();
          }
        } else {
          if (k == 0 && (b & 63) == 0) bv = b + lane < n ? (uint32_t)bytes[b + lane] : 0u;
          cur = (int)__builtin_amdgcn_readlane((int)bv, b & 63);
        }
        FxBit u;
        u.q = q;
        u.y = (cur >> (7 - k)) & 1;
        u.bpos = (k + 1) & 7;
        u.boundary = (k == 7);
        u.c0 = u.boundary ? 1 : ((1 << (k + 1)) | (cur >> (7 - k)));
        u.blpos = blpos0 + b + (u.boundary ? 1 : 0);
        u.lastbyte = u.boundary ? cur : prev_byte;
        u.sscmrate = 0; u.rate = 0; u.lstmpr = 0; u.lstmex = 0;   // (not read by the maps)
        const int ri = u.boundary ? b : b - 1;
        u.normal = ri >= 0 ? 1 : have0;
        u.rec = ri >= 0 ? &wrec[ri & 1] : &d->rec;
        u.orow = loc.ex[0];
        float* const real_row = q + 1 < nbits ? out + (long)(q + 1) * ostride : gd->pending;
        unsigned* const row = rows + (size_t)q * FX_ROW_WORDS;
        const int par = q & 1;
        const bool carry = u.bpos == 4 || u.bpos == 7;   // the keys of a bit between two lookups are those of the bit before it (or fewer): its flags carry over
        if (!LATE && k == 6) for (int i = lane; i < (int)(sizeof(FxByteRec) / 4); i += 64) ((uint32_t*)&wrec[b & 1])[i] = ((const uint32_t*)&recs[b])[i];
        if (carry) { if (k0 + lane < k1) sh.mconf[par][k0 + lane] = sh.mconf[par ^ 1][k0 + lane]; }
        else if (msl < FX_NSLOTS) fxd_map_touch(d, &sh, u, msl, ht[par], 127u);
        FX_TICK(0);
        // the wavefront's barrier of the bit: the previous run's table stores are complete before a serial walk may read them
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        FX_TICK(1);
        if (msl < FX_NSLOTS) fxd_map_run(d, &sh, u, msl);
        FX_TICK(2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        FX_TICK(3);
        ht[par ^ 1][lane] = 0; ht[par ^ 1][lane + 64] = 0;
        if (k0 + lane < k1) sh.mconf[par ^ 1][k0 + lane] = 0;
        if (u.normal) {
          for (int i = txlo + lane; i < txhi; i += 64)
            __hip_atomic_store(reinterpret_cast<unsigned short*>(row) + i, (unsigned short)sh.tx[1][i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          for (int i = exlo + lane; i < exhi; i += 64) real_row[i] = loc.ex[0][i];
        }
        if (res_lane) fx_st_u(row + FX_ROW_RES + lane, (unsigned)fx_res8(d, &sh, u, lane));
        FX_TICK(4);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wavefront's row stores have reached the coherence point before its counter moves
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) fx_st_u(&X->mw_done[mw], (unsigned)(q + 1));
        FX_TICK(5);
        if (k == 7) prev_byte = cur;
      }
    }
  } else
  for (int q = 0; q < nbits; q++) {
    if (LATE) {   // the role's helper wavefront stages what the update needs as it arrives: bit q, the LSTM hints of update q, at a byte's last bit its record
      if (wave == (role == 1 ? 3 : 5)) {
        const int b = q >> 3, k = q & 7;
        const int y = late_y(LB, q + 1);
        const bool ok = y >= 0 && late_wait_cnt(LB, LC_BM2, (unsigned)(q + 2));
        if (ok) {
          late_part = (k == 0 ? 0u : late_part) * 2u + (unsigned)y;
          if (lane == 8) ah.byte[b & 3] = (uint8_t)(late_part << (7 - k));
          else if (lane == 0) { ah.pr[b & 3][k] = *(const volatile int16_t*)&lstmpr[q]; ah.ex[b & 3][k] = *(const volatile uint8_t*)&lstmex[q]; }
          if (k == 7) for (int i = lane; i < (int)(sizeof(FxByteRec) / 4); i += 64) ((uint32_t*)&loc.rec[b & 1])[i] = ((const volatile uint32_t*)&recs[b])[i];
        }
        if (lane == 0) late_go = ok ? 1 : 0;
      }
      __syncthreads();
      if (!late_go) return;
    }
    FxBit u = fx_bit_dev(d, &ah, &loc, q, blpos0, lastbyte0, have0);
    float* const real_row = q + 1 < nbits ? out + (long)(q + 1) * ostride : gd->pending;
    unsigned* const row = rows + (size_t)q * FX_ROW_WORDS;
    const FxLayout l = u.normal ? ln : fxd_layout(d, 0);
    if (role == 1) {
      // ================= role U: match models, SSCMs, run map, LSTM input =================
      u.orow = loc.ex[0];
      int16_t* txn = sh.tx[1];
      if (wave == 0) {
        if (lane == 0) {
          if (u.boundary) { d->buffer[(uint32_t)d->pos & FX_BMASK] = (uint8_t)u.lastbyte; d->pos++; }   // fxd_match_unit's first line (:3806-3807)
          fxd_match2(d, &sh, u, txn + 2 * FX_NSSCM, u.orow + FX_NSSCM, &sh.isMatch);
        }
        __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): the contexts are in LDS
        __builtin_amdgcn_wave_barrier();
        if (lane < 3) fxd_match2_sm(d, &sh, u, lane);
        if (lane == 0) fxd_sparse(d, u, txn + 2 * FX_NSSCM + 7, u.orow + FX_NSSCM + 7);
      } else if (wave == 1) { if (lane < FX_NSSCM) fxd_sscm_unit(d, &sh, u, lane); }
      else if (wave == 2) {
        if (lane == 0) fxd_rcm_unit(d, &sh, u);
        else if (lane == 1) txn[l.tx_lstm] = d->stretch[u.lstmpr];
      } else if (wave == 3) fx_stage_ahead(&ah, &loc, bytes, recs, lstmpr, lstmex, n, q, lane);
      FX_TICK(0);
      fx_lds_barrier();
      FX_TICK(1);
      if (tid < 12) fx_st_u(row + FX_ROW_UTX + tid, reinterpret_cast<const uint32_t*>(txn)[tid]);
      else if (tid == 12) fx_st_u(row + FX_ROW_UTX + 12, (uint32_t)(uint16_t)txn[l.tx_rcm] | ((uint32_t)(uint16_t)txn[l.tx_lstm] << 16));
      else if (tid == 13) fx_st_u(row + FX_ROW_MATCH, (unsigned)sh.isMatch);
      else if (tid >= 64 && tid < 64 + FX_NSSCM + 9) real_row[tid - 64] = loc.ex[0][tid - 64];
      else if (tid >= 128 && tid < 130) real_row[l.exp_rcm + (tid - 128)] = loc.ex[0][l.exp_rcm + (tid - 128)];
      FX_TICK(2);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // as in role M: the row is complete before its counter moves
      __syncthreads();
      if (tid == 0) fx_st_u(&X->u_done, (unsigned)(q + 1));
      FX_TICK(3);
    } else {
      // ================= role X: trainers, selectors, mixers, APM chain =================
      u.orow = loc.ex[q & 1];
      const unsigned long long wt0 = (prof && lane == 0) ? __builtin_readcyclecounter() : 0ull;
      if (!LATE) {   // what learns from the coded bit needs nothing of this bit's rows: it runs under the wait for them
        if (tid < FX_THREADS) {   // thread t OWNS word t (two weights) of every first-layer row: it trains the words it used in the last dot products (registers), stores them
          // without a wait and reloads them in phase 3 -- a word is only ever loaded and stored by its owner, in order, so nothing has to be drained (see below)
          int e[FX_NMIX1];
#pragma unroll
          for (int k = 0; k < FX_NMIX1; k++) e[k] = merr[k];
#pragma unroll
          for (int k = 0; k < FX_NMIX1; k++)
            if (e[k]) { xw[k] = fx_pair_train(xt, xw[k], (int)(int16_t)e[k]); reinterpret_cast<uint32_t*>(d->wx[k] + (size_t)xcn[k] * FX_TX)[tid] = xw[k]; }
        } else if (wave == 6) {
          if (lane < 32) {   // fxd_train_small on the LDS copy of the selected rows: lane 16 j + i OWNS weight i of final mixer 10 + j (in the table as well)
            const int j = lane >> 4, i = lane & 15;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int err = merr[10 + j];
            if (err) {
              const int16_t w = fxd_train1(sh.in2[i], msw[lane], (int16_t)err);
              msw[lane] = w;
              (d->wx[10 + j] + (size_t)mscx[j] * 16)[i] = w;
            }
          }
        } else if (wave == 7) {
          unsigned long long w7a = 0, w7b = 0;
          if (lane < 6) {   // fxd_apm_update with the two cells taken from the row in LDS: this wave fetched it after its own last store, nobody else writes the table
            const int j = lane, rate = j == 0 ? 3 : j == 1 ? u.rate + 1 : u.rate;
            const int idx = sh.apm_index[j];
            if (xapm_ok) {
              const int i = idx - (int)loc.apm.cx[j] * 33;
              const int g = (u.y << 16) + (u.y << rate) - u.y * 2;
              const int t0 = loc.apm.row[j][i], t1 = loc.apm.row[j][i + 1];
              uint16_t* t = d->apm_t[j] + idx;
              t[0] = (uint16_t)(t0 + ((g - t0) >> rate));
              t[1] = (uint16_t)(t1 + ((g - t1) >> rate));
            } else fxd_apm_update(d, &sh, u, lane);   // the chunk's first update: the row of the previous chunk's last bit is not in LDS
            if (prof && lane == 0) w7a = __builtin_readcyclecounter();
            if (xapm_ok) apmcx[j] = apmspec[fx_fail_combo(&sh, u)][j];   // computed for the three possible outcomes under the previous bit's phase 5
            else {
              uint32_t cx[6];
              fx_apm_ctx(&sh, u, cx);
              apmcx[j] = j == 0 ? cx[0] : j == 1 ? cx[1] : j == 2 ? cx[2] : j == 3 ? cx[3] : j == 4 ? cx[4] : cx[5];
            }
          }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the row numbers are in LDS; the cell stores above and the fetches below are the same wave's: performed in order)
          __builtin_amdgcn_wave_barrier();
          if (prof && lane == 0) w7b = __builtin_readcyclecounter();
          fx_apm_issue(d, apmcx, &xapm, lane);
          if (prof && lane == 0) { const unsigned long long w7c = __builtin_readcyclecounter(); w7[0] += w7a - wt0; w7[1] += w7b - w7a; w7[2] += w7c - w7b; }
        }
        else if (wave == 5) fx_stage_ahead(&ah, &loc, bytes, recs, lstmpr, lstmex, n, q, lane);
        xapm_ok = true;
      }
      if (prof && lane == 0 && role == 2) wtop += __builtin_readcyclecounter() - wt0;   // this wave's share of the top block (before the wait for the rows)
      if (!LATE) {   // one lane per row counter (role M's wavefronts, role U) on wave 4, which has nothing else to do up here; the value the poll returned is kept
        if (wave == 4 && lane <= FX_M_WAVES && have_row < (unsigned)(q + 1)) {
          unsigned* const c = lane < FX_M_WAVES ? &X->mw_done[lane] : &X->u_done;
          unsigned it = 0, v;
          while ((v = fx_ld_u(c)) < (unsigned)(q + 1))
            if ((++it & 4095u) == 0 && (it > (unsigned)FX_ROLE_SPIN || fx_ld_u(&X->fail))) { fx_st_u(&X->fail, 1u); break; }
          have_row = v;
        }
      } else
      if (tid <= FX_M_WAVES && have_row < (unsigned)(q + 1)) {   // one lane per row counter: role M's wavefronts, role U
        unsigned* const c = tid < FX_M_WAVES ? &X->mw_done[tid] : &X->u_done;
        fx_wait_ge(c, (unsigned)(q + 1), &X->fail);
        have_row = fx_ld_u(c);
      }
      FX_TICK(0);
      fx_lds_barrier();
      FX_TICK(1);
      {   // the bit's inputs: the maps' range from M's row, the units' from U's
        int16_t* txn = sh.tx[sh.parity ^ 1];
        if (tid < FX_THREADS) {
          if (u.normal) {
            const uint32_t w = fx_ld_u(row + tid);
            const int i0 = 2 * tid, lo = FX_NSSCM * 2 + 9;
            if (i0 >= lo && i0 < l.tx_rcm) txn[i0] = (int16_t)(w & 0xffff);
            if (i0 + 1 >= lo && i0 + 1 < l.tx_rcm) txn[i0 + 1] = (int16_t)(w >> 16);
          }
        } else if (tid < FX_THREADS + 12) {
          const int i = tid - FX_THREADS;
          const uint32_t w = fx_ld_u(row + FX_ROW_UTX + i);
          txn[2 * i] = (int16_t)(w & 0xffff);
          if (2 * i + 1 < FX_NSSCM * 2 + 9) txn[2 * i + 1] = (int16_t)(w >> 16);
        } else if (tid == FX_THREADS + 12) {
          const uint32_t w = fx_ld_u(row + FX_ROW_UTX + 12);
          txn[l.tx_rcm] = (int16_t)(w & 0xffff); txn[l.tx_lstm] = (int16_t)(w >> 16);
        } else if (tid == FX_THREADS + 13) sh.isMatch = (int)fx_ld_u(row + FX_ROW_MATCH);
        else if (tid >= FX_THREADS + 16 && tid < FX_THREADS + 24) res8_s[tid - FX_THREADS - 16] = (int)fx_ld_u(row + FX_ROW_RES + (tid - FX_THREADS - 16));
      }
      if (!LATE) {
        // phase 2 (the selectors) in the same interval as the gather: its two lanes fetch the nine words of the row they need themselves
        if (tid == 64) {
          int r8[8];
#pragma unroll
          for (int i = 0; i < 8; i++) r8[i] = (int)fx_ld_u(row + FX_ROW_RES + i);
          sh.isMatch = (int)fx_ld_u(row + FX_ROW_MATCH);
#pragma unroll
          for (int i = 0; i < 8; i++) res8_s[i] = r8[i];
          fx_phase2b_tail(d, &sh, u, res8_s);
        } else if (tid == 320) {
          const int im = (int)fx_ld_u(row + FX_ROW_MATCH);
          fx_phase2a_dev(d, &sh, u, im);
        }
        FX_TICK(2);
        fx_lds_barrier();   // (an LDS barrier: nothing stored to the tables in this bit is read from them in this bit)
      } else {
      if (wave == 2 || wave == 3) fx_train_rows_fast(d, &sh, u, tid - 128);
      else if (wave == 6) { if (lane == 2 || lane == 3) fxd_train_small(d, &sh, u, 10 + lane - 2); }
      else if (wave == 7) { if (lane < 6) { fxd_apm_update(d, &sh, u, lane); fx_apm_prefetch(d, &sh, u, &loc.apm, lane); } }
      FX_TICK(2);
      // the full barrier of the bit: the trained rows and APM cells are stored before phase 3 / 5 read them
      __syncthreads();
      }
      FX_TICK(3);
      if (LATE) {
        if (tid == 0) fx_phase2b_tail(d, &sh, u, res8_s);
        else if (tid == 256) fx_phase2a_dev(d, &sh, u);
        fx_lds_barrier();
      }
      FX_TICK(4);
      const unsigned long long wt3 = (prof && lane == 0) ? __builtin_readcyclecounter() : 0ull;
      if (!LATE) {
        if (tid < FX_THREADS) {   // phase 3: the ten selected rows' words from the table (each by its owner: what it stored above comes back, in order), the pair products
          const uint32_t t = reinterpret_cast<const uint32_t*>(sh.tx[sh.parity ^ 1])[tid];
#pragma unroll
          for (int k = 0; k < FX_NMIX1; k++) xcn[k] = sh.mx_cxt[k];
#pragma unroll
          for (int k = 0; k < FX_NMIX1; k++) xw[k] = reinterpret_cast<const uint32_t*>(d->wx[k] + (size_t)xcn[k] * FX_TX)[tid];
#pragma unroll
          for (int k = 0; k < FX_NMIX1; k++) sh.part[k][tid] = fx_pair_dot(t, xw[k]);
          xt = t;
        } else if (wave == 6) {
          if (lane < 32) {   // the final mixers' rows for phase 5, fetched here if their selector moved (each lane its own weight: it stored the old one itself)
            const int j = lane >> 4, i = lane & 15, c = sh.mx_cxt[10 + j];
            if (c != mscx[j]) msw[lane] = (d->wx[10 + j] + (size_t)c * 16)[i];
          }
        }
      } else if (tid < FX_THREADS) fx_phase3_dots(d, &sh, u, tid);
      if (prof && lane == 0 && role == 2) wp3 += __builtin_readcyclecounter() - wt3;
      fx_lds_barrier();
      if (tid < FX_THREADS) fxd_phase4(d, &sh, u, tid);
      else if (!LATE && wave == 7) { fx_apm_commit(&loc.apm, apmcx, &xapm, lane); }
      else if (!LATE && wave == 6) { if (lane >= 32 && lane < 34) mscx[lane - 32] = sh.mx_cxt[10 + lane - 32]; }
      fx_lds_barrier();
      FX_TICK(5);
      if (tid < 64) { if (!LATE) fx_phase5_dev(d, &sh, u, &loc.apm, tid, scr_s, msw, mscx); else fx_phase5_dev(d, &sh, u, &loc.apm, tid, scr_s); }
      else if (!LATE && wave == 7 && q + 1 < nbits) {   // the NEXT update's APM contexts for the three possible outcomes of this bit's probability (fx_apm_ctx)
        if (lane < 18) {
          const FxBit un = fx_bit_dev(d, &ah, &loc, q + 1, blpos0, lastbyte0, have0);
          const int combo = lane / 6, j = lane - 6 * combo;
          uint32_t cx[6];
          fx_apm_ctx(&sh, un, cx, combo);
          apmspec[combo][j] = j == 0 ? cx[0] : j == 1 ? cx[1] : j == 2 ? cx[2] : j == 3 ? cx[3] : j == 4 ? cx[4] : cx[5];
        }
      }
      else if (!LATE && wave == 4 && q + 1 < nbits) {   // the NEXT update's mixer errors (fxd_mixer_err: the coded bit is known, the outputs are this bit's) -- once for all trainers.
        // mx_pr[0..9] / [10..11] are written by wave 0 during phase 5: this wave waits for its two flags (scr_s[12], scr_s[13] = bit + 1)
        if (lane < 12) {
          const int* const flag = &scr_s[lane < FX_NMIX1 ? 12 : 13];
          unsigned spins = 0;
          bool gone = false;
          while (*(volatile const int*)flag != q + 1) {   // bounded like every other in-launch wait: a flag that never comes fails the stage (fx_wait_ge's rule)
            __builtin_amdgcn_s_sleep(1);
            if ((++spins & 0xFFFFu) == 0 && (spins > (1u << 26) || __hip_atomic_load(&X->fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) { gone = true; break; }
          }
          if (gone) __hip_atomic_store(&X->fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const FxBit un = fx_bit_dev(d, &ah, &loc, q + 1, blpos0, lastbyte0, have0);
          merr[lane] = fxd_mixer_err(&sh, d, un, lane);
        }
      }
      fx_lds_barrier();
      FX_TICK(6);
      for (int i = l.exp_mix + tid; i < FX_OUTPUTS; i += FX_DEV_THREADS) real_row[i] = loc.ex[q & 1][i];
      FX_TICK(7);
      if (LATE && q + 1 < nbits) {   // row q + 1 is complete (roles M and U finished theirs before this update began)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every wave's stores of the row have landed before the barrier (s_barrier alone does not wait for them)
        __syncthreads();
        if (tid == 0) late_publish(LB, LC_FX, (unsigned)(q + 2));
      }
    }
  }
  __syncthreads();
  if (prof && tid == 0 && blockIdx.x != 1) for (int k = 0; k < 8; k++) prof[16 * role + k] += pacc[k];
  if (prof && role == 2 && lane == 0) { prof[48 + wave] += wtop; prof[56 + wave] += wp3; if (wave == 7) { prof[40] += w7[0]; prof[41] += w7[1]; prof[42] += w7[2]; } }
  if (prof && blockIdx.x == 0 && tid < 64) prof[64 + tid] += pbp[tid >> 3][tid & 7];
#undef FX_TICK
  // ---- every role writes its own part of the stream's state back ----
  if (role == 0) {   // (the maps of this workgroup's wavefronts)
    const int ka = d->mw_map[8 * mwg], kb = d->mw_map[8 * mwg + 8];
    for (int i = 8 * ka + tid; i < 8 * kb; i += FX_DEV_THREADS) {
      FxMapDev* x = &gd->maps[i >> 3];
      const int j = i & 7;
      x->cp[j] = sh.mcp[i >> 3][j]; x->cp0[j] = sh.mcp0[i >> 3][j]; x->runp[j] = sh.mrunp[i >> 3][j]; x->cxt[j] = sh.mcxt[i >> 3][j]; x->sm_cxt[j] = sh.msmc[i >> 3][j];
    }
    for (int k = ka; k < kb; k++) {
      const FxMapDev* x = &d->maps[k];
      for (int i = tid; i < x->C * 256; i += FX_DEV_THREADS) x->sm[i] = (&sh.sm[x->slot_base][0])[i];
    }
  } else if (role == 1) {
    const int w0 = (int)(offsetof(FxDev, sscm_data) / 4), w1 = (int)(offsetof(FxDev, wx) / 4);
    for (int i = w0 + tid; i < w1; i += FX_DEV_THREADS) ((uint32_t*)gd)[i] = ((const uint32_t*)d)[i];
  } else {
    if (tid < FX_THREADS) {   // the mixer / APM part of fxd_store_shared, into the LDS copy first
      for (int i = tid; i < 2 * FX_TX; i += FX_THREADS) (&d->tx[0][0])[i] = (&sh.tx[0][0])[i];
      if (tid < 16) d->in2[tid] = sh.in2[tid];
      if (tid < 12) { d->mx_elim[tid] = sh.mx_elim[tid]; d->mx_cxt[tid] = sh.mx_cxt[tid]; d->mx_pr[tid] = sh.mx_pr[tid]; }
      if (tid < 6) d->apm_index[tid] = sh.apm_index[tid];
      if (tid == 0) {
        d->pr = sh.pr; d->parity = sh.parity; d->fails = sh.fails; d->failz = sh.failz; d->failcount = sh.failcount;
        d->blpos = blpos0 + n; d->lastbyte = ah.byte[(n - 1) & 3]; d->have_rec = 1; d->rec = loc.rec[(n - 1) & 1];
      }
    }
    __syncthreads();
    const int w0 = (int)(offsetof(FxDev, mx_elim) / 4), w1 = (int)(offsetof(FxDev, pending) / 4);
    for (int i = w0 + tid; i < w1; i += FX_DEV_THREADS) ((uint32_t*)gd)[i] = ((const uint32_t*)d)[i];
    const int r0 = (int)(offsetof(FxDev, rec) / 4), r1 = r0 + (int)(sizeof(FxByteRec) / 4);
    for (int i = r0 + tid; i < r1; i += FX_DEV_THREADS) ((uint32_t*)gd)[i] = ((const uint32_t*)d)[i];
  }
}

__global__ __launch_bounds__(FX_DEV_THREADS) void cmx_fxcm_roles_kernel(FxDev* gd, FxXfer* X, unsigned* rows, const uint8_t* bytes, const FxByteRec* recs,
                                                                         const int16_t* lstmpr, const uint8_t* lstmex, float* out, long ostride, int n,
                                                                         unsigned long long* prof) {
  fx_roles_body<false>(gd, X, rows, bytes, recs, lstmpr, lstmex, out, ostride, n, prof, CmxLate());
}
__global__ __launch_bounds__(FX_DEV_THREADS) void cmx_fxcm_roles_late_kernel(FxDev* gd, FxXfer* X, unsigned* rows, CmxLate box, const FxByteRec* recs,
                                                                              const int16_t* lstmpr, const uint8_t* lstmex, float* out, long ostride, int n) {
  fx_roles_body<true>(gd, X, rows, nullptr, recs, lstmpr, lstmex, out, ostride, n, nullptr, box);
}

__global__ void cmx_fxcm_pattern16_kernel(uint16_t* p, size_t n, const uint16_t* pat, int plen) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = pat[i % (size_t)plen];
}

static const size_t FX_LDS_BYTES = ((sizeof(FxShared) + 15) & ~(size_t)15) + sizeof(FxLocal);

namespace {
struct DevPolicy {
  std::vector<void*> blocks;
  bool ok = true;
  void* zalloc(size_t bytes) {
    void* p = nullptr;
    if (!ok || hipMalloc(&p, bytes + 64) != hipSuccess || hipMemset(p, 0, bytes + 64) != hipSuccess) { ok = false; return nullptr; }
    blocks.push_back(p);
    return p;
  }
  void fill16(void* p, size_t n, uint16_t v) { if (p && hipMemsetD16((hipDeviceptr_t)p, v, n) != hipSuccess) ok = false; }
  void fill32(void* p, size_t n, uint32_t v) { if (p && hipMemsetD32((hipDeviceptr_t)p, (int)v, n) != hipSuccess) ok = false; }
  void pattern16(void* p, size_t n, const uint16_t* pat, int plen) {
    if (!p) return;
    uint16_t* dp = (uint16_t*)zalloc((size_t)plen * 2);
    if (!dp) return;
    upload(dp, pat, (size_t)plen * 2);
    hipLaunchKernelGGL(cmx_fxcm_pattern16_kernel, dim3(1024), dim3(256), 0, 0, (uint16_t*)p, n, dp, plen);
    if (hipDeviceSynchronize() != hipSuccess) ok = false;
  }
  void upload(void* dst, const void* src, size_t bytes) { if (dst && hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) ok = false; }
};
enum { FX_STAGE_BUFS = CMX_PIPELINE_SLOTS };
}  // namespace

struct cmx_fxcm {
  int device = 0;
  FxParser* parser = nullptr;
  DevPolicy pol;
  FxDev* d_dev = nullptr;
  // record staging: page-locked host buffers and their device twins, recycled once the copy that read them is done
  FxByteRec* h_recs[FX_STAGE_BUFS] = {};
  FxByteRec* d_recs[FX_STAGE_BUFS] = {};
  hipEvent_t done[FX_STAGE_BUFS] = {};
  size_t cap[FX_STAGE_BUFS] = {};
  bool used[FX_STAGE_BUFS] = {};
  int next = 0;
  uint64_t bytes_done = 0;
  unsigned long long* d_prof = nullptr;   // CMX_FXCM_PROFILE=1: thread 0's clocks per phase of each role (prof[16 role + k])
  FxXfer* d_xfer = nullptr;               // three-role kernel: hand-off counters and the rows M / U publish for X
  unsigned* d_rows = nullptr; size_t rows_cap = 0;
  hipStream_t s_up = nullptr; bool own_up = false;   // record uploads: a stream that never has a kernel in front of a copy (cmx_fxcm_set_upload_stream)
  hipEvent_t ev_up[FX_STAGE_BUFS] = {};
  FxByteRec* late_recs[3] = {}; FxByteRec* late_drecs[3] = {}; size_t late_cap[3] = {};   // the decoder's form (cmx_late.h): host-coherent records and their device mirrors (the relay wave copies a byte's record over when the byte is complete), three chunk slots
};

extern "C" {

void cmx_fxcm_destroy(cmx_fxcm_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->pol.blocks) (void)hipFree(p);
  if (h->d_dev) (void)hipFree(h->d_dev);
  for (int i = 0; i < FX_STAGE_BUFS; i++) {
    if (h->h_recs[i]) (void)hipHostFree(h->h_recs[i]);
    if (h->d_recs[i]) (void)hipFree(h->d_recs[i]);
    if (h->done[i]) (void)hipEventDestroy(h->done[i]);
  }
  if (h->d_prof) (void)hipFree(h->d_prof);
  if (h->d_xfer) (void)hipFree(h->d_xfer);
  for (hipEvent_t e : h->ev_up) if (e) (void)hipEventDestroy(e);
  if (h->own_up && h->s_up) (void)hipStreamDestroy(h->s_up);
  if (h->d_rows) (void)hipFree(h->d_rows);
  for (FxByteRec* r : h->late_recs) cmx_late_free(r);
  for (FxByteRec* r : h->late_drecs) cmx_late_free_dev(r);
  if (h->parser) fxp_destroy(h->parser);
  delete h;
}

cmx_fxcm_t* cmx_fxcm_create(const char* dictionary_path, int device) {
  if (cmx_device_count() <= 0) { cmx_set_err("cmx_fxcm_create: no HIP device visible (a gfx950 GPU is required)"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return nullptr; }
  cmx_fxcm_t* h = new cmx_fxcm();
  h->device = device;
  FxDev host;
  fxb::build(host, h->pol);
  const char* serial = getenv("CMX_FXCM_SERIAL_MAPS");   // A/B switch: one lane per map instead of one per context slot
  if (serial && serial[0] == '1') host.slot_parallel = 0;
  bool ok = h->pol.ok;
  ok = ok && hipMalloc((void**)&h->d_dev, sizeof(FxDev)) == hipSuccess;
  ok = ok && hipMemcpy(h->d_dev, &host, sizeof(FxDev), hipMemcpyHostToDevice) == hipSuccess;
  for (int i = 0; ok && i < FX_STAGE_BUFS; i++) ok = hipEventCreateWithFlags(&h->done[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&h->ev_up[i], hipEventDisableTiming) == hipSuccess;
  ok = ok && hipFuncSetAttribute((const void*)cmx_fxcm_roles_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FX_LDS_BYTES) == hipSuccess;
  ok = ok && hipFuncSetAttribute((const void*)cmx_fxcm_roles_late_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)FX_LDS_BYTES) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_xfer, sizeof(FxXfer)) == hipSuccess && hipMemset(h->d_xfer, 0, sizeof(FxXfer)) == hipSuccess;
  const char* prof = getenv("CMX_FXCM_PROFILE");
  if (ok && prof && prof[0] == '1') ok = hipMalloc((void**)&h->d_prof, 1024) == hipSuccess && hipMemset(h->d_prof, 0, 1024) == hipSuccess;
  ok = ok && hipDeviceSynchronize() == hipSuccess;
  if (!ok) { cmx_set_err("cmx_fxcm_create: allocation / init failed (the stage needs ~4.4 GB of HBM)"); cmx_fxcm_destroy(h); return nullptr; }
  h->parser = fxp_create(dictionary_path);
  return h;
}

int cmx_fxcm_run(cmx_fxcm_t* h, const uint8_t* bytes, const uint8_t* d_bytes, size_t nbytes, const int16_t* d_lstmpr, const uint8_t* d_lstmex,
                 float* d_probs, size_t pstride, void* stream) {
  if (!h) { cmx_set_err("cmx_fxcm_run: null handle"); return 1; }
  if (nbytes == 0) return 0;
  if (!bytes || !d_bytes || !d_lstmpr || !d_lstmex || !d_probs || pstride < 3 + FX_OUTPUTS || nbytes > (1u << 24)) { cmx_set_err("cmx_fxcm_run: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  const int b = h->next;
  h->next = (h->next + 1) % FX_STAGE_BUFS;
  if (h->used[b] && hipEventSynchronize(h->done[b]) != hipSuccess) { cmx_set_err("cmx_fxcm_run: staging buffer wait failed"); return 1; }
  if (h->cap[b] < nbytes) {
    if (h->h_recs[b]) (void)hipHostFree(h->h_recs[b]);
    if (h->d_recs[b]) (void)hipFree(h->d_recs[b]);
    h->h_recs[b] = nullptr; h->d_recs[b] = nullptr; h->cap[b] = 0;
    if (hipHostMalloc((void**)&h->h_recs[b], nbytes * sizeof(FxByteRec), hipHostMallocDefault) != hipSuccess ||
        hipMalloc((void**)&h->d_recs[b], nbytes * sizeof(FxByteRec)) != hipSuccess) { cmx_set_err("cmx_fxcm_run: staging allocation failed"); return 1; }
    h->cap[b] = nbytes;
  }
  if (fxp_run(h->parser, bytes, (int)nbytes, h->h_recs[b]) != 0) { cmx_set_err("cmx_fxcm_run: parser emitted a context count a map does not expect"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  // the records go up on the upload stream (nothing in front of them), the kernel's stream waits for the copy: enqueued on
  // `s` the copy would sit behind the previous chunk's kernel, and a host-to-device copy that waits in stream order holds up
  // every later copy of the process
  if (!h->s_up) { if (hipStreamCreateWithFlags(&h->s_up, hipStreamNonBlocking) != hipSuccess) { cmx_set_err("cmx_fxcm_run: stream creation failed"); return 1; } h->own_up = true; }
  if (hipMemcpyAsync(h->d_recs[b], h->h_recs[b], nbytes * sizeof(FxByteRec), hipMemcpyHostToDevice, h->s_up) != hipSuccess ||
      hipEventRecord(h->ev_up[b], h->s_up) != hipSuccess || hipStreamWaitEvent(s, h->ev_up[b], 0) != hipSuccess) { cmx_set_err("cmx_fxcm_run: record upload failed"); return 1; }
  if (h->rows_cap < nbytes) {   // grown between chunks: nothing of this stream may be in flight on the old buffer
    if (hipDeviceSynchronize() != hipSuccess) { cmx_set_err("cmx_fxcm_run: device error"); return 1; }
    if (h->d_rows) (void)hipFree(h->d_rows);
    h->d_rows = nullptr; h->rows_cap = 0;
    if (hipMalloc((void**)&h->d_rows, nbytes * 8 * FX_ROW_WORDS * 4) != hipSuccess) { cmx_set_err("cmx_fxcm_run: row buffer allocation failed"); return 1; }
    h->rows_cap = nbytes;
  }
  if (hipMemsetAsync((char*)h->d_xfer + 16, 0, sizeof(FxXfer) - 16, s) != hipSuccess) { cmx_set_err("cmx_fxcm_run: hipMemsetAsync failed"); return 1; }
  hipLaunchKernelGGL(cmx_fxcm_roles_kernel, dim3(FX_M_WGS + 2), dim3(FX_DEV_THREADS), FX_LDS_BYTES, s, h->d_dev, h->d_xfer, h->d_rows, d_bytes, h->d_recs[b], d_lstmpr, d_lstmex,
                     d_probs + 3, (long)pstride, (int)nbytes, h->d_prof);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_fxcm_run: ") + hipGetErrorString(e)); return 1; }
  if (hipEventRecord(h->done[b], s) != hipSuccess) { cmx_set_err("cmx_fxcm_run: event record failed"); return 1; }
  h->used[b] = true;
  h->bytes_done += nbytes;
  return 0;
}

// ---- the decoder's form of a chunk (cmx_late.h): the kernel is launched for the next nbytes bytes, which arrive bit by bit through
// `box`; the parser's record of a byte is written by cmx_fxcm_late_byte() when the decoder has completed it (before the byte's last
// bit is published). hint_pr / hint_ex: the ByteModel kernel's per-update LSTM hints (host-coherent, counter LC_BM2); d_probs: rows
// in memory the mixing network's kernel sees while both run. slot 0..2: the stage's set of record buffers for this chunk.
// everything the decoder's form allocates, for chunks of up to nbytes bytes: before the first chunk's kernels are launched
int cmx_fxcm_late_prepare(cmx_fxcm_t* h, size_t nbytes) {
  if (!h || nbytes == 0 || nbytes > (1u << 16)) { cmx_set_err("cmx_fxcm_late_prepare: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  for (int slot = 0; slot < 3; slot++) {
    if (h->late_cap[slot] >= nbytes) continue;
    if (h->late_cap[slot]) { cmx_set_err("cmx_fxcm_late_prepare: the chunk size may not grow"); return 1; }
    h->late_recs[slot] = (FxByteRec*)cmx_late_alloc(nbytes * sizeof(FxByteRec));
    h->late_drecs[slot] = (FxByteRec*)cmx_late_alloc_dev(h->device, nbytes * sizeof(FxByteRec));
    if (!h->late_recs[slot] || !h->late_drecs[slot]) { cmx_set_err("cmx_fxcm_late_prepare: record buffer allocation failed"); return 1; }
    h->late_cap[slot] = nbytes;
  }
  if (h->rows_cap < nbytes) {
    if (hipDeviceSynchronize() != hipSuccess) { cmx_set_err("cmx_fxcm_late_prepare: device error"); return 1; }
    if (h->d_rows) (void)hipFree(h->d_rows);
    h->d_rows = nullptr; h->rows_cap = 0;
    if (hipMalloc((void**)&h->d_rows, nbytes * 8 * FX_ROW_WORDS * 4) != hipSuccess) { cmx_set_err("cmx_fxcm_late_prepare: row buffer allocation failed"); return 1; }
    h->rows_cap = nbytes;
  }
  return 0;
}
int cmx_fxcm_run_late(cmx_fxcm_t* h, void* box, size_t nbytes, const int16_t* hint_pr, const uint8_t* hint_ex, float* d_probs, size_t pstride, int slot, void* stream) {
  if (!h || !box || !hint_pr || !hint_ex || !d_probs || nbytes == 0 || nbytes > (1u << 16) || pstride < 3 + FX_OUTPUTS || slot < 0 || slot > 2) { cmx_set_err("cmx_fxcm_run_late: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  if (h->late_cap[slot] < nbytes || h->rows_cap < nbytes) { cmx_set_err("cmx_fxcm_run_late: call cmx_fxcm_late_prepare first (nothing may be allocated while the stream's kernels run)"); return 1; }
  hipStream_t s = (hipStream_t)stream;
  if (hipMemsetAsync((char*)h->d_xfer + 16, 0, sizeof(FxXfer) - 16, s) != hipSuccess) { cmx_set_err("cmx_fxcm_run_late: hipMemsetAsync failed"); return 1; }
  hipLaunchKernelGGL(cmx_fxcm_roles_late_kernel, dim3(FX_M_WGS + 2), dim3(FX_DEV_THREADS), FX_LDS_BYTES, s, h->d_dev, h->d_xfer, h->d_rows, *(const CmxLate*)box,
                     (const FxByteRec*)h->late_drecs[slot], hint_pr, hint_ex, d_probs + 3, (long)pstride, (int)nbytes);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_fxcm_run_late: ") + hipGetErrorString(e)); return 1; }
  h->bytes_done += nbytes;
  return 0;
}
static_assert(sizeof(FxByteRec) % 4 == 0, "the relay copies whole words");
// what the stream's relay wave has to bring over for this stage (cmx_late.h): the parser's record of every completed byte
int cmx_fxcm_late_relay(cmx_fxcm_t* h, int slot, void* out_, int max) {
  if (!h || slot < 0 || slot > 2 || !h->late_cap[slot] || !out_ || max < 1) { cmx_set_err("cmx_fxcm_late_relay: bad argument (prepare first)"); return -1; }
  cmx_late_relay_t* out = (cmx_late_relay_t*)out_;
  out[0].src = h->late_recs[slot]; out[0].dst = h->late_drecs[slot]; out[0].stride = (uint32_t)sizeof(FxByteRec); out[0].kind = 2;
  return 1;
}
// byte number b of the chunk in `slot` is complete: the parser's record of it
int cmx_fxcm_late_byte(cmx_fxcm_t* h, int slot, size_t b, uint8_t byte) {
  if (!h || slot < 0 || slot > 2 || b >= h->late_cap[slot]) { cmx_set_err("cmx_fxcm_late_byte: bad argument"); return 1; }
  if (fxp_run(h->parser, &byte, 1, h->late_recs[slot] + b) != 0) { cmx_set_err("cmx_fxcm_late_byte: parser emitted a context count a map does not expect"); return 1; }
  return 0;
}

// CMX_FXCM_PROFILE=1: thread 0's clocks per phase of each role since creation: out64[16 role + k], role 0 = M, 1 = U, 2 = X
int cmx_fxcm_profile(cmx_fxcm_t* h, unsigned long long out64[128]) {
  if (!h || !h->d_prof) return 1;
  return hipMemcpy(out64, h->d_prof, 1024, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}

int cmx_fxcm_set_upload_stream(cmx_fxcm_t* h, void* stream) {
  if (!h) { cmx_set_err("cmx_fxcm_set_upload_stream: null handle"); return 1; }
  if (h->own_up && h->s_up) (void)hipStreamDestroy(h->s_up);
  h->s_up = (hipStream_t)stream; h->own_up = false;
  return 0;
}

// 1 = a bounded in-launch wait of the three-role kernel ran out (the stream's fxcm columns are void); synchronises the device
int cmx_fxcm_failed(cmx_fxcm_t* h) {
  if (!h) return 1;
  (void)hipSetDevice(h->device);
  if (hipDeviceSynchronize() != hipSuccess) return 1;
  unsigned f = 0;
  if (hipMemcpy(&f, &h->d_xfer->fail, 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  return f ? 1 : 0;
}

// DEVICE address of that flag (see cmx_lstm_fail_flag)
const unsigned* cmx_fxcm_fail_flag(cmx_fxcm_t* h) { return h ? &h->d_xfer->fail : nullptr; }

int cmx_fxcm_sync(cmx_fxcm_t* h) {
  if (!h) { cmx_set_err("cmx_fxcm_sync: null handle"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_fxcm_sync: ") + hipGetErrorString(e)); return 1; }
  return 0;
}

}  // extern "C"
