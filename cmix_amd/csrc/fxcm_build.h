// fxcm_build.h -- host-side construction of a stream's FxDev (fxcm_dev.h): table geometry, start-up values and the
// derived lookup tables of every component, following fxcm's Predictor constructor and PredictorInit (reference
// src/models/fxcmv1.cpp:4845-4876, :3313-3405) and the Init() of each class. Memory comes from a policy object so the
// same code fills device memory (fxcm_stage.hip: hipMalloc / hipMemset / fill kernels) and host memory
// (tests/host/fxcm_emul.cpp: calloc), which is how the kernel body is checked against the oracle without a GPU.
#ifndef CMX_FXCM_BUILD_H
#define CMX_FXCM_BUILD_H
#include <cstring>
#include <vector>

#include "cmx_fxcm_tables.h"
#include "fxcm_dev.h"

namespace fxb {
inline int squash(int d) { return d < -2047 ? 1 : d > 2047 ? 4095 : FX_SQUASH[d + 2047]; }
inline int clp(int z) { return z < -2047 ? -2047 : z > 2047 ? 2047 : z; }
inline int sc(int p) { return p > 0 ? p >> 7 : (p + 127) >> 7; }   // :906-909
inline const uint8_t* sta(int which) {
  static const uint8_t* const tabs[6] = {FX_STA1, FX_STA2, FX_STA4, FX_STA5, FX_STA6, FX_STA7};
  return tabs[which];
}
inline int pre(const uint8_t* nn, int state) {
  const uint32_t n0 = nn[state * 4 + 2] * 3u + 1, n1 = nn[state * 4 + 3] * 3u + 1;
  return (int)((n1 << 12) / (n0 + n1));
}

// Policy: void* zalloc(size_t); void fill16(void*, size_t count, uint16_t); void fill32(void*, size_t count, uint32_t);
//         void pattern16(void*, size_t count, const uint16_t* pat, int plen); void upload(void* dst, const void* src, size_t)
template <class Policy>
void build(FxDev& h, Policy& P) {
  static const uint32_t c_r[27] = {3, 4, 6, 4, 6, 6, 2, 3, 3, 3, 6, 4, 3, 4, 5, 6, 2, 6, 4, 4, 4, 4, 4, 4, 4, 4, 4};                       // :3222-3230
  static const uint32_t c_s[27] = {28, 26, 28, 31, 34, 31, 33, 33, 35, 35, 29, 32, 33, 34, 30, 36, 31, 32, 32, 32, 32, 32, 33, 32, 32, 32, 32};
  static const uint32_t c_s3[27] = {43, 33, 34, 28, 34, 29, 32, 33, 37, 35, 33, 28, 31, 35, 28, 30, 33, 34, 32, 32, 32, 32, 32, 32, 32, 32, 32};
  static const uint32_t c_s4[27] = {9, 8, 9, 5, 8, 12, 15, 8, 8, 12, 10, 7, 7, 8, 8, 13, 13, 14, 8, 8, 12, 12, 12, 12, 12, 12, 12};
  memset(&h, 0, sizeof h);
  auto up = [&](const void* src, size_t bytes) { void* p = P.zalloc(bytes); P.upload(p, src, bytes); return p; };
  h.squash = (const int16_t*)up(FX_SQUASH, sizeof FX_SQUASH);
  h.stretch = (const int16_t*)up(FX_STRETCH, sizeof FX_STRETCH);
  uint8_t wrt[512];
  memcpy(wrt, FX_WRT_2B, 256); memcpy(wrt + 256, FX_WRT_3B, 256);
  h.wrt = (const uint8_t*)up(wrt, 512);
  const uint8_t* sta_dev[6];
  for (int k = 0; k < 6; k++) h.sta[k] = sta_dev[k] = (const uint8_t*)up(sta(k), 1024);

  int slot = 0, tx = 2 * FX_NSSCM + 7 + 2, ex = FX_NSSCM + 7 + 2;
  for (int k = 0; k < FX_NMAPS; k++) {
    const FxMapDef& df = FX_MAPS[k];
    FxMapDev& x = h.maps[k];
    x.A = df.kind == 0 ? 7 : df.kind == 1 ? 3 : 14;
    x.B = df.kind == 0 ? 64 : df.kind == 1 ? 32 : 128;
    x.C = df.C; x.kep = df.keep; x.u = df.u;
    size_t buckets;
    if (df.kind == 2) { const uint32_t m2 = df.size * 2; x.tmask = (m2 >> 7) - 1; buckets = (size_t)(m2 >> 7) + 128; }
    else { x.tmask = (df.size >> 6) - 1; buckets = (size_t)(df.size >> 6) + 64; }
    x.t = (uint8_t*)P.zalloc(buckets * (size_t)x.B);
    x.slot_base = slot; x.tx_off = tx; x.exp_off = ex;
    for (int i = 0; i < x.C; i++) { h.slot_map[slot + i] = (uint8_t)k; h.slot_idx[slot + i] = (uint8_t)i; }
    slot += x.C; tx += x.C * (5 + x.u); ex += x.C * (4 + x.u);
    const uint8_t* nn = sta(df.sta);
    x.nn = sta_dev[df.sta];
    // the per-map lookup tables (ContextMap::Init :1005-1055)
    std::vector<int16_t> tab(FX_TAB_LEN, 0);
    const int cmul = (int)c_r[df.prm], cms = (int)c_s[df.prm], cms3 = (int)c_s3[df.prm], cms4 = (int)c_s4[df.prm];
    for (int rc = 0; rc < 256; rc++) {
      int v = FX_ILOG[rc];
      v = v << (2 + (~rc & 1));
      if ((rc & 1) == 0) v = v * cmul / 4;
      tab[FX_TAB_RC1 + rc + 256] = (int16_t)clp(v);
      tab[FX_TAB_RC1 + rc] = (int16_t)clp(-v);
    }
    for (int i = 0; i < 4096; i++) {
      tab[FX_TAB_ST1 + i] = (int16_t)clp(sc(cms * FX_STRETCH[i]));
      tab[FX_TAB_ST2 + i] = df.st2 == 0 ? 0 : (int16_t)clp(sc((df.st2 == 1 ? 12 : 14) * (i - 2048)));
    }
    for (int s = 0; s < 256; s++) {
      const int n0 = -!nn[s * 4 + 2], n1 = -!nn[s * 4 + 3];
      int r = 0, sp0 = 0;
      if (n1 - n0 == 1) { sp0 = 0; r = 1; }
      if (n1 - n0 == -1) { sp0 = 4095; r = 1; }
      if (r) {
        tab[FX_TAB_ST8 + s] = (int16_t)clp(sc(cms4 * (pre(nn, s) - sp0)));
        tab[FX_TAB_ST32 + s] = (int16_t)clp(sc(cms3 * FX_STRETCH[pre(nn, s)]));
        if (s < 8) tab[FX_TAB_ST32 + s] = 0;
      }
    }
    x.tab = (const int16_t*)up(tab.data(), tab.size() * 2);
    std::vector<uint32_t> sm((size_t)x.C * 256);   // StateMap::Init :680-692
    for (size_t i = 0; i < sm.size(); ++i) {
      const uint32_t n0 = nn[(i & 255) * 4 + 2] * 3u + 1, n1 = nn[(i & 255) * 4 + 3] * 3u + 1;
      sm[i] = ((n1 << 20) / (n0 + n1)) << 12;
    }
    x.sm = (uint32_t*)up(sm.data(), sm.size() * 4);
    const uint32_t first = (uint32_t)(2 * x.A + 1);
    for (int i = 0; i < 8; ++i) { x.cp0[i] = x.cp[i] = first; x.runp[i] = first + 3; }
  }
  static const int scm_bits[FX_NSSCM] = {8, 8, 8, 9, 8, 8, 7};   // :3306-3312
  for (int j = 0; j < FX_NSSCM; j++) {
    const size_t n = ((size_t)1 << scm_bits[j]) * 255;
    h.sscm_data[j] = (uint16_t*)P.zalloc(n * 2);
    P.fill16(h.sscm_data[j], n, 0x7FFF);
    h.sscm_mask[j] = (1 << scm_bits[j]) - 1;
  }
  static const int sm1_n[3] = {1 << 9, 1 << 19, 1 << 16};
  for (int j = 0; j < 3; j++) {
    h.sm1_t[j] = (uint32_t*)P.zalloc((size_t)sm1_n[j] * 4);
    P.fill32(h.sm1_t[j], (size_t)sm1_n[j], 1u << 31);
    h.sm1_mask[j] = sm1_n[j] - 1;
  }
  const uint32_t G = 4096u * 4096u;
  h.rcm_t = (uint8_t*)P.zalloc((size_t)G + 64);
  h.rcm_n = G / 4 - 1; h.rcm_cp = 1;
  for (int k = 0; k < 256; k++) {
    int c = FX_ILOG[k] * 8;
    if ((k & 1) == 0) c = c * 6 / 4;
    h.rcm_rc[k + 256] = (int16_t)clp(c);
    h.rcm_rc[k] = (int16_t)clp(-c);
  }
  h.mhashmask = 0x200000 - 1;
  h.mhash = (uint32_t*)P.zalloc((size_t)(0x200000 + 32) * 16);
  h.sp_table = (uint32_t*)P.zalloc((size_t)1024 * 1024 * 4);
  for (int i = 0; i < 4; i++) { h.sp_list.prev[i] = i - 1; h.sp_list.next[i] = i + 1; }
  h.sp_list.next[3] = -1;
  h.buffer = (uint8_t*)P.zalloc((size_t)FX_BMASK + 1);
  static const int mx[12][4] = {{2048, 237, 8, 69}, {6 * 256, 204, 8, 19}, {6 * 256 * 4, 70, 1, 34}, {8 * 256, 54, 1, 23}, {6 * 256, 55, 1, 24}, {7 * 256 * 4, 55, 1, 24},
                                {0x4000, 70, 1, 34}, {0x4000, 55, 1, 24}, {0x20000, 55, 1, 24}, {0x20000, 55, 1, 24}, {8 * 7 * 2 * 2, 6, 0, 4}, {1, 6, 0, 4}};   // :3325-3336
  for (int k = 0; k < 12; k++) {
    const size_t n = (size_t)(k < FX_NMIX1 ? FX_TX : 16) * (size_t)mx[k][0];
    h.wx[k] = (int16_t*)P.zalloc(n * 2);
    P.fill16(h.wx[k], n, 129);   // setTxWx :653
    h.mx_M[k] = mx[k][0]; h.mx_shift[k] = mx[k][1]; h.mx_elim[k] = mx[k][2]; h.mx_uperr[k] = mx[k][3];
    h.mx_pr[k] = 2048;
  }
  static const int apm_n[6] = {256, 0x10000, 0x10000, 0x40000, 0x40000, 0x40000};   // :3287-3292
  uint16_t pat[33];
  for (int j = 0; j < 33; ++j) pat[j] = (uint16_t)(squash((j - 16) * 128) * 16);
  for (int j = 0; j < 6; j++) {
    const size_t n = (size_t)apm_n[j] * 33;
    h.apm_t[j] = (uint16_t*)P.zalloc(n * 2 + 4);
    P.pattern16(h.apm_t[j], n, pat, 33);
  }
  h.pr = 2048;
  h.slot_parallel = 1;
  // role M's wavefronts: consecutive whole maps, as few slots per wavefront as FX_M_WAVES groups allow
  for (int cap = 1;; cap++) {
    int w = 0, cnt = 0;
    h.mw_slot[0] = 0; h.mw_map[0] = 0;
    bool fits = true;
    for (int k = 0; k < FX_NMAPS && fits; k++) {
      if (cnt && cnt + h.maps[k].C > cap) { ++w; if (w >= FX_M_WAVES) { fits = false; break; } h.mw_slot[w] = (uint8_t)h.maps[k].slot_base; h.mw_map[w] = (uint8_t)k; cnt = 0; }
      cnt += h.maps[k].C;
    }
    if (!fits) continue;
    for (int v = w + 1; v <= FX_M_WAVES; v++) { h.mw_slot[v] = FX_NSLOTS; h.mw_map[v] = FX_NMAPS; }
    break;
  }
  for (int i = 0; i < FX_OUTPUTS; i++) h.pending[i] = 0.5f;   // model_predictions(0.5f, num_models) :94
  h.rec.AH2 = 0x765BA55C;                                      // :3262
}
}  // namespace fxb
#endif
