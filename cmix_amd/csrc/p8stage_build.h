// p8stage_build.h -- host-side construction of the paq8 stage's device state from the front end's P8Layout (p8_rec.h):
// the ContextMap family and the three ContextMap2 instances at the places contextModel2's walk gives them in the
// 1552-vector (reference src/models/paq8.cpp:8101-8207), the small lanes (constructors :893-898, :937-942, :978-983,
// :651-658, :3846-3850), the DMC forest, the mixer (Mixer m(NUM_INPUTS, 77472, NUM_SETS, 32) :8109) and the APM chains
// (Predictor :8216-8240). Memory comes from a policy object (hipMalloc in p8stage.hip, calloc in tests/host/p8stage_emul.cpp).
#ifndef CMX_P8STAGE_BUILD_H
#define CMX_P8STAGE_BUILD_H
#include <cstring>
#include <vector>

#include "p8cm_build.h"
#include "p8dmc_build.h"
#include "p8fam_dev.h"
#include "p8stage_dev.h"

// ---- a model's ContextMap family that contains generic instances (the audio models call recordModel, :5861) ----------------------------------
// The model's family runs the first design's per-context body (p8cm_dev.h) on a P8CmDev whose later instances ARE the generic family's: the
// same tables; the per-context registers and StateMaps are handed over at every switch between the generic family (second design: P8FamHome
// + the cached slot / run bytes, which equal the table between kernels) and the model's.
struct P8ViewMap { int n, gen_first[P8_XL_MAXG], view_first[P8_XL_MAXG], count[P8_XL_MAXG]; };
P8_HD void p8v_slot_in(const P8FamHome* gh, const uint16_t* gsm, P8CmRegs* vr, uint16_t* vsm, int g, int v) {
  vr->cp0[v] = gh->cp0[g]; vr->runp[v] = gh->runp[g];
  vr->cp[v] = gh->cpo[g] == P8F_NIL ? P8_NIL : gh->cp0[g] + gh->cpo[g];
  vr->sm_cxt[v] = gh->smc[g];
  for (int i = 0; i < 256; i++) vsm[(size_t)v * 256 + i] = gsm[(size_t)g * 256 + i];
}
P8_HD void p8v_slot_out(P8FamHome* gh, uint16_t* gsm, const P8CmRegs* vr, const uint16_t* vsm, const uint8_t* T, int g, int v) {
  gh->cp0[g] = vr->cp0[v]; gh->runp[g] = vr->runp[v];
  gh->cpo[g] = vr->cp[v] == P8_NIL ? (uint8_t)P8F_NIL : (uint8_t)(vr->cp[v] - vr->cp0[v]);
  gh->smc[g] = (uint8_t)vr->sm_cxt[v];
  gh->rc[g] = T[vr->runp[v]]; gh->rb[g] = T[vr->runp[v] + 1];
  for (int k = 0; k < 7; k++) gh->slot[g][k] = T[vr->cp0[v] + k];
  for (int i = 0; i < 256; i++) gsm[(size_t)g * 256 + i] = vsm[(size_t)v * 256 + i];
}

struct P8MixDev {
  int16_t* wx;             // [P8_NROWS][P8_NX]
  int16_t* wx2;            // [32]: the second layer's one row
  const int16_t* squash; const int16_t* stretch;
  int nx_first;
  int16_t first_map[P8_NX];
};
// host-side description of one stream's state; every pointer inside the members is policy memory
struct P8StageState {
  P8CmDev fam;
  P8FamHome* fam_home;     // policy memory: the family's per-context registers between chunks (p8fam_dev.h)
  P8Cm2Dev cm2[P8_NCM2];
  P8LanesDev lanes;
  P8DmcDev dmc;
  P8TailDev tail;
  P8MixDev mix;
  // the image models (p8_rec.h P8XLayout): one ContextMap each -- a one-instance family of the first design (p8cm_dev.h: the reference's
  // loop body per context, the shared rnd() stream handed over at every switch between the generic family and this one) -- and
  // their own lane tables
  P8CmDev xfam[P8_NMODEL - 1];
  P8XLanesDev xlanes[P8_NMODEL - 1];
  P8ViewMap xview[P8_NMODEL - 1];   // the generic instances inside a model's family
};

namespace p8b {
inline uint32_t sm32_prior(const uint8_t* nex1024, int i) {   // StateMap32(256) :651-656
  uint32_t n0 = nex1024[4 * i + 2], n1 = nex1024[4 * i + 3];
  if (n0 == 0) n1 *= 64;
  if (n1 == 0) n0 *= 64;
  return ((n1 << 16) / (n0 + n1 + 1)) << 16;
}
template <class Policy>
bool build_stage(P8StageState& S, Policy& P, const P8Layout& L, int level, const uint8_t* nex1024, const int16_t* stretch4096, const int16_t* squash4096,
                 const uint8_t* ilog65536) {
  auto up = [&](const void* src, size_t bytes) { void* p = P.zalloc(bytes); P.upload(p, src, bytes); return p; };
  // ---- ContextMap family ----
  if (!build_family(S.fam, P, L.fam_ninst, L.fam_size, L.fam_count, nex1024, stretch4096, ilog65536)) return false;
  S.fam.row_stride = P8_NX; S.fam.order_slot = L.order_slot;
  for (int s = 0; s < L.fam_slots; s++) S.fam.slot_off[s] = L.fam_off[s];
  for (int o = 0; o <= P8_ORDER_MAX; o++) { S.fam.order_ctx[o] = L.order_ctx[o]; S.fam.order_chk[o] = L.order_chk[o]; }
  {   // the second-design kernel's home state: ContextMap's constructor (:1049-1062) in cached form
    P8FamHome* hh = new P8FamHome();
    memset(hh, 0, sizeof *hh);
    for (int s = 0; s < S.fam.nslots; s++) { hh->cp0[s] = P8_B_STATE; hh->cpo[s] = 0; hh->runp[s] = P8_B_STATE + 3; }
    S.fam_home = (P8FamHome*)up(hh, sizeof *hh);
    delete hh;
  }
  // ---- ContextMap2 x 3 ----
  for (int k = 0; k < P8_NCM2; k++) {
    if (!build(S.cm2[k], P, L.cm2_size[k], L.cm2_count[k], nex1024, stretch4096, ilog65536)) return false;
    S.cm2[k].row_stride = P8_NX; S.cm2[k].out_off = L.cm2_off[k];
  }
  // ---- small lanes ----
  P8LanesDev& D = S.lanes;
  memset(&D, 0, sizeof D);
  D.nlanes = L.nlanes;
  D.nex = (const uint8_t*)up(nex1024, 1024);
  D.stretch = (const int16_t*)up(stretch4096, 4096 * 2);
  D.ilog = S.fam.ilog;
  uint32_t prior[256];
  for (int i = 0; i < 256; i++) prior[i] = sm32_prior(nex1024, i);
  for (int l = 0; l < L.nlanes; l++) {
    P8LaneDev& q = D.lane[l];
    q.q = L.lane[l];
    const size_t n = q.q.cells;
    switch (q.q.kind) {
      case P8L_SSCM: {
        std::vector<uint16_t> v(n, (uint16_t)q.q.init);
        q.c16 = (uint16_t*)up(v.data(), n * 2);
        break;
      }
      case P8L_STAT: {
        std::vector<uint32_t> v(n, q.q.init);
        q.c32 = (uint32_t*)up(v.data(), n * 4);
        break;
      }
      case P8L_IND:
        q.c8 = (uint8_t*)P.zalloc(n);
        q.sm = (uint32_t*)up(prior, sizeof prior);
        break;
      case P8L_SM32: {
        std::vector<uint32_t> v(n, q.q.init);
        if (n == 256) for (int i = 0; i < 256; i++) v[i] = prior[i];
        q.c32 = (uint32_t*)up(v.data(), n * 4);
        break;
      }
      case P8L_RCM:   // RunContextMap(m): BH<4> t(m / 4); cp = t[0] + 1 :862
        q.c8 = (uint8_t*)P.zalloc(n + 64);
        q.mask = (uint32_t)(n / 4 - 1);
        D.regs[l].cp = 2;
        break;
      case P8L_PIC: {
        q.c8 = (uint8_t*)P.zalloc(n);
        uint16_t sm[256];
        for (int i = 0; i < 256; i++) {   // StateMap :626-635
          int n0 = nex1024[4 * i + 2], n1 = nex1024[4 * i + 3];
          if (n0 == 0) n1 *= 64;
          if (n1 == 0) n0 *= 64;
          sm[i] = (uint16_t)(65536 * (n1 + 1) / (n0 + n1 + 2));
        }
        q.sm16 = (uint16_t*)up(sm, sizeof sm);
        break;
      }
      default: break;
    }
    if (q.q.kind == P8L_SSCM || q.q.kind == P8L_STAT || q.q.kind == P8L_IND) {
      q.stride = (1u << q.q.bits_per_ctx) - 1;
      q.mask = q.stride ? (uint32_t)(n / q.stride) - 1 : 0;
    }
  }
  // ---- the image models ----
  for (int m = 0; m < P8_NMODEL - 1; m++) {
    const P8XLayout& X = L.xl[m];
    memset(&S.xfam[m], 0, sizeof S.xfam[m]);
    memset(&S.xlanes[m], 0, sizeof S.xlanes[m]);
    if (X.nx == 0) continue;   // (a model the front end has not got)
    {   // the model's family: its own ContextMap (if it has one), then the generic instances it calls -- the generic family's own tables
      P8CmDev& h = S.xfam[m];
      P8ViewMap& V = S.xview[m];
      memset(&V, 0, sizeof V);
      h.slot_parallel = 1;
      int s = 0, k = 0;
      auto add = [&](uint8_t* table, uint32_t mask, int count) {
        h.inst[k].table = table; h.inst[k].mask = mask; h.inst[k].first = s; h.inst[k].count = count;
        for (int i = 0; i < count; i++, s++) { h.slot_inst[s] = (uint8_t)k; h.regs.cp0[s] = h.regs.cp[s] = P8_B_STATE; h.regs.runp[s] = P8_B_STATE + 3; }
        k++;
      };
      if (X.fam_count > 0) {
        const uint64_t sz = X.fam_size;
        if (sz < 4096 || (sz & (sz - 1)) || (sz >> 6) > 0x4000000ull) return false;
        add((uint8_t*)P.zalloc((size_t)sz), (uint32_t)((sz >> 6) - 1), X.fam_count);
      }
      for (int g = 0; g < X.ngen; g++) {
        const P8CmInst& gi = S.fam.inst[X.gen_inst[g]];
        V.gen_first[g] = gi.first; V.view_first[g] = s; V.count[g] = gi.count; V.n = g + 1;
        add(gi.table, gi.mask, gi.count);
      }
      if (s > P8_XL_MAXS - 2 || k > P8CM_MAXI) return false;   // (s == 0: a model without ContextMaps, im1bitModel)
      h.ninst = k; h.nslots = s;
      h.row_stride = P8_NX; h.order_slot = -1;
      for (int i = 0; i < s; i++) h.slot_off[i] = X.fam_off[i];
      h.nex = S.fam.nex; h.stretch = S.fam.stretch; h.ilog = S.fam.ilog;
      std::vector<uint16_t> sm((size_t)s * 256);
      for (size_t i = 0; i < sm.size(); ++i) {   // StateMap :626-635
        int n0 = nex1024[4 * (i & 255) + 2], n1 = nex1024[4 * (i & 255) + 3];
        if (n0 == 0) n1 *= 64;
        if (n1 == 0) n0 *= 64;
        sm[i] = (uint16_t)(65536 * (n1 + 1) / (n0 + n1 + 2));
      }
      h.sm = (uint16_t*)up(sm.data(), sm.size() * 2);
      h.rnd = S.fam.rnd;
    }
    P8XLanesDev& XD = S.xlanes[m];
    XD.nlanes = X.nlanes; XD.model = m + 1;
    XD.nex = D.nex; XD.stretch = D.stretch;
    XD.squash = (const int16_t*)up(squash4096, 4096 * 2);
    for (int l = 0; l < X.nlanes; l++) {
      P8LaneDev& q = XD.lane[l];
      q.q = X.lane[l];
      const size_t n = q.q.cells;
      if (q.q.kind == P8L_SSCM) { std::vector<uint16_t> v(n, (uint16_t)q.q.init); q.c16 = (uint16_t*)up(v.data(), n * 2); }
      else if (q.q.kind == P8L_STAT) { std::vector<uint32_t> v(n, q.q.init); q.c32 = (uint32_t*)up(v.data(), n * 4); }
      else if (q.q.kind == P8L_PIC || q.q.kind == P8L_PIC2) {
        q.c8 = (uint8_t*)P.zalloc(n);
        uint16_t sm[512];
        for (int i = 0; i < 512; i++) {   // StateMap :626-635 (two of them for a pair)
          int n0 = nex1024[4 * (i & 255) + 2], n1 = nex1024[4 * (i & 255) + 3];
          if (n0 == 0) n1 *= 64;
          if (n1 == 0) n0 *= 64;
          sm[i] = (uint16_t)(65536 * (n1 + 1) / (n0 + n1 + 2));
        }
        q.sm16 = (uint16_t*)up(sm, sizeof sm);
        continue;
      }
      else if (q.q.kind == P8L_NONE) continue;
      else if (q.q.kind == P8L_SM32) { std::vector<uint32_t> v(n, q.q.init); q.c32 = (uint32_t*)up(v.data(), n * 4); continue; }
      else if (q.q.kind == P8L_JPG) {   // jpegModel's tables :6484-6490 (BH<9> t(MEM()), StateMap sm[32], Mixer m1(33, 2050, 3), APM a1(0x8000), a2(0x20000))
        P8JpgDev J;
        memset(&J, 0, sizeof J);
        J.t = (uint8_t*)P.zalloc((size_t)n * 9 + 64); J.mask = (uint32_t)(n - 1);
        std::vector<uint16_t> sm(32 * 256);
        for (size_t i = 0; i < sm.size(); i++) {   // StateMap :626-635
          int n0 = nex1024[4 * (i & 255) + 2], n1 = nex1024[4 * (i & 255) + 3];
          if (n0 == 0) n1 *= 64;
          if (n1 == 0) n0 *= 64;
          sm[i] = (uint16_t)(65536 * (n1 + 1) / (n0 + n1 + 2));
        }
        J.sm = (uint16_t*)up(sm.data(), sm.size() * 2);
        J.w1 = (int16_t*)P.zalloc((size_t)2050 * 40 * 2);
        for (int i = 0; i < 8; i++) J.w2[i] = 0x7fff;
        std::vector<uint32_t> av((size_t)0x20000 * 24);
        for (size_t i = 0; i < av.size(); i++) { const int pp = (((int)(i % 24) * 2 + 1) * 4096) / 48 - 2048; av[i] = ((uint32_t)(pp > 2047 ? 4095 : pp < -2047 ? 0 : (int)squash4096[pp + 2048]) << 20) + 6; }   // APM :693-698
        J.a1 = (uint32_t*)up(av.data(), (size_t)0x8000 * 24 * 4);
        J.a2 = (uint32_t*)up(av.data(), av.size() * 4);
        q.jpg = (P8JpgDev*)up(&J, sizeof J);
        continue;
      }
      else if (q.q.kind == P8L_HT16) {
        q.c8 = (uint8_t*)P.zalloc(n);          // the HashTable<16>
        q.c32 = (uint32_t*)P.zalloc(64 * 4);   // the 14 contexts' pointers and StateMap contexts
        std::vector<uint16_t> sm(14 * 256);
        for (size_t i = 0; i < sm.size(); i++) {   // StateMap :626-635
          int n0 = nex1024[4 * (i & 255) + 2], n1 = nex1024[4 * (i & 255) + 3];
          if (n0 == 0) n1 *= 64;
          if (n1 == 0) n0 *= 64;
          sm[i] = (uint16_t)(65536 * (n1 + 1) / (n0 + n1 + 2));
        }
        q.sm16 = (uint16_t*)up(sm.data(), sm.size() * 2);
        continue;
      }
      else return false;   // the models hold no other kind
      q.stride = (1u << q.q.bits_per_ctx) - 1;
      q.mask = q.stride ? (uint32_t)(n / q.stride) - 1 : 0;
    }
  }
  // ---- DMC forest ----
  build_dmc(S.dmc, P, level, nex1024, stretch4096);
  // ---- mixer ----
  P8MixDev& M = S.mix;
  memset(&M, 0, sizeof M);
  {
    const size_t n = (size_t)P8_NROWS * P8_NX;
    M.wx = (int16_t*)P.zalloc(n * 2);
    P.fill16(M.wx, 32, n);                 // rows are created lazily with init_w = 32 (:530-537)
    int16_t w2[32];
    for (int i = 0; i < 32; i++) w2[i] = 0x7fff;   // :589
    M.wx2 = (int16_t*)up(w2, sizeof w2);
    M.squash = (const int16_t*)up(squash4096, 4096 * 2);
    M.stretch = D.stretch;
    M.nx_first = L.nx_first;
    for (int i = 0; i < P8_NX; i++) M.first_map[i] = L.first_map[i];
  }
  // ---- APM chains ----
  P8TailDev& T = S.tail;
  memset(&T, 0, sizeof T);
  T.stretch = D.stretch; T.squash = M.squash;
  T.pr = 2048;
  for (int i = 0; i < P8_NOUT; i++) T.out[i] = 0.5f;   // model_predictions(0.5, ...) :500
  auto sq = [&](int d) { return d > 2047 ? 4095 : d < -2047 ? 0 : (int)squash4096[d + 2048]; };
  {
    std::vector<uint32_t> v((size_t)0x10000 * 24);
    for (size_t i = 0; i < v.size(); i++) {   // APM :693-698
      const int p = (((int)(i % 24) * 2 + 1) * 4096) / 48 - 2048;
      v[i] = ((uint32_t)sq(p) << 20) + 6;
    }
    for (int k = 0; k < 4; k++) T.apm[k] = (uint32_t*)up(v.data(), v.size() * 4);
  }
  {
    std::vector<uint16_t> v((size_t)0x10000 * 33);
    for (size_t i = 0; i < v.size(); i++) v[i] = (uint16_t)(sq(((int)(i % 33) - 16) * 128) * 16);   // APM1 :603-607
    for (int k = 0; k < 3; k++) T.apm1[k] = (uint16_t*)up(v.data(), v.size() * 2);
    T.gen[0] = (uint16_t*)up(v.data(), (size_t)0x2000 * 33 * 2);
    for (int k = 1; k < 7; k++) T.gen[k] = (uint16_t*)up(v.data(), v.size() * 2);
    for (int k = 0; k < 2; k++) T.col_apm1[k] = (uint16_t*)up(v.data(), v.size() * 2);   // Image.Color's APM1s :8224
    for (int k = 0; k < 2; k++) T.pal_apm1[k] = (uint16_t*)up(v.data(), v.size() * 2);   // Image.Palette's
  }
  {
    std::vector<uint32_t> v((size_t)0x10000 * 24);
    for (size_t i = 0; i < v.size(); i++) { const int p = (((int)(i % 24) * 2 + 1) * 4096) / 48 - 2048; v[i] = ((uint32_t)sq(p) << 20) + 6; }
    T.col_apm[0] = (uint32_t*)up(v.data(), (size_t)0x1000 * 24 * 4);                      // Image.Color's APMs {0x1000}, 3 x {0x10000} :8223
    for (int k = 1; k < 4; k++) T.col_apm[k] = (uint32_t*)up(v.data(), v.size() * 4);
    T.pal_apm[0] = (uint32_t*)up(v.data(), (size_t)0x1000 * 24 * 4);
    for (int k = 1; k < 4; k++) T.pal_apm[k] = (uint32_t*)up(v.data(), v.size() * 4);
    T.gray_apm[0] = (uint32_t*)up(v.data(), (size_t)0x1000 * 24 * 4);
    for (int k = 1; k < 3; k++) T.gray_apm[k] = (uint32_t*)up(v.data(), v.size() * 4);
  }
  return true;
}
}  // namespace p8b
#endif
