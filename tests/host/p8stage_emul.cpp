// tests/host/p8stage_emul.cpp -- TEST INFRASTRUCTURE ONLY. The whole paq8 stage on the host: the PRODUCT's front end
// (cmix_amd/csrc/p8front/*.c, linked in) and the PRODUCT's device bodies (p8cm_dev.h, p8cm2_dev.h, p8dmc_dev.h,
// p8stage_dev.h: the same step functions the kernels of p8stage.hip call), lanes looped per barrier step, state built
// by the same p8stage_build.h with a calloc policy. Only the mixer's dot products / training are plain loops here (on
// the device they are wave-parallel code, p8stage.hip). tests/test_p8stage_host.py compares its 1591 values per step
// with columns 434..2024 of traces of the unmodified reference. Nothing in cmix_amd/ loads it.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../cmix_amd/csrc/p8front/p8f_front.h"
#include "../../cmix_amd/csrc/p8cm2v2_dev.h"
#include "../../cmix_amd/csrc/p8stage_build.h"

namespace {
struct HostPolicy {
  std::vector<void*> blocks;
  void* zalloc(size_t bytes) { void* p = calloc(bytes + 64, 1); blocks.push_back(p); return p; }
  void upload(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
  void fill16(int16_t* dst, int16_t v, size_t n) { for (size_t i = 0; i < n; i++) dst[i] = v; }
};
struct Emul {
  P8Front* front = nullptr;
  P8StageState S;
  HostPolicy pol;
  P8CmShared fsh;
  std::vector<unsigned char> f2mem;   // P8FamShared + StateMaps (second-design family body)
  P8FamShared* f2 = nullptr;
  int f2_lk = 0; uint32_t f2_prev_i = 0, f2_i = 0;
  int use_v1 = 0;
  P8Cm2Shared csh[P8_NCM2];
  P8Cm2V2Shared* c2[P8_NCM2] = {};   // second-design ContextMap2 body
  int c2_lk[P8_NCM2] = {};
  P8DmcShared dsh;
  uint64_t steps = 0;
  int last_bit = 0;
  // the image models: their one ContextMap each runs the first design's per-context body (p8cm_dev.h) in the reference's order; the
  // process-wide rnd() stream changes hands whenever a byte's model differs from the previous byte's (as between two kernels)
  P8CmShared xsh[P8_NMODEL - 1];
  int fam_owner = 0;      // whose copy of the generator state is current: 0 the generic family, m an image model
  int last_byte = 0;      // the last whole byte of the stream (ContextMap's c1)
  uint64_t fam_serial = 0, cm2_serial = 0;
  int late = 0;           // 1: the DECODER's order of operations (cmx_late.h): the front end emits a step's records only after the bit before it has been
  int late_models = 0;    // with late: model steps (images, audio, JPEG) in the decoder's order too
                          // handed in, and the maps' uniform registers take that bit at the top of the step (p8d_bit_y, p8f_uni_tail + p8f_uni_head)
  int fam_miniwalk = 1;   // 0: whole-instance walks only; 1: the kernel's narrowed walk; 2: with every second visit treated as unlisted (the fall-back path)
  uint64_t fam_mini = 0, fam_mini_full = 0;
  // diagnostics: per family instance, lookup bits with an overlap / with an overlap AND a pending rnd() draw in the instance
  uint64_t inst_conf[P8CM_MAXI] = {}, inst_risky[P8CM_MAXI] = {}, lookups = 0, bits_with_conf = 0, bits_with_risky = 0, draws_total = 0, multi_conf = 0;
};
int16_t sat16(int v) { return (int16_t)(v > 32767 ? 32767 : v < -32768 ? -32768 : v); }
int dot(const int16_t* t, const int16_t* w, int n) {
  uint32_t sum = 0;
  for (int i = 0; i + 1 < n; i += 2) {
    const uint32_t pair = (uint32_t)((int32_t)t[i] * w[i]) + (uint32_t)((int32_t)t[i + 1] * w[i + 1]);
    sum += (uint32_t)((int32_t)pair >> 8);
  }
  return (int32_t)sum;
}
void train(const int16_t* t, int16_t* w, int n, int e) {
  if (!e) return;
  const int16_t err = (int16_t)e;
  for (int i = 0; i < n; ++i) {
    int v = sat16(2 * (int)t[i]);
    v = (v * (int)err) >> 16;
    v = sat16(v + 1) >> 1;
    w[i] = sat16(v + (int)w[i]);
  }
}
}  // namespace

extern "C" {
void* p8s_create(int level) {
  Emul* e = new Emul();
  e->front = p8f_front_new(level);
  if (!e->front) { delete e; return nullptr; }
  if (!p8b::build_stage(e->S, e->pol, *p8f_front_layout(e->front), level, p8f_state_table(), p8f_stretch_table(), p8f_squash_table(), p8f_ilog_table())) {
    p8f_front_free(e->front); delete e; return nullptr;
  }
  e->fsh.r = e->S.fam.regs; e->fsh.rnd = e->S.fam.rnd;
  for (int m = 0; m < P8_NMODEL - 1; m++) { e->xsh[m].r = e->S.xfam[m].regs; e->xsh[m].rnd = e->S.xfam[m].rnd; }
  e->use_v1 = getenv("CMX_P8FAM_V1") != nullptr;
  for (int k = 0; k < P8_NCM2; k++) e->c2[k] = new P8Cm2V2Shared();
  e->f2mem.resize(sizeof(P8FamShared) + (size_t)e->S.fam.nslots * 512 + 64);
  e->f2 = (P8FamShared*)e->f2mem.data();
  for (int k = 0; k < P8_NCM2; k++) e->csh[k].r = e->S.cm2[k].regs;
  return e;
}
void p8s_destroy(void* h) {
  Emul* e = (Emul*)h;
  if (!e) return;
  p8f_front_free(e->front);
  for (void* p : e->pol.blocks) free(p);
  delete e;
}
void p8s_conflict_report(void* h) {
  Emul* e = (Emul*)h;
  printf("lookup bits %llu: with an overlap %llu (two or more instances %llu), with overlap + pending draw %llu; draws %llu\n", (unsigned long long)e->lookups,
         (unsigned long long)e->bits_with_conf, (unsigned long long)e->multi_conf, (unsigned long long)e->bits_with_risky, (unsigned long long)e->draws_total);
  for (int k = 0; k < e->S.fam.ninst; k++)
    printf("  inst %2d (%3d ctx, %8u buckets): overlap %llu, + draw %llu\n", k, e->S.fam.inst[k].count, e->S.fam.inst[k].mask + 1, (unsigned long long)e->inst_conf[k],
           (unsigned long long)e->inst_risky[k]);
}
// diagnostics: per family context cp, cp0, runp (byte offsets; cp = 0xFFFFFFFF: none), state byte at cp, StateMap context
void p8s_dump_fam(void* h, uint32_t* out /* [nslots][5] */) {
  Emul* e = (Emul*)h;
  const P8CmDev* d = &e->S.fam;
  for (int s = 0; s < d->nslots; s++) {
    const uint8_t* T = d->inst[d->slot_inst[s]].table;
    uint32_t cp, cp0, runp, smc;
    if (e->use_v1) { cp = e->fsh.r.cp[s]; cp0 = e->fsh.r.cp0[s]; runp = e->fsh.r.runp[s]; smc = (uint32_t)e->fsh.r.sm_cxt[s]; }
    else { const P8FamHome* r = &e->f2->r; cp0 = r->cp0[s]; cp = r->cpo[s] == P8F_NIL ? 0xFFFFFFFFu : cp0 + r->cpo[s]; runp = r->runp[s]; smc = r->smc[s]; }
    out[5 * s] = cp; out[5 * s + 1] = cp0; out[5 * s + 2] = runp; out[5 * s + 3] = cp == 0xFFFFFFFFu ? 0 : T[cp]; out[5 * s + 4] = smc;
  }
}
void p8s_dump_table(void* h, int inst, uint8_t* out) { Emul* e = (Emul*)h; memcpy(out, e->S.fam.inst[inst].table, ((size_t)e->S.fam.inst[inst].mask + 1) * 64); }
// diagnostics: where the tables' inputs sit
void p8s_layout_dump(void* h) {
  const P8Layout& L = *p8f_front_layout(((Emul*)h)->front);
  int s = 0;
  for (int k = 0; k < L.fam_ninst; k++) { printf("family inst %d: %d contexts, inputs %d..%d\n", k, L.fam_count[k], L.fam_off[s], L.fam_off[s + L.fam_count[k] - 1] + 4); s += L.fam_count[k]; }
  for (int k = 0; k < P8_NCM2; k++) printf("cm2 %d: %d contexts at %d\n", k, L.cm2_count[k], L.cm2_off[k]);
  for (int l = 0; l < L.nlanes; l++) printf("lane %d kind %d off %d nout %d\n", l, L.lane[l].kind, L.lane[l].off, L.lane[l].nout);
  for (int m = 0; m < P8_NMODEL - 1; m++) printf("image model %d: prefix %d nx %d lanes %d contexts %d (first at %d)\n", m + 1, L.xl[m].prefix_nx, L.xl[m].nx, L.xl[m].nlanes, L.xl[m].fam_count, L.xl[m].fam_off[0]);
}
void p8s_set_miniwalk(void* h, int mode) { ((Emul*)h)->fam_miniwalk = mode; }
void p8s_set_late(void* h, int on) { ((Emul*)h)->late = on & 1; ((Emul*)h)->late_models = (on >> 1) & 1; }
void p8s_miniwalk_stats(void* h, uint64_t* out2) { out2[0] = ((Emul*)h)->fam_mini; out2[1] = ((Emul*)h)->fam_mini_full; }
uint32_t p8s_rnd_i(void* h) { return ((Emul*)h)->f2_i; }   // how many values of the shared generator the family has drawn (mod 2^32)
void p8s_set_rnd_i(void* h, uint32_t i) { Emul* e = (Emul*)h; e->S.fam.rnd.i = (int)i; e->f2_i = e->f2_prev_i = i; }   // test hook: place the generator's counter (its 64 values stay) -- to reach the counter's sign change without 8 MB of input
void p8s_set_pos(void* h, int pos) { p8f_front_set_pos(((Emul*)h)->front, pos); }   // test hook: the front end's byte position (before the first byte)
void p8s_stats(void* h, uint64_t* out3) { Emul* e = (Emul*)h; out3[0] = e->steps; out3[1] = e->fam_serial; out3[2] = e->cm2_serial; }
// nbytes more bytes of the stream; out [8 nbytes][1591] f32 = PAQ8::Predict() before each of their bits. 0 or a negative front-end code.
int p8s_run(void* h, const uint8_t* bytes, int nbytes, float* out) {
  Emul* e = (Emul*)h;
  const P8Layout& L = *p8f_front_layout(e->front);
  const size_t n = (size_t)nbytes, T = 8 * n;
  std::vector<uint32_t> fctx(n * L.fam_slots), c2ctx[P8_NCM2], ops(T * P8_NLANE);
  std::vector<uint16_t> fchk(n * L.fam_slots), c2chk[P8_NCM2];
  std::vector<int32_t> sel(T * P8_NSEL);
  std::vector<P8ApmRec> apm(T);
  P8Chunk c = P8Chunk();
  c.fam_ctx = fctx.data(); c.fam_chk = fchk.data();
  for (int k = 0; k < P8_NCM2; k++) { c2ctx[k].resize(n * L.cm2_count[k]); c2chk[k].resize(n * L.cm2_count[k]); c.cm2_ctx[k] = c2ctx[k].data(); c.cm2_chk[k] = c2chk[k].data(); }
  c.ops = ops.data(); c.sel = sel.data(); c.apm = apm.data();
  std::vector<uint8_t> model(n, 0);
  std::vector<uint32_t> xops, xfctx;
  std::vector<uint16_t> xfchk;
  c.model = model.data();
  if (!e->late || e->late_models) {   // (the DEVICE's decoder form of the image models is not built: there the front end refuses such a byte -- it has nowhere to put its
    // records; late_models: the emulation of what that form has to do, step by step)
    xops.assign(T * P8_XL_NLANE, 0); xfctx.assign(n * P8_XL_MAXS, 0); xfchk.assign(n * P8_XL_MAXS, 0);
    c.xops = xops.data(); c.xfam_ctx = xfctx.data(); c.xfam_chk = xfchk.data();
  }
  std::vector<uint8_t> bits(T), order(T, 0);
  for (size_t i = 0; i < T; i++) bits[i] = (bytes[i >> 3] >> (7 - (i & 7))) & 1;
  if (!e->late) {
    const int rc = p8f_front_run(e->front, bytes, n, &c);
    if (rc) return rc;
  }
  std::vector<int16_t> x(T * P8_NX, 0);
  P8StageState& S = e->S;
  // per-family uniform registers, carried between calls
  int f_last_y = S.fam.last_y, f_c1 = S.fam.c1;
  P8FamRun f_run; f_run.last_y = S.fam.last_y; f_run.c1 = S.fam.c1; f_run.lk = 0; f_run.c0 = 1; f_run.bits8 = 0; f_run.order = 0; f_run.nslots = S.fam.nslots; f_run.row_stride = S.fam.row_stride;   // as cmx_p8s_fam2_kernel starts a chunk
  uint32_t run_bits[P8_NCM2]; int c_last_y[P8_NCM2];
  for (int k = 0; k < P8_NCM2; k++) { run_bits[k] = S.cm2[k].bits; c_last_y[k] = S.cm2[k].last_y; }
  if (!e->use_v1)
    for (int k = 0; k < P8_NCM2; k++) {
      for (int tid = 0; tid < P8CM2_MAXC; tid++) p8c2_load(&S.cm2[k], e->c2[k], tid, P8CM2_MAXC);
      for (int i = 0; i < S.cm2[k].C; i++) p8c2_reload(&S.cm2[k], e->c2[k], i);
      e->c2_lk[k] = 0;
    }
  if (!e->use_v1) {
    for (int tid = 0; tid < 256; tid++) p8f_load(&S.fam, S.fam_home, S.fam.sm, e->f2, tid, 256);
    e->f2_lk = 0; e->f2_i = e->f2_prev_i = (uint32_t)S.fam.rnd.i;
  }
  for (size_t t = 0; t < T; t++) {
    const uint64_t g = e->steps + t;
    const int y = t ? bits[t - 1] : e->last_bit;
    if (e->late) {   // the records of step t exist only now: every bit before the step has been decoded and handed in
      if (t) p8f_front_set_bit(e->front, bits[t - 1]);
      const int rc = p8f_front_emit_step(e->front, &c, t);
      if (rc) return rc;
    }
    int16_t* xr = x.data() + t * P8_NX;
    float* orow = out + t * P8_NOUT;
    const int md = c.model[t >> 3];                      // the byte's model (its record exists from the byte's first step on)
    if (md != e->fam_owner) {   // the generator -- and the per-context state of the generic instances a model's family contains -- changes hands: what the kernels between two segments do on the device
      auto view_out = [&](int m) {   // model m's family -> the generic family's home
        const P8ViewMap& V = S.xview[m - 1];
        P8CmShared* xs = &e->xsh[m - 1];
        for (int g = 0; g < V.n; g++) for (int i = 0; i < V.count[g]; i++)
          p8v_slot_out(S.fam_home, S.fam.sm, &xs->r, S.xfam[m - 1].sm, S.fam.inst[S.fam.slot_inst[V.gen_first[g]]].table, V.gen_first[g] + i, V.view_first[g] + i);
      };
      auto view_in = [&](int m) {
        const P8ViewMap& V = S.xview[m - 1];
        P8CmShared* xs = &e->xsh[m - 1];
        for (int g = 0; g < V.n; g++) for (int i = 0; i < V.count[g]; i++) p8v_slot_in(S.fam_home, S.fam.sm, &xs->r, S.xfam[m - 1].sm, V.gen_first[g] + i, V.view_first[g] + i);
      };
      if (e->use_v1) { fprintf(stderr, "p8stage_emul: models with their own tables need the second-design family (unset CMX_P8FAM_V1)\n"); return -99; }
      if (e->fam_owner == 0) {
        for (int tid = 0; tid < 256; tid++) p8f_store(&S.fam, S.fam_home, S.fam.sm, e->f2, e->f2_i, tid, 256);
        e->xsh[md - 1].rnd = S.fam.rnd;
        view_in(md);
      } else {
        view_out(e->fam_owner);
        if (md == 0) {
          S.fam.rnd = e->xsh[e->fam_owner - 1].rnd;
          for (int tid = 0; tid < 256; tid++) p8f_load(&S.fam, S.fam_home, S.fam.sm, e->f2, tid, 256);
          e->f2_lk = 0; e->f2_i = e->f2_prev_i = (uint32_t)S.fam.rnd.i; f_run.lk = 0;
        } else { e->xsh[md - 1].rnd = e->xsh[e->fam_owner - 1].rnd; view_in(md); }
      }
      e->fam_owner = md;
    }
    P8Cm2Bit cu[P8_NCM2];
    const int md_pre = c.model[t >> 3];
    for (int k = 0; k < P8_NCM2; k++) {
      // TextModel's and exeModel's ContextMap2 keep their OWN partial byte (ContextMap2::mix :1305-1309): a step that does not call them
      // leaves it where it was -- only the bit that precedes their next call is the stream's
      if (k > 0 && md_pre) { cu[k] = P8Cm2Bit(); c_last_y[k] = bits[t]; continue; }
      if (e->late) { cu[k] = p8d_bit_y(&S.cm2[k], c.cm2_ctx[k], c.cm2_chk[k], t ? y : c_last_y[k], x.data(), (int)t, &run_bits[k]); if (t + 1 == T) c_last_y[k] = bits[t]; }
      else cu[k] = p8d_bit(&S.cm2[k], c.cm2_ctx[k], c.cm2_chk[k], bits.data(), x.data(), (int)t, &run_bits[k], &c_last_y[k]);
    }
    const P8CmBit fu_pre = P8CmBit();
    (void)fu_pre;
    if (g >= 8 && e->use_v1) {
      for (int k = 0; k < P8_NCM2; k++) {   // instance 0 first: the family needs its return value
        if (k > 0 && md) continue;          // TextModel's and exeModel's maps are not called in an image model's step (:8161-8166)
        P8Cm2Dev* d = &S.cm2[k];
        for (int i = d->C - 1; i >= 0; i--) p8d_touch(d, &e->csh[k], cu[k], i);
        for (int i = d->C - 1; i >= 0; i--) p8d_conflict(d, &e->csh[k], i);
        e->cm2_serial += e->csh[k].conflict != 0;
        for (int i = d->C - 1; i >= 0; i--) p8d_run(d, &e->csh[k], cu[k], i);
        if (k == 0) { int o = 0; for (int i = 0; i < d->C; i++) o += e->csh[0].nz[i]; order[t] = (uint8_t)o; }
      }
    } else if (g >= 8) {   // second design: the control flow of cmx_p8s_cm2v2_kernel
      for (int k = 0; k < P8_NCM2; k++) {
        if (k > 0 && md) continue;
        P8Cm2Dev* d = &S.cm2[k];
        P8Cm2V2Shared* sh = e->c2[k];
        const bool look = cu[k].bpos == 0 || cu[k].bpos == 2 || cu[k].bpos == 5;
        static P8Cm2Tmp tmp[P8CM2_MAXC];
        if (look) { ++e->c2_lk[k]; for (int i = d->C - 1; i >= 0; i--) p8c2_phase1(d, sh, cu[k], e->c2_lk[k], i, &tmp[i]); }
        const bool force = getenv("CMX_P8C2_FORCE_WALK") && (g % 5) == 2;   // test hook: exercise walk + reload
        if ((look && sh->conf[e->c2_lk[k] & 1]) || sh->shared || force) {
          e->cm2_serial++;
          p8c2_walk(d, sh, cu[k], look);
          for (int i = d->C - 1; i >= 0; i--) p8c2_reload(d, sh, i);
        } else for (int i = d->C - 1; i >= 0; i--) p8c2_run(d, sh, cu[k], i, &tmp[i]);
        if (k == 0) { int o = 0; for (int i = 0; i < d->C; i++) o += sh->base.nz[i]; order[t] = (uint8_t)o; }
        if (look) for (int tid = P8CM2_MAXC - 1; tid >= 0; tid--) p8c2_clear_next(sh, e->c2_lk[k], tid, P8CM2_MAXC);
      }
    }
    if (e->use_v1) {
    const P8CmBit fu = p8d_cm_bit(&S.fam, c.fam_ctx, c.fam_chk, bits.data(), x.data(), order.data(), (int)t, &f_last_y, &f_c1);
    if (g >= 8 && !md) {
      P8CmDev* d = &S.fam;
      for (int s = d->nslots - 1; s >= 0; s--) p8d_cm_touch(d, &e->fsh, fu, s);
      for (int s = d->nslots - 1; s >= 0; s--) p8d_cm_check(d, &e->fsh, s);
      for (int s = d->nslots - 1; s >= 0; s--) p8d_cm_draw(d, &e->fsh, s);
      e->fam_serial += e->fsh.conflict != 0;
      for (int s = d->nslots - 1; s >= 0; s--) p8d_cm_run(d, &e->fsh, fu, s);
    }
    } else {   // second design (p8fam_dev.h): the control flow of cmx_p8s_fam2_kernel, lanes looped per phase
      P8CmDev* d = &S.fam;
      P8FamShared* sh = e->f2;
      const int SS = d->nslots;
      if (e->late && t > 0) p8f_uni_tail(&f_run, (int)((t - 1) & 7), y);   // (t == 0: f_run.last_y is the carried bit, as the kernel takes it from the box)
      P8FamUni fu = P8FamUni();
      if (md) {   // an image model's step: the generic family only follows the bits (last bit, partial byte, last whole byte)
        if ((t & 7) == 0) f_run.c0 = 1;
        if (!e->late) p8f_uni_tail(&f_run, (int)(t & 7), bits[t]);   // (a decoder does not know bits[t] yet: the tail of this step is taken at the top of the next one, above)
      } else fu = e->late ? p8f_uni_head(c.fam_ctx, c.fam_chk, x.data(), order.data(), (int)t, &f_run, e->f2_i)
                          : p8f_uni_inc(d, c.fam_ctx, c.fam_chk, bits.data(), x.data(), order.data(), (int)t, &f_run, e->f2_i);
      if (g >= 8 && !md) {
        static P8FamTmp tmp[P8CM_MAXS];
        for (int s = SS - 1; s >= 0; s--) { p8f_lane(d, s, &tmp[s]); tmp[s].cx = p8f_ctx(d, fu, s); tmp[s].ck = p8f_chk(d, fu, s); p8f_phase1(d, sh, fu, s, &tmp[s]); }
        for (uint32_t k = 0, nn = e->f2_i - e->f2_prev_i; k < nn; k += 24)   // (P8F_REFILL with the lanes of a group looped)
          for (int l = 23; l >= 0; l--) p8f_refill_group(sh, e->f2_prev_i + P8F_LOOK + 1u + k, nn - k, l);
        const bool look = fu.bp == 0 || fu.bp == 2 || fu.bp == 5;
        int total = 0;
        if (getenv("CMX_P8_TRACE_G") && g == (uint64_t)atoi(getenv("CMX_P8_TRACE_G"))) {
          const int k = d->slot_inst[atoi(getenv("CMX_P8_TRACE_SLOT"))];
          printf("g %llu: anyconf %u conflict[inst %d] %d\n", (unsigned long long)g, sh->anyconf[fu.lk & 1], k, sh->conflict[fu.lk & 1][k]);
          for (int s = d->inst[k].first; s < d->inst[k].first + d->inst[k].count; s++)
            printf("  slot %d: look %d nb %u | old cp0 %u (bucket %u) cpo %d runp %u (bucket %u) ctx %u chk %u\n", s, tmp[s].look, tmp[s].nb, sh->r.cp0[s], sh->r.cp0[s] >> 6, sh->r.cpo[s], sh->r.runp[s],
                   sh->r.runp[s] >> 6, p8f_ctx(d, fu, s), p8f_chk(d, fu, s));
        }
        if (!(look && sh->anyconf[fu.lk & 1]) && !sh->anyshared) {
          for (int s = SS - 1; s >= 0; s--) p8f_run(d, sh, fu, s, &tmp[s], p8f_count(sh, (int)t, 0, s));
          total = p8f_count(sh, (int)t, 0, SS);
        } else {
          e->fam_serial++;
          int anys = 0;
          auto walked = [&](int q) { return (look && sh->conflict[fu.lk & 1][q]) || sh->shared[q]; };
          // as in the kernel: the walked instances first, in order (narrowed at a lookup bit: p8f_miniwalk; sh->db is exact afterwards), then every
          // other context in one pass with its rank taken from db
          for (int k = 0; k < d->ninst; k++) {
            if (!walked(k)) continue;
            const int first = d->inst[k].first, end = first + d->inst[k].count;
            const int whole = !look || sh->shared[k] || !e->fam_miniwalk;
            if (!whole) for (int s = end - 1; s >= first; s--) p8f_register(sh, fu, k, s, &tmp[s]);
            uint32_t full = 0;
            (void)p8f_miniwalk(d, sh, fu, k, p8f_count(sh, (int)t, 0, first), whole, e->fam_miniwalk == 2, &full);
            sh->wfull[k] = (uint8_t)full;
            if (!whole) { e->fam_mini++; e->fam_mini_full += full; }
          }
          for (int s = SS - 1; s >= 0; s--) {
            const int k = d->slot_inst[s];
            if (walked(k) && (sh->wfull[k] || sh->ink[s] == 2)) p8f_reload(d, sh, s);
            else p8f_run(d, sh, fu, s, &tmp[s], p8f_count(sh, (int)t, 0, s));
          }
          const int base = p8f_count(sh, (int)t, 0, SS);
          if (look) for (int k = 0; k < d->ninst; k++) if (walked(k)) sh->shared[k] = (uint8_t)p8f_shares(d, sh, k, !(sh->shared[k] || !e->fam_miniwalk));   // slots change hands at lookup bits only
          for (int q = 0; q < d->ninst; q++) anys |= sh->shared[q];
          sh->anyshared = (uint32_t)anys;
          total = base;
          e->multi_conf += anys != 0;
        }
        if (look) for (int tid = 255; tid >= 0; tid--) p8f_clear_next(sh, fu.lk, tid, 256);
        e->f2_prev_i = e->f2_i; e->f2_i += (uint32_t)total;
        if (getenv("CMX_P8FAM_CHECK")) {   // invariant: the cached bytes equal the table
          for (int s = 0; s < SS; s++) {
            const uint8_t* T = d->inst[d->slot_inst[s]].table;
            const P8FamHome* r = &sh->r;
            int bad = r->rc[s] != T[r->runp[s]] || r->rb[s] != T[r->runp[s] + 1];
            for (int k = 0; k < 7; k++) bad |= r->slot[s][k] != T[r->cp0[s] + k];
            if (bad) {
              printf("step %llu (bp %d) slot %d inst %d: cache != table: cp0 %u cpo %d runp %u rc %d/%d rb %d/%d slot", (unsigned long long)g, fu.bp, s, d->slot_inst[s], r->cp0[s], r->cpo[s],
                     r->runp[s], r->rc[s], T[r->runp[s]], r->rb[s], T[r->runp[s] + 1]);
              for (int k = 0; k < 7; k++) printf(" %d/%d", r->slot[s][k], T[r->cp0[s] + k]);
              printf("  conflict-bit %d\n", (int)(look && sh->anyconf[fu.lk & 1]));
              fflush(stdout);
              setenv("CMX_P8FAM_CHECK_HIT", "1", 1);
            }
          }
          if (getenv("CMX_P8FAM_CHECK_HIT")) { unsetenv("CMX_P8FAM_CHECK"); }
        }
      }
    }
    if (const char* ts = getenv("CMX_P8_TRACE_SLOT")) {
      const int s = atoi(ts);
      const P8CmDev* d = &S.fam;
      const uint8_t* T = d->inst[d->slot_inst[s]].table;
      uint32_t cp, cp0, runp, smc;
      if (e->use_v1) { cp = e->fsh.r.cp[s]; cp0 = e->fsh.r.cp0[s]; runp = e->fsh.r.runp[s]; smc = (uint32_t)e->fsh.r.sm_cxt[s]; }
      else { const P8FamHome* r = &e->f2->r; cp0 = r->cp0[s]; cp = r->cpo[s] == P8F_NIL ? 0xFFFFFFFFu : cp0 + r->cpo[s]; runp = r->runp[s]; smc = r->smc[s]; }
      if (g >= (uint64_t)atoi(getenv("CMX_P8_TRACE_FROM")) && g < (uint64_t)atoi(getenv("CMX_P8_TRACE_FROM")) + 24) {
        if (getenv("CMX_P8_TRACE_ADDR")) printf("[T[%d]=%d] ", atoi(getenv("CMX_P8_TRACE_ADDR")), T[atoi(getenv("CMX_P8_TRACE_ADDR"))]);
        printf("g %llu bp %d: cp %d cp0 %u runp %u T[cp] %d rc %d rb %d smc %u | x %d %d %d %d %d\n", (unsigned long long)g, (int)(g & 7), (int)cp, cp0, runp, cp == 0xFFFFFFFFu ? -1 : T[cp], T[runp], T[runp + 1], smc,
               xr[d->slot_off[s]], xr[d->slot_off[s] + 1], xr[d->slot_off[s] + 2], xr[d->slot_off[s] + 3], xr[d->slot_off[s] + 4]);
      }
    }
    if (g == 0) { memcpy(orow, S.tail.out, sizeof S.tail.out); continue; }   // no step 0: the constructor's 0.5
    {
      int c0g = 1;
      for (int j = 0; j < (int)(t & 7); j++) c0g = c0g * 2 + bits[t - (t & 7) + j];
      for (int l = S.lanes.nlanes - 1; l >= 0; l--) p8s_glane_step(&S.lanes, &S.lanes.regs[l], l, &c.ops[t * P8_NLANE + l], y, order[t], (int)(t & 7), c0g, xr, md);
    }
    if (!md) {
      for (int tid = P8DMC_THREADS - 1; tid >= 0; tid--) p8d_dmc_step1(&S.dmc, &e->dsh, tid, y);
      for (int tid = P8DMC_THREADS - 1; tid >= 0; tid--) p8d_dmc_step2(&S.dmc, &e->dsh, tid, (int)(g & 7), xr + L.dmc_off);
      for (int tid = P8DMC_THREADS - 1; tid >= 0; tid--) p8d_dmc_step3(&S.dmc, &e->dsh, tid);
    } else {   // the image model's own maps: its ContextMap in the reference's context order, then its lane table
      P8CmDev* xd = &S.xfam[md - 1];
      P8CmBit u;
      const int bp = (int)(t & 7);
      int c0 = 1;
      for (int j = 0; j < bp; j++) c0 = c0 * 2 + bits[t - bp + j];
      u.y = y; u.bp = bp; u.c0 = c0; u.c1 = (t >> 3) ? bytes[(t >> 3) - 1] : e->last_byte; u.order = 0;
      u.ctx = c.xfam_ctx + (t >> 3) * P8_XL_MAXS; u.chk = c.xfam_chk + (t >> 3) * P8_XL_MAXS; u.out = xr;
      int a0 = (int)u.ctx[P8_XL_MAXS - 2], a1 = (int)u.ctx[P8_XL_MAXS - 1];   // the slots whose map is called with a context this byte
      P8CmShared* xs = &e->xsh[md - 1];   // the four phases of cmx_p8s_xfam_kernel, lanes looped per phase
      if (xd->nslots == 0) a0 = a1 = 0;    // (a model without ContextMaps: im1bitModel)
      xs->act_lo = a0; xs->act_hi = a1;
      for (int sl = a1 - 1; sl >= a0; sl--) p8d_cm_touch(xd, xs, u, sl);
      for (int sl = a1 - 1; sl >= a0; sl--) p8d_cm_check(xd, xs, sl);
      for (int sl = a1 - 1; sl >= a0; sl--) p8d_cm_draw(xd, xs, sl);
      for (int sl = a1 - 1; sl >= a0; sl--) p8d_cm_run(xd, xs, u, sl);
      P8XLanesDev* XD = &S.xlanes[md - 1];
      const P8LaneTabs tb = {XD->nex, XD->stretch, nullptr};
      for (int l = XD->nlanes - 1; l >= 0; l--) {   // a map the step does not call is not touched and writes nothing: its positions may be another face's (im8bitModel: gray / palette)
        const uint32_t op = c.xops[t * P8_XL_NLANE + l];
        if (XD->lane[l].q.kind == P8L_JPG) p8s_lane_jpg(&XD->lane[l], &tb, XD->squash, &c.xops[t * P8_XL_NLANE + l], y, xr);
        else if (XD->lane[l].q.kind == P8L_HT16) p8s_lane_ht16(&XD->lane[l], &tb, &c.xops[t * P8_XL_NLANE + l], y, (int)(t & 7), xr);
        else if (XD->lane[l].q.kind == P8L_PIC2) p8s_lane_pic2(&XD->lane[l], &tb, &XD->regs[l], op, c.xops[t * P8_XL_NLANE + l + 1], y, xr);
        else if (op & P8OP_MIX) p8s_lane_step_t(&XD->lane[l], &tb, &XD->regs[l], op, y, order[t], xr, P8_NX);
      }
    }
    // ---- mixer + tail (Mixer::p :553-581, Predictor::update :8281-8358) ----
    P8TailDev& Tl = S.tail;
    Tl.misses += Tl.misses + (uint64_t)((Tl.pr >> 11) != y);
    int16_t xs[P8_NX + 8];
    memset(xs, 0, sizeof xs);
    int nx = P8_NX;
    if (g < 8) { nx = S.mix.nx_first; for (int i = 0; i < nx; i++) xs[i] = xr[S.mix.first_map[i]]; }
    else if (md) {   // a model's step: its inputs in add() order through the model's map (p8_rec.h), fewer weight sets
      const P8XLayout& X = L.xl[md - 1];
      nx = c.apm[t].m[1];
      if (nx < 0 || nx > P8_NX || c.apm[t].model != md) { fprintf(stderr, "p8stage_emul: step %zu (byte %zu of the chunk, stream step %llu): model %d, record says model %d with %d inputs\n", t, t >> 3, (unsigned long long)g, md, (int)c.apm[t].model, nx); return -98; }
      const int skip = c.apm[t].m[0] ? X.opt_n : 0;   // the model's own ContextMap is silent this byte: its inputs are not there
      const int skp = c.apm[t].m[0] == 1 ? skip : 0;
      for (int i = 0; i < nx; i++) xs[i] = xr[X.map[(skp && i >= X.opt_lo) ? i + skp : i]];
      if (md == P8_MODEL_JPEG && c.apm[t].m[3]) xs[nx - 1] = (int16_t)c.apm[t].m[3];   // a stuffed / restart step's one constant input (jpegModel :6466, :6473)
    }
    else memcpy(xs, xr, P8_NX * 2);
    const int nsel = md ? (int)c.apm[t].m[2] : P8_NSEL;
    const float cf = (float)(1.0 / 4095);
    int ne = nx;   // exported values in front of the second layer's: the inputs in order -- or, a coded JPEG step, through the model's export map
    if (md && c.apm[t].m[0] == 2) {
      const P8XLayout& X = L.xl[md - 1];
      ne = X.exp_n;
      for (int i = 0; i < ne; i++) Tl.out[i] = (float)p8s_squash(Tl.squash, xr[X.exp[i]]) * cf;
    } else
    for (int i = 0; i < nx; i++) Tl.out[i] = (float)p8s_squash(Tl.squash, xs[i]) * cf;
    const int npad = (nx + 7) & ~7;
    int row[P8_NSEL], pr[P8_NSEL];
    int16_t st[32];
    memset(st, 0, sizeof st);
    for (int i = 0; i < nsel; i++) {
      row[i] = md ? c.sel[t * P8_NSEL + i] : p8s_sel(i, c.sel[t * P8_NSEL + i], order[t], Tl.pr);
      const int dsum = dot(xs, S.mix.wx + (size_t)row[i] * P8_NX, npad);
      pr[i] = p8s_squash(Tl.squash, (int32_t)((uint32_t)dsum * 9u) >> 9);
      st[i] = Tl.stretch[pr[i]];
      Tl.out[ne + i] = (float)p8s_squash(Tl.squash, st[i]) * cf;
    }
    const int p2 = p8s_squash(Tl.squash, dot(st, S.mix.wx2, 32) >> 9);
    int res[8];
    int fin;
    if (c.apm[t].text >= P8_APM_COLOR) fin = p8s_tail_image(&Tl, &c.apm[t], y, p2, Tl.out + ne + nsel);
    else {
      for (int j = 3; j >= 0; j--) p8s_tail_a(&Tl, &c.apm[t], y, p2, j, res);
      for (int j = 2; j >= 0; j--) p8s_tail_b(&Tl, &c.apm[t], y, p2, j, res);
      fin = p8s_tail_c(&c.apm[t], p2, res, Tl.out + ne + nsel);
    }
    Tl.pr = fin;
    memcpy(orow, Tl.out, sizeof Tl.out);
    // training with this step's bit (the reference does it at the start of the next step: nothing reads the rows in between)
    const int yb = bits[t];
    for (int i = 0; i < nsel; i++) train(xs, S.mix.wx + (size_t)row[i] * P8_NX, npad, ((yb << 12) - pr[i]) * 7);
    train(st, S.mix.wx2, 32, ((yb << 12) - p2) * 7);
  }
  if (e->late && T) {   // the chunk's last bit: into the run registers a later chunk starts from, and to the front end
    p8f_uni_tail(&f_run, (int)((T - 1) & 7), bits[T - 1]);
    p8f_front_set_bit(e->front, bits[T - 1]);
  }
  if (!e->use_v1) { f_last_y = f_run.last_y; f_c1 = f_run.c1; }
  S.fam.last_y = f_last_y; S.fam.c1 = f_c1;
  if (!e->use_v1) {
    if (e->fam_owner == 0) for (int tid = 0; tid < 256; tid++) p8f_store(&S.fam, S.fam_home, S.fam.sm, e->f2, e->f2_i, tid, 256);
    else {   // (the generic family's own state was stored when the generator left it; what the model's family shares with it goes back now)
      const int m = e->fam_owner;
      const P8ViewMap& V = S.xview[m - 1];
      for (int g = 0; g < V.n; g++) for (int i = 0; i < V.count[g]; i++)
        p8v_slot_out(S.fam_home, S.fam.sm, &e->xsh[m - 1].r, S.xfam[m - 1].sm, S.fam.inst[S.fam.slot_inst[V.gen_first[g]]].table, V.gen_first[g] + i, V.view_first[g] + i);
      S.fam.rnd = e->xsh[m - 1].rnd; e->fam_owner = 0;
    }
  }
  if (n) e->last_byte = bytes[n - 1];
  for (int k = 0; k < P8_NCM2; k++) { S.cm2[k].bits = run_bits[k]; S.cm2[k].last_y = c_last_y[k]; if (!e->use_v1) S.cm2[k].regs = e->c2[k]->base.r; }
  e->steps += T; e->last_bit = T ? bits[T - 1] : e->last_bit;
  return 0;
}
}
