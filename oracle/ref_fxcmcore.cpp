// oracle/ref_fxcmcore.cpp -- TEST INFRASTRUCTURE ONLY. Compiled by oracle/Makefile into oracle/_ref/libcmixreffxcm.so
// together with the UNMODIFIED reference source src/models/fxcmv1.cpp (included below from where it lies under
// /root/reference; nothing is copied). Exposes the vendored fxcm model's own building blocks -- the tables its
// constructor computes, Mixer1, StateMap, StateMap1, APM, RunContextMap, SmallStationaryContextMap, DirectStateMap --
// so that the C restatement in oracle/fxcm_core.c can be pinned against the reference's objects, bit for bit.
#include <stdint.h>
#include <string.h>

#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <math.h>
#include <ctype.h>
#include <assert.h>
#include <algorithm>
#include <unordered_map>
#include <memory>
#include <valarray>
#include <vector>

// The reference's alloc1() (fxcmv1.cpp:146-150) aligns the pointer it got from calloc() upwards WITHOUT having asked
// for the extra bytes: harmless for the model's huge, mmap-backed tables, but the small instances the tests build
// would write past their allocation into their neighbours. The harness pads every allocation the reference makes;
// the reference source itself is untouched.
static void* reffx_padded_calloc(size_t count, size_t size) { return calloc(count * size + 128, 1); }
#define calloc(count, size) reffx_padded_calloc(count, size)
#define private public
#define protected public
#include "models/fxcmv1.cpp"
#undef private
#undef protected
#undef calloc

char* dictionary_path = NULL;  // hidden inputs of the model (predictor.cpp:359, runner.cpp:17)
int lstmpr = 0, lstmex = 0;

namespace fx = fxcmv1;

static int drain(int16_t* inputs, float* exported, int* nexported) {
  const int n = fx::x.mxInputs1.ncount;
  for (int i = 0; i < n; ++i) inputs[i] = fx::x.mxInputs1.n[i];
  if (exported) {
    *nexported = (int)fx::prediction_index;
    for (unsigned i = 0; i < fx::prediction_index; ++i) exported[i] = fx::model_predictions[i];
  }
  fx::x.mxInputs1.ncount = 0;
  fx::prediction_index = 0;
  return n;
}
static void set_bit_state(int y, int bpos, int c0) {  // update1 :4776-4779
  fx::x.y = y; fx::x.bpos = bpos; fx::x.c0 = c0;
  fx::x.bposshift = 7 - bpos;
  fx::x.c0shift_bpos = (c0 << 1) ^ (256 >> fx::x.bposshift);
}

extern "C" {

// what Predictor::Predictor() precomputes (:4846-4875), without building the 3.7 GB of model tables
void reffx_init() {
  static bool done = false;
  if (done) return;
  done = true;
  int o = 2;
  for (int i = 0; i < 1024; ++i) fx::dt[i] = 4096 / (o), o++;
  fx::dt[1023] = 1;
  for (int i = 0; i <= 4095; i++) fx::strt[i] = fx::stretchc(i);
  for (int i = -2047; i <= 2047; i++) fx::sqt[i + 2047] = fx::squashc(i);
  fx::InitIlog();
  fx::x.Init();
  for (int i = 0; i < 4096; i++) { fx::st2_p1[i] = fx::clp(fx::sc(12 * (i - 2048))); fx::st2_p2[i] = fx::clp(fx::sc(14 * (i - 2048))); }
  fx::StateTable st;
  st.Init(28, 28, 31, 29, 23, 4, 17, &fx::STA1[0][0]);
  st.Init(32, 28, 31, 28, 21, 5, 6, &fx::STA2[0][0]);
  st.Init(31, 27, 30, 27, 24, 4, 27, &fx::STA4[0][0]);
  st.Init(33, 31, 31, 24, 20, 4, 33, &fx::STA5[0][0]);
  st.Init(28, 29, 30, 30, 23, 3, 22, &fx::STA6[0][0]);
  st.Init(28, 29, 33, 23, 23, 6, 14, &fx::STA7[0][0]);
  fx::pre2(&fx::STA7[0][0]);
}
void reffx_tables(int16_t* squash4095, int16_t* stretch4096, uint8_t* ilog256, int32_t* dt1024, uint8_t* sta6x1024, int16_t* pre1_256,
                  int16_t* st2_p1_4096, int16_t* st2_p2_4096) {
  reffx_init();
  memcpy(squash4095, fx::sqt, 4095 * 2);
  memcpy(stretch4096, fx::strt, 4096 * 2);
  memcpy(ilog256, fx::ilog, 256);
  memcpy(dt1024, fx::dt, 1024 * 4);
  const uint8_t* tabs[6] = {&fx::STA1[0][0], &fx::STA2[0][0], &fx::STA4[0][0], &fx::STA5[0][0], &fx::STA6[0][0], &fx::STA7[0][0]};
  for (int k = 0; k < 6; ++k) memcpy(sta6x1024 + 1024 * k, tabs[k], 1024);
  memcpy(pre1_256, fx::pre1, 256 * 2);
  memcpy(st2_p1_4096, fx::st2_p1, 4096 * 2);
  memcpy(st2_p2_4096, fx::st2_p2, 4096 * 2);
}
static const uint8_t* sta(int which) {
  const uint8_t* tabs[6] = {&fx::STA1[0][0], &fx::STA2[0][0], &fx::STA4[0][0], &fx::STA5[0][0], &fx::STA6[0][0], &fx::STA7[0][0]};
  return tabs[which];
}

// Mixer1 (:472-660): one coded bit = update(y) on the previous inputs, new inputs, context, p() or p1()
struct MixerBox { fx::Mixer1 m; short* tx; };
void* reffx_mixer_new(int n, int m, int shift, int elim, int uperr) {
  reffx_init();
  MixerBox* b = new MixerBox();
  b->tx = (short*)aligned_alloc(64, (size_t)n * 2 + 64);
  memset(b->tx, 0, (size_t)n * 2 + 64);
  b->m.Init(m, shift, elim, uperr);
  b->m.setTxWx(n, b->tx);
  return b;
}
int reffx_mixer_step(void* h, int y, const int16_t* in, int cxt, int elim, int use_p1, int* pr_out) {
  MixerBox* b = (MixerBox*)h;
  b->m.elim = elim;  // adapted per byte by the caller (:4766-4771)
  b->m.update(y);
  memcpy(b->tx, in, (size_t)b->m.N * 2);
  b->m.cxt = cxt;
  const int r = use_p1 ? b->m.p1() : b->m.p();
  *pr_out = b->m.pr;
  return r;
}

// StateMap (:672-705), StateMap1 (:707-736)
void* reffx_statemap_new(int n, int which_sta) { reffx_init(); fx::StateMap* s = new fx::StateMap(); s->Init(n, sta(which_sta)); return s; }
int reffx_statemap_set(void* h, int y, int c) { fx::x.y = y; fx::StateMap* s = (fx::StateMap*)h; s->set(c); return s->pr; }
void* reffx_statemap1_new(int n, int limit) { reffx_init(); fx::StateMap1* s = new fx::StateMap1(); s->Init(n, limit); return s; }
int reffx_statemap1_set(void* h, int y, int c) { fx::x.y = y; fx::StateMap1* s = (fx::StateMap1*)h; s->set(c); return s->pr; }

// APM<S> (:1622-1643); the model uses S = 0x10000 ... see :3287-3292; one size is enough to pin the arithmetic
void* reffx_apm_new() { reffx_init(); fx::APM<1024>* a = new fx::APM<1024>(); a->Init(); return a; }
int reffx_apm_p(void* h, int pr, int cxt, int rate, int y) { return ((fx::APM<1024>*)h)->p(pr, cxt, rate, y); }

// RunContextMap (:756-829)
void* reffx_rcm_new(int m, int ml) { reffx_init(); fx::RunContextMap* r = new fx::RunContextMap(); r->Init(m, ml); return r; }
void reffx_rcm_set(void* h, uint32_t cx, int c1) { ((fx::RunContextMap*)h)->set(cx, (fx::U8)c1); }
int reffx_rcm_mix(void* h, int y, int bpos, int c0, int16_t* out) {
  set_bit_state(y, bpos, c0);
  const int r = ((fx::RunContextMap*)h)->mix();
  drain(out, nullptr, nullptr);
  return r;
}

// SmallStationaryContextMap (:831-863): note the second add() re-uses the exported slot (prediction_index--)
void* reffx_sscm_new(int bits_of_context, int input_bits) { reffx_init(); fx::SmallStationaryContextMap* s = new fx::SmallStationaryContextMap(); s->Init(bits_of_context, input_bits); return s; }
void reffx_sscm_set(void* h, uint32_t ctx) { ((fx::SmallStationaryContextMap*)h)->set(ctx); }
int reffx_sscm_mix(void* h, int y, int rate, int16_t* out, float* exported, int* nexported) {
  fx::x.y = y;
  ((fx::SmallStationaryContextMap*)h)->mix(rate);
  return drain(out, exported, nexported);
}

// DirectStateMap (:1646-1683)
void* reffx_dsm_new(int m, int c, int which_sta) { reffx_init(); fx::DirectStateMap* d = new fx::DirectStateMap(); d->Init(m, c, sta(which_sta)); return d; }
int reffx_dsm_step(void* h, int y, const uint32_t* cx, int n, int16_t* out, float* exported, int* nexported) {
  fx::x.y = y;
  fx::DirectStateMap* d = (fx::DirectStateMap*)h;
  for (int i = 0; i < n; ++i) d->set(cx[i], y);
  d->mix();
  return drain(out, exported, nexported);
}

// ContextMap (:971-1174, 64-byte buckets of 7), ContextMap1 (:1176-1379, 32-byte buckets of 3), ContextMap2 (:1408-1612,
// 128-byte buckets of 14). One coded bit: at bpos == 0 the byte's contexts first (set(), or sets() = "skip this slot").
}  // extern "C"
template <class CM> static void* cm_new(uint32_t m, int c, int s3, int which_sta, int cs4, int k, int u, int which_st2) {
  reffx_init();
  CM* cm = new CM();
  memset(cm, 0, sizeof(CM));
  cm->Init(m, c, s3, sta(which_sta), cs4, k, u, which_st2 == 0 ? fx::st2_p0 : which_st2 == 1 ? fx::st2_p1 : fx::st2_p2);
  return cm;
}
template <class CM> static int cm_step(void* h, int y, int bpos, int c0, uint32_t c4, const uint32_t* cx, const uint8_t* skip, int n, int16_t* out,
                                       float* exported, int* nexported, int* ninputs) {
  set_bit_state(y, bpos, c0);
  fx::x.c4 = c4;
  CM* cm = (CM*)h;
  if (bpos == 0)
    for (int i = 0; i < n; ++i) { if (skip[i]) cm->sets(); else cm->set(cx[i]); }
  const int r = cm->mix();
  *ninputs = drain(out, exported, nexported);
  return r;
}
extern "C" {
void* reffx_cm_new(int kind, uint32_t m, int c, int s3, int which_sta, int cs4, int k, int u, int which_st2) {
  return kind == 0 ? cm_new<fx::ContextMap>(m, c, s3, which_sta, cs4, k, u, which_st2)
       : kind == 1 ? cm_new<fx::ContextMap1>(m, c, s3, which_sta, cs4, k, u, which_st2)
                   : cm_new<fx::ContextMap2>(m, c, s3, which_sta, cs4, k, u, which_st2);
}
int reffx_cm_step(int kind, void* h, int y, int bpos, int c0, uint32_t c4, const uint32_t* cx, const uint8_t* skip, int n, int16_t* out,
                  float* exported, int* nexported, int* ninputs) {
  return kind == 0 ? cm_step<fx::ContextMap>(h, y, bpos, c0, c4, cx, skip, n, out, exported, nexported, ninputs)
       : kind == 1 ? cm_step<fx::ContextMap1>(h, y, bpos, c0, c4, cx, skip, n, out, exported, nexported, ninputs)
                   : cm_step<fx::ContextMap2>(h, y, bpos, c0, c4, cx, skip, n, out, exported, nexported, ninputs);
}

// the byte history the match models read through buf() / bufr() (:3251-3253, :3411-3416)
void reffx_buf_reset() { memset(fx::buffer, 0, sizeof fx::buffer); fx::pos = 0; }
void reffx_buf_push(int byte) { fx::buffer[fx::pos & fx::BMASK] = (fx::U8)byte; fx::pos++; }

// SparseMatchModel (:1742-1829)
void* reffx_sparsematch_new() { reffx_init(); fx::SparseMatchModel* m = new fx::SparseMatchModel(); m->Init(); return m; }
int reffx_sparsematch_p(void* h, int y, int bpos, int c0, int16_t* out, int* state4) {
  set_bit_state(y, bpos, c0);
  fx::SparseMatchModel* m = (fx::SparseMatchModel*)h;
  const int r = m->p();
  drain(out, nullptr, nullptr);
  state4[0] = (int)m->hashIndex; state4[1] = (int)m->index; state4[2] = m->expectedByte; state4[3] = m->valid;
  return r;
}

// The stemmer's DATA tables (:2417-2653: suffix lists, word lists, per-suffix flag words) written out as C arrays --
// numbers and strings only -- for scripts/gen_fxcm_tables.py.
static int emit_strs(char* o, const char* name, const char** a, int n) {
  int k = sprintf(o, "static const char* const FXW_%s[] = {", name);
  for (int i = 0; i < n; ++i) k += sprintf(o + k, "%s\"%s\"", i ? "," : "", a[i]);
  return k + sprintf(o + k, "};\n");
}
static int emit_pairs(char* o, const char* name, const char* (*a)[2], int n) {
  int k = sprintf(o, "static const char* const FXW_%s[][2] = {", name);
  for (int i = 0; i < n; ++i) k += sprintf(o + k, "%s{\"%s\",\"%s\"}", i ? "," : "", a[i][0], a[i][1]);
  return k + sprintf(o + k, "};\n");
}
static int emit_u32(char* o, const char* name, const fx::U32* a, int n) {
  int k = sprintf(o, "static const uint32_t FXW_%s[] = {", name);
  for (int i = 0; i < n; ++i) k += sprintf(o + k, "%s%u", i ? "," : "", a[i]);
  return k + sprintf(o + k, "};\n");
}
int reffx_dump_stem_tables(char* o) {
  int k = 0;
#define STRS(x, n) k += emit_strs(o + k, #x, fx::x, n)
#define U32S(x, n) k += emit_u32(o + k, #x, fx::x, n)
  STRS(VerbWords1, NUM_VERB); STRS(Numbers, NUM_NUM); STRS(ConjWords, NUM_CONJ_WORDS); STRS(ApoWords, NUM_APO_WORDS); STRS(PrepWords, NUM_PREP_WORDS);
  STRS(ConAdVerPrepWords, NUM_CAVER_WORDS); STRS(VerbWords, NUM_VERB_WORDS); STRS(MaleWords, NUM_MALE_WORDS); STRS(FemaleWords, NUM_FEMALE_WORDS);
  STRS(ArticleWords, NUM_ARTICLE_WORDS); STRS(SuffixesStep0, NUM_SUFFIXES_STEP0); STRS(SuffixesStep1b, NUM_SUFFIXES_STEP1b);
  U32S(TypesStep1b, NUM_SUFFIXES_STEP1b);
  k += emit_pairs(o + k, "SuffixesStep2", fx::SuffixesStep2, NUM_SUFFIXES_STEP2);
  U32S(TypesStep2, NUM_SUFFIXES_STEP2); U32S(TypesStep2Suffix, NUM_SUFFIXES_STEP2);
  k += emit_pairs(o + k, "SuffixesStep3", fx::SuffixesStep3, NUM_SUFFIXES_STEP3);
  U32S(TypesStep3, NUM_SUFFIXES_STEP3); U32S(TypesStep3Suffix, NUM_SUFFIXES_STEP3);
  STRS(SuffixesStep4, NUM_SUFFIXES_STEP4); U32S(TypesStep4, NUM_SUFFIXES_STEP4); U32S(TypesStep4Suffix, NUM_SUFFIXES_STEP4);
  STRS(ExceptionsRegion1, NUM_EXCEPTION_REGION1);
  k += emit_pairs(o + k, "Exceptions1", fx::Exceptions1, NUM_EXCEPTIONS1);
  U32S(TypesExceptions1, NUM_EXCEPTIONS1); STRS(Exceptions2, NUM_EXCEPTIONS2); U32S(TypesExceptions2, NUM_EXCEPTIONS2);
#undef STRS
#undef U32S
  return k;
}

// fxcm's Word + EnglishStemmer (:2302-3216) on one word; letters added the way the model adds them (Word::operator+=)
int reffx_stem_word(const char* s, int blpos, uint8_t* letters64, int* start_end, uint32_t* hash_type_suffix_prefix) {
  static fx::EnglishStemmer stemmer;
  fx::x.blpos = blpos;
  fx::Word w;
  for (const char* p = s; *p; ++p) w += *p;
  const int r = stemmer.Stem(&w);
  memcpy(letters64, w.Letters, 64);
  start_end[0] = w.Start; start_end[1] = w.End;
  hash_type_suffix_prefix[0] = w.Hash; hash_type_suffix_prefix[1] = w.Type; hash_type_suffix_prefix[2] = w.Suffix; hash_type_suffix_prefix[3] = w.Preffix;
  return r;
}

void reffx_wrt_tables(uint8_t* out768) { memcpy(out768, wrt_2b, 256); memcpy(out768 + 256, wrt_3b, 256); memcpy(out768 + 512, fx::wrt_4b, 256); }

// The whole model as cmix drives it: fxcmv1::Predictor (:4837-4890) behind FXCM::Perceive / FXCM::Predict (:4893-4911).
// All of its state is in namespace-level globals: ONE model per loaded copy of this library, and none of the
// single-block entries above may be used in the same copy.
void* reffx_model_new() { return new fx::Predictor(); }
void* reffx_model_new_dict(const char* path) { dictionary_path = strdup(path); return new fx::Predictor(); }  // cmix's WRT dictionary (runner.cpp:290)
int reffx_model_update(void* h, int bit, int hint_pr, int hint_ex, float* out431) {
  lstmpr = hint_pr; lstmex = hint_ex;
  fx::x.y = bit;
  ((fx::Predictor*)h)->update();
  for (int i = 0; i < 431; ++i) out431[i] = fx::model_predictions[i];
  return fx::pr;
}
// state injection (round 6): the model's position in its block -- what fxcm's thresholds read (update1 :4772-4774: 3.67 MB and 14.7 MB; modelPrediction
// :3200, :3918, :3965: 448 MB .. 463 MB). The twin of orc_fx_model_set_blpos / fxe_set_blpos.
// The two rates the model keeps (recomputed at every byte boundary from the position alone) are set to what that recomputation gives for the position.
void reffx_model_set_blpos(void* h, int blpos) {
  fx::x.blpos = blpos;
  fx::sscmrate = (blpos > 14 * 256 * 1024);
  (void)h; fx::rate = 6 + (blpos > 14 * 256 * 1024) + (blpos > 28 * 512 * 1024);
}
int reffx_model_debug(uint32_t* out) {
  int n = 0;
  for (int i = 0; i < 12; i++) out[n++] = (uint32_t)fx::mxA[i].cxt;
  out[n++] = fx::stream2b; out[n++] = fx::stream3b; out[n++] = fx::stream2bR; out[n++] = fx::stream3bR; out[n++] = fx::word0; out[n++] = (uint32_t)fx::fc;
  out[n++] = fx::BrFcIdx; out[n++] = fx::FcIdx; out[n++] = (uint32_t)fx::isParagraph; out[n++] = fx::fccxt.context; out[n++] = fx::brcxt.context;
  out[n++] = fx::qocxt.context; out[n++] = fx::worcxt.fword; out[n++] = fx::worcxt.Word(1); out[n++] = fx::worcxt.Type(1); out[n++] = (uint32_t)fx::ordX;
  out[n++] = (uint32_t)fx::ordW; out[n++] = fx::isMatch; out[n++] = fx::fails; out[n++] = (uint32_t)fx::col;
  out[n++] = fx::colcxt.colb(1, 0); out[n++] = fx::colcxt.colb(1, 1); out[n++] = fx::colcxt.nlChar; out[n++] = (uint32_t)fx::colcxt.rows; out[n++] = (uint32_t)fx::colcxt.collen(1);
  out[n++] = (uint32_t)fx::nl1; out[n++] = (uint32_t)fx::colcxt.abovecellpos; out[n++] = (uint32_t)fx::numlen0;
  out[n++] = (uint32_t)(fx::isText | fx::isMath << 1 | fx::isPre << 2 | fx::isNowiki << 3); out[n++] = (uint32_t)fx::deccode; out[n++] = (uint32_t)fx::lastCW;
  return n;
}
// the byte contexts the maps currently hold: cmC[0..5] x 8, cmC1[0..7] x 8, cmC2[0..17] x 8 (unused slots as they are)
void reffx_model_contexts(uint32_t* out) {
  int n = 0;
  for (int k = 0; k < 6; k++) for (int i = 0; i < 8; i++) out[n++] = fx::cmC[k].cxt[i];
  for (int k = 0; k < 8; k++) for (int i = 0; i < 8; i++) out[n++] = fx::cmC1[k].cxt[i];
  for (int k = 0; k < 18; k++) for (int i = 0; i < 8; i++) out[n++] = fx::cmC2[k].cxt[i];
}

}  // extern "C"
