// oracle/ref_paq8core.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// C-ABI window onto the numeric building blocks of the UNMODIFIED paq8 model (reference src/models/paq8.cpp, compiled
// here as part of this translation unit, from where it lies): squash/stretch tables, the two-layer int16 Mixer
// (:513-598, dot_product/train :403-432), APM1 (:600-621), StateMap (:623-645), StateMap32 (:645-690), APM (:691-712).
// SURVEY.md 8(a') lists them as what the paq8/fxcm device stages must reproduce; oracle/paq8_core.c restates them and
// tests/test_oracle_paq8core.py pins the restatement against these entry points. No reference logic is re-implemented
// here; only member access is widened. Built as its own shared object (oracle/_ref/libcmixrefpaq8.so).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <unordered_map>
#include <valarray>
#include <vector>
#include <ctype.h>
#include <dirent.h>
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>

#define private public
#define protected public
#include "models/paq8.cpp"
#undef private
#undef protected

char* dictionary_path = NULL;  // referenced by other reference objects' externs; unused here

// Mixer's members are private by default (no access keyword to widen): pointers to them are taken through explicit
// template instantiation, which is exempt from access checking.
namespace {
template <typename Tag, typename Tag::type M> struct Rob { friend typename Tag::type get(Tag) { return M; } };
struct MixerNx { typedef int paq8::Mixer::*type; friend type get(MixerNx); };
struct MixerTx { typedef paq8::Array<short, 16> paq8::Mixer::*type; friend type get(MixerTx); };
template struct Rob<MixerNx, &paq8::Mixer::nx>;
template struct Rob<MixerTx, &paq8::Mixer::tx>;
struct MixerCxt { typedef paq8::Array<int> paq8::Mixer::*type; friend type get(MixerCxt); };
struct MixerNcxt { typedef int paq8::Mixer::*type; friend type get(MixerNcxt); };
struct MixerBase { typedef int paq8::Mixer::*type; friend type get(MixerBase); };
template struct Rob<MixerCxt, &paq8::Mixer::cxt>;
template struct Rob<MixerNcxt, &paq8::Mixer::ncxt>;
template struct Rob<MixerBase, &paq8::Mixer::base>;
struct RndTable { typedef paq8::Array<paq8::U32> paq8::Random::*type; friend type get(RndTable); };
struct RndI { typedef int paq8::Random::*type; friend type get(RndI); };
template struct Rob<RndTable, &paq8::Random::table>;
template struct Rob<RndI, &paq8::Random::i>;
}  // namespace

extern "C" {

void refp8_ilog_table(uint8_t* out65536) {
  for (int i = 0; i < 65536; ++i) out65536[i] = (uint8_t)paq8::ilog((paq8::U16)i);
}

void refp8_tables(int16_t* squash4096, int16_t* stretch4096, int32_t* dt1024, uint8_t* state_table_256x4) {
  for (int i = 0; i < 4096; ++i) squash4096[i] = (int16_t)paq8::squash(i - 2048);
  for (int i = 0; i < 4096; ++i) stretch4096[i] = (int16_t)paq8::stretch(i);
  for (int i = 0; i < 1024; ++i) dt1024[i] = 16384 / (i + i + 3);  // what paq8::Predictor() fills (:8243-8245)
  for (int i = 0; i < 256; ++i)
    for (int j = 0; j < 4; ++j) state_table_256x4[4 * i + j] = paq8::State_table[i][j];
}
void refp8_init_dt() {
  for (int i = 0; i < 1024; ++i) paq8::dt[i] = 16384 / (i + i + 3);
}

// ---- Mixer ----
void* refp8_mixer_new(int n, int m, int s, int w) { return new paq8::Mixer(n, m, s, w); }
void refp8_mixer_free(void* h) { delete (paq8::Mixer*)h; }
// One bit: update() with the previous bit, the inputs, the weight-set selectors, p(). exported[] receives what the
// add() calls of both layers hand to cmix through AddPrediction (squash(x) / 4095), *nexp their number.
int refp8_mixer_step(void* h, int y_prev, const int16_t* x, int nx, const int* cx, const int* range, int ncx,
                     float* exported, int* nexp) {
  paq8::Mixer& m = *(paq8::Mixer*)h;
  paq8::y = y_prev;
  paq8::ResetPredictions();
  m.update();
  for (int i = 0; i < nx; ++i) m.add(x[i]);
  for (int i = 0; i < ncx; ++i) m.set(cx[i], range[i]);
  const int p = m.p();
  *nexp = (int)paq8::prediction_index;
  for (int i = 0; i < *nexp; ++i) exported[i] = paq8::model_predictions[i];
  return p;
}

// ---- APM1 / StateMap / StateMap32 / APM: p() = update with the previous bit, then predict ----
void* refp8_apm1_new(int n) { return new paq8::APM1(n); }
int refp8_apm1_p(void* h, int y_prev, int pr, int cx, int rate) { paq8::y = y_prev; return ((paq8::APM1*)h)->p(pr, cx, rate); }
void* refp8_statemap_new() { return new paq8::StateMap(); }
int refp8_statemap_p(void* h, int y_prev, int cx) { paq8::y = y_prev; return ((paq8::StateMap*)h)->p(cx); }
void* refp8_statemap32_new(int n) { return new paq8::StateMap32(n); }
int refp8_statemap32_p(void* h, int y_prev, int cx, int limit) { paq8::y = y_prev; return ((paq8::StateMap32*)h)->p(cx, limit); }
void* refp8_apm_new(int n) { return new paq8::APM(n); }
int refp8_apm_p(void* h, int y_prev, int pr, int cx, int limit) { paq8::y = y_prev; return ((paq8::APM*)h)->p(pr, cx, limit); }


// ---- the models below hand their outputs to a Mixer through add(): a recording mixer collects them ----
static paq8::Mixer* sink() {
  static paq8::Mixer* m = new paq8::Mixer(4096, 1, 64, 0);  // room for 64 set() calls per step
  return m;
}
// the weight-set selectors a model handed to Mixer::set(): the cumulative values the mixer stored (base + cx)
static int drain_sets(int* out) {
  paq8::Mixer* m = sink();
  int& ncxt = m->*get(MixerNcxt());
  paq8::Array<int>& cxt = m->*get(MixerCxt());
  const int n = ncxt;
  for (int i = 0; i < n; ++i) out[i] = cxt[i];
  ncxt = 0;
  m->*get(MixerBase()) = 0;
  return n;
}
static int drain(int16_t* out) {
  paq8::Mixer* m = sink();
  int& nx = m->*get(MixerNx());
  paq8::Array<short, 16>& tx = m->*get(MixerTx());
  const int n = nx;
  for (int i = 0; i < n; ++i) out[i] = tx[i];
  nx = 0;
  paq8::ResetPredictions();
  return n;
}

// ContextMap2 (:1164-1358): one coded bit. At bpos == 0 the nset byte contexts are handed to set() first.
void* refp8_cm2_new(uint64_t size_bytes, uint32_t count) { return new paq8::ContextMap2(size_bytes, count); }
void refp8_cm2_free(void* h) { delete (paq8::ContextMap2*)h; }
int refp8_cm2_step(void* h, int y_prev, int bpos, const uint64_t* ctx, int nset, int16_t* out, int* nout) {
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::ContextMap2& cm = *(paq8::ContextMap2*)h;
  if (bpos == 0) for (int i = 0; i < nset; ++i) cm.set(ctx[i]);
  const int r = cm.mix(*sink());
  *nout = drain(out);
  return r;
}

// SmallStationaryContextMap (:891-933), StationaryMap (:935-974), IndirectMap (:976-1008)
void* refp8_sscm_new(int boc, int bpc) { return new paq8::SmallStationaryContextMap(boc, bpc); }
void refp8_sscm_set(void* h, uint32_t ctx) { ((paq8::SmallStationaryContextMap*)h)->set(ctx); }
int refp8_sscm_mix(void* h, int y_prev, int rate, int mul, int div, int16_t* out) {
  paq8::y = y_prev;
  ((paq8::SmallStationaryContextMap*)h)->mix(*sink(), rate, mul, div);
  return drain(out);
}
void* refp8_smap_new(int boc, int bpc, int rate) { return new paq8::StationaryMap(boc, bpc, rate); }
void refp8_smap_set_direct(void* h, uint32_t ctx) { ((paq8::StationaryMap*)h)->set_direct(ctx); }
void refp8_smap_set(void* h, uint64_t ctx) { ((paq8::StationaryMap*)h)->set(ctx); }
int refp8_smap_mix(void* h, int y_prev, int mul, int div, int limit, int16_t* out) {
  paq8::y = y_prev;
  ((paq8::StationaryMap*)h)->mix(*sink(), mul, div, (paq8::U16)limit);
  return drain(out);
}
void* refp8_imap_new(int boc, int bpc) { return new paq8::IndirectMap(boc, bpc); }
void refp8_imap_set_direct(void* h, uint32_t ctx) { ((paq8::IndirectMap*)h)->set_direct(ctx); }
void refp8_imap_set(void* h, uint64_t ctx) { ((paq8::IndirectMap*)h)->set(ctx); }
int refp8_imap_mix(void* h, int y_prev, int mul, int div, int limit, int16_t* out) {
  paq8::y = y_prev;
  ((paq8::IndirectMap*)h)->mix(*sink(), mul, div, (paq8::U16)limit);
  return drain(out);
}
// ContextMap (:1010-1145): one coded bit; globals c0 / buf(1) / y / bpos are what mix() reads. The process-global rnd
// (:152-165) advances inside -- the harness exposes a reset so that a test starts from the constructor's state.
void* refp8_cm_new(uint64_t size_bytes, int count) { return new paq8::ContextMap(size_bytes, count); }
void refp8_cm_free(void* h) { delete (paq8::ContextMap*)h; }
// back to the state Random::Random() left: a snapshot of the live object taken when this library was loaded (paq8's
// globals are constructed first: they come earlier in this translation unit), before anything could draw from it
static std::vector<paq8::U32> rnd_pristine = [] {
  paq8::Array<paq8::U32>& t = paq8::rnd.*get(RndTable());
  std::vector<paq8::U32> v(64);
  for (int j = 0; j < 64; ++j) v[j] = t[j];
  return v;
}();
void refp8_rnd_reset() {
  paq8::Array<paq8::U32>& t = paq8::rnd.*get(RndTable());
  for (int j = 0; j < 64; ++j) t[j] = rnd_pristine[j];
  paq8::rnd.*get(RndI()) = 0;
}
uint32_t refp8_rnd_next() { return paq8::rnd(); }
int refp8_cm_step(void* h, int y_prev, int bpos, int c0, int c1, const uint64_t* ctx, int nset, int16_t* out, int* nout) {
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  if (paq8::buf.size() == 0) paq8::buf.setsize(1 << 16);
  paq8::buf[paq8::pos - 1] = (paq8::U8)c1;  // buf(1)
  paq8::ContextMap& cm = *(paq8::ContextMap*)h;
  if (bpos == 0) for (int i = 0; i < nset; ++i) cm.set(ctx[i]);
  const int r = cm.mix(*sink());
  *nout = drain(out);
  return r;
}

// RunContextMap over BH<4> (:778-813, :857-889): set() at byte boundaries (reads buf(1)), mix() every bit (reads c0,
// bpos). c1 = the byte just coded.
void* refp8_rcm_new(int m) { return new paq8::RunContextMap(m); }
void refp8_rcm_set(void* h, uint64_t cx, int c1) {
  if (paq8::buf.size() == 0) paq8::buf.setsize(1 << 16);
  paq8::buf[paq8::pos - 1] = (paq8::U8)c1;
  ((paq8::RunContextMap*)h)->set(cx);
}
int refp8_rcm_mix(void* h, int bpos, int c0, int16_t* out) {
  paq8::bpos = bpos;
  paq8::c0 = c0;
  const int r = ((paq8::RunContextMap*)h)->mix(*sink());
  drain(out);
  return r;
}

// dmcForest (:7637-7822): ten DMC state graphs; only y and bpos go in. `level` sets MEM() and with it the graph sizes.
void* refp8_dmc_new(int level) { paq8::level = level; return new paq8::dmcForest(); }
void refp8_dmc_free(void* h) { delete (paq8::dmcForest*)h; }
int refp8_dmc_mix(void* h, int y_prev, int bpos, int16_t* out) {
  paq8::y = y_prev;
  paq8::bpos = bpos;
  ((paq8::dmcForest*)h)->mix(*sink());
  return drain(out);
}

// linearPredictionModel (:4476-4502): three OLS<double,U8> predictors (:1364-1466) + two fixed ones over the last
// 64 bytes, each read through a SmallStationaryContextMap. last[i-1] = buf(i). One instance per process (statics).
int refp8_lpm_step(int y_prev, int bpos, int c0, const uint8_t* last, int nlast, int16_t* out) {
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  if (paq8::buf.size() == 0) paq8::buf.setsize(1 << 16);
  for (int i = 1; i <= nlast; ++i) paq8::buf[paq8::pos - i] = last[i - 1];
  paq8::linearPredictionModel(*sink());
  return drain(out);
}

// Context models that read only the byte history and a few plain globals: nestModel (:4107-4181), distanceModel
// (:4598-4612), indirectModel (:7548-7599). The caller marshals the globals they read: c4, f4, pos, buf(1..8); `level`
// sizes their ContextMaps. One instance of each per process (function-local statics).
int refp8_ctxmodel_step(int which, int level, int y_prev, int bpos, int c0, uint32_t c4, uint32_t f4, int pos,
                        const uint8_t* last, int nlast, int16_t* out) {
  paq8::level = level;
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  paq8::c4 = c4;
  paq8::f4 = f4;
  paq8::pos = pos;
  if (paq8::buf.size() == 0) paq8::buf.setsize(1 << 16);
  for (int i = 1; i <= nlast; ++i) paq8::buf[paq8::pos - i] = last[i - 1];
  if (which == 0) paq8::nestModel(*sink());
  else if (which == 1) paq8::distanceModel(*sink());
  else paq8::indirectModel(*sink());
  return drain(out);
}

// sparseModel (:4504-4535) and sparseModel1 (:4539-4596). g[] = the plain globals they read, in this order:
// c4, f4, x4, w4, tt, words, spaces, frstchar, spafdo; seenbefore / howmany are their arguments.
int refp8_sparse_step(int which, int level, int y_prev, int bpos, int c0, const uint32_t* g, int seenbefore, int howmany,
                      const uint8_t* last, int nlast, int16_t* out) {
  paq8::level = level;
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  paq8::c4 = g[0]; paq8::f4 = g[1]; paq8::x4 = g[2]; paq8::w4 = g[3]; paq8::tt = g[4];
  paq8::words = g[5]; paq8::spaces = g[6]; paq8::frstchar = g[7]; paq8::spafdo = g[8];
  if (paq8::buf.size() == 0) paq8::buf.setsize(1 << 16);
  for (int i = 1; i <= nlast; ++i) paq8::buf[paq8::pos - i] = last[i - 1];
  if (which == 0) paq8::sparseModel(*sink(), seenbefore, howmany);
  else paq8::sparseModel1(*sink(), seenbefore, howmany);
  return drain(out);
}

// MatchModel (:3520-3692): needs random access to the whole history. refp8_buf_reset / refp8_buf_push keep the
// reference's own buffer and `pos` the way Predictor::update does (buf[pos++] = byte, :8254).
void refp8_buf_reset(int log2size) {
  paq8::buf.setsize(1u << log2size);
  for (uint32_t i = 0; i < paq8::buf.size(); ++i) paq8::buf[i] = 0;
  paq8::pos = 0;
}
void refp8_buf_push(int byte) { paq8::buf[paq8::pos++] = (paq8::U8)byte; }
void* refp8_match_new(uint32_t size) { return new paq8::MatchModel(size); }
int refp8_match_step(void* h, int y_prev, int bpos, int c0, int16_t* out, int* nout, int* expected_byte) {
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  paq8::ModelStats st;
  memset(&st, 0, sizeof st);
  const int len = ((paq8::MatchModel*)h)->Predict(*sink(), paq8::buf, &st);
  *nout = drain(out);
  *expected_byte = st.Match.expectedByte;
  return len;
}

// SparseMatchModel (:3694-3843): over the same buffer as MatchModel (refp8_buf_reset / _push). sets[] receives the
// two mixer weight-set selectors (cumulative, as Mixer::set stores them).
void* refp8_sparsematch_new(uint64_t size) { return new paq8::SparseMatchModel(size); }
int refp8_sparsematch_step(void* h, int y_prev, int bpos, int c0, int16_t* out, int* nout, int* sets, int* nsets) {
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  const int len = ((paq8::SparseMatchModel*)h)->Predict(*sink(), paq8::buf, nullptr);
  *nout = drain(out);
  *nsets = drain_sets(sets);
  return len;
}

// picModel (:3844-3864) and recordModel1 (:4435-4474) over the reference's buffer (refp8_buf_reset / _push).
int refp8_small_step(int which, int y_prev, int bpos, int c0, uint32_t c4, uint32_t f4, uint32_t w5, int16_t* out) {
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  paq8::c4 = c4;
  paq8::f4 = f4;
  paq8::w5 = w5;
  if (which == 0) paq8::picModel(*sink());
  else paq8::recordModel1(*sink());
  return drain(out);
}

// recordModel (:4204-4433) over the reference's buffer. io[] in: blpos, grp0, filetype, Stats.Record,
// Stats.Match.length, Stats.Match.expectedByte; out: io[3] = Stats.Record as the model leaves it.
int refp8_record_step(int level, int y_prev, int bpos, int c0, uint32_t c4, uint32_t* io, int16_t* out, int* sets, int* nsets) {
  paq8::level = level;
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  paq8::c4 = c4;
  paq8::blpos = (int)io[0];
  paq8::grp0 = (paq8::U8)io[1];
  paq8::ModelStats st;
  memset(&st, 0, sizeof st);
  st.Record = io[3];
  st.Match.length = io[4];
  st.Match.expectedByte = (paq8::U8)io[5];
  paq8::recordModel(*sink(), (paq8::Filetype)io[2], &st);
  io[3] = st.Record;
  const int n = drain(out);
  *nsets = drain_sets(sets);
  return n;
}

// XMLModel (:7823-8096) over the reference's buffer; returns Stats.XML in *xml.
int refp8_xml_step(int level, int y_prev, int bpos, int c0, uint32_t c4, int16_t* out, uint32_t* xml) {
  paq8::level = level;
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  paq8::c4 = c4;
  paq8::ModelStats st;
  memset(&st, 0, sizeof st);
  paq8::XMLModel(*sink(), &st);
  *xml = st.XML;
  return drain(out);
}

// exeModel (:6560-7546): the x86 decoder's opcode tables as data, and the model itself (forced on, as contextModel2
// calls it) over the reference's buffer.
void refp8_exe_tables(uint8_t* out /* 4*256 + 32 + 4*256 + 32 + 19 + 8 + 1 */) {
  int n = 0;
  for (int i = 0; i < 256; ++i) out[n++] = paq8::Table1[i];
  for (int i = 0; i < 256; ++i) out[n++] = paq8::Table2[i];
  for (int i = 0; i < 256; ++i) out[n++] = paq8::Table3_38[i];
  for (int i = 0; i < 256; ++i) out[n++] = paq8::Table3_3A[i];
  for (int i = 0; i < 32; ++i) out[n++] = paq8::TableX[i];
  for (int i = 0; i < 256; ++i) out[n++] = paq8::TypeOp1[i];
  for (int i = 0; i < 256; ++i) out[n++] = paq8::TypeOp2[i];
  for (int i = 0; i < 256; ++i) out[n++] = paq8::TypeOp3_38[i];
  for (int i = 0; i < 256; ++i) out[n++] = paq8::TypeOp3_3A[i];
  for (int i = 0; i < 32; ++i) out[n++] = paq8::TypeOpX[i];
  for (int i = 0; i < 19; ++i) out[n++] = paq8::InvalidX64Ops[i];
  for (int i = 0; i < 8; ++i) out[n++] = paq8::X64Prefixes[i];
  out[n++] = (uint8_t)paq8::OP_GEN_BRANCH;
}
int refp8_exe_step(int level, int y_prev, int bpos, int c0, uint32_t c4, int blpos, int16_t* out, int* sets, int* nsets, uint32_t* x86) {
  paq8::level = level;
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  paq8::c4 = c4;
  paq8::blpos = blpos;
  paq8::ModelStats st;
  memset(&st, 0, sizeof st);
  paq8::exeModel(*sink(), true, &st);
  *x86 = st.x86_64;
  const int n = drain(out);
  *nsets = drain_sets(sets);
  return n;
}

// EnglishStemmer (:1764-2431) / FrenchStemmer (:2433-2822) / GermanStemmer (:2831-3004) on one word, letters added the
// way wordModel and TextModel add them (Word::operator+=). lang: 1 English, 2 French, 3 German.
int refp8_stem_word(int lang, const char* s, uint8_t* letters64, int* start_end, uint64_t* type_lang, uint64_t* hash4_after_stem,
                    uint64_t* hash4_gethashes) {
  static paq8::EnglishStemmer en;
  static paq8::FrenchStemmer fr;
  static paq8::GermanStemmer de;
  paq8::Stemmer* stemmer = lang == 2 ? (paq8::Stemmer*)&fr : lang == 3 ? (paq8::Stemmer*)&de : (paq8::Stemmer*)&en;
  paq8::Word w;
  w.Language = 0;  // not initialised by Word::Word()
  for (const char* p = s; *p; ++p) w += *p;
  const int r = stemmer->Stem(&w);
  memcpy(letters64, w.Letters, 64);
  start_end[0] = w.Start; start_end[1] = w.End;
  type_lang[0] = w.Type; type_lang[1] = w.Language;
  memcpy(hash4_after_stem, w.Hash, 32);
  w.GetHashes();
  memcpy(hash4_gethashes, w.Hash, 32);
  return r;
}
int refp8_en_stem_word(const char* s, uint8_t* letters64, int* start_end, uint64_t* type_lang, uint64_t* hash4_after_stem,
                       uint64_t* hash4_gethashes) {
  return refp8_stem_word(1, s, letters64, start_end, type_lang, hash4_after_stem, hash4_gethashes);
}

// the word-level globals are process-wide and other harness entries (refp8_sparse_step) set them as inputs
void refp8_word_globals_reset() {
  paq8::spaces = paq8::spacecount = paq8::words = paq8::wordcount = paq8::wordlen = paq8::wordlen1 = 0;
  paq8::frstchar = 0; paq8::spafdo = 0; paq8::col = 0;
}
// wordModel (:3873-4105) over the reference's buffer. g_out receives the word-level globals it maintains for other
// models: spaces, spacecount, words, wordcount, wordlen, wordlen1, frstchar, spafdo, col.
int refp8_word_step(int level, int y_prev, int bpos, int c0, uint32_t c4, uint32_t f4, uint32_t b3, int blpos, int16_t* out,
                    uint32_t* g_out) {
  paq8::level = level;
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  paq8::c4 = c4;
  paq8::f4 = f4;
  paq8::b3 = b3;
  paq8::blpos = blpos;
  paq8::wordModel(*sink());
  g_out[0] = paq8::spaces; g_out[1] = paq8::spacecount; g_out[2] = paq8::words; g_out[3] = paq8::wordcount; g_out[4] = paq8::wordlen;
  g_out[5] = paq8::wordlen1; g_out[6] = paq8::frstchar; g_out[7] = paq8::spafdo; g_out[8] = paq8::col;
  return drain(out);
}

// TextModel (:3006-3518) over the reference's buffer. sets: the eight mixer selectors; stats: ModelStats::Text.
void refp8_text_tables(uint8_t* out /* 254 + 128 */) {
  memcpy(out, paq8::AsciiGroupC0, 254);
  memcpy(out + 254, paq8::AsciiGroup, 128);
}
void* refp8_text_new(uint32_t size) { return new paq8::TextModel(size); }
int refp8_text_step(void* h, int y_prev, int bpos, int c0, int16_t* out, int* sets, int* nsets, uint32_t* stats) {
  paq8::y = y_prev;
  paq8::bpos = bpos;
  paq8::c0 = c0;
  paq8::grp0 = (bpos > 0) ? paq8::AsciiGroupC0[(1 << bpos) - 2 + (c0 & ((1 << bpos) - 1))] : 0;  // Predictor::update :8274
  paq8::ModelStats st;
  memset(&st, 0, sizeof st);
  ((paq8::TextModel*)h)->Predict(*sink(), paq8::buf, &st);
  stats[0] = st.Text.state; stats[1] = st.Text.lastPunct; stats[2] = st.Text.wordLength; stats[3] = st.Text.boolmask;
  stats[4] = st.Text.firstLetter; stats[5] = st.Text.mask;
  const int n = drain(out);
  *nsets = drain_sets(sets);
  return n;
}

// The whole model as cmix drives it: paq8::Predictor (:8208-8362) behind PAQ8::Perceive / PAQ8::Predict (:8366-8385).
// contextModel2 keeps its sub-models in function-local statics sized by the level at the first call: ONE predictor per
// loaded copy of this library, and none of the single-model entries above may be used in the same copy.
void* refp8_predictor_new(int level) {
  paq8::level = level;
  paq8::buf.setsize(paq8::MEM() * 8);
  return new paq8::Predictor();
}
// state injection (round 6's wrap / threshold audit): paq8's byte position `pos` (:167) -- the index into the 2^30-byte history ring at level 11 (Buf :169-186,
// every read is (pos - i) & (size - 1)) and the value the match models and the detectors store and compare. A stream reaches 2^30 after 1 GB.
void refp8_set_pos(int pos) { paq8::pos = pos; }
int refp8_predictor_update(void* h, int bit, float* out1591) {
  paq8::y = bit;
  ((paq8::Predictor*)h)->update();
  for (int i = 0; i < 1591; ++i) out1591[i] = paq8::model_predictions[i];
  return ((paq8::Predictor*)h)->p();
}

uint64_t refp8_hash2(uint64_t a, uint64_t b) { return paq8::hash(a, b); }
uint64_t refp8_hash3(uint64_t a, uint64_t b, uint64_t c) { return paq8::hash(a, b, c); }
uint64_t refp8_hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d) { return paq8::hash(a, b, c, d); }
uint64_t refp8_hash5(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e) { return paq8::hash(a, b, c, d, e); }
uint64_t refp8_hash6(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e, uint64_t f) { return paq8::hash(a, b, c, d, e, f); }
uint64_t refp8_combine64(uint64_t seed, uint64_t x) { return paq8::combine64(seed, x); }
uint32_t refp8_finalize64(uint64_t h, int bits) { return paq8::finalize64(h, bits); }
uint64_t refp8_checksum64(uint64_t h, int hashbits, int checksumbits) { return paq8::checksum64(h, hashbits, checksumbits); }

}  // extern "C"
