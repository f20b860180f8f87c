// oracle/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin C-ABI window onto the *unmodified* reference Predictor compiled from
// /root/reference (see oracle/Makefile, target _ref/libcmixref.so).  It drives
// the real Predictor::Predict()/Perceive() (reference src/predictor.cpp:361,
// :421) and copies out the state those calls leave behind, so that the CPU
// restatement in oracle/*.c and the HIP engine can be compared stage by stage:
//   raw model probabilities, layer-0/1/2 stretched inputs, every mixer's
//   selector key and output, final probability, ContextManager registers,
//   PPMd / LSTM byte distributions.
// No reference logic is re-implemented here; only member access is widened.
//
// One Predictor per process: paq8/fxcm keep state in namespace globals
// (SURVEY.md section 5), so ref_create() may be called once.

#include <algorithm>
#include <array>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <numeric>
#include <set>
#include <string>
#include <unordered_map>
#include <valarray>
#include <vector>
#include <math.h>

#define private public
#define protected public
#include "predictor.h"
#include "mixer/lstm.h"
#include "mixer/byte-mixer.h"
#include "models/byte-model.h"
#include "models/ppmd.h"
#include "models/match.h"
#include "contexts/interval.h"
#include "contexts/interval-hash.h"
#include "states/nonstationary.h"
#include "preprocess/preprocessor.h"
#undef private
#undef protected

// The reference defines this in runner.cpp (which is not linked into the .so);
// fxcm reads it (reference src/models/fxcmv1.cpp:412-428).
char* dictionary_path = NULL;

namespace {
Predictor* g_p = nullptr;
std::vector<bool> g_vocab(256, true);
}

extern "C" {

// Layout constants so the Python side can size its buffers.
int ref_num_inputs(int layer) { return g_p ? (int)g_p->layers_[layer]->Inputs().size() : -1; }
int ref_num_mixers(int layer) { return g_p ? (int)g_p->mixers_[layer].size() : -1; }
int ref_num_models(void) { return g_p ? (int)g_p->models_.size() : -1; }
int ref_num_contexts(void) { return g_p ? (int)g_p->manager_.contexts_.size() : -1; }
int ref_num_bit_contexts(void) { return g_p ? (int)g_p->manager_.bit_contexts_.size() : -1; }
int ref_auxiliary(int i) { return g_p ? (int)g_p->auxiliary_[i] : -1; }

int ref_create(const uint8_t* vocab256, const char* dict_path) {
  if (g_p) return -1;
  for (int i = 0; i < 256; ++i) g_vocab[i] = vocab256[i] != 0;
  if (dict_path && dict_path[0]) dictionary_path = strdup(dict_path);
  g_p = new Predictor(g_vocab);
  return 0;
}

float ref_predict(void) { return g_p->Predict(); }
void ref_perceive(int bit) { g_p->Perceive(bit); }
void ref_pretrain(int bit) { g_p->Pretrain(bit); }

// ---- state readout, valid after ref_predict() and before ref_perceive() ----

// Raw (pre-stretch) outputs of every model in layer-0 order (2078 floats).
int ref_get_model_probs(float* out) {
  int n = 0;
  for (auto& m : g_p->models_) {
    // PAQ8/FXCM override Predict() with a side-effect-free accessor; all other
    // models keep the value produced by the last Predict() in outputs_.
    const std::valarray<float>* o = &m->outputs_;
    if (m->NumOutputs() != o->size()) o = &m->Predict();
    for (size_t j = 0; j < o->size(); ++j) out[n++] = (*o)[j];
  }
  for (auto& m : g_p->byte_models_) out[n++] = m->outputs_[0];
  for (auto& m : g_p->byte_mixers_) out[n++] = m->outputs_[0];
  return n;
}

int ref_get_layer_inputs(int layer, float* out) {
  const std::valarray<float>& in = g_p->layers_[layer]->Inputs();
  for (size_t i = 0; i < in.size(); ++i) out[i] = in[i];
  return (int)in.size();
}

// Selector key (as the Mixer sees it: the referenced 64-bit context) and the
// mixer's own stretch-domain output p_ for every mixer of a layer.
int ref_get_mixers(int layer, uint64_t* ctx, float* out) {
  auto& v = g_p->mixers_[layer];
  for (size_t k = 0; k < v.size(); ++k) {
    ctx[k] = v[k]->context_;
    out[k] = v[k]->p_;
  }
  return (int)v.size();
}

uint64_t ref_mixer_steps(int layer, int k) { return g_p->mixers_[layer][k]->steps_; }
uint64_t ref_mixer_rows(int layer, int k) { return g_p->mixers_[layer][k]->context_map_.size(); }
float ref_mixer_lr(int layer, int k) { return g_p->mixers_[layer][k]->learning_rate_; }

// Weight row currently selected by mixer (layer,k): weights then extra weights.
int ref_get_mixer_row(int layer, int k, float* w, float* ew, uint64_t* row_steps) {
  Mixer* m = g_p->mixers_[layer][k].get();
  unsigned int key = (unsigned int)m->context_;
  auto it = m->context_map_.find(key);
  if (it == m->context_map_.end() || !it->second) {
    it = m->context_map_.find(0xDEADBEEF);
    if (it == m->context_map_.end() || !it->second) return -1;
  }
  ContextData* d = it->second.get();
  for (size_t i = 0; i < d->weights.size(); ++i) w[i] = d->weights[i];
  for (size_t i = 0; i < d->extra_weights.size(); ++i) ew[i] = d->extra_weights[i];
  *row_steps = d->steps;
  return (int)d->weights.size();
}

// ContextManager registers (reference src/context-manager.h:21-27).
// regs: bit_context, long_bit_context, zero, history_pos, line_break,
//       longest_match, auxiliary_context, wrt_context, wrt_state,
//       recent_bytes[0..7], words[0..7]            => 25 values
int ref_get_manager(uint64_t* regs, uint64_t* ctx, uint64_t* bit_ctx) {
  ContextManager& m = g_p->manager_;
  int n = 0;
  regs[n++] = m.bit_context_;
  regs[n++] = m.long_bit_context_;
  regs[n++] = m.zero_context_;
  regs[n++] = m.history_pos_;
  regs[n++] = m.line_break_;
  regs[n++] = m.longest_match_;
  regs[n++] = m.auxiliary_context_;
  regs[n++] = m.wrt_context_;
  regs[n++] = m.wrt_state_;
  for (int i = 0; i < 8; ++i) regs[n++] = m.recent_bytes_[i];
  for (int i = 0; i < 8; ++i) regs[n++] = m.words_[i];
  for (size_t i = 0; i < m.contexts_.size(); ++i) ctx[i] = m.contexts_[i]->context_;
  for (size_t i = 0; i < m.bit_contexts_.size(); ++i) bit_ctx[i] = m.bit_contexts_[i]->context_;
  return n;
}

uint64_t ref_context_size(int i) { return g_p->manager_.contexts_[i]->size_; }

// 256-way byte distributions: which = 0 -> PPMd (byte_models_[0]),
// 1 -> LSTM byte mixer (byte_mixers_[0]), 2 -> Bracket (models_[0]).
int ref_get_byte_probs(int which, float* out, int* top_bot_ex) {
  ByteModel* b = nullptr;
  if (which == 0) b = g_p->byte_models_[0].get();
  else if (which == 1) b = g_p->byte_mixers_[0].get();
  else b = dynamic_cast<ByteModel*>(g_p->models_[0].get());
  if (!b) return -1;
  for (int i = 0; i < 256; ++i) out[i] = b->probs_[i];
  top_bot_ex[0] = b->top_;
  top_bot_ex[1] = b->bot_;
  top_bot_ex[2] = b->ex;
  return 256;
}

// LSTM internals for stage-level parity (sizes: hidden = 2*cells+1).
int ref_lstm_dims(int* dims) {
  Lstm* l = g_p->byte_mixers_[0]->lstm_.get();
  dims[0] = l->input_size_;
  dims[1] = l->output_size_;
  dims[2] = l->num_cells_;
  dims[3] = (int)l->layers_.size();
  dims[4] = l->horizon_;
  dims[5] = l->epoch_;
  return 6;
}
int ref_lstm_hidden(float* out) {
  Lstm* l = g_p->byte_mixers_[0]->lstm_.get();
  for (size_t i = 0; i < l->hidden_.size(); ++i) out[i] = l->hidden_[i];
  return (int)l->hidden_.size();
}
// gate: 0 forget, 1 input node, 2 output gate. Copies weights_[cell][*].
int ref_lstm_gate_weights(int layer, int gate, float* out) {
  LstmLayer* L = g_p->byte_mixers_[0]->lstm_->layers_[layer].get();
  NeuronLayer* n = gate == 0 ? &L->forget_gate_ : gate == 1 ? &L->input_node_ : &L->output_gate_;
  size_t k = 0;
  for (size_t i = 0; i < n->weights_.size(); ++i)
    for (size_t j = 0; j < n->weights_[i].size(); ++j) out[k++] = n->weights_[i][j];
  return (int)k;
}
int ref_lstm_gate_row_len(int layer) {
  LstmLayer* L = g_p->byte_mixers_[0]->lstm_->layers_[layer].get();
  return (int)L->forget_gate_.weights_[0].size();
}
// Output layer weights for time slot `epoch` (V x hidden).
int ref_lstm_output_layer(int epoch, float* out) {
  Lstm* l = g_p->byte_mixers_[0]->lstm_.get();
  size_t k = 0;
  for (size_t i = 0; i < l->output_layer_[epoch].size(); ++i)
    for (size_t j = 0; j < l->output_layer_[epoch][i].size(); ++j) out[k++] = l->output_layer_[epoch][i][j];
  return (int)k;
}

// Reference DATA tables (dumped by scripts/gen_ref_tables.py, never transcribed by hand):
// the Nonstationary transition table (src/states/nonstationary.cpp:3) ...
int ref_nonstationary_table(uint8_t* out512) {
  for (int s = 0; s < 256; ++s)
    for (int b = 0; b < 2; ++b) out512[2 * s + b] = (uint8_t)g_p->manager_.nonstationary_.Next(s, b);
  return 512;
}
// ... and the byte-class map an Interval / IntervalHash context was built with (predictor.cpp:230-304).
int ref_interval_map(int ctx_index, int* out256) {
  Context* c = g_p->manager_.contexts_[ctx_index].get();
  const std::vector<int>* m = nullptr;
  if (Interval* i = dynamic_cast<Interval*>(c)) m = &i->map_;
  else if (IntervalHash* h = dynamic_cast<IntervalHash*>(c)) m = &h->map_;
  if (!m) return -1;
  for (int i = 0; i < 256; ++i) out256[i] = (*m)[i];
  return 256;
}

// A stand-alone instance of the reference's PPMd byte model with an arbitrary arena size (the Predictor's own
// is fixed at 14000 MB, predictor.cpp:101): a small arena makes the memory-exhaustion path -- cut-off and
// restore, ppmd.cpp:562-640,686-727 -- run every few tens of KB, so it can be pinned by short traces.
namespace {
PPMD::PPMD* g_ppmd = nullptr;
unsigned int g_ppmd_byte = 0;
std::vector<bool> g_ppmd_vocab(256, true);
}
int ref_ppmd_create(int order, int memory_mb, const uint8_t* vocab256) {
  if (g_ppmd) return -1;
  for (int i = 0; i < 256; ++i) g_ppmd_vocab[i] = vocab256[i] != 0;
  g_ppmd = new PPMD::PPMD(order, memory_mb, g_ppmd_byte, g_ppmd_vocab);
  return 0;
}
int ref_ppmd_update(int byte, float* out256) {
  g_ppmd_byte = (unsigned int)byte;
  g_ppmd->ByteUpdate();
  for (int i = 0; i < 256; ++i) out256[i] = g_ppmd->probs_[i];
  return 0;
}

// ---- state injection (round 6: wrap / threshold regression traces, tests/golden/make_wrap_traces.py) ----------------------------------
// Counters that only reach their interesting values hundreds of megabytes into a stream are PLACED there; everything that then happens
// is the unmodified reference's own code. Only member access is widened, as everywhere in this file.
// Mixer::steps_ of all 47 mixers (they tick together, mixer.cpp:61): the argument of the decay schedule's pow() (mixer.cpp:58).
int ref_debug_set_mixer_steps(uint64_t steps) {
  int n = 0;
  for (auto& layer : g_p->mixers_)
    for (auto& m : layer) { m->steps_ = steps; ++n; }
  return n;
}
// ContextManager::history_pos_ (the write position in the 100 MB ring, context-manager.cpp:24-27) and every Match model's own history_pos_
// (bytes seen so far, match.cpp:43-46: NOT reduced modulo the ring -- the two agree until the ring wraps): as after `pos` bytes of a stream
// whose last n bytes were `tail` (written to the ring in front of pos). Returns the number of Match models.
int ref_debug_set_history(uint64_t pos, const uint8_t* tail, uint64_t n) {
  ContextManager& m = g_p->manager_;
  const uint64_t size = m.history_.size();
  for (uint64_t i = 0; i < n; ++i) m.history_[(pos - n + i) % size] = tail[i];
  m.history_pos_ = pos % size;
  int k = 0;
  for (auto& md : g_p->models_)
    if (Match* mm = dynamic_cast<Match*>(md.get())) { mm->history_pos_ = pos; ++k; }
  return k;
}
uint64_t ref_history_size(void) { return g_p->manager_.history_.size(); }

// libm probes: the exact host functions the reference's float path resolves to,
// exported so tests can compare the device re-implementations against the very
// same glibc variant (ifunc-selected) that the oracle process uses.
float ref_logistic(float x) { return Sigmoid::Logistic(x); }
float ref_logit(float p) { return g_p->sigmoid_.Logit(p); }

// The reference's preprocessor (src/preprocess/preprocessor.cpp:568 Encode) on a file: the block-framed stream the predictor codes
// (what runner.cpp writes to <out>.cmix.temp), for fixtures whose block types only the detector can produce (IMAGE*, AUDIO, JPEG).
int ref_preprocess_encode(const char* in_path, const char* out_path, const char* temp_path) {
  FILE* in = fopen(in_path, "rb");
  if (!in) return -1;
  fseek(in, 0, SEEK_END);
  const unsigned long long n = (unsigned long long)ftell(in);
  fseek(in, 0, SEEK_SET);
  FILE* out = fopen(out_path, "wb");
  if (!out) { fclose(in); return -2; }
  preprocessor::Encode(in, out, false, n, std::string(temp_path), NULL);
  fclose(in);
  fclose(out);
  return 0;
}

}  // extern "C"
