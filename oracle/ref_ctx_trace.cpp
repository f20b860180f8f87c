// oracle/ref_ctx_trace.cpp -- TEST INFRASTRUCTURE (never linked into the product): the UNMODIFIED reference Predictor with its three heavy members
// replaced by constant stand-ins -- PAQ8 and FXCM return 0.5 in every column, the Lstm a uniform distribution (their member functions are DEFINED here
// instead of linking paq8.o / fxcmv1.o / lstm.o / lstm-layer.o; predictor.o, the context manager, every context, Direct / DirectHash / Indirect / Match /
// Bracket, PPMD, Mixer, SSE are the reference's own objects, oracle/Makefile) -- so that the 54 columns of the small models (0..2, 2025..2075), PPMd's
// column 2076 and the 47 mixer selectors, none of which reads another model's output, can be recorded over a LONG stream at a tenth of the full
// reference's cost. (Not valid here: the auxiliary-context selector, which averages stand-in columns; the final probability.)
// Output per 64 KB block: position, then ref_long_trace.cpp's 131 group digests (only groups 127 and 128 are pure small-model columns), then one digest
// per small-model column (55: columns 0, 1, 2, 2025..2076) and one per selector (47):
//   column digest = sum over the block's bits t of (bits(p[t]) + 1) * B[t mod 2^19];  selector digest = sum of (context[t] + 1) * B[t mod 2^19]   (mod 2^64)
// usage: ref_ctx_trace stream.bin vocab256.bin out.txt
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define private public
#define protected public
#include "predictor.h"
#include "models/paq8.h"
#include "models/fxcmv1.h"
#include "mixer/lstm.h"
#undef private
#undef protected

namespace paq8 { class Predictor {}; }
namespace fxcmv1 { class Predictor {}; }
PAQ8::PAQ8(int) { outputs_.resize(1591, 0.5f); }
const std::valarray<float>& PAQ8::Predict() { return outputs_; }
unsigned int PAQ8::NumOutputs() { return 1591; }
void PAQ8::Perceive(int) {}
FXCM::FXCM() { outputs_.resize(431, 0.5f); }
const std::valarray<float>& FXCM::Predict() { return outputs_; }
unsigned int FXCM::NumOutputs() { return 431; }
void FXCM::Perceive(int) {}
#ifndef REAL_LSTM   // -DREAL_LSTM: link the reference's lstm.o / lstm-layer.o instead (column 2077 and with it group 129 become valid; 2 ms per byte more)
Lstm::Lstm(unsigned int, unsigned int output_size, unsigned int, unsigned int, int, float, float) : output_(std::valarray<float>(1.0f / output_size, output_size), 1) {}
Lstm::~Lstm() {}
std::valarray<float>& Lstm::Perceive(unsigned int) { return output_[0]; }
std::valarray<float>& Lstm::Predict(unsigned int) { return output_[0]; }
void Lstm::SetInput(const std::valarray<float>&) {}
#endif

#ifdef WITH_ORACLE_MIXNET   // -DWITH_ORACLE_MIXNET -L_build -lcmixoracle: every row and selector set is also fed to the oracle's restatement of the final
extern "C" {                // mixing network + SSE (oracle/mixnet.c), whose result must be the float Predictor::Predict() returned -- the mixers'
#include "cmix_oracle.h"    // long-stream behaviour (row cap, decay schedule, periodic shrink, SSE tables) against the reference's own Mixer / SSE objects
}
#endif
static uint64_t splitmix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  std::vector<uint8_t> s; { uint8_t buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) s.insert(s.end(), buf, buf + k); } fclose(f);
  uint8_t vb[256]; f = fopen(argv[2], "rb"); if (!f || fread(vb, 1, 256, f) != 256) return 4; fclose(f);
  FILE* out = fopen(argv[3], "w"); if (!out) return 5;
  std::vector<bool> vocab(256); for (int i = 0; i < 256; ++i) vocab[i] = vb[i] != 0;
  Predictor* P = new Predictor(vocab);
  static uint64_t A[2078], B[1 << 19];
  for (int c = 0; c < 2078; ++c) A[c] = splitmix64((uint64_t)c) | 1ull;
  for (int i = 0; i < (1 << 19); ++i) B[i] = splitmix64(0x1000000ull + (uint64_t)i) | 1ull;
  int cols[55]; for (int i = 0; i < 3; ++i) cols[i] = i; for (int i = 0; i < 52; ++i) cols[3 + i] = 2025 + i;
  static uint64_t h[131], hc[55], hs[47];
  static float probs[2078];
  const size_t nbits = s.size() * 8;
#ifdef WITH_ORACLE_MIXNET
  orc_mixnet* om = orc_mixnet_create();
  uint64_t mix_bad = 0; long long mix_first = -1;
#endif
  for (size_t t = 0; t < nbits; ++t) {
    const float p_ref = P->Predict();
    int n = 0;   // (as ref_harness.cpp's ref_get_model_probs)
    for (auto& m : P->models_) { const std::valarray<float>* o = &m->outputs_; if (m->NumOutputs() != o->size()) o = &m->Predict(); for (size_t j = 0; j < o->size(); ++j) probs[n++] = (*o)[j]; }
    for (auto& m : P->byte_models_) probs[n++] = m->outputs_[0];
    for (auto& m : P->byte_mixers_) probs[n++] = m->outputs_[0];
    if (n != 2078) return 7;
    const uint64_t b = B[t & ((1u << 19) - 1)];
    uint64_t g[130]; for (int k = 0; k < 130; ++k) g[k] = 0;
    for (int c = 0; c < 2078; ++c) { uint32_t u; memcpy(&u, &probs[c], 4); g[c >> 4] += ((uint64_t)u + 1ull) * A[c]; }
    for (int k = 0; k < 130; ++k) h[k] += g[k] * b;
    for (int i = 0; i < 55; ++i) { uint32_t u; memcpy(&u, &probs[cols[i]], 4); hc[i] += ((uint64_t)u + 1ull) * b; }
    int k = 0;
    for (auto& layer : P->mixers_) for (auto& m : layer) { if (k < 47) hs[k] += ((uint64_t)m->context_ + 1ull) * b; ++k; }
    if (k != 47) return 8;
#ifdef WITH_ORACLE_MIXNET
    {
      uint64_t sel[47]; int q = 0;
      for (auto& layer : P->mixers_) for (auto& m : layer) sel[q++] = (uint64_t)m->context_;
      const float p_orc = orc_mixnet_step(om, probs, sel, (s[t >> 3] >> (7 - (t & 7))) & 1, nullptr);
      if (memcmp(&p_orc, &p_ref, 4) != 0) {
        if (mix_first < 0) { mix_first = (long long)t; fprintf(stderr, "oracle mixing network != reference first at bit %zu (byte %zu): %.9g vs %.9g\n", t, t >> 3, p_orc, p_ref); }
        ++mix_bad;
      }
    }
#endif
    P->Perceive((s[t >> 3] >> (7 - (t & 7))) & 1);
    if (((t + 1) & ((1u << 19) - 1)) == 0 || t + 1 == nbits) {
      fprintf(out, "%zu", (t + 1) >> 3);
      for (int q = 0; q < 131; ++q) { fprintf(out, " %016llx", (unsigned long long)h[q]); h[q] = 0; }
      for (int q = 0; q < 55; ++q) { fprintf(out, " %016llx", (unsigned long long)hc[q]); hc[q] = 0; }
      for (int q = 0; q < 47; ++q) { fprintf(out, " %016llx", (unsigned long long)hs[q]); hs[q] = 0; }
#ifdef WITH_ORACLE_MIXNET
      fprintf(out, " mixnet_bits_differing_so_far %llu first %lld", (unsigned long long)mix_bad, mix_first);
#endif
      fprintf(out, "\n"); fflush(out);
    }
  }
  fclose(out);
  return 0;
}
