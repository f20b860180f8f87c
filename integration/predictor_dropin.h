// integration/predictor_dropin.h -- the drop-in `class Predictor` (src/predictor.h:17-22) for a build that links NO
// reference model object: the four members forward to the C ABI, and compression runs in the library's look-ahead mode
// (cmx_stage_input): every model family is a device stage and Predict() / Perceive() pop and check the probabilities of
// chunks that have already left the mixing network. src/coder/*, src/preprocess/* and src/runner.cpp compile against
// it unmodified; the one thing the reference does not have is the look-ahead itself, so a maintainer adds ONE line to
// RunCompression (runner.cpp:205-206), after the pretraining call and before Compress():
//
//       Predictor p(vocab);
//       if (enable_preprocess) preprocessor::Pretrain(&p, dictionary);
//   +   p.StageInput(&temp_in, temp_bytes);          // the bytes Compress() is about to code
//       Compress(temp_bytes, &temp_in, &data_out, output_bytes, &p);
//
// (oracle/Makefile target `dropin_engine` applies exactly that line to a scratch copy of runner.cpp at build time.)
// Decompression cannot look ahead: without StageInput the handle builds the per-bit stages, which take the fxcm / paq8
// columns from the caller -- integration/predictor.h is the shim for that direction.
#ifndef PREDICTOR_H
#define PREDICTOR_H

#include <cstdio>
#include <cstdlib>
#include <istream>
#include <vector>

#include "cmix_amd.h"

extern char* dictionary_path;  // runner.cpp:17

// SURVEY.md 8f-3 without touching main(): the vocabulary-independent stages the engine needs whatever the input is (mixing network,
// paq8 stage: ~12 GB of tables) start being built on a library thread when the program starts, i.e. while runner.cpp parses its
// arguments and preprocesses the input into the temp file; the Predictor constructed afterwards adopts them (cmx_prewarm). The fxcm
// stage waits for the dictionary path, which only main() knows. CMIX_NO_PREWARM=1 switches this off (decompression does not use
// the chunk pipeline's stages).
inline void cmx_dropin_prewarm_once() {
  static const int once = []() {
    if (!getenv("CMIX_NO_PREWARM")) { const char* dev = getenv("CMIX_DEVICE"); (void)cmx_prewarm(dev ? atoi(dev) : 0, NULL, 0); }
    return 0;
  }();
  (void)once;
}
namespace { struct CmxAtStart { CmxAtStart() { cmx_dropin_prewarm_once(); } } cmx_at_start; }

class Predictor {
 public:
  explicit Predictor(const std::vector<bool>& vocab) {
    unsigned char v[256];
    for (int i = 0; i < 256; ++i) v[i] = vocab[i] ? 1 : 0;
    const char* dev = getenv("CMIX_DEVICE");
    h_ = cmx_create(v, dictionary_path, dev ? atoi(dev) : 0);
    if (!h_) Die();
  }
  ~Predictor() { cmx_destroy(h_); }
  Predictor(const Predictor&) = delete;
  Predictor& operator=(const Predictor&) = delete;

  // The next n bytes of *is, handed to the engine ahead of the coder; the stream is left where it was.
  void StageInput(std::istream* is, unsigned long long n) {
    const std::istream::pos_type at = is->tellg();
    std::vector<unsigned char> buf(1 << 20);
    while (n) {
      const size_t m = n < buf.size() ? (size_t)n : buf.size();
      is->read((char*)buf.data(), (std::streamsize)m);
      if ((size_t)is->gcount() != m) { fprintf(stderr, "cmix_amd shim: short read while staging the input\n"); abort(); }
      if (cmx_stage_input(h_, buf.data(), m)) Die();
      n -= m;
    }
    if (cmx_stage_input(h_, NULL, 0)) Die();   // end of input: the ragged last chunk goes now
    is->clear();
    is->seekg(at);
  }

  float Predict() {  // predictor.cpp:361-419
    const float p = cmx_predict(h_);
    if (p < 0) Die();
    return p;
  }
  void Perceive(int bit) {  // predictor.cpp:421-469
    if (cmx_perceive(h_, bit)) Die();
  }
  void Pretrain(int bit) {  // predictor.cpp:471-487
    if (cmx_pretrain(h_, bit)) Die();
  }

 private:
  static void Die() {  // the reference has no error path: report and stop
    fprintf(stderr, "cmix_amd: %s\n", cmx_last_error());
    abort();
  }
  cmx_t* h_;
};

#endif
