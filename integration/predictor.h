// integration/predictor.h -- the shim of INTEGRATION.md section 1, as a compilable file: a `class Predictor` with
// exactly the reference's four public members (src/predictor.h:17-22) that forwards to the C ABI of libcmixamd.so.
// src/coder/*, src/preprocess/* and src/runner.cpp compile against it unmodified (build: see integration/README).
//
// The two vendored model families whose device stages only exist in chunk (look-ahead) form (fxcm, paq8) are owned here -- this
// is the strict per-bit surface a DEcoder needs; a compressor uses integration/predictor_engine.h instead --, constructed in the
// reference's order (predictor.cpp:28-36: Bracket, FXCM, PAQ8, ...), and their outputs are handed to the library
// per bit; everything else behind Predict()/Perceive() runs in the library (MI355X + the PPMd host stage).
#ifndef PREDICTOR_H
#define PREDICTOR_H

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <valarray>
#include <vector>

#include "cmix_amd.h"
#include "models/fxcmv1.h"
#include "models/paq8.h"

extern char* dictionary_path;  // runner.cpp:17
extern int lstmpr, lstmex;     // predictor.cpp:359 in the reference; integration/predictor_shim.cpp here

class Predictor {
 public:
  explicit Predictor(const std::vector<bool>& vocab) {
    unsigned char v[256];
    for (int i = 0; i < 256; ++i) v[i] = vocab[i] ? 1 : 0;
    const char* dev = getenv("CMIX_DEVICE");
    h_ = cmx_create(v, dictionary_path, dev ? atoi(dev) : 0);
    if (!h_) Die();
    fxcm_.reset(new FXCM());    // predictor.cpp:77-82
    paq8_.reset(new PAQ8(11));  // predictor.cpp:84-88
  }
  ~Predictor() { cmx_destroy(h_); }
  Predictor(const Predictor&) = delete;
  Predictor& operator=(const Predictor&) = delete;

  float Predict() {  // predictor.cpp:361-419
    float cols[2022];  // layer-0 columns 3..2024
    const std::valarray<float>& a = fxcm_->Predict();
    const std::valarray<float>& b = paq8_->Predict();
    if (a.size() + b.size() != 2022) { fprintf(stderr, "cmix_amd shim: unexpected model widths\n"); abort(); }
    for (size_t i = 0; i < a.size(); ++i) cols[i] = a[i];
    for (size_t i = 0; i < b.size(); ++i) cols[a.size() + i] = b[i];
    if (cmx_set_model_outputs(h_, cols)) Die();
    const float p = cmx_predict(h_);
    if (p < 0) Die();
    return p;
  }

  void Perceive(int bit) {  // predictor.cpp:421-469
    if (cmx_perceive(h_, bit)) Die();  // enqueues the device side of :422-461 and returns
    paq8_->Perceive(bit);              // host model, overlapped with it (the models are independent of each other)
    if (cmx_get_lstm_hint(h_, &lstmpr, &lstmex)) Die();  // :462-465 -- waits for the device
    fxcm_->Perceive(bit);   // fxcm last (:466)
  }

  void Pretrain(int bit) {  // predictor.cpp:471-487
    fxcm_->Predict();
    paq8_->Predict();
    fxcm_->Perceive(bit);
    paq8_->Perceive(bit);
    if (cmx_pretrain(h_, bit)) Die();
  }

 private:
  static void Die() {  // the reference has no error path: report and stop
    fprintf(stderr, "cmix_amd: %s\n", cmx_last_error());
    abort();
  }
  cmx_t* h_;
  std::unique_ptr<Model> fxcm_, paq8_;  // as the reference holds them (predictor.h: models_), deleted through ~Model
};

#endif
