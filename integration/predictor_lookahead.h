// integration/predictor_lookahead.h -- the look-ahead counterpart of integration/predictor.h for COMPRESSION, where
// every byte is known before it is coded (runner.cpp:101-119): `class Predictor` here only has to serve
// preprocessor::Pretrain (the one caller besides the coder, preprocessor.cpp:37-69); the coding itself goes a chunk
// at a time through cmx_pipeline_begin / _hints / _finish (integration/compress_lookahead.cpp). The two vendored
// model families are owned here and run on host threads (the hybrid path that preceded the whole-engine compressor,
// integration/predictor_engine.h): paq8 always; fxcm unless
// CMX_FXCM_DEVICE=1 asks for the device stage (cmx_pipeline_enable_fxcm), in which case no host fxcm object exists.
#ifndef PREDICTOR_H
#define PREDICTOR_H

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <valarray>
#include <vector>

#include "cmix_amd.h"
#include "models/fxcmv1.h"
#include "models/paq8.h"

extern char* dictionary_path;  // runner.cpp:17 in the reference; compress_lookahead.cpp here
extern int lstmpr, lstmex;     // predictor.cpp:359

class Predictor {
 public:
  explicit Predictor(const std::vector<bool>& vocab, size_t chunk_bytes) : chunk_(chunk_bytes) {
    for (int i = 0; i < 256; ++i) vocab_[i] = vocab[i] ? 1 : 0;
    const char* dev = getenv("CMIX_DEVICE");
    device_ = dev ? atoi(dev) : 0;
    pipe_ = cmx_pipeline_create(vocab_, device_, chunk_);
    if (!pipe_) Die();
    const char* fxd = getenv("CMX_FXCM_DEVICE");
    fxcm_on_device_ = fxd && fxd[0] == '1';
    if (fxcm_on_device_) { if (cmx_pipeline_enable_fxcm(pipe_, dictionary_path)) Die(); }
    else fxcm_.reset(new FXCM());    // predictor.cpp:77-82
    paq8_.reset(new PAQ8(11));  // predictor.cpp:84-88
  }
  ~Predictor() { cmx_pipeline_destroy(pipe_); }
  Predictor(const Predictor&) = delete;
  Predictor& operator=(const Predictor&) = delete;

  // predictor.cpp:471-487: the host models learn bit by bit; the device's share is trained in one batch before the
  // first chunk (FlushPretrain), which is equivalent because nothing downstream is trained during pretraining.
  void Pretrain(int bit) {
    if (fxcm_) fxcm_->Predict();
    paq8_->Predict();
    if (fxcm_) fxcm_->Perceive(bit);   // on the device: cmx_pipeline_pretrain (FlushPretrain) covers it
    paq8_->Perceive(bit);
    pre_partial_ = (pre_partial_ << 1) | (bit ? 1u : 0u);
    if (++pre_j_ == 8) { pre_.push_back((uint8_t)pre_partial_); pre_j_ = 0; pre_partial_ = 0; }
  }
  void FlushPretrain() {
    if (!pre_.empty() && cmx_pipeline_pretrain(pipe_, pre_.data(), pre_.size())) Die();
    pre_.clear();
  }
  float Predict() { fprintf(stderr, "look-ahead build: coding goes through CompressLookahead()\n"); abort(); }
  void Perceive(int) { Predict(); }

  cmx_pipeline_t* pipe() { return pipe_; }
  Model* fxcm() { return fxcm_.get(); }
  bool fxcm_on_device() const { return fxcm_on_device_; }
  Model* paq8() { return paq8_.get(); }
  int device() const { return device_; }
  size_t chunk() const { return chunk_; }
  static void Die() {
    fprintf(stderr, "cmix_amd: %s\n", cmx_last_error());
    abort();
  }

 private:
  size_t chunk_;
  int device_ = 0;
  bool fxcm_on_device_ = false;
  unsigned char vocab_[256];
  cmx_pipeline_t* pipe_ = nullptr;
  std::unique_ptr<Model> fxcm_, paq8_;
  std::vector<uint8_t> pre_;
  int pre_j_ = 0;
  unsigned pre_partial_ = 0;
};

#endif
