// integration/predictor_engine.h -- `class Predictor` for COMPRESSION with the whole engine on the device: every model
// family of src/predictor.cpp:77-102 has a stage behind cmx_pipeline_* (contexts + small models, PPMd host stage, LSTM,
// fxcm, paq8), so this shim owns NO reference model object and the build links none (oracle/Makefile target `engine`:
// the reference's preprocessor + integration/compress_engine.cpp + libcmixamd.so). It only has to serve
// preprocessor::Pretrain (preprocessor.cpp:37-69), which feeds the dictionary bit by bit: the bytes are collected and
// handed to cmx_pipeline_pretrain in one batch before the first chunk (nothing downstream learns during pretraining,
// predictor.cpp:471-487). Coding goes a chunk at a time through cmx_pipeline_submit (compress_engine.cpp).
#ifndef PREDICTOR_H
#define PREDICTOR_H

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "cmix_amd.h"

extern char* dictionary_path;  // runner.cpp:17 in the reference; compress_engine.cpp here

class Predictor {
 public:
  explicit Predictor(const std::vector<bool>& vocab, size_t chunk_bytes) : chunk_(chunk_bytes) {
    for (int i = 0; i < 256; ++i) vocab_[i] = vocab[i] ? 1 : 0;
    const char* dev = getenv("CMIX_DEVICE");
    device_ = dev ? atoi(dev) : 0;
    pipe_ = cmx_pipeline_create(vocab_, device_, chunk_);
    if (!pipe_ || cmx_pipeline_enable_fxcm(pipe_, dictionary_path) || cmx_pipeline_enable_paq8(pipe_)) Die();
  }
  ~Predictor() { cmx_pipeline_destroy(pipe_); }
  Predictor(const Predictor&) = delete;
  Predictor& operator=(const Predictor&) = delete;

  void Pretrain(int bit) {
    pre_partial_ = (pre_partial_ << 1) | (bit ? 1u : 0u);
    if (++pre_j_ == 8) { pre_.push_back((uint8_t)pre_partial_); pre_j_ = 0; pre_partial_ = 0; }
  }
  void FlushPretrain() {
    if (!pre_.empty() && cmx_pipeline_pretrain(pipe_, pre_.data(), pre_.size())) Die();
    pre_.clear();
  }
  float Predict() { fprintf(stderr, "engine build: coding goes through CompressEngine()\n"); abort(); }
  void Perceive(int) { Predict(); }

  cmx_pipeline_t* pipe() { return pipe_; }
  int device() const { return device_; }
  size_t chunk() const { return chunk_; }
  static void Die() {
    fprintf(stderr, "cmix_amd: %s\n", cmx_last_error());
    abort();
  }

 private:
  size_t chunk_;
  int device_ = 0;
  unsigned char vocab_[256];
  cmx_pipeline_t* pipe_ = nullptr;
  std::vector<uint8_t> pre_;
  int pre_j_ = 0;
  unsigned pre_partial_ = 0;
};

#endif
