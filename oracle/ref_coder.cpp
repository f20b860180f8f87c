// oracle/ref_coder.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// Compiles the UNMODIFIED reference arithmetic coder (src/coder/encoder.cpp, src/coder/decoder.cpp) from where it
// lies, against a stand-in `Predictor` that replays a recorded probability sequence: the coder only ever calls
// Predict() and Perceive() (encoder.cpp:15,24; decoder.cpp:21,31), so pre-defining predictor.h's include guard and
// supplying those two members leaves every line of the coder itself untouched. Built as its own shared object
// (oracle/_ref/libcmixrefcoder.so, -Bsymbolic) so its Encoder/Decoder never meet the real ones in libcmixref.so.
#define PREDICTOR_H
#include <stddef.h>
#include <stdint.h>
#include <fstream>

class Predictor {
 public:
  Predictor(const float* p, size_t n) : p_(p), n_(n), i_(0) {}
  float Predict() { return i_ < n_ ? p_[i_] : 0.5f; }
  void Perceive(int) { ++i_; }

 private:
  const float* p_;
  size_t n_, i_;
};

#include "coder/encoder.cpp"
#include "coder/decoder.cpp"

extern "C" {

// Encoder over p[0..n) / bits[0..n) followed by Flush(), written to `path`. Returns 0 on success.
int refcoder_encode(const float* p, const uint8_t* bits, size_t n, const char* path) {
  std::ofstream os(path, std::ios::out | std::ios::binary);
  if (!os.is_open()) return 1;
  Predictor pr(p, n);
  Encoder e(&os, &pr);
  for (size_t t = 0; t < n; ++t) e.Encode(bits[t] & 1);
  e.Flush();
  os.close();
  return 0;
}

// Decoder over the file at `path`, replaying p[0..n).
int refcoder_decode(const float* p, size_t n, const char* path, uint8_t* bits_out) {
  std::ifstream is(path, std::ios::in | std::ios::binary);
  if (!is.is_open()) return 1;
  Predictor pr(p, n);
  Decoder d(&is, &pr);
  for (size_t t = 0; t < n; ++t) bits_out[t] = (uint8_t)d.Decode();
  return 0;
}

}  // extern "C"
