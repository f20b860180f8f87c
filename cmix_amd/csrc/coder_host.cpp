// coder_host.cpp -- the callers either side of the prediction path (last row): the 32-bit binary
// arithmetic coder that turns Predictor::Predict()'s float into code bytes (src/coder/encoder.cpp:10-39,
// src/coder/decoder.cpp:3-39) and the container header of runner.cpp:34-84. HOST code: one multiply-add and a
// compare per bit, consuming the p[] a chunk of the device pipeline has produced (cmx_pipeline_submit).
//
// Numeric contract. Discretize() is float arithmetic: 65534 -> float, float product (rounded), + 1.0f (rounded),
// truncation to unsigned. The interval split is 32-bit unsigned arithmetic with wrap-around exactly as written in
// the reference. Built without FP contraction (cmix_amd/build.py).
#include <stddef.h>
#include <stdint.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/cmix_amd.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip

namespace {
inline uint32_t discretize(float p) {
  volatile float prod = 65534.0f * p;  // volatile: the product is rounded to float before the add, on any compiler
  float s = 1.0f + prod;
  return (uint32_t)s;
}
inline uint32_t split(uint32_t x1, uint32_t x2, uint32_t p) {
  const uint32_t range = x2 - x1;
  return x1 + (range >> 16) * p + (((range & 0xffffu) * p) >> 16);
}
}  // namespace

struct cmx_encoder {
  uint32_t x1 = 0, x2 = 0xffffffffu;
  bool flushed = false;
  std::vector<uint8_t> out;
  inline void shift_out() {
    while (((x1 ^ x2) & 0xff000000u) == 0) {
      out.push_back((uint8_t)(x2 >> 24));
      x1 <<= 8;
      x2 = (x2 << 8) + 255;
    }
  }
  inline void encode(float p, int bit) {
    const uint32_t xmid = split(x1, x2, discretize(p));
    if (bit) x2 = xmid; else x1 = xmid + 1;
    shift_out();
  }
};

struct cmx_decoder {
  const uint8_t* in = nullptr;
  size_t len = 0, pos = 0;
  uint32_t x1 = 0, x2 = 0xffffffffu, x = 0;
  inline uint32_t read_byte() { return pos < len ? in[pos++] : 0u; }  // past the end the reference reads zeros
  inline int decode(float p) {
    const uint32_t xmid = split(x1, x2, discretize(p));
    int bit = 0;
    if (x <= xmid) { bit = 1; x2 = xmid; } else { x1 = xmid + 1; }
    while (((x1 ^ x2) & 0xff000000u) == 0) {
      x1 <<= 8;
      x2 = (x2 << 8) + 255;
      x = (x << 8) + read_byte();
    }
    return bit;
  }
};

extern "C" {

cmx_encoder_t* cmx_encoder_create(void) { return new cmx_encoder(); }
void cmx_encoder_destroy(cmx_encoder_t* e) { delete e; }

int cmx_encoder_encode_bits(cmx_encoder_t* e, const float* p, const uint8_t* bits, size_t nbits) {
  if (!e || (nbits && (!p || !bits))) { cmx_set_err("cmx_encoder_encode_bits: bad argument"); return 1; }
  if (e->flushed) { cmx_set_err("cmx_encoder_encode_bits: encoder already flushed"); return 1; }
  for (size_t t = 0; t < nbits; ++t) {
    if (!(p[t] >= 0.0f && p[t] <= 1.0f)) { cmx_set_err("cmx_encoder_encode_bits: probability outside [0,1]"); return 1; }
    e->encode(p[t], bits[t] & 1);
  }
  return 0;
}

int cmx_encoder_encode_bytes(cmx_encoder_t* e, const float* p, const uint8_t* bytes, size_t nbytes) {
  if (!e || (nbytes && (!p || !bytes))) { cmx_set_err("cmx_encoder_encode_bytes: bad argument"); return 1; }
  if (e->flushed) { cmx_set_err("cmx_encoder_encode_bytes: encoder already flushed"); return 1; }
  for (size_t i = 0; i < nbytes; ++i)
    for (int j = 7; j >= 0; --j) {  // runner.cpp:106-108
      const float q = p[8 * i + (7 - j)];
      if (!(q >= 0.0f && q <= 1.0f)) { cmx_set_err("cmx_encoder_encode_bytes: probability outside [0,1]"); return 1; }
      e->encode(q, (bytes[i] >> j) & 1);
    }
  return 0;
}

int cmx_encoder_flush(cmx_encoder_t* e) {
  if (!e) { cmx_set_err("cmx_encoder_flush: null handle"); return 1; }
  if (e->flushed) return 0;
  e->shift_out();
  e->out.push_back((uint8_t)(e->x2 >> 24));
  e->flushed = true;
  return 0;
}

size_t cmx_encoder_size(const cmx_encoder_t* e) { return e ? e->out.size() : 0; }
const uint8_t* cmx_encoder_data(const cmx_encoder_t* e) { return e ? e->out.data() : nullptr; }

cmx_decoder_t* cmx_decoder_create(const uint8_t* code, size_t len) {
  if (len && !code) { cmx_set_err("cmx_decoder_create: bad argument"); return nullptr; }
  cmx_decoder* d = new cmx_decoder();
  d->in = code;
  d->len = len;
  for (int i = 0; i < 4; ++i) d->x = (d->x << 8) + (d->read_byte() & 0xff);
  return d;
}
void cmx_decoder_destroy(cmx_decoder_t* d) { delete d; }

int cmx_decoder_decode(cmx_decoder_t* d, float p) {
  if (!d || !(p >= 0.0f && p <= 1.0f)) { cmx_set_err("cmx_decoder_decode: bad argument"); return -1; }
  return d->decode(p);
}

int cmx_decoder_decode_bits(cmx_decoder_t* d, const float* p, size_t nbits, uint8_t* bits_out) {
  if (!d || (nbits && (!p || !bits_out))) { cmx_set_err("cmx_decoder_decode_bits: bad argument"); return 1; }
  for (size_t t = 0; t < nbits; ++t) {
    if (!(p[t] >= 0.0f && p[t] <= 1.0f)) { cmx_set_err("cmx_decoder_decode_bits: probability outside [0,1]"); return 1; }
    bits_out[t] = (uint8_t)d->decode(p[t]);
  }
  return 0;
}

size_t cmx_header_write(uint64_t length, const uint8_t vocab[256], int dictionary_used, uint8_t out[CMX_HEADER_MAX]) {
  if (!out || (length >= CMX_MIN_VOCAB_FILE_SIZE && !vocab)) { cmx_set_err("cmx_header_write: bad argument"); return 0; }
  if (length >> 39) { cmx_set_err("cmx_header_write: length does not fit 39 bits"); return 0; }
  size_t n = 0;
  for (int i = 4; i >= 0; --i) {
    uint8_t c = (uint8_t)(length >> (8 * i));
    if (i == 4) { c &= 0x7F; if (dictionary_used) c |= 0x80; }
    out[n++] = c;
  }
  if (length < CMX_MIN_VOCAB_FILE_SIZE) return n;
  for (int i = 0; i < 32; ++i) {
    uint8_t c = 0;
    for (int j = 0; j < 8; ++j) if (vocab[i * 8 + j]) c |= (uint8_t)(1u << j);
    out[n++] = c;
  }
  return n;
}

size_t cmx_header_read(const uint8_t* in, size_t len, uint64_t* length, int* dictionary_used, uint8_t vocab[256]) {
  if (!in || !length || !dictionary_used || !vocab || len < 5) { cmx_set_err("cmx_header_read: bad argument / short header"); return 0; }
  uint64_t L = 0;
  for (int i = 0; i <= 4; ++i) {
    uint8_t c = in[i];
    if (i == 0) { *dictionary_used = (c & 0x80) ? 1 : 0; c &= 0x7F; }
    L = (L << 8) + c;
  }
  *length = L;
  memset(vocab, 0, 256);
  if (L == 0) return 5;
  if (L < CMX_MIN_VOCAB_FILE_SIZE) { memset(vocab, 1, 256); return 5; }
  if (len < 37) { cmx_set_err("cmx_header_read: short header"); return 0; }
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 8; ++j) vocab[i * 8 + j] = (in[5 + i] >> j) & 1;
  return 37;
}

}  // extern "C"
