/* oracle/coder_oracle.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * CPU restatement of the reference's binary arithmetic coder: Encoder (src/coder/encoder.cpp:3-39) and Decoder
 * (src/coder/decoder.cpp:3-39), and of the container header (src/runner.cpp:34-84). Pinned against the reference's
 * own coder compiled from its sources (oracle/ref_coder.cpp -> oracle/_ref/libcmixrefcoder.so) in
 * tests/test_coder.py and against tests/golden/coder_vectors.npz.
 */
#include <stddef.h>
#include <stdint.h>
#include <string.h>

/* encoder.cpp:10-12 / decoder.cpp:16-18: `1 + 65534 * p` with p float: int -> float conversions, a float product, a
 * float sum, then truncation to unsigned. */
static uint32_t orc_discretize(float p) {
  volatile float prod = (float)65534 * p;
  volatile float sum = (float)1 + prod;
  return (uint32_t)sum;
}

/* encoder.cpp:16-17 */
static uint32_t orc_xmid(uint32_t x1, uint32_t x2, uint32_t p) {
  return x1 + ((x2 - x1) >> 16) * p + (((x2 - x1) & 0xffff) * p >> 16);
}

/* Encode nbits then Flush. Returns the number of code bytes, or (size_t)-1 when `cap` is too small. */
size_t orc_coder_encode(const float* p, const uint8_t* bits, size_t nbits, uint8_t* out, size_t cap) {
  uint32_t x1 = 0, x2 = 0xffffffffu;  /* encoder.cpp:3-4 */
  size_t n = 0;
  for (size_t t = 0; t < nbits; ++t) {
    uint32_t xmid = orc_xmid(x1, x2, orc_discretize(p[t]));
    if (bits[t]) x2 = xmid; else x1 = xmid + 1;                 /* encoder.cpp:18-22 */
    while (((x1 ^ x2) & 0xff000000u) == 0) {                    /* encoder.cpp:26-30 */
      if (n >= cap) return (size_t)-1;
      out[n++] = (uint8_t)(x2 >> 24);
      x1 <<= 8;
      x2 = (x2 << 8) + 255;
    }
  }
  while (((x1 ^ x2) & 0xff000000u) == 0) {                      /* Flush, encoder.cpp:33-39 */
    if (n >= cap) return (size_t)-1;
    out[n++] = (uint8_t)(x2 >> 24);
    x1 <<= 8;
    x2 = (x2 << 8) + 255;
  }
  if (n >= cap) return (size_t)-1;
  out[n++] = (uint8_t)(x2 >> 24);
  return n;
}

void orc_coder_decode(const float* p, size_t nbits, const uint8_t* code, size_t len, uint8_t* bits_out) {
  uint32_t x1 = 0, x2 = 0xffffffffu, x = 0;
  size_t pos = 0;
  for (int i = 0; i < 4; ++i) x = (x << 8) + (pos < len ? code[pos++] : 0);  /* decoder.cpp:5-7; ReadByte :10-14 */
  for (size_t t = 0; t < nbits; ++t) {
    uint32_t xmid = orc_xmid(x1, x2, orc_discretize(p[t]));
    if (x <= xmid) { bits_out[t] = 1; x2 = xmid; } else { bits_out[t] = 0; x1 = xmid + 1; }  /* decoder.cpp:24-30 */
    while (((x1 ^ x2) & 0xff000000u) == 0) {                                                  /* decoder.cpp:33-37 */
      x1 <<= 8;
      x2 = (x2 << 8) + 255;
      x = (x << 8) + (pos < len ? code[pos++] : 0);
    }
  }
}

/* runner.cpp:34-52 */
size_t orc_header_write(uint64_t length, const uint8_t* vocab, int dictionary_used, uint8_t* out) {
  size_t n = 0;
  for (int i = 4; i >= 0; --i) {
    uint8_t c = (uint8_t)(length >> (8 * i));
    if (i == 4) { c &= 0x7F; if (dictionary_used) c |= 0x80; }
    out[n++] = c;
  }
  if (length < 10000) return n;  /* kMinVocabFileSize, runner.cpp:14 */
  for (int i = 0; i < 32; ++i) {
    uint8_t c = 0;
    for (int j = 0; j < 8; ++j) if (vocab[i * 8 + j]) c += (uint8_t)(1 << j);
    out[n++] = c;
  }
  return n;
}
