// oracle/ref_paq8_trace.cpp -- TEST INFRASTRUCTURE (never linked into the product): the UNMODIFIED paq8::Predictor alone (oracle/_ref/libcmixrefpaq8.so,
// oracle/ref_paq8core.cpp, cmix's level 11) over a long stream, printing ref_long_trace.cpp's digests for the column groups that lie wholly inside PAQ8's
// layer-0 columns 434..2024 (groups 28..125 of 16 columns; the other fields of a line are 0). paq8 reads nothing but the bits, so its columns in the full
// predictor are these -- at a fifth of the full reference's cost. Compared with scripts/gpu_stage_hashes.py's file by `--compare-groups 28 125`.
// usage: ref_paq8_trace stream.bin out.txt
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
extern "C" {
void* refp8_predictor_new(int level);
int refp8_predictor_update(void* h, int bit, float* out1591);
}
static uint64_t splitmix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
int main(int argc, char** argv) {
  if (argc < 3) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  std::vector<uint8_t> s; { uint8_t buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) s.insert(s.end(), buf, buf + k); } fclose(f);
  FILE* out = fopen(argv[2], "w"); if (!out) return 5;
  void* P = refp8_predictor_new(11);
  static uint64_t A[2078], B[1 << 19];
  for (int c = 0; c < 2078; ++c) A[c] = splitmix64((uint64_t)c) | 1ull;
  for (int i = 0; i < (1 << 19); ++i) B[i] = splitmix64(0x1000000ull + (uint64_t)i) | 1ull;
  static uint64_t h[131];
  static float probs[1591];
  for (int j = 0; j < 1591; ++j) probs[j] = 0.5f;   // PAQ8::Predict() before the first Perceive
  const size_t nbits = s.size() * 8;
  for (size_t t = 0; t < nbits; ++t) {
    const uint64_t b = B[t & ((1u << 19) - 1)];
    uint64_t g[130];
    for (int k = 0; k < 130; ++k) g[k] = 0;
    for (int j = 0; j < 1591; ++j) {
      const int c = 434 + j;
      uint32_t u; memcpy(&u, &probs[j], 4);
      g[c >> 4] += ((uint64_t)u + 1ull) * A[c];
    }
    for (int k = 28; k <= 125; ++k) h[k] += g[k] * b;
    refp8_predictor_update(P, (s[t >> 3] >> (7 - (t & 7))) & 1, probs);
    if (((t + 1) & ((1u << 19) - 1)) == 0 || t + 1 == nbits) {
      fprintf(out, "%zu", (t + 1) >> 3);
      for (int k = 0; k < 131; ++k) { fprintf(out, " %016llx", (unsigned long long)h[k]); h[k] = 0; }
      fprintf(out, "\n");
      fflush(out);
    }
  }
  fclose(out);
  return 0;
}
