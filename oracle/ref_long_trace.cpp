// oracle/ref_long_trace.cpp -- TEST INFRASTRUCTURE (never linked into the product): drives the UNMODIFIED reference Predictor (oracle/_ref/libcmixref.so,
// oracle/ref_harness.cpp) over a long stream and prints, for every block of 64 KB of input, one order-independent 64-bit digest per group of SIXTEEN
// consecutive layer-0 inputs (130 groups: SURVEY.md appendix A maps columns to models) and one of the final probability -- so that a long run of the engine (scripts/gpu_stage_hashes.py, the same digests on the
// device) can be compared block by block and stage by stage without storing 139 G floats. Round 5: the engine's 8 MiB file differs from the reference
// binary's by two bytes (profiles/r05_long_run_8m.json) -- which stage, where?
//   digest(block, group c / 16) = sum over the block's bits t and the group's columns c of (bits(p[t][c]) + 1) * A[c] * B[t mod 2^19]   (mod 2^64)
//   A[c] = splitmix64(c) | 1, B[i] = splitmix64(0x1000000 + i) | 1
// usage: ref_long_trace stream.bin vocab256.bin out.txt
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
extern "C" {
int ref_create(const uint8_t* vocab256, const char* dict_path);
float ref_predict(void);
void ref_perceive(int bit);
int ref_get_model_probs(float* out);
}
static uint64_t splitmix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  std::vector<uint8_t> s; { uint8_t buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) s.insert(s.end(), buf, buf + k); } fclose(f);
  uint8_t vocab[256]; f = fopen(argv[2], "rb"); if (!f || fread(vocab, 1, 256, f) != 256) return 4; fclose(f);
  FILE* out = fopen(argv[3], "w"); if (!out) return 5;
  if (ref_create(vocab, "")) return 6;
  static uint64_t A[2078], B[1 << 19];
  for (int c = 0; c < 2078; ++c) A[c] = splitmix64((uint64_t)c) | 1ull;
  for (int i = 0; i < (1 << 19); ++i) B[i] = splitmix64(0x1000000ull + (uint64_t)i) | 1ull;
  static uint64_t h[131];
  static float probs[2078];
  const size_t nbits = s.size() * 8;
  for (size_t t = 0; t < nbits; ++t) {
    const float p = ref_predict();
    ref_get_model_probs(probs);
    const uint64_t b = B[t & ((1u << 19) - 1)];
    uint64_t g[130];
    for (int k = 0; k < 130; ++k) g[k] = 0;
    for (int c = 0; c < 2078; ++c) {
      uint32_t u; memcpy(&u, &probs[c], 4);
      g[c >> 4] += ((uint64_t)u + 1ull) * A[c];
    }
    for (int k = 0; k < 130; ++k) h[k] += g[k] * b;
    uint32_t u; memcpy(&u, &p, 4);
    h[130] += ((uint64_t)u + 1ull) * b;
    ref_perceive((s[t >> 3] >> (7 - (t & 7))) & 1);
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
