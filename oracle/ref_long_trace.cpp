// oracle/ref_long_trace.cpp -- TEST INFRASTRUCTURE (never linked into the product): drives the UNMODIFIED reference Predictor (oracle/_ref/libcmixref.so,
// oracle/ref_harness.cpp) over a long stream and prints, for every block of 64 KB of input, one order-independent 64-bit digest per group of SIXTEEN
// consecutive layer-0 inputs (130 groups: SURVEY.md appendix A maps columns to models) and one of the final probability -- so that a long run of the
// engine (scripts/gpu_stage_hashes.py, the same digests on the device) can be compared block by block and model by model without storing 139 G floats.
// Round 5: the engine's 8 MiB file differs from the reference binary's by two bytes (profiles/r05_long_run_8m.json) -- which stage, where?
//   digest(block, group c / 16) = sum over the block's bits t and the group's columns c of (bits(p[t][c]) + 1) * A[c] * B[t mod 2^19]   (mod 2^64)
//   A[c] = splitmix64(c) | 1, B[i] = splitmix64(0x1000000 + i) | 1
//
// With the engine's digest file as a fourth argument the comparison happens ON LINE, and the reference cannot be rewound, so from block `first_snapshot`
// on the process fork()s at every block start: the child sleeps on a pipe holding the predictor's state of that moment (copy-on-write). A block whose
// digests equal the engine's releases the child; the first block that differs wakes it up instead, and the child runs the block AGAIN, this time writing
//   <out>.block<k>.digests   the 131 digests of each 4 KB piece of the block (16 lines)
//   <out>.block<k>.g<g>.f32  the block's rows of each differing group (2^19 x 16 floats; at most 12 groups), <out>.block<k>.p.f32 the final probabilities
// and then goes on to the end of the stream without snapshots, so that <out> lists every later block too.
// usage: ref_long_trace stream.bin vocab256.bin out.txt [engine_digests.txt [first_snapshot_block]]
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <sys/wait.h>
#include <unistd.h>
extern "C" {
int ref_create(const uint8_t* vocab256, const char* dict_path);
float ref_predict(void);
void ref_perceive(int bit);
int ref_get_model_probs(float* out);
}
#ifdef WITH_ORACLE_MIXNET   // -DWITH_ORACLE_MIXNET -L_build -lcmixoracle: every row is also fed to the oracle's restatement of the final mixing network + SSE, whose
extern "C" {                // result must be the float Predictor::Predict() returned; the selectors come from ref_get_mixers (ref_harness.cpp)
#include "cmix_oracle.h"
int ref_get_mixers(int layer, uint64_t* ctx, float* out);
int ref_num_mixers(int layer);
}
static orc_mixnet* g_om = nullptr;
static unsigned long long g_mix_bad = 0; static long long g_mix_first = -1;
#endif
static uint64_t splitmix64(uint64_t x) { x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull; x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; return x ^ (x >> 31); }
enum { NG = 130, BLOCK_BITS = 1 << 19 };
static uint64_t A[2078], B[BLOCK_BITS];
static FILE* g_pfile = nullptr;   // REF_TRACE_P_FILE=path: every final probability, 4 bytes per bit (to run the coder over them: is this harness the binary?)
static std::vector<uint8_t> s;

struct Detail {   // what the woken child writes while it repeats a block
  FILE* dig = nullptr; FILE* p = nullptr; std::vector<FILE*> rows; std::vector<int> groups;
};

// one block from bit t0; returns the digests in h[131]
static void run_block(size_t t0, size_t t1, uint64_t* h, Detail* d) {
  static float probs[2078];
  uint64_t piece[NG + 1];
  for (int k = 0; k <= NG; ++k) { h[k] = 0; piece[k] = 0; }
  for (size_t t = t0; t < t1; ++t) {
    const float p = ref_predict();
    ref_get_model_probs(probs);
    const uint64_t b = B[t & (BLOCK_BITS - 1)];
    uint64_t g[NG];
    for (int k = 0; k < NG; ++k) g[k] = 0;
    for (int c = 0; c < 2078; ++c) {
      uint32_t u; memcpy(&u, &probs[c], 4);
      g[c >> 4] += ((uint64_t)u + 1ull) * A[c];
    }
#ifdef WITH_ORACLE_MIXNET
    {
      uint64_t sel[64]; float mo[64]; int q = 0;
      for (int layer = 0; layer < 3; ++layer) { ref_get_mixers(layer, sel + q, mo + q); q += ref_num_mixers(layer); }
      const float p_orc = orc_mixnet_step(g_om, probs, sel, (s[t >> 3] >> (7 - (t & 7))) & 1, nullptr);
      if (memcmp(&p_orc, &p, 4) != 0) {
        if (g_mix_first < 0) g_mix_first = (long long)t;
        if (g_mix_bad < 20) fprintf(stderr, "oracle mixing network != reference at bit %zu (byte %zu bit %zu): oracle %.9g (%08x) reference %.9g (%08x)\n", t, t >> 3, t & 7, p_orc, *(uint32_t*)&p_orc, p, *(const uint32_t*)&p);
        ++g_mix_bad;
      }
    }
#endif
    if (g_pfile) fwrite(&p, 4, 1, g_pfile);
    uint32_t u; memcpy(&u, &p, 4);
    const uint64_t hp = ((uint64_t)u + 1ull) * b;
    for (int k = 0; k < NG; ++k) h[k] += g[k] * b;
    h[NG] += hp;
    if (d) {
      for (int k = 0; k < NG; ++k) piece[k] += g[k] * b;
      piece[NG] += hp;
      fwrite(&p, 4, 1, d->p);
      for (size_t i = 0; i < d->groups.size(); ++i) {
        float row[16] = {0};
        for (int j = 0; j < 16 && 16 * d->groups[i] + j < 2078; ++j) row[j] = probs[16 * d->groups[i] + j];
        fwrite(row, 4, 16, d->rows[i]);
      }
      if (((t + 1) & 32767) == 0 || t + 1 == t1) {   // a 4 KB piece is complete
        fprintf(d->dig, "%zu", (t + 1) >> 3);
        for (int k = 0; k <= NG; ++k) { fprintf(d->dig, " %016llx", (unsigned long long)piece[k]); piece[k] = 0; }
        fprintf(d->dig, "\n");
      }
    }
    ref_perceive((s[t >> 3] >> (7 - (t & 7))) & 1);
  }
}

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  FILE* f = fopen(argv[1], "rb"); if (!f) return 3;
  { uint8_t buf[65536]; size_t k; while ((k = fread(buf, 1, sizeof buf, f)) > 0) s.insert(s.end(), buf, buf + k); } fclose(f);
  uint8_t vocab[256]; f = fopen(argv[2], "rb"); if (!f || fread(vocab, 1, 256, f) != 256) return 4; fclose(f);
  const std::string outp = argv[3];
  std::vector<std::vector<uint64_t>> eng;   // the engine's digests, one line per block
  if (argc > 4) {
    FILE* e = fopen(argv[4], "r"); if (!e) return 7;
    unsigned long long pos;
    while (fscanf(e, "%llu", &pos) == 1) {
      std::vector<uint64_t> row(NG + 1);
      for (int k = 0; k <= NG; ++k) { unsigned long long v; if (fscanf(e, "%llx", &v) != 1) return 8; row[k] = v; }
      eng.push_back(row);
    }
    fclose(e);
  }
  const size_t first_snapshot = argc > 5 ? (size_t)atol(argv[5]) : 0;
  FILE* out = fopen(outp.c_str(), "w"); if (!out) return 5;
  if (ref_create(vocab, "")) return 6;
  if (getenv("REF_TRACE_P_FILE")) g_pfile = fopen(getenv("REF_TRACE_P_FILE"), "wb");
#ifdef WITH_ORACLE_MIXNET
  g_om = orc_mixnet_create();
#endif
  for (int c = 0; c < 2078; ++c) A[c] = splitmix64((uint64_t)c) | 1ull;
  for (int i = 0; i < BLOCK_BITS; ++i) B[i] = splitmix64(0x1000000ull + (uint64_t)i) | 1ull;
  const size_t nbits = s.size() * 8, nblocks = (nbits + BLOCK_BITS - 1) / BLOCK_BITS;
  bool snapshots = !eng.empty();
  for (size_t k = 0; k < nblocks; ++k) {
    const size_t t0 = k * BLOCK_BITS, t1 = t0 + BLOCK_BITS < nbits ? t0 + BLOCK_BITS : nbits;
    int pfd[2] = {-1, -1};
    pid_t child = -1;
    if (snapshots && k >= first_snapshot && k < eng.size()) {
      fflush(out);
      if (pipe(pfd) != 0) return 9;
      child = fork();
      if (child == 0) {   // the snapshot: sleeps until the parent has run the block
        close(pfd[1]);
        unsigned char verdict[1 + 17] = {0};
        if (read(pfd[0], verdict, sizeof verdict) < 1 || verdict[0] != 'g') _exit(0);
        close(pfd[0]);
        Detail d;
        char name[600];
        for (int g = 0; g <= NG && d.groups.size() < 12; ++g) if (g < NG && ((verdict[1 + g / 8] >> (g & 7)) & 1)) d.groups.push_back(g);
        snprintf(name, sizeof name, "%s.block%zu.digests", outp.c_str(), k); d.dig = fopen(name, "w");
        snprintf(name, sizeof name, "%s.block%zu.p.f32", outp.c_str(), k); d.p = fopen(name, "wb");
        for (int g : d.groups) { snprintf(name, sizeof name, "%s.block%zu.g%d.f32", outp.c_str(), k, g); d.rows.push_back(fopen(name, "wb")); }
        uint64_t h[NG + 1];
        run_block(t0, t1, h, &d);
        fclose(d.dig); fclose(d.p); for (FILE* r : d.rows) fclose(r);
        fprintf(out, "%zu", t1 >> 3);
        for (int g = 0; g <= NG; ++g) fprintf(out, " %016llx", (unsigned long long)h[g]);
        fprintf(out, "\n"); fflush(out);
        snapshots = false;   // this process carries on to the end of the stream, the parent has gone
        continue;
      }
      close(pfd[0]);
    }
    uint64_t h[NG + 1];
    run_block(t0, t1, h, nullptr);
    if (child > 0) {
      unsigned char verdict[1 + 17] = {0};
      bool bad = false;
      for (int g = 0; g <= NG; ++g) if (h[g] != eng[k][g]) { bad = true; verdict[1 + g / 8] |= (unsigned char)(1u << (g & 7)); }
      verdict[0] = bad ? 'g' : 'k';
      if (write(pfd[1], verdict, sizeof verdict) != (ssize_t)sizeof verdict) return 10;
      close(pfd[1]);
      if (bad) {   // the child repeats the block with the detail files and finishes the stream; this process is done
        FILE* note = fopen((outp + ".first_difference").c_str(), "w");
        fprintf(note, "block %zu (bytes %zu..%zu): groups", k, t0 >> 3, (t1 >> 3) - 1);
        for (int g = 0; g <= NG; ++g) if (h[g] != eng[k][g]) fprintf(note, " %d", g);
        fprintf(note, "\n"); fclose(note);
        int st; waitpid(child, &st, 0);
        return 0;
      }
      int st; waitpid(child, &st, 0);
    }
    fprintf(out, "%zu", t1 >> 3);
    for (int g = 0; g <= NG; ++g) fprintf(out, " %016llx", (unsigned long long)h[g]);
#ifdef WITH_ORACLE_MIXNET
    fprintf(out, " mixnet_bits_differing_so_far %llu first %lld", g_mix_bad, g_mix_first);
#endif
    fprintf(out, "\n"); fflush(out);
  }
  fclose(out);
  if (g_pfile) fclose(g_pfile);
  return 0;
}
