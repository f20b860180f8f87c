// tests/host/p8match_emul.cpp -- TEST INFRASTRUCTURE ONLY. The body of cmx_p8match_kernel (cmix_amd/csrc/p8match_dev.h) on
// the host: same step functions and construction, the two lanes looped per barrier step (sparse lane first). Checked against
// the oracle in tests/test_p8match_host.py. Nothing in cmix_amd/ loads it.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../cmix_amd/csrc/p8match_build.h"

namespace {
struct HostPolicy {
  std::vector<void*> blocks;
  void* zalloc(size_t bytes) { void* p = calloc(bytes + 64, 1); blocks.push_back(p); return p; }
  void upload(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
};
struct Emul { P8MatchDev dev; HostPolicy pol; };
}  // namespace

extern "C" {
void* p8m_create(uint64_t match_bytes, uint64_t sparse_bytes, int hist_log2, const uint8_t* nex, const int16_t* stretch, const uint8_t* ilog) {
  Emul* e = new Emul();
  if (!p8b::build_match(e->dev, e->pol, match_bytes, sparse_bytes, hist_log2, nex, stretch, ilog)) { delete e; return nullptr; }
  return e;
}
void p8m_destroy(void* h) { Emul* e = (Emul*)h; for (void* p : e->pol.blocks) free(p); delete e; }
// whole bytes; out [8n][28] i16, stats [8n][3] i32 (match length, expected byte or -1, sparse length), sets [8n][2] i32
// t0 = 1 starts a stream the way paq8's Predictor does: its first contextModel2 call already has one coded bit (bpos 1)
void p8m_run_from(void* h, const uint8_t* bytes, int n, int t0, int16_t* out, int* stats, int* sets);
void p8m_run(void* h, const uint8_t* bytes, int n, int16_t* out, int* stats, int* sets) { p8m_run_from(h, bytes, n, 0, out, stats, sets); }
void p8m_run_from(void* h, const uint8_t* bytes, int n, int t0, int16_t* out, int* stats, int* sets) {
  P8MatchDev* d = &((Emul*)h)->dev;
  int y = d->last_y;
  for (int t = 0; t < t0; t++) y = (bytes[t >> 3] >> (7 - (t & 7))) & 1;
  for (int t = t0; t < 8 * n; t++) {
    const int bpos = t & 7, cur = bytes[t >> 3];
    const int c0 = (1 << bpos) | (cur >> (8 - bpos));
    for (int tid = 1; tid >= 0; tid--) p8d_match_step2(d, tid, y, bpos, c0, out + (size_t)t * 28, stats + (size_t)t * 3, sets + (size_t)t * 2);
    for (int tid = 1; tid >= 0; tid--) p8d_match_step1(d, tid, bpos == 7, cur);
    y = (cur >> (7 - bpos)) & 1;
  }
  d->last_y = y;
}
}
