// tests/host/p8dmc_emul.cpp -- TEST INFRASTRUCTURE ONLY. The step functions of cmx_p8s_dmc_kernel (cmix_amd/csrc/p8dmc_dev.h) on the
// host: same step functions and construction, lanes looped per barrier step in reverse order. Checked against the oracle in
// tests/test_p8dmc_host.py. Nothing in cmix_amd/ loads it.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../cmix_amd/csrc/p8dmc_build.h"

namespace {
struct HostPolicy {
  std::vector<void*> blocks;
  void* zalloc(size_t bytes) { void* p = calloc(bytes + 64, 1); blocks.push_back(p); return p; }
  void upload(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
};
struct Emul { P8DmcDev dev; P8DmcShared sh; HostPolicy pol; uint64_t resets = 0; };
}  // namespace

extern "C" {
void* p8x_create(int level, const uint8_t* nex, const int16_t* stretch) { Emul* e = new Emul(); p8b::build_dmc(e->dev, e->pol, level, nex, stretch); return e; }
void p8x_destroy(void* h) { Emul* e = (Emul*)h; for (void* p : e->pol.blocks) free(p); delete e; }
// start mid-stream: the last coded bit and the number of bits coded so far
void p8x_seed(void* h, int last_y, uint32_t bits_done) { ((Emul*)h)->dev.last_y = last_y; ((Emul*)h)->dev.bits_done = bits_done; }
uint64_t p8x_resets(void* h) { return ((Emul*)h)->resets; }
void p8x_run(void* h, const uint8_t* bits, int nbits, int16_t* out) {
  Emul* e = (Emul*)h;
  int y = e->dev.last_y;
  const uint32_t done = e->dev.bits_done;
  for (int t = 0; t < nbits; t++) {
    for (int tid = P8DMC_THREADS - 1; tid >= 0; tid--) p8d_dmc_step1(&e->dev, &e->sh, tid, y);
    for (int tid = P8DMC_THREADS - 1; tid >= 0; tid--) p8d_dmc_step2(&e->dev, &e->sh, tid, (int)((done + (uint32_t)t) & 7), out + (size_t)t * 6);
    for (int k = 0; k < 8; k++) e->resets += e->sh.reset[k] != 0;
    for (int tid = P8DMC_THREADS - 1; tid >= 0; tid--) p8d_dmc_step3(&e->dev, &e->sh, tid);
    y = bits[t];
  }
  e->dev.last_y = y; e->dev.bits_done = done + (uint32_t)nbits;
}
}
