// tests/host/p8cm_emul.cpp -- TEST INFRASTRUCTURE ONLY. The step functions of the ContextMap family (cmix_amd/csrc/p8cm_dev.h: first design, used by cmx_p8s_xfam_kernel; p8fam_dev.h: cmx_p8s_fam2_kernel) on the host:
// same step functions and construction, the workgroup replaced by a loop over lanes per barrier step in a seeded shuffled
// order. Checked against the oracle in tests/test_p8cm_host.py. Nothing in cmix_amd/ loads it.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../cmix_amd/csrc/p8cm_build.h"

namespace {
struct HostPolicy {
  std::vector<void*> blocks;
  void* zalloc(size_t bytes) { void* p = calloc(bytes + 64, 1); blocks.push_back(p); return p; }
  void upload(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
};
struct Emul { P8CmDev dev; P8CmShared sh; HostPolicy pol; uint32_t rng; int order[P8CM_MAXS]; uint64_t steps = 0, serial = 0, draws = 0; };
}  // namespace

extern "C" {
void* p8f_create(int ninst, const uint64_t* sizes, const int* counts, const uint8_t* nex, const int16_t* stretch, const uint8_t* ilog, uint32_t seed, int serial) {
  Emul* e = new Emul();
  if (!p8b::build_family(e->dev, e->pol, ninst, sizes, counts, nex, stretch, ilog)) { delete e; return nullptr; }
  if (serial) e->dev.slot_parallel = 0;
  e->rng = seed;
  for (int i = 0; i < P8CM_MAXS; i++) e->order[i] = i;
  return e;
}
void p8f_destroy(void* h) { Emul* e = (Emul*)h; for (void* p : e->pol.blocks) free(p); delete e; }
void p8f_stats(void* h, uint64_t* out3) { Emul* e = (Emul*)h; out3[0] = e->steps; out3[1] = e->serial; out3[2] = e->draws; }
int p8f_run(void* h, const uint32_t* ctx, const uint16_t* chk, const uint8_t* bits, int nbytes, int16_t* out) {
  Emul* e = (Emul*)h;
  P8CmDev* d = &e->dev;
  P8CmShared* sh = &e->sh;
  sh->r = d->regs; sh->rnd = d->rnd;
  int last_y = d->last_y, c1 = d->c1;
  const int S = d->nslots;
  for (int t = 0; t < 8 * nbytes; t++) {
    const P8CmBit u = p8d_cm_bit(d, ctx, chk, bits, out, nullptr, t, &last_y, &c1);
    if (e->rng)
      for (int i = S - 1; i > 0; i--) {
        e->rng = e->rng * 1664525u + 1013904223u;
        const int j = (int)((e->rng >> 8) % (uint32_t)(i + 1)), tmp = e->order[i];
        e->order[i] = e->order[j]; e->order[j] = tmp;
      }
    for (int k = 0; k < S; k++) p8d_cm_touch(d, sh, u, e->order[k]);
    for (int k = 0; k < S; k++) p8d_cm_check(d, sh, e->order[k]);
    for (int k = 0; k < S; k++) p8d_cm_draw(d, sh, e->order[k]);
    e->steps++; e->serial += sh->conflict != 0; e->draws += (uint64_t)sh->ndraws;
    for (int k = 0; k < S; k++) p8d_cm_run(d, sh, u, e->order[k]);
  }
  d->regs = sh->r; d->rnd = sh->rnd; d->last_y = last_y; d->c1 = c1;
  return 0;
}
}
