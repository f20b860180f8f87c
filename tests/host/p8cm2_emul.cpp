// tests/host/p8cm2_emul.cpp -- TEST INFRASTRUCTURE ONLY. The step functions of cmx_p8s_cm2v2_kernel (cmix_amd/csrc/p8cm2_dev.h; the stand-alone cmx_p8cm2 kernel is gone since round 3) on the
// host: same step functions, same construction (p8cm2_build.h), the workgroup replaced by a loop over lanes per barrier
// step in a seeded shuffled order. Checked against the oracle in tests/test_p8cm2_host.py. Nothing in cmix_amd/ loads it.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../cmix_amd/csrc/p8cm2_build.h"

namespace {
struct HostPolicy {
  std::vector<void*> blocks;
  void* zalloc(size_t bytes) { void* p = calloc(bytes + 64, 1); blocks.push_back(p); return p; }
  void upload(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
};
struct Emul { P8Cm2Dev dev; P8Cm2Shared sh; HostPolicy pol; uint32_t rng; int order[P8CM2_MAXC]; uint64_t steps = 0, serial = 0; };
}  // namespace

extern "C" {
void* p8e_create(uint64_t size_bytes, int count, const uint8_t* nex, const int16_t* stretch, const uint8_t* ilog, uint32_t seed, int serial) {
  Emul* e = new Emul();
  if (!p8b::build(e->dev, e->pol, size_bytes, count, nex, stretch, ilog)) { delete e; return nullptr; }
  if (serial) e->dev.slot_parallel = 0;
  e->rng = seed;
  for (int i = 0; i < P8CM2_MAXC; i++) e->order[i] = i;
  return e;
}
void p8e_destroy(void* h) { Emul* e = (Emul*)h; for (void* p : e->pol.blocks) free(p); delete e; }
void p8e_hash(uint64_t ctx, uint32_t index, uint64_t size_bytes, uint32_t* ctx32, uint16_t* chk16) { p8b::hash(ctx, index, p8b::hashbits(size_bytes), ctx32, chk16); }
// start in the middle of a stream: the partial-byte register and the last coded bit as ContextMap2 would hold them
void p8e_seed(void* h, uint32_t bits, int last_y) { ((Emul*)h)->dev.bits = bits; ((Emul*)h)->dev.last_y = last_y; }
void p8e_stats(void* h, uint64_t* out2) { out2[0] = ((Emul*)h)->steps; out2[1] = ((Emul*)h)->serial; }
int p8e_run(void* h, const uint32_t* ctx, const uint16_t* chk, const uint8_t* bits, int nbytes, int16_t* out) {
  Emul* e = (Emul*)h;
  P8Cm2Dev* d = &e->dev;
  P8Cm2Shared* sh = &e->sh;
  sh->r = d->regs;
  uint32_t run_bits = d->bits;
  int last_y = d->last_y;
  const int C = d->C;
  for (int t = 0; t < 8 * nbytes; t++) {
    const P8Cm2Bit u = p8d_bit(d, ctx, chk, bits, out, t, &run_bits, &last_y);
    if (e->rng)
      for (int i = C - 1; i > 0; i--) {
        e->rng = e->rng * 1664525u + 1013904223u;
        const int j = (int)((e->rng >> 8) % (uint32_t)(i + 1)), tmp = e->order[i];
        e->order[i] = e->order[j]; e->order[j] = tmp;
      }
    for (int k = 0; k < P8CM2_MAXC; k++) if (e->order[k] < C) p8d_touch(d, sh, u, e->order[k]);
    for (int k = 0; k < P8CM2_MAXC; k++) if (e->order[k] < C) p8d_conflict(d, sh, e->order[k]);
    e->steps++; e->serial += sh->conflict != 0;
    for (int k = 0; k < P8CM2_MAXC; k++) if (e->order[k] < C) p8d_run(d, sh, u, e->order[k]);
  }
  d->regs = sh->r; d->bits = run_bits; d->last_y = last_y;
  return 0;
}
}
