// tests/host/fxcm_emul.cpp -- TEST INFRASTRUCTURE ONLY. Runs the body of cmx_fxcm_chunk_kernel on the host: the same
// fxcm_dev.h phase functions, the same fxcm_build.h construction and the same host parser as the product, with the
// workgroup replaced by a loop over thread ids per phase -- in a seeded shuffled order, so that a phase in which one
// thread reads what another writes shows up as a mismatch against the oracle (tests/test_fxcm_stage_host.py).
// Not a fallback: nothing in cmix_amd/ loads this library.
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../cmix_amd/csrc/fxcm_build.h"

namespace {
struct HostPolicy {
  std::vector<void*> blocks;
  void* zalloc(size_t bytes) { void* p = calloc(bytes + 64, 1); blocks.push_back(p); return p; }
  void fill16(void* p, size_t n, uint16_t v) { uint16_t* q = (uint16_t*)p; for (size_t i = 0; i < n; i++) q[i] = v; }
  void fill32(void* p, size_t n, uint32_t v) { uint32_t* q = (uint32_t*)p; for (size_t i = 0; i < n; i++) q[i] = v; }
  void pattern16(void* p, size_t n, const uint16_t* pat, int plen) { uint16_t* q = (uint16_t*)p; for (size_t i = 0; i < n; i++) q[i] = pat[i % (size_t)plen]; }
  void upload(void* dst, const void* src, size_t bytes) { memcpy(dst, src, bytes); }
};
struct Emul { FxDev dev; FxShared sh; HostPolicy pol; FxParser* parser; uint32_t rng; int order[FX_THREADS]; uint64_t bits_maps = 0, bits_serial = 0; uint64_t serial_by[FX_NMAPS][8] = {}; };
void shuffle(Emul* e) {
  for (int i = FX_THREADS - 1; i > 0; i--) {
    e->rng = e->rng * 1664525u + 1013904223u;
    const int j = (int)((e->rng >> 8) % (uint32_t)(i + 1));
    const int t = e->order[i]; e->order[i] = e->order[j]; e->order[j] = t;
  }
}
}  // namespace

extern "C" {
void* fxe_create(const char* dictionary_path, uint32_t shuffle_seed) {
  Emul* e = new Emul();
  fxb::build(e->dev, e->pol);
  e->parser = fxp_create(dictionary_path);
  e->rng = shuffle_seed;
  for (int i = 0; i < FX_THREADS; i++) e->order[i] = i;
  return e;
}
void fxe_destroy(void* h) {
  Emul* e = (Emul*)h;
  for (void* p : e->pol.blocks) free(p);
  fxp_destroy(e->parser);
  delete e;
}
// n bytes; lstmpr / lstmex: the LSTM hints of each of the 8n updates; out: [8n][ostride] floats, row q = the 431 values
// FXCM::Predict() returns before bit q of the chunk is coded
int fxe_run(void* h, const uint8_t* bytes, int n, const int16_t* lstmpr, const uint8_t* lstmex, float* out, long ostride) {
  Emul* e = (Emul*)h;
  FxDev* d = &e->dev;
  FxShared* sh = &e->sh;
  std::vector<FxByteRec> recs((size_t)n);
  if (fxp_run(e->parser, bytes, n, recs.data()) != 0) return -1;
  const int nbits = 8 * n, blpos0 = d->blpos, lastbyte0 = d->lastbyte, have0 = d->have_rec;
  for (int t = 0; t < FX_THREADS; t++) fxd_load_shared(d, sh, t);
  for (int i = 0; i < FX_OUTPUTS; i++) out[i] = d->pending[i];
  for (int q = 0; q < nbits; q++) {
    const FxBit u = fxd_bit(d, bytes, recs.data(), lstmpr, lstmex, out, ostride, nbits, q, blpos0, lastbyte0, have0);
    if (e->rng) shuffle(e);
    for (int t = 0; t < FX_THREADS; t++) fxd_phase1a(d, sh, u, e->order[t]);
    if (d->slot_parallel) for (int k = 0; k < FX_NMAPS; k++) { e->bits_maps++; e->bits_serial += sh->mconf[u.q & 1][k] != 0; e->serial_by[k][u.bpos] += sh->mconf[u.q & 1][k] != 0; }
    for (int t = 0; t < FX_THREADS; t++) fxd_phase1c(d, sh, u, e->order[t]);
    for (int t = 0; t < FX_THREADS; t++) fxd_phase2(d, sh, u, e->order[t]);
    for (int t = 0; t < FX_THREADS; t++) fxd_phase3(d, sh, u, e->order[t]);
    for (int t = 0; t < FX_THREADS; t++) fxd_phase4(d, sh, u, e->order[t]);
    for (int t = 0; t < FX_THREADS; t++) fxd_phase5(d, sh, u, e->order[t]);
  }
  for (int t = 0; t < FX_THREADS; t++) fxd_store_shared(d, sh, t);
  d->blpos = blpos0 + n; d->lastbyte = bytes[n - 1]; d->have_rec = 1; d->rec = recs[(size_t)n - 1];
  return 0;
}
void fxe_set_serial_maps(void* h, int serial) { ((Emul*)h)->dev.slot_parallel = !serial; }
// how often a map fell back to its serial walk: [0] map-bits in total, [1] serial ones
void fxe_conflict_stats(void* h, uint64_t* out2) { out2[0] = ((Emul*)h)->bits_maps; out2[1] = ((Emul*)h)->bits_serial; }
// ... by map and bit position: out[31][8]
void fxe_conflict_by(void* h, uint64_t* out) { for (int k = 0; k < FX_NMAPS; k++) for (int b = 0; b < 8; b++) out[8 * k + b] = ((Emul*)h)->serial_by[k][b]; }
// role M's wavefront ownership (FxDev::mw_slot / mw_map), for the layout test: out[0..16] slots, out[17..33] maps, then C of every map
void fxe_wave_layout(void* h, int* out) {
  Emul* e = (Emul*)h;
  for (int i = 0; i <= FX_M_WAVES; i++) { out[i] = e->dev.mw_slot[i]; out[FX_M_WAVES + 1 + i] = e->dev.mw_map[i]; }
  for (int k = 0; k < FX_NMAPS; k++) out[2 * (FX_M_WAVES + 1) + k] = e->dev.maps[k].C;
}
void fxe_set_blpos(void* h, int blpos) { Emul* e = (Emul*)h; e->dev.blpos = blpos; fxp_set_blpos(e->parser, blpos); }
int fxe_debug(void* h, uint32_t* out) {   // the twelve mixer selectors + a few registers
  Emul* e = (Emul*)h;
  int n = 0;
  for (int i = 0; i < 12; i++) out[n++] = (uint32_t)e->dev.mx_cxt[i];
  out[n++] = e->dev.fails; out[n++] = (uint32_t)e->dev.pr;
  return n;
}
}
