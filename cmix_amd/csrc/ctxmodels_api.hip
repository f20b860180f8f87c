// ctxmodels_api.hip -- host side of the context / small-model stage (C ABI: cmx_ctxmodels_* in
// include/cmix_amd.h).
//
// Builds the per-lane configuration in the reference's construction order (Predictor::Predictor,
// src/predictor.cpp:28-36 -> AddBracket :90-98, AddWord :104-131, AddDirect :133-148, AddMatch
// :150-164, AddDoubleIndirect :166-178, AddMixers :199-356), including the structural de-duplication
// of contexts (ContextManager::AddContext, context-manager.cpp:6-12) and the rand() draw each
// Indirect constructor makes (indirect.cpp:10), allocates the tables in HBM and launches the kernel.
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/cmix_amd.h"
#include "cmx_glibc_rand.h"
#include "ctxmodels_state.h"
#include "cmx_late.h"

extern "C" __global__ void cmx_ctxmodels_kernel(const CtxDev, const uint8_t*, size_t, float*, size_t, uint32_t*, float*);
extern "C" __global__ void cmx_ctxmodels_peek_kernel(const CtxDev, const uint8_t*, float*, size_t, uint32_t*);
extern "C" __global__ void cmx_ctxmodels_late_kernel(const CtxDev, CmxLate, size_t, float*, size_t, uint32_t*, float*);
extern "C" __global__ void cmx_bytemodel_bits(const float*, const float*, const uint8_t*, size_t, float*, int*, size_t, int,
                                              float*);
extern "C" unsigned cmx_ctxmodels_lds_bytes();

void cmx_set_err(const std::string& s);  // cmx_api.hip

struct cmx_ctxmodels {
  int device = 0;
  CtxDev dev;
  std::vector<void*> allocs;
  float* d_prev_dist = nullptr;      // Bracket byte distribution at chunk start
  float* d_bracket_dist = nullptr;   // [cap][256] scratch
  size_t dist_cap = 0;
  int* d_err = nullptr;
  std::vector<CtxLane> lanes;
};

namespace {

unsigned pack_orders(const unsigned* o) {  // o[0] = count, then the orders
  unsigned v = o[0], sh = 4;
  for (unsigned i = 1; i <= o[0]; ++i) { v |= o[i] << sh; sh += 4; }
  return v;
}

int size_kind(unsigned long long size) {
  if ((size & (size - 1)) == 0) return SZ_POW2;
  if (size == 10000000ull) return SZ_10M;
  if (size == 20000000ull) return SZ_20M;
  if (size == 500000ull) return SZ_500K;
  if (size == 100000ull) return SZ_100K;
  return -1;
}

}  // namespace

extern "C" {

void cmx_ctxmodels_destroy(cmx_ctxmodels_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->allocs) (void)hipFree(p);
  if (h->d_bracket_dist) (void)hipFree(h->d_bracket_dist);
  delete h;
}

cmx_ctxmodels_t* cmx_ctxmodels_create(const uint8_t vocab[256], int device) {
  int ndev = cmx_device_count();
  if (ndev <= 0) { cmx_set_err("cmx_ctxmodels_create: no HIP device visible (a gfx950 GPU is required)"); return nullptr; }
  if (device < 0 || device >= ndev) { cmx_set_err("cmx_ctxmodels_create: bad device index"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return nullptr; }
  cmx_ctxmodels_t* h = new cmx_ctxmodels();
  h->device = device;
  bool fail = false;
  auto dalloc = [&](size_t bytes, int fill_byte) -> void* {
    void* p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { fail = true; return nullptr; }
    h->allocs.push_back(p);
    (void)hipMemsetAsync(p, fill_byte, bytes, 0);
    return p;
  };
  auto dalloc_half = [&](size_t count) -> float* {  // float table filled with 0.5f
    void* p = nullptr;
    if (hipMalloc(&p, count * 4) != hipSuccess) { fail = true; return nullptr; }
    h->allocs.push_back(p);
    (void)hipMemsetD32Async((hipDeviceptr_t)p, 0x3F000000, count, 0);
    return (float*)p;
  };

  std::vector<CtxLane>& L = h->lanes;
  L.assign(64, CtxLane{});
  for (auto& l : L) { l.ctype = CT_NONE; l.mtype = MT_NONE; l.sel_kind = SEL_ZERO; }
  GlibcRand rng(0xDEADBEEFu);  // predictor.cpp:26
  int nc = 0, nm = 0, n_ind = 0, n_match = 0;
  unsigned long long ctx_size[64] = {0};

  auto add_hash = [&](unsigned order, unsigned hs) {  // context-hash.cpp:3-7
    L[nc].ctype = CT_HASH; L[nc].hash_size = hs; ctx_size[nc] = 1ull << (hs * order); L[nc].mask = ctx_size[nc] - 1;
    return nc++;
  };
  auto add_sparse = [&](const unsigned* o) {  // sparse.cpp:5-15
    L[nc].ctype = CT_SPARSE; L[nc].orders = pack_orders(o); ctx_size[nc] = ~0ull;
    return nc++;
  };
  auto add_interval = [&](int map_id, unsigned max_value, unsigned num_bits) {  // interval.cpp:3-15
    L[nc].ctype = CT_INTERVAL; L[nc].map_id = map_id;
    unsigned shift = 1; while ((1u << shift) <= max_value) ++shift;
    L[nc].shift = shift; ctx_size[nc] = 1ull << num_bits; L[nc].mask = ctx_size[nc] - 1;
    return nc++;
  };
  auto add_direct = [&](int mtype, int ctx, int col, unsigned long long rows) {  // direct.cpp:3-13, direct-hash.cpp:3-14
    CtxLane& m = L[nm];
    m.mtype = mtype; m.mctx = ctx; m.col = col; m.limit = 30; m.divtab = 0; m.size = rows;
    m.size_kind = size_kind(rows);
    m.divisor = (float)(1.0 / (30 + 0.0f));
    m.pred = dalloc_half(rows * 256);
    m.cnt = (uint8_t*)dalloc(rows * 256, 0);
    if (mtype == MT_DIRECTHASH) m.chk = (unsigned long long*)dalloc(rows * 8, 0);
    return nm++;
  };
  auto add_indirect = [&](int ctx, int col, float delta, int run_map) {  // indirect.cpp:4-14
    CtxLane& m = L[nm];
    m.mtype = MT_INDIRECT; m.mctx = ctx; m.col = col; m.run_map = run_map; m.slot = n_ind++;
    m.divisor = (float)(1.0 / delta);
    m.offset = (unsigned long long)rng.next() % (CTX_SHARED - 257);
    return nm++;
  };
  auto add_match = [&](int ctx, int col, unsigned long long map_size) {  // match.cpp:3-15
    CtxLane& m = L[nm];
    m.mtype = MT_MATCH; m.mctx = ctx; m.col = col; m.limit = 200; m.divtab = 31; m.slot = n_match++;
    m.size = map_size; m.size_kind = size_kind(map_size);
    m.divisor = (float)(1.0 / (200 + 0.5f));
    m.map = (uint32_t*)dalloc(map_size * 4, 0);
    return nm++;
  };

  // AddBracket (predictor.cpp:90-98)
  L[nc].ctype = CT_BRACKET; ctx_size[nc] = 257 * 256; nc++;            // ctx 0 = BracketContext(256, 15)
  L[nm].mtype = MT_BRACKET; L[nm].col = 0; nm++;                        // model 0 = Bracket(200, 10, 100000)
  add_direct(MT_DIRECT, 0, 1, ctx_size[0]);
  add_indirect(0, 2, 300, 0);
  // AddWord (predictor.cpp:104-131)
  int col = 2025;
  static const unsigned P1[18][7] = {{1, 0}, {2, 0, 1}, {2, 7, 2}, {1, 7}, {1, 1}, {2, 1, 2}, {3, 1, 2, 3}, {2, 1, 3},
      {2, 1, 4}, {2, 1, 5}, {2, 2, 3}, {2, 3, 4}, {3, 1, 2, 4}, {4, 1, 2, 3, 4}, {3, 2, 3, 4}, {1, 2},
      {5, 1, 2, 3, 4, 5}, {6, 1, 2, 3, 4, 5, 6}};  // count, orders
  for (int i = 0; i < 18; ++i) { int c = add_sparse(P1[i]); add_indirect(c, col++, 200, 0); }
  // model_params2 {0},{1},{7},{1,3},{1,2,3},{7,2} are structurally equal to contexts 1,5,4,8,7,3 (sparse.cpp:24-33)
  const int P2CTX[6] = {1, 5, 4, 8, 7, 3};
  for (int i = 0; i < 6; ++i) {
    add_match(P2CTX[i], col++, 10000000ull);
    if (i == 1) {
      add_indirect(P2CTX[i], col++, 200, 1);
      add_direct(MT_DIRECTHASH, P2CTX[i], col++, 500000ull);
    }
  }
  // AddDirect (predictor.cpp:133-148)
  for (unsigned ord = 0; ord < 4; ++ord) {
    int c = add_hash(ord, 8);
    if (ord < 3) add_direct(MT_DIRECT, c, col++, ctx_size[c]);
    else add_direct(MT_DIRECTHASH, c, col++, 100000ull);
  }
  // AddMatch (predictor.cpp:150-164): (0,8),(1,8),(2,8) are contexts 19..21 again
  const unsigned PM[10][2] = {{0, 8}, {1, 8}, {2, 8}, {7, 4}, {11, 3}, {13, 2}, {15, 2}, {17, 2}, {20, 1}, {25, 1}};
  for (int i = 0; i < 10; ++i) {
    int c = i < 3 ? 19 + i : add_hash(PM[i][0], PM[i][1]);
    add_match(c, col++, ctx_size[c] < 20000000ull ? ctx_size[c] : 20000000ull);
  }
  // AddDoubleIndirect (predictor.cpp:166-178)
  const unsigned PI[11][4] = {{1, 8, 1, 8}, {2, 8, 1, 8}, {1, 8, 2, 8}, {2, 8, 2, 8}, {1, 8, 3, 8}, {3, 8, 1, 8},
      {4, 6, 4, 8}, {5, 5, 5, 5}, {1, 8, 4, 8}, {1, 8, 5, 6}, {6, 4, 6, 4}};
  for (int i = 0; i < 11; ++i) {  // indirect-hash.cpp:3-11
    CtxLane& c = L[nc];
    c.ctype = CT_INDIRECT; c.hash_size1 = PI[i][1]; c.hash_size = PI[i][3];
    const unsigned long long size1 = 1ull << (PI[i][1] * PI[i][0]);
    ctx_size[nc] = 1ull << (PI[i][3] * PI[i][2]);
    c.mask1 = (unsigned)(size1 - 1); c.mask = ctx_size[nc] - 1;
    c.ihash = (uint32_t*)dalloc(size1 * 4, 0);
    add_indirect(nc, col++, 400, 0);
    nc++;
  }
  // AddMixers contexts (predictor.cpp:199-328)
  add_hash(2, 4);                      // 41
  add_hash(3, 2);                      // 42
  add_interval(0, 10, 8);              // 43 interval1
  add_interval(1, 15, 8);              // 44 interval2
  add_interval(2, 1, 7);               // 45 interval3
  add_interval(3, 3, 10);              // 46 interval4
  add_interval(3, 3, 15);              // 47 interval5
  add_interval(3, 3, 7);               // 48 interval8
  add_interval(4, 7, 9);               // 49 interval6
  {                                    // 50 interval7 = IntervalHash(map, 8, 7, 2) (interval-hash.cpp:3-16)
    CtxLane& c = L[nc];
    c.ctype = CT_INTERVALHASH; c.map_id = 4; c.shift = 3; c.mask1 = 255; c.hash_size = 2;
    ctx_size[nc] = 1ull << 14; c.mask = ctx_size[nc] - 1; nc++;
  }
  add_interval(4, 7, 7);               // 51 interval9
  L[nc].ctype = CT_COMBINED; L[nc].orders = 1 | (0 << 4); ctx_size[nc] = 65536; nc++;  // 52: (recent[0]<<8)+recent[1]
  L[nc].ctype = CT_COMBINED; L[nc].orders = 2 | (1 << 4); ctx_size[nc] = 65536; nc++;  // 53: (recent[1]<<8)+recent[2]
  if (nc != CTX_N || nm != CTX_NM || col != 2076 || n_ind != CTX_N_INDIRECT || n_match != CTX_N_MATCH) {
    cmx_set_err("cmx_ctxmodels_create: internal layout error");
    cmx_ctxmodels_destroy(h);
    return nullptr;
  }
  // selectors (predictor.cpp:199-356). BitContext i reads context BC[i].
  {
    const int C = R_CTX;
    struct { int kind, src; } S[CTX_NSEL] = {
        {SEL_BITCTX, C + 19}, {SEL_BITCTX, C + 19}, {SEL_BITCTX, C + 20}, {SEL_BITCTX, C + 20}, {SEL_BITCTX, C + 41},
        {SEL_BITCTX, C + 42}, {SEL_PLAIN, R_RECENT + 2}, {SEL_PLAIN, R_RECENT + 3}, {SEL_PLAIN, R_ZERO},
        {SEL_PLAIN, R_LINE_BREAK}, {SEL_PLAIN, R_LONGEST_MATCH}, {SEL_PLAIN, R_WRT_CONTEXT}, {SEL_ZERO, 0},
        {SEL_PLAIN, C + 43}, {SEL_PLAIN, C + 44}, {SEL_PLAIN, C + 45}, {SEL_BITCTX, C + 45}, {SEL_PLAIN, C + 46},
        {SEL_PLAIN, C + 47}, {SEL_BITCTX, C + 48}, {SEL_PLAIN, C + 49}, {SEL_PLAIN, C + 50}, {SEL_BITCTX, C + 51},
        {SEL_BITCTX, R_RECENT + 1}, {SEL_PLAIN, C + 52}, {SEL_PLAIN, C + 53},
        {SEL_PLAIN, R_ZERO}, {SEL_PLAIN, R_ZERO}, {SEL_LBC, 0}, {SEL_LBC, 0}, {SEL_LBC, 0}, {SEL_PLAIN, R_RECENT + 0},
        {SEL_PLAIN, R_RECENT + 1}, {SEL_PLAIN, R_RECENT + 2}, {SEL_PLAIN, R_LONGEST_MATCH}, {SEL_PLAIN, R_WRT_CONTEXT},
        {SEL_PLAIN, C + 43}, {SEL_PLAIN, C + 44}, {SEL_PLAIN, C + 45}, {SEL_PLAIN, C + 46}, {SEL_PLAIN, C + 47},
        {SEL_PLAIN, C + 49}, {SEL_PLAIN, C + 50}, {SEL_BITCTX, C + 48}, {SEL_BITCTX, C + 45}, {SEL_BITCTX, C + 51},
        {SEL_PLAIN, R_ZERO}};
    for (int i = 0; i < CTX_NSEL; ++i) { L[i].sel_kind = S[i].kind; L[i].sel_src = S[i].src; }
  }

  CtxDev& D = h->dev;
  memset(&D, 0, sizeof D);
  for (int i = 0; i < 256; ++i) D.vocab[i] = vocab[i] != 0;
  D.history = (uint8_t*)dalloc(CTX_HISTORY, 0);
  D.shared_map = (uint8_t*)dalloc(CTX_SHARED, 0);
  D.bstack_cap = 1u << 26;
  D.bstack = (uint16_t*)dalloc((size_t)D.bstack_cap * 2, 0);
  h->d_err = (int*)dalloc(16, 0);
  D.err = h->d_err;
  h->d_prev_dist = (float*)dalloc(256 * 4, 0);
  {
    std::vector<float> dt(240, 0.0f);
    for (int c = 1; c <= 30; ++c) dt[c] = (float)(1.0 / (c + 0.0f));           // direct.cpp:24
    for (int c = 1; c <= 200; ++c) dt[31 + c] = (float)(1.0 / (c + 0.5f));     // match.cpp:32
    float* d = (float*)dalloc(dt.size() * 4, 0);
    if (d) (void)hipMemcpy(d, dt.data(), dt.size() * 4, hipMemcpyHostToDevice);
    D.divtabs = d;
  }
  {
    CtxLane* d = (CtxLane*)dalloc(64 * sizeof(CtxLane), 0);
    if (d) (void)hipMemcpy(d, L.data(), 64 * sizeof(CtxLane), hipMemcpyHostToDevice);
    D.lanes = d;
  }
  {
    CtxPersist* P = new CtxPersist();
    memset(P, 0, sizeof *P);
    for (int i = 0; i < 6; ++i) for (int d = 0; d < 200; ++d) { P->br_stats[i][d][0] = 1; P->br_stats[i][d][1] = 256; }  // bracket.cpp:7-8
    for (int i = 0; i < 256; ++i) P->br_probs[i] = (float)(1.0 / 256);                     // byte-model.cpp:5-6
    int slot = 0, mslot = 0;
    for (int l = 0; l < CTX_NM; ++l) {
      if (L[l].mtype == MT_INDIRECT) {
        for (int i = 0; i < 256; ++i) {
          float v = 0.5f;                                                                  // nonstationary.cpp:9-11
          if (L[l].run_map) v = i < 128 ? (float)((128.0 - i) / 256) : (float)(i / 256.0);  // run-map.cpp:17-20
          P->ipred[slot][i] = v;
        }
        ++slot;
      } else if (L[l].mtype == MT_MATCH) {
        for (int i = 0; i < 256; ++i) P->mpred[mslot][i] = (float)(0.5 + (i + 0.5) / 512);  // match.cpp:11-13
        ++mslot;
      }
    }
    CtxPersist* d = (CtxPersist*)dalloc(sizeof(CtxPersist), 0);
    if (d) (void)hipMemcpy(d, P, sizeof *P, hipMemcpyHostToDevice);
    delete P;
    D.persist = d;
  }
  if (fail) {
    cmx_set_err("cmx_ctxmodels_create: hipMalloc failed");
    cmx_ctxmodels_destroy(h);
    return nullptr;
  }
  (void)hipFuncSetAttribute((const void*)cmx_ctxmodels_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)cmx_ctxmodels_lds_bytes());
  (void)hipFuncSetAttribute((const void*)cmx_ctxmodels_peek_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)cmx_ctxmodels_lds_bytes());
  (void)hipFuncSetAttribute((const void*)cmx_ctxmodels_late_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)cmx_ctxmodels_lds_bytes());
  if (hipDeviceSynchronize() != hipSuccess) {
    cmx_set_err("cmx_ctxmodels_create: init failed");
    cmx_ctxmodels_destroy(h);
    return nullptr;
  }
  return h;
}

static int ctxmodels_launch(cmx_ctxmodels_t* h, const uint8_t* d_bytes, size_t nbytes, float* d_probs, size_t pstride,
                            uint32_t* d_sel, void* stream, int dry = 0, int only_k = -1);

int cmx_ctxmodels_run(cmx_ctxmodels_t* h, const uint8_t* d_bytes, size_t nbytes, float* d_probs, size_t pstride,
                      uint32_t* d_sel, void* stream) {
  if (!h) { cmx_set_err("cmx_ctxmodels_run: null handle"); return 1; }
  if (nbytes == 0) return 0;
  if (!d_bytes || !d_probs || !d_sel || pstride < CMX_N_INPUTS) { cmx_set_err("cmx_ctxmodels_run: bad argument"); return 1; }
  return ctxmodels_launch(h, d_bytes, nbytes, d_probs, pstride, d_sel, stream);
}

// Predictor::Pretrain (predictor.cpp:471-487) for the models this stage owns: Predict, Perceive,
// UpdateContexts and ByteUpdate exactly as in normal coding -- only nothing downstream (mixers, SSE,
// LSTM, PPMd) is trained, so no outputs are produced.
int cmx_ctxmodels_pretrain(cmx_ctxmodels_t* h, const uint8_t* d_bytes, size_t nbytes, void* stream) {
  if (!h) { cmx_set_err("cmx_ctxmodels_pretrain: null handle"); return 1; }
  if (nbytes == 0) return 0;
  if (!d_bytes) { cmx_set_err("cmx_ctxmodels_pretrain: bad argument"); return 1; }
  return ctxmodels_launch(h, d_bytes, nbytes, nullptr, CMX_N_INPUTS, nullptr, stream);
}

// Bit-synchronous mode (what a decoder needs): the 8 rows the stage would write if the next byte were *d_byte,
// leaving every piece of state untouched. Row j only depends on the top j bits of *d_byte, so with j bits coded
// (and anything below) row j is what Predict() sees for the next bit (predictor.cpp:362-369).
int cmx_ctxmodels_peek(cmx_ctxmodels_t* h, const uint8_t* d_byte, int bit_index, float* d_probs, size_t pstride,
                       uint32_t* d_sel, void* stream) {
  if (!h) { cmx_set_err("cmx_ctxmodels_peek: null handle"); return 1; }
  if (!d_byte || !d_probs || !d_sel || pstride < CMX_N_INPUTS || bit_index < -1 || bit_index > 7) {
    cmx_set_err("cmx_ctxmodels_peek: bad argument");
    return 1;
  }
  return ctxmodels_launch(h, d_byte, 1, d_probs, pstride, d_sel, stream, 1, bit_index);
}

static int ctxmodels_launch(cmx_ctxmodels_t* h, const uint8_t* d_bytes, size_t nbytes, float* d_probs, size_t pstride,
                            uint32_t* d_sel, void* stream, int dry, int only_k) {
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipStream_t st = (hipStream_t)stream;
  if (h->dist_cap < nbytes) {
    if (h->d_bracket_dist) { (void)hipDeviceSynchronize(); (void)hipFree(h->d_bracket_dist); h->d_bracket_dist = nullptr; }
    size_t cap = nbytes < 4096 ? 4096 : nbytes;
    if (hipMalloc((void**)&h->d_bracket_dist, cap * 256 * 4) != hipSuccess) { cmx_set_err("cmx_ctxmodels_run: hipMalloc failed"); return 1; }
    h->dist_cap = cap;
  }
  // the Bracket model's distribution going into the first byte of this chunk (a dry pass leaves it where it is)
  const float* prev_dist = (const float*)((const char*)h->dev.persist + offsetof(CtxPersist, br_probs));
  if (!dry) {
    (void)hipMemcpyAsync(h->d_prev_dist, prev_dist, 256 * 4, hipMemcpyDeviceToDevice, st);
    prev_dist = h->d_prev_dist;
  }
  if (dry)
    hipLaunchKernelGGL(cmx_ctxmodels_peek_kernel, dim3(1), dim3(64), cmx_ctxmodels_lds_bytes(), st, h->dev, d_bytes,
                       d_probs, pstride, d_sel);
  else
    hipLaunchKernelGGL(cmx_ctxmodels_kernel, dim3(1), dim3(64), cmx_ctxmodels_lds_bytes(), st, h->dev, d_bytes, nbytes,
                       d_probs, pstride, d_sel, h->d_bracket_dist);
  // column 0: ByteModel::Predict of the Bracket model along the known bytes (byte-model.cpp:8-37)
  if (d_probs)
    hipLaunchKernelGGL(cmx_bytemodel_bits, dim3((unsigned)nbytes), dim3(64), 0, st, prev_dist, h->d_bracket_dist,
                       d_bytes, nbytes, d_probs, (int*)nullptr, pstride, only_k, (float*)nullptr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_ctxmodels_run: ") + hipGetErrorString(e)); return 1; }
  return 0;
}

// The decoder's form of a chunk (cmx_late.h): the kernel is launched for the next nbytes bytes; their bits arrive through `box`.
// probs / sel / brk_dist ([nbytes][256]: the Bracket model's distribution after each byte) are memory that the kernels reading them
// (mixing network, ByteModel kernel) see coherently while this one runs. *brk_dist0_out: where the distribution going into the
// chunk's first byte is (valid when every earlier kernel of this stage has ended).
int cmx_ctxmodels_run_late(cmx_ctxmodels_t* h, void* box, size_t nbytes, float* probs, size_t pstride, uint32_t* sel, float* brk_dist,
                           const float** brk_dist0_out, void* stream) {
  if (!h || !box || !probs || !sel || !brk_dist || nbytes == 0 || pstride < CMX_N_INPUTS) { cmx_set_err("cmx_ctxmodels_run_late: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipStream_t st = (hipStream_t)stream;
  // (the stage's own copy in its persistent state: the kernel rewrites it only in its epilogue, long after the first byte's reader has it)
  if (brk_dist0_out) *brk_dist0_out = (const float*)((const char*)h->dev.persist + offsetof(CtxPersist, br_probs));
  hipLaunchKernelGGL(cmx_ctxmodels_late_kernel, dim3(1), dim3(64), cmx_ctxmodels_lds_bytes(), st, h->dev, *(const CmxLate*)box, nbytes, probs, pstride, sel, brk_dist);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_ctxmodels_run_late: ") + hipGetErrorString(e)); return 1; }
  return 0;
}

int cmx_ctxmodels_sync(cmx_ctxmodels_t* h) {
  if (!h) return 1;
  (void)hipSetDevice(h->device);
  if (hipDeviceSynchronize() != hipSuccess) { cmx_set_err("cmx_ctxmodels_sync: device error"); return 1; }
  int err = 0;
  (void)hipMemcpy(&err, h->d_err, 4, hipMemcpyDeviceToHost);
  if (err) { cmx_set_err("cmx_ctxmodels: bracket-context stack overflow (more than 2^26 unclosed brackets)"); return 1; }
  return 0;
}

// Test readout: how many bytes took the serial (overlapping Indirect maps) path, committed / in dry passes
int cmx_ctxmodels_debug_slow_bytes(cmx_ctxmodels_t* h, uint64_t out2[2]) {
  if (!h || !out2) return 1;
  if (cmx_ctxmodels_sync(h)) return 1;
  return hipMemcpy(out2, (const char*)h->dev.persist + offsetof(CtxPersist, slow_bytes), 16, hipMemcpyDeviceToHost) ==
                 hipSuccess ? 0 : 1;
}

// Test hook (state injection; tests/golden/make_wrap_traces.py, the twin of the reference harness's ref_debug_set_history): the stage as after `pos`
// bytes of a stream whose last n bytes were `tail` -- the history ring's write position (context-manager.cpp:24-27: modulo 100 000 000), every
// Match model's own byte counter (match.cpp:43-46: not reduced), the ring's bytes in front of the position. Between chunks only.
int cmx_ctxmodels_debug_set_history(cmx_ctxmodels_t* h, uint64_t pos, const uint8_t* tail, uint64_t n) {
  if (!h || (n && !tail) || n > CTX_HISTORY) { cmx_set_err("cmx_ctxmodels_debug_set_history: bad argument"); return 1; }
  if (cmx_ctxmodels_sync(h)) return 1;
  CtxPersist* P = new CtxPersist();
  bool ok = hipMemcpy(P, h->dev.persist, sizeof *P, hipMemcpyDeviceToHost) == hipSuccess;
  P->history_pos = pos % CTX_HISTORY;
  for (int l = 0; l < 64; ++l) P->m_history_pos[l] = pos;
  ok = ok && hipMemcpy(h->dev.persist, P, sizeof *P, hipMemcpyHostToDevice) == hipSuccess;
  delete P;
  const uint64_t first = (pos + CTX_HISTORY - n % CTX_HISTORY) % CTX_HISTORY;   // ring index of tail[0]
  const uint64_t n1 = n < CTX_HISTORY - first ? n : CTX_HISTORY - first;
  if (n1) ok = ok && hipMemcpy(h->dev.history + first, tail, n1, hipMemcpyHostToDevice) == hipSuccess;
  if (n > n1) ok = ok && hipMemcpy(h->dev.history, tail + n1, n - n1, hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { cmx_set_err("cmx_ctxmodels_debug_set_history: device error"); return 1; }
  return 0;
}

// Test readout in the layout the parity tests use for ContextManager state: regs25, ctx54, bitctx8
int cmx_ctxmodels_get_manager(cmx_ctxmodels_t* h, uint64_t* regs25, uint64_t* ctx54, uint64_t* bitctx8) {
  if (!h) return 1;
  if (cmx_ctxmodels_sync(h)) return 1;
  CtxPersist* P = new CtxPersist();
  (void)hipMemcpy(P, h->dev.persist, sizeof *P, hipMemcpyDeviceToHost);
  int n = 0;
  regs25[n++] = 1;  // bit_context_ is 1 between bytes (predictor.cpp:468)
  regs25[n++] = 1;  // long_bit_context_
  regs25[n++] = P->regs[R_ZERO];
  regs25[n++] = P->history_pos;
  regs25[n++] = P->regs[R_LINE_BREAK];
  regs25[n++] = P->regs[R_LONGEST_MATCH];
  regs25[n++] = 0;  // auxiliary_context_: produced by the mixing-network stage
  regs25[n++] = P->regs[R_WRT_CONTEXT];
  regs25[n++] = P->wrt_state;
  for (int i = 0; i < 8; ++i) regs25[n++] = P->regs[R_RECENT + i];
  for (int i = 0; i < 8; ++i) regs25[n++] = P->regs[R_WORDS + i];
  for (int i = 0; i < CTX_N; ++i) ctx54[i] = P->regs[R_CTX + i];
  const int BC[8] = {19, 20, 41, 42, 45, 48, 51, -1};
  for (int i = 0; i < 8; ++i) {
    uint64_t b = BC[i] >= 0 ? P->regs[R_CTX + BC[i]] : P->regs[R_RECENT + 1];
    bitctx8[i] = P->bytes_done ? (b << 8) + 1 : 0;
  }
  delete P;
  return 0;
}

}  // extern "C"
