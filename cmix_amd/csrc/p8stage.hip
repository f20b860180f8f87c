// p8stage.hip -- the paq8 stage of the pipeline: layer-0 columns 434..2024 of the cmix predictor, i.e. the 1591 values
// PAQ8::Predict() returns per bit (reference src/models/paq8.cpp: Predictor::update :8248-8362 around contextModel2
// :8101-8207; wired at src/predictor.cpp:85-97). Two halves:
//   * host:   the front end (p8front/: the 15 sub-models' parsers and contexts, everything that is a function of the
//             byte stream alone) runs on the calling thread and leaves, per chunk, hashed contexts per byte and one op
//             word per small map and step in page-locked memory (p8_rec.h);
//   * device: every table that learns, as role kernels over the chunk's known bits, all writing one int16 row of 1552
//             mixer inputs per step into HBM:
//               cm2     3 x ContextMap2 (p8cm2_dev.h): contextModel2's order-N map (whose return value, the "order",
//                       feeds the family and the mixer selectors), TextModel's, exeModel's
//               fam     the ContextMap family, 204 contexts of 16 instances with the shared rnd() stream (p8cm_dev.h)
//               lanes   one wavefront: every small learner, one lane each (p8stage_dev.h)
//               dmc     the DMC forest (p8dmc_dev.h)
//               mix     the 1552 x 28 int16 mixer (rows in registers from dot product to training), the second layer,
//                       the APM / APM1 chains, squash(x)/4095 export of all 1591 values into the layer-0 matrix
//             Streams: s_d = upload, cm2[0]; s_a = fam (after cm2[0] of the same chunk, beside cm2[0] of the next); s_b = cm2[1];
//             s_e = cm2[2]; s_c = lanes (after cm2[0]); s_f = dmc (bits only); s_m = mix after all.
// Integer work, latency-bound by construction (dependent table accesses per bit); algorithmic HBM traffic per input
// byte: mixer 28 rows x 1552 x 2 B x 2 (read + write) x 8 = 1.39 MB, buckets ~270 contexts x 3 probes x 64 B x 2 = 0.1 MB.
// Parity: tests/test_p8stage_host.py runs the bodies on the host, tests/test_zgpu_p8stage.py the kernels, both against
// columns 434..2024 of traces of the unmodified reference.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/cmix_amd.h"
#include "p8front/p8f_front.h"
#include "p8cm2v2_dev.h"
#include "p8stage_build.h"
#include "cmx_late.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip
extern "C" int cmx_device_count(void);
extern "C" int cmx_make_stream(hipStream_t* st, int which);   // cmx_api.hip: a stage's kernel stream (compute-unit mask when the mixing network owns an XCD)

// ---------------------------------------------------------------- role kernels
// skip: chunk-local steps before the stream's first byte boundary (the maps have no contexts yet, reference :1072 loop over cn == 0)
// model (TextModel's and exeModel's maps in a chunk that holds image-model bytes, else nullptr): a step of an image model does not call these
// maps -- their state, their own partial byte included (ContextMap2::mix :1305-1309), stays where it was; only the bit before their
// next call is the stream's.
__global__ __launch_bounds__(P8CM2_MAXC) void cmx_p8s_cm2v2_kernel(P8Cm2Dev* d, const uint32_t* ctx, const uint16_t* chk, const uint8_t* bits, int16_t* x,
                                                                  uint8_t* order_out, int nbits, int skip, const uint8_t* model) {
  __shared__ __attribute__((aligned(16))) P8Cm2V2Shared sh;
  const int i = threadIdx.x, C = d->C;
  p8c2_load(d, &sh, i, P8CM2_MAXC);
  uint32_t run_bits = d->bits;
  int last_y = d->last_y, lk = 0;
  __syncthreads();
  if (i < C) p8c2_reload(d, &sh, i);
  __syncthreads();
  for (int t = 0; t < nbits; t++) {
    if (model && model[t >> 3]) { last_y = bits[t]; continue; }
    const P8Cm2Bit u = p8d_bit(d, ctx, chk, bits, x, t, &run_bits, &last_y);
    if (t < skip) { if (order_out && i == 0) order_out[t] = 0; continue; }
    const bool look = u.bpos == 0 || u.bpos == 2 || u.bpos == 5;
    P8Cm2Tmp tmp;
    if (look) {
      ++lk;
      if (i < C) p8c2_phase1(d, &sh, u, lk, i, &tmp);
      __syncthreads();
    }
    if ((look && sh.conf[lk & 1]) || sh.shared || !d->slot_parallel) {   // slot_parallel == 0: A/B switch, always the serial walk
      if (i == 0) p8c2_walk(d, &sh, u, look);
      __syncthreads();
      if (i < C) p8c2_reload(d, &sh, i);
    } else if (i < C) p8c2_run(d, &sh, u, i, &tmp);
    __syncthreads();
    if (order_out && i == 0) { int o = 0; for (int k = 0; k < C; k++) o += sh.base.nz[k]; order_out[t] = (uint8_t)o; }
    if (look) p8c2_clear_next(&sh, lk, i, P8CM2_MAXC);
  }
  __syncthreads();
  if (i == 0) { d->regs = sh.base.r; d->bits = run_bits; d->last_y = last_y; }
}

// ---- the role kernels for a DECODER (cmx_late.h): the same step functions, but the bit before a step arrives through the box
// (thread 0 waits for it, one barrier hands it to the workgroup), the host records are read from host-mapped memory, the input rows
// go to memory every kernel of the stream sees, and every role counts the rows it has completed.
#define P8_LATE_Y(t_)                                             \
  if (threadIdx.x == 0) late_y_s = late_y(B, (t_));               \
  __syncthreads();                                                \
  const int y_ = late_y_s;                                        \
  if (y_ < 0) return;                                             \
  __syncthreads()
__global__ __launch_bounds__(P8CM2_MAXC) void cmx_p8s_cm2v2_late_kernel(P8Cm2Dev* d, CmxLate B, int counter, const uint32_t* ctx, const uint16_t* chk, int16_t* x,
                                                                       uint8_t* order_out, int nbits, int skip) {
  __shared__ __attribute__((aligned(16))) P8Cm2V2Shared sh;
  __shared__ int late_y_s;
  __shared__ uint32_t ctx_s[P8CM2_MAXC];   // the byte's hashed contexts / checksums: host records that arrive while the kernel runs, read ONCE with
  __shared__ uint16_t chk_s[P8CM2_MAXC];   // loads the compiler cannot merge, hoist or turn into cached scalar loads (volatile), kept here for the byte
  const int i = threadIdx.x, C = d->C;
  p8c2_load(d, &sh, i, P8CM2_MAXC);
  uint32_t run_bits = d->bits;
  int lk = 0;
  __syncthreads();
  if (i < C) p8c2_reload(d, &sh, i);
  __syncthreads();
  for (int t = 0; t < nbits; t++) {
    P8_LATE_Y(t);
    if ((t & 7) == 0) {
      if (i < C) { ctx_s[i] = *(volatile const uint32_t*)(ctx + (size_t)(t >> 3) * C + i); chk_s[i] = *(volatile const uint16_t*)(chk + (size_t)(t >> 3) * C + i); }
      __syncthreads();
    }
    P8Cm2Bit u = p8d_bit_y(d, ctx, chk, y_, x, t, &run_bits);
    u.ctx = ctx_s; u.chk = chk_s;
    if (t < skip) { if (i == 0) { if (order_out) order_out[t] = 0; late_publish(B, counter, (uint32_t)(t + 1)); } continue; }
    const bool look = u.bpos == 0 || u.bpos == 2 || u.bpos == 5;
    P8Cm2Tmp tmp;
    if (look) {
      ++lk;
      if (i < C) p8c2_phase1(d, &sh, u, lk, i, &tmp);
      __syncthreads();
    }
    if ((look && sh.conf[lk & 1]) || sh.shared || !d->slot_parallel) {
      if (i == 0) p8c2_walk(d, &sh, u, look);
      __syncthreads();
      if (i < C) p8c2_reload(d, &sh, i);
    } else if (i < C) p8c2_run(d, &sh, u, i, &tmp);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY wave's stores have landed before the barrier (s_barrier alone does not wait for them) -- then one thread counts the row
    __syncthreads();
    if (i == 0) {
      if (order_out) { int o = 0; for (int k = 0; k < C; k++) o += sh.base.nz[k]; order_out[t] = (uint8_t)o; }
      late_publish(B, counter, (uint32_t)(t + 1));
    }
    if (look) p8c2_clear_next(&sh, lk, i, P8CM2_MAXC);
  }
  P8_LATE_Y(nbits);   // the chunk's last bit: the state a later chunk starts from
  if (i == 0) { d->regs = sh.base.r; d->bits = run_bits; d->last_y = y_; }
}

// The family kernel (p8fam_dev.h): per-context bytes and StateMaps in LDS, one barrier per bit on the common path, overlaps resolved
// in rounds over the instance order. 512 threads: the (up to 256) contexts on 8 wavefronts of 32 lanes -- a context's walk through
// phase 1 / run is a chain of data-dependent branches (lookup bit or not, hit / replace, second visit, draw), and lanes of one
// wavefront that take different branches run them one after the other, so fewer lanes per wavefront cost less --; 24 more lanes of the
// last wavefront keep the rnd() ring filled.
constexpr int P8FAM_THREADS = 512;
// LATE (a decoder): B is the box, `bits` is unused; the order-N map's value of a byte's first step is awaited on its counter.
template <bool LATE>
__device__ __forceinline__ void p8s_fam2_body(P8CmDev* d, P8FamHome* home, const uint32_t* ctx, const uint16_t* chk, const uint8_t* bits, int16_t* x,
                                              const uint8_t* order, int nbits, int skip, unsigned long long* prof, CmxLate B) {
  __shared__ int late_y_s;
  __shared__ uint32_t fctx_s[LATE ? P8CM_MAXS : 1];
  __shared__ uint16_t fchk_s[LATE ? P8CM_MAXS : 1];
  extern __shared__ __attribute__((aligned(16))) unsigned char p8f_smem[];
  P8FamShared& sh = *(P8FamShared*)p8f_smem;
  const int tid = threadIdx.x, S = d->nslots, ninst = d->ninst;
  const int wv = tid >> 6, ln = tid & 63;
  const int sl = ln < 32 ? 32 * wv + ln : P8CM_MAXS;          // this thread's context slot (>= S: none)
  const int rl = (wv == 7 && ln >= 32 && ln < 56) ? ln - 32 : -1;   // refill lane 0..23
  unsigned long long pc0 = __builtin_readcyclecounter();
#define P8F_TICK(k) do { if (prof && tid == 0) { const unsigned long long c_ = __builtin_readcyclecounter(); __hip_atomic_fetch_add(&prof[8 * u.bp + (k)], c_ - pc0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pc0 = c_; } } while (0)
#define P8F_COUNT(i, v) do { if (prof && tid == 0) __hip_atomic_fetch_add(&prof[(i)], (unsigned long long)(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); } while (0)
  p8f_load(d, home, d->sm, &sh, tid, P8FAM_THREADS);
  P8FamRun frun; frun.last_y = d->last_y; frun.c1 = d->c1; frun.lk = 0; frun.c0 = 1; frun.bits8 = 0; frun.order = 0; frun.nslots = d->nslots; frun.row_stride = d->row_stride;
  P8FamTmp tmp;   // (its per-context constants are filled once)
  tmp.tab = nullptr; tmp.mask = 0; tmp.inst = 0; tmp.off = 0;
  if (sl < S) p8f_lane(d, sl, &tmp);
  const int order_slot = d->order_slot;
  uint32_t rnd_i = (uint32_t)d->rnd.i, prev_i = rnd_i;
  uint32_t my_cx = 0, nx_cx = 0; uint16_t my_ck = 0, nx_ck = 0; bool have_nx = false, have_cur = false;
  __syncthreads();
  for (int t = 0; t < nbits; t++) {
    if (LATE) {
      if (tid == 0) { int y = late_y(B, t); if (y >= 0 && (t & 7) == 0 && order && !late_wait_cnt(B, LC_CM2_0, (uint32_t)(t + 1))) y = -1; late_y_s = y; }
      __syncthreads();
      const int y = late_y_s;
      if (y < 0) return;
      __syncthreads();
      if (t > 0) p8f_uni_tail(&frun, (t - 1) & 7, y); else frun.last_y = y;
    }
    P8FamUni u = LATE ? p8f_uni_head(ctx, chk, x, nullptr, t, &frun, rnd_i) : p8f_uni_inc(d, ctx, chk, bits, x, order, t, &frun, rnd_i);
    if (LATE) {
      // a decoder: the byte's hashed contexts / checksums (host records that arrived with the last bit) and the order-N map's value of the step (written by
      // its kernel a moment ago) are read ONCE, with loads the compiler cannot merge, hoist or turn into cached scalar loads, and kept in LDS for the byte
      if ((t & 7) == 0) {
        for (int q = tid; q < S; q += P8FAM_THREADS) { fctx_s[q] = *(volatile const uint32_t*)(ctx + (size_t)(t >> 3) * S + q); fchk_s[q] = *(volatile const uint16_t*)(chk + (size_t)(t >> 3) * S + q); }
        if (order) frun.order = (int)*(volatile const uint8_t*)(order + t);
        __syncthreads();
      }
      u.order = frun.order; u.ctx = fctx_s; u.chk = fchk_s;
    }
    const int lk = frun.lk;
    if (t < skip) { if (LATE && tid == 0) late_publish(B, LC_FAM, (uint32_t)(t + 1)); continue; }
    P8F_TICK(0);
    // the context's hash / checksum of the byte: read once per byte -- the next byte's during bit 6, an ordinary bit, so that no lookup bit
    // starts with a global round trip before it can even compute its bucket's address
    if (sl < S) {
      if (u.bp == 0 && have_nx) { my_cx = nx_cx; my_ck = nx_ck; }
      else if (u.bp == 0 || !have_cur) {   // (a chunk's first byte; a stream's first step is bit 1; a decoder: every byte)
        my_cx = p8f_ctx(d, u, sl); my_ck = p8f_chk(d, u, sl);
      }
      have_cur = true;
      if (u.bp == 0) have_nx = false;
      tmp.cx = my_cx; tmp.ck = my_ck;
      if (!LATE && u.bp == 6 && t + 2 < nbits) {
        const size_t nb_ = (size_t)((t >> 3) + 1) * (size_t)S;
        const int ord_ = order ? order[t + 2] : 0;
        nx_cx = sl == order_slot ? d->order_ctx[ord_] : ctx[nb_ + sl];
        nx_ck = sl == order_slot ? d->order_chk[ord_] : chk[nb_ + sl];
        have_nx = true;
      }
    }
    if (sl < S) p8f_phase1(d, &sh, u, sl, &tmp);
    if (rl >= 0)   // 24 lanes of one wavefront in lockstep, a group of 24 values per iteration
      P8F_REFILL(&sh, prev_i, rnd_i, rl);
    P8F_TICK(1);
    __syncthreads();
    P8F_TICK(2);
    const bool look = u.bp == 0 || u.bp == 2 || u.bp == 5;
    int total;
    const bool all_serial = !d->slot_parallel;   // A/B switch (CMX_P8CM_SERIAL=1): every instance walked by its first lane
    if (!(look && sh.anyconf[lk & 1]) && !sh.anyshared && !all_serial) {
      if (sl < S) p8f_run(d, &sh, u, sl, &tmp, p8f_count(&sh, t, 0, sl));
      total = p8f_count(&sh, t, 0, S);
    } else {
      // An overlap (or two contexts sitting on one slot). The walked instances first, in order -- at a lookup bit only their contexts that
      // share a key (p8f_miniwalk), which leaves sh.db exact -- then every other context on its lane in ONE pass, its rank a prefix count of db.
      P8F_COUNT(64 + u.bp, 1);
      for (int k = 0; k < ninst; k++) {
        const bool walked = (look && sh.conflict[lk & 1][k]) || sh.shared[k] || all_serial;
        if (!walked) continue;
        const int first = d->inst[k].first, cnt = d->inst[k].count;
        const bool whole = !look || sh.shared[k] || all_serial;
        P8F_COUNT(72 + u.bp, 1);
        if (!whole && sl >= first && sl < first + cnt) p8f_register(&sh, u, k, sl, &tmp);
        __syncthreads();
        if (sl == first) { uint32_t full; (void)p8f_miniwalk(d, &sh, u, k, p8f_count(&sh, t, 0, first), whole, d->slot_parallel == 2, &full); sh.wfull[k] = (uint8_t)full; }
        __syncthreads();
      }
      if (sl < S) {
        const int k = tmp.inst;
        const bool walked = (look && sh.conflict[lk & 1][k]) || sh.shared[k] || all_serial;
        if (walked && (sh.wfull[k] || sh.ink[sl] == 2)) p8f_reload(d, &sh, sl);
        else p8f_run(d, &sh, u, sl, &tmp, p8f_count(&sh, t, 0, sl));
      }
      total = p8f_count(&sh, t, 0, S);
      __syncthreads();   // every context has its new registers
      if (look && sl < S && sl == d->inst[d->slot_inst[sl]].first) {   // slots change hands at lookup bits only
        const int k = d->slot_inst[sl];
        if (sh.conflict[lk & 1][k] || sh.shared[k] || all_serial) sh.shared[k] = (uint8_t)p8f_shares(d, &sh, k, !(sh.shared[k] || all_serial));
      }
      __syncthreads();
      if (tid == 0) { int anys = 0; for (int q = 0; q < ninst; q++) anys |= sh.shared[q]; sh.anyshared = (uint32_t)anys; }
      __syncthreads();
    }
    P8F_TICK(3);
    if (look) p8f_clear_next(&sh, lk, tid, P8FAM_THREADS);
    prev_i = rnd_i; rnd_i += (uint32_t)total;
    P8F_COUNT(80 + u.bp, 1);
    P8F_TICK(4);
    if (LATE) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY wave's stores have landed before the barrier (s_barrier alone does not wait for them) -- then one thread counts the row
      __syncthreads();   // every context's inputs of the step are in the row
      if (tid == 0) late_publish(B, LC_FAM, (uint32_t)(t + 1));
    }
  }
#undef P8F_TICK
#undef P8F_COUNT
  if (LATE && nbits > 0) {   // the chunk's last bit goes into the run registers a later chunk starts from
    if (tid == 0) late_y_s = late_y(B, nbits);
    __syncthreads();
    if (late_y_s < 0) return;
    p8f_uni_tail(&frun, (nbits - 1) & 7, late_y_s);
  }
  __syncthreads();
  p8f_store(d, home, d->sm, &sh, rnd_i, tid, P8FAM_THREADS);
  if (tid == 0) { d->last_y = frun.last_y; d->c1 = frun.c1; }
}
__global__ __launch_bounds__(P8FAM_THREADS) void cmx_p8s_fam2_kernel(P8CmDev* d, P8FamHome* home, const uint32_t* ctx, const uint16_t* chk, const uint8_t* bits, int16_t* x,
                                                                    const uint8_t* order, int nbits, int skip, unsigned long long* prof) {
  p8s_fam2_body<false>(d, home, ctx, chk, bits, x, order, nbits, skip, prof, CmxLate());
}
__global__ __launch_bounds__(P8FAM_THREADS) void cmx_p8s_fam2_late_kernel(P8CmDev* d, P8FamHome* home, CmxLate B, const uint32_t* ctx, const uint16_t* chk, int16_t* x,
                                                                         const uint8_t* order, int nbits, int skip) {
  p8s_fam2_body<true>(d, home, ctx, chk, nullptr, x, order, nbits, skip, nullptr, B);
}

// t0: 1 for the chunk that starts the stream (there is no step 0), else 0
// The (up to 64) learners are independent of each other and of six different kinds (p8s_lane_step switches on the lane's kind): on one
// wavefront the kinds run one after the other every step. 8 wavefronts of 8 learners each keep most kinds in wavefronts of their own.
constexpr int P8LANES_THREADS = 512;
// model (a chunk that holds bytes of an image / audio model; else nullptr): in such a byte's steps only the lanes the model calls (P8Lane.modes:
// the common prefix; recordModel's maps under the audio models) run; the others neither run nor write (their positions may be the model's).
__global__ __launch_bounds__(P8LANES_THREADS) void cmx_p8s_lanes_kernel(P8LanesDev* d, const uint32_t* ops, const uint8_t* bits, const uint8_t* order, int16_t* x,
                                                                       int nbits, int t0, const uint8_t* model) {
  const int ln = threadIdx.x & 63, l = ln < 8 ? 8 * (int)(threadIdx.x >> 6) + ln : P8_NLANE;
  if (l >= P8_NLANE) return;
  P8LaneRegs r = d->regs[l];
  const int last_y = d->last_y;
  int c0 = 1;   // the partial byte (chunks are whole bytes; a stream's first step is bit 1 of byte 0: c0 = 1 as well)
  for (int t = t0; t < nbits; t++) {
    const int y = t ? bits[t - 1] : last_y;
    const int md = model ? model[t >> 3] : 0;
    c0 = (t & 7) ? c0 * 2 + y : 1;
    if (l < d->nlanes) p8s_glane_step(d, &r, l, ops + (size_t)t * P8_NLANE + l, y, order[t], t & 7, c0, x + (size_t)t * P8_NX, md);
  }
  d->regs[l] = r;
  if (l == 0) d->last_y = bits[nbits - 1];
}

__global__ __launch_bounds__(P8DMC_THREADS) void cmx_p8s_dmc_kernel(P8DmcDev* d, const uint8_t* bits, int16_t* x, int off, int nbits, int t0, const uint8_t* model) {
  __shared__ P8DmcShared sh;
  const int tid = threadIdx.x;
  const uint32_t done = d->bits_done;
  const int last_y = d->last_y;
  for (int t = t0; t < nbits; t++) {
    if (model && model[t >> 3]) continue;   // a step of an image model does not call the forest (uniform: no barrier is skipped by part of the workgroup)
    const int y = t ? bits[t - 1] : last_y;
    p8d_dmc_step1(d, &sh, tid, y);
    __syncthreads();
    p8d_dmc_step2(d, &sh, tid, (int)((done + (uint32_t)t) & 7), x + (size_t)t * P8_NX + off);
    __syncthreads();
    p8d_dmc_step3(d, &sh, tid);
    __syncthreads();
  }
  if (tid == 0) { d->last_y = bits[nbits - 1]; d->bits_done = done + (uint32_t)nbits; }
}

__global__ __launch_bounds__(P8LANES_THREADS) void cmx_p8s_lanes_late_kernel(P8LanesDev* d, CmxLate B, const uint32_t* ops, const uint8_t* order, int16_t* x, int nbits, int t0) {
  __shared__ int late_y_s;
  const int ln = threadIdx.x & 63, l = ln < 8 ? 8 * (int)(threadIdx.x >> 6) + ln : P8_NLANE;
  const bool act = l < P8_NLANE;
  P8LaneRegs r = d->regs[act ? l : 0];
  int c0 = 1;
  for (int t = 0; t < nbits; t++) {
    if (threadIdx.x == 0) late_y_s = late_y(B, t);
    __syncthreads();
    const int y = late_y_s;
    if (y < 0) return;
    __syncthreads();
    c0 = (t & 7) ? c0 * 2 + y : 1;
    if (t >= t0 && act && l < d->nlanes) {
      const uint32_t op = *(volatile const uint32_t*)(ops + (size_t)t * P8_NLANE + l);
      int ord = 0;
      if ((op & P8OP_SET) && (op & P8OP_ORDER)) {   // the one kind of op that reads the order-N map's value of the SAME step: only that lane waits for its kernel
        if (!late_wait_cnt(B, LC_CM2_0, (uint32_t)(t + 1))) ord = 0;
        else ord = (int)*(volatile const uint8_t*)(order + t);
      }
      const uint32_t op2[2] = {op, l + 1 < P8_NLANE ? *(volatile const uint32_t*)(ops + (size_t)t * P8_NLANE + l + 1) : 0u};
      p8s_glane_step(d, &r, l, op2, y, ord, t & 7, c0, x + (size_t)t * P8_NX, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY wave's stores have landed before the barrier (s_barrier alone does not wait for them) -- then one thread counts the row
    __syncthreads();
    if (threadIdx.x == 0) late_publish(B, LC_LANES, (uint32_t)(t + 1));
  }
  if (threadIdx.x == 0) late_y_s = late_y(B, nbits);
  __syncthreads();
  if (late_y_s < 0) return;
  if (act) d->regs[l] = r;
  if (l == 0) d->last_y = late_y_s;
}

__global__ __launch_bounds__(P8DMC_THREADS) void cmx_p8s_dmc_late_kernel(P8DmcDev* d, CmxLate B, int16_t* x, int off, int nbits, int t0) {
  __shared__ P8DmcShared sh;
  __shared__ int late_y_s;
  const int tid = threadIdx.x;
  const uint32_t done = d->bits_done;
  for (int t = 0; t < nbits; t++) {
    P8_LATE_Y(t);
    if (t >= t0) {
      p8d_dmc_step1(d, &sh, tid, y_);
      __syncthreads();
      p8d_dmc_step2(d, &sh, tid, (int)((done + (uint32_t)t) & 7), x + (size_t)t * P8_NX + off);
      __syncthreads();
      p8d_dmc_step3(d, &sh, tid);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY wave's stores have landed before the barrier (s_barrier alone does not wait for them) -- then one thread counts the row
      __syncthreads();
    }
    if (tid == 0) late_publish(B, LC_DMC, (uint32_t)(t + 1));
  }
  P8_LATE_Y(nbits);
  if (tid == 0) { d->last_y = y_; d->bits_done = done + (uint32_t)nbits; }
}

// ---------------------------------------------------------------- mixer + APM chains + export
namespace {
constexpr int MX_THREADS = 448, MX_GROUPS = P8_NX / 8;   // 7 waves x 4 sets; 194 groups of 8 int16
__device__ __forceinline__ int sat16(int v) { return v > 32767 ? 32767 : v < -32768 ? -32768 : v; }
__device__ __forceinline__ int lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (int)(int16_t)(v >> 16); }
typedef short mx_s2 __attribute__((ext_vector_type(2)));
// one dword = one pair of int16: ((t0*w0 + t1*w1) >> 8), the sum wrapping as pmaddwd's does (:403-413): v_dot2_i32_i16 + a shift
__device__ __forceinline__ uint32_t pair_dot(uint32_t t, uint32_t w) {
  const int s = __builtin_amdgcn_sdot2(__builtin_bit_cast(mx_s2, t), __builtin_bit_cast(mx_s2, w), 0, false);
  return (uint32_t)(s >> 8);
}
// train (:415-430) on a pair, in packed 16-bit math: w += ((sat(2 t) * err >> 16) + 1 sat >> 1), saturating -- paddsw /
// pmulhw / psraw map onto v_pk_add_i16 clamp, two 24-bit multiplies + v_perm, v_pk_ashrrev_i16 (7 instructions per pair)
__device__ __forceinline__ uint32_t pair_train(uint32_t t, uint32_t w, int err) {
  const mx_s2 tv = __builtin_bit_cast(mx_s2, t), wv = __builtin_bit_cast(mx_s2, w);
  const mx_s2 v = __builtin_elementwise_add_sat(tv, tv);
  mx_s2 r;
  r.x = (short)(__mul24((int)v.x, err) >> 16);
  r.y = (short)(__mul24((int)v.y, err) >> 16);
  const mx_s2 one = {1, 1};
  r = __builtin_elementwise_add_sat(r, one) >> 1;
  r = __builtin_elementwise_add_sat(r, wv);
  return __builtin_bit_cast(uint32_t, r);
}
}  // namespace

// Pointers read out of structures are "generic" to the compiler: it has to use flat_load / flat_store for them, which count
// on BOTH memory counters (they might address LDS), so an LDS-only barrier would still wait for every outstanding table
// access. Everything the kernels below reach through such pointers is HBM: say so.
#define MX_GLOBAL __attribute__((address_space(1)))
typedef uint32_t mx_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 mx_gload4(const MX_GLOBAL int16_t* row, int grp) {   // 16-byte group grp of a weight row
  const mx_u4 v = reinterpret_cast<const MX_GLOBAL mx_u4*>(row)[grp];
  return make_uint4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void mx_gstore4(MX_GLOBAL int16_t* row, int grp, uint4 v) {
  mx_u4 o; o.x = v.x; o.y = v.y; o.z = v.z; o.w = v.w;
  reinterpret_cast<MX_GLOBAL mx_u4*>(row)[grp] = o;
}

// A workgroup barrier that orders LDS traffic only. __syncthreads() also waits for every outstanding global load and
// store of the wave (s_waitcnt vmcnt(0)), which would serialise the one-bit-ahead requests below with the barriers of
// the bit; the kernel's cross-lane traffic through global memory is handled where it occurs.
__device__ __forceinline__ void mx_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// The mixer kernel. out: row t of the caller's matrix (ld floats per row) receives the 1591 values before bit t; first: chunk-local steps
// that belong to the stream's first byte (compacted input vector, P8Layout.first_map). Everything a bit needs from HBM / L2 is requested
// one bit ahead:
//   * weight rows: the 28 selectors of bit t+1 are known while bit t is being mixed (host part + the order-N map's
//     value; only set 26 needs this bit's final probability): the rows of t+1 are loaded into a second register set
//     right after the dot products of t were issued; a row that bit t trains and bit t+1 selects again is taken from the
//     trained registers instead (rows of different sets never coincide: cumulative bases, :548-551);
//   * the input row of t+1 (3104 B) the same way, LDS double-buffered;
//   * the APM / APM1 chains (:8281-8358): the 24 / 33 cells of each table's context row are fetched into LDS at the top
//     of the bit by the lane that owns the table (the contexts are the host's, plus the miss history); the cell(s) a
//     table has to update were read one bit earlier and are kept in registers, so the update is a store, not a
//     load-modify-store; the chain itself then runs on LDS.
namespace {
struct MxApmLane {       // one lane per table: j = 0..3 first group, 4..6 second group
  int idx;               // cell index chosen by the previous lookup (APM: cxt; APM1: index)
  uint32_t v0, v1;       // its value(s) as of the last lookup (APM uses v0 only)
};
__device__ __forceinline__ uint32_t apm_upd(uint32_t p0, int y, int limit) {   // StateMap32 cell update (APM::p :699-703)
  const int n = p0 & 1023, q = p0 >> 10;
  if (n < limit) ++p0; else p0 = (p0 & 0xfffffc00u) | (uint32_t)limit;
  p0 += ((uint32_t)(((y << 22) - q) >> 3) * (uint32_t)(16384 / (n + n + 3))) & 0xfffffc00u;
  return p0;
}
__device__ __forceinline__ uint32_t apm1_upd(uint32_t v, int y, int rate) {   // APM1 cell update :611-613
  const int g = (y << 16) + (y << rate) - y - y;
  return (uint16_t)(v + ((g - (int)v) >> rate));
}
}  // namespace

// ---- the mixer on FOUR workgroups ------------------------------------------------------------------------------------------------
// The 28 first-layer weight sets are independent of each other: a set needs the step's inputs, its selector and -- to train -- the coded
// bit and its OWN output. Only set 26 (selected by the previous step's final probability) and everything behind the first layer
// (second layer, APM chains, export) need the chain's result. So workgroup 0 keeps one set per wavefront (sets 2, 6, .., 26), the second
// layer, the chains and the export, and workgroups 1..3 run the other 21 sets -- one per wavefront, its row in registers from the dot
// product to the training step as in the one-workgroup kernel -- and hand their 12-bit outputs over through a per-step exchange row
// (value | tag words, agent scope). They need nothing back, so they run ahead of workgroup 0 as far as the input rows let them: no
// hand-off on anybody's serial path, and the VALU work of a step (28 x 1552 int16 MACs + as many training operations, which bound the
// one-workgroup kernel: 8 k + 7.5 k of its 19.1 k clocks per bit) is spread over four compute units.
// BLK = workgroup; it owns set 4 wave + QSEL of every wavefront, QSEL = (BLK + 2) & 3 (so that set 26 = 4 * 6 + 2 stays with workgroup 0).
constexpr unsigned MX4_SPIN = 1u << 24;
// an exchange word = tag << 12 | the 12-bit output; tag = launch number (24 bits, of this stage) << 28 | step + 1 (a chunk is at most 2^27 steps)
__device__ __forceinline__ unsigned long long mx4_tag(unsigned epoch, int t) { return ((unsigned long long)(epoch & 0xFFFFFFu) << 28) | (unsigned long long)(unsigned)(t + 1); }
template <int BLK>
__device__ __forceinline__ void mx4_body(const P8MixDev* M, P8TailDev* T, const int16_t* __restrict__ x, const int32_t* __restrict__ sel,
                                         const P8ApmRec* __restrict__ apm, const uint8_t* __restrict__ order, const uint8_t* __restrict__ bits,
                                         float* __restrict__ out, size_t ld, int nbits, int t0, int first, int last_y, unsigned long long* prx,
                                         unsigned epoch, unsigned* fail) {
  constexpr int QSEL = (BLK + 2) & 3;
  constexpr bool MAIN = BLK == 0;
  __shared__ __attribute__((aligned(16))) uint32_t xs[2][P8_NX / 2];   // the step's inputs as pairs, double-buffered
  __shared__ float outs[MAIN ? P8_NOUT : 1];
  __shared__ int pr_s[32], res_s[8];
  __shared__ uint32_t st_s[16];
  __shared__ uint32_t arow[MAIN ? 7 : 1][36];    // the context rows of the seven chain tables (24 u32 or 33 u16 cells, widened)
  __shared__ int p_s, fin_s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int si = 4 * wave + QSEL;                  // this wavefront's weight set
  __shared__ int16_t squash[4096], stretch[MAIN ? 4096 : 1];
  for (int i = threadIdx.x; i < 4096; i += MX_THREADS) { squash[i] = M->squash[i]; if (MAIN) stretch[i] = M->stretch[i]; }
  const float cf = (float)(1.0 / 4095);
  if (MAIN) for (int i = tid; i < P8_NOUT; i += MX_THREADS) outs[i] = T->out[i];
  if (MAIN && tid == 0) fin_s = T->pr;
  unsigned long long misses = MAIN ? T->misses : 0;   // every thread keeps the miss history itself (uniform)
  MX_GLOBAL int16_t* const wx = (MX_GLOBAL int16_t*)M->wx; MX_GLOBAL int16_t* const wx2 = (MX_GLOBAL int16_t*)M->wx2;
  const int nx_first = M->nx_first;
  __shared__ int32_t ring_sel[4][P8_NSEL];
  __shared__ uint32_t ring_apm[4][6];     // P8ApmRec = 24 bytes
  __shared__ int ring_ord[4], ring_bit[4];
  const bool chain = MAIN && wave == 1 && lane < 7;
  const int fl = tid - 64, fj = fl >= 0 && fl < 7 * 36 ? fl / 36 : (chain ? lane : 0), fk = fl >= 0 ? fl % 36 : 0;
  MX_GLOBAL uint32_t* const tb_apm = (MX_GLOBAL uint32_t*)(MAIN && fj < 4 ? T->apm[fj] : nullptr);
  MX_GLOBAL uint16_t* const tb_apm1 = (MX_GLOBAL uint16_t*)(MAIN && fj >= 4 ? T->apm1[fj - 4] : nullptr);
  MX_GLOBAL uint16_t* const tb_gen = (MX_GLOBAL uint16_t*)(MAIN ? T->gen[fj] : nullptr);
  MX_GLOBAL uint32_t* const my_apm = (MX_GLOBAL uint32_t*)(chain && lane < 4 ? T->apm[lane] : nullptr);
  MX_GLOBAL uint16_t* const my_apm1 = (MX_GLOBAL uint16_t*)(chain && lane >= 4 ? T->apm1[lane - 4] : nullptr);
  MX_GLOBAL uint16_t* const my_gen = (MX_GLOBAL uint16_t*)(chain ? T->gen[lane] : nullptr);
  auto ring_load = [&](int ts, uint32_t& v) {   // wave 5: lanes 0..27 selectors, 28 order, 29 bit, 32..37 APM record (workgroup 0)
    if (wave != 5 || ts >= nbits) return;
    if (lane < P8_NSEL) v = (uint32_t)sel[(size_t)ts * P8_NSEL + lane];
    else if (lane == 28) v = order[ts];
    else if (lane == 29) v = bits[ts];
    else if (MAIN && lane >= 32 && lane < 38) v = reinterpret_cast<const uint32_t*>(apm + ts)[lane - 32];
  };
  auto ring_store = [&](int ts, uint32_t v) {
    if (wave != 5 || ts >= nbits) return;
    const int slot = ts & 3;
    if (lane < P8_NSEL) ring_sel[slot][lane] = (int32_t)v;
    else if (lane == 28) ring_ord[slot] = (int)v;
    else if (lane == 29) ring_bit[slot] = (int)v;
    else if (MAIN && lane >= 32 && lane < 38) ring_apm[slot][lane - 32] = v;
  };
  MxApmLane al_txt = {0, 0, 0}, al_gen = {0, 0, 0};
  if (chain) {
    const int j = lane;
    if (j < 4) { al_txt.idx = T->apm_cxt[j]; al_txt.v0 = my_apm[al_txt.idx]; }
    else { al_txt.idx = T->apm1_idx[j - 4]; al_txt.v0 = my_apm1[al_txt.idx]; al_txt.v1 = my_apm1[al_txt.idx + 1]; }
    al_gen.idx = T->gen_idx[j]; al_gen.v0 = my_gen[al_gen.idx]; al_gen.v1 = my_gen[al_gen.idx + 1];
  }
  __syncthreads();
  if (MAIN && t0) for (int i = tid; i < P8_NOUT; i += MX_THREADS) out[i] = outs[i];   // no step 0: the constructor's values
  // ---- this wavefront's one set: its row now and at the next step ----
  uint4 w[4], wn[4];
  int row = 0, rown = 0;
  uint4 xn = make_uint4(0, 0, 0, 0);
#define MX4_LOAD_ROW(dst, rowid)                                                                  \
  {                                                                                               \
    const MX_GLOBAL int16_t* wr_ = wx + (size_t)(rowid) * P8_NX;                                \
    _Pragma("unroll") for (int g_ = 0; g_ < 4; ++g_) { const int grp_ = lane + 64 * g_; dst[g_] = grp_ < MX_GROUPS ? mx_gload4(wr_, grp_) : make_uint4(0, 0, 0, 0); } \
  }
  // the set's row of step ts -- unless it is set 26, whose row needs the previous step's final probability (load_late)
  auto load_early = [&](uint4* dst, int& r, int ts) {
    if (si != P8_SEL_LASTPR) { r = p8s_sel(si, ring_sel[ts & 3][si], ring_ord[ts & 3], 0); MX4_LOAD_ROW(dst, r) }
  };
  auto load_late = [&](uint4* dst, int& r, int ts, int lastpr) {
    if (MAIN && si == P8_SEL_LASTPR) { r = p8s_sel(P8_SEL_LASTPR, ring_sel[ts & 3][P8_SEL_LASTPR], 0, lastpr); MX4_LOAD_ROW(dst, r) }
  };
  auto stage_x = [&](int t, int buf) {   // the input row of step t -> LDS (compacted during the first byte)
    const int16_t* xr = x + (size_t)t * P8_NX;
    if (t < first) {
      int16_t* xh = reinterpret_cast<int16_t*>(xs[buf]);
      for (int i = tid; i < P8_NX; i += MX_THREADS) xh[i] = i < nx_first ? xr[M->first_map[i]] : (int16_t)0;
    } else if (tid < MX_GROUPS) reinterpret_cast<uint4*>(xs[buf])[tid] = reinterpret_cast<const uint4*>(xr)[tid];
  };
  int a_base = 0, upd_idx = -1; uint32_t upd_v0 = 0, upd_v1 = 0;
  auto row_ctx = [&](const P8ApmRec* a, int j, unsigned long long ms) {
    if (a->text == P8_APM_TEXT) return j == 0 ? (a->c[0] | (int)((ms & 0xF) << 4)) : j == 1 ? (int)a->c[1 + (int)(ms & 3)] : (int)a->c[3 + j];
    return j == 0 ? (a->c[0] | (int)(ms & 7)) : j < 4 ? (int)a->c[j] : j == 4 ? (int)a->c[4] : (int)a->c[j - 3];
  };
  auto chain_fetch = [&](int ts, int y, unsigned long long ms) {   // see cmx_p8s_mix2_kernel's notes: update of the chosen cells, fetch of the step's rows
    const P8ApmRec* ap = reinterpret_cast<const P8ApmRec*>(ring_apm[ts & 3]);
    const int a_text = ap->text == P8_APM_TEXT, a_limit = ap->limit;
    if (fl >= 0 && fl < 7 * 36) {
      const int ncell = (a_text && fj < 4) ? 24 : 33;
      if (fk < ncell) {
        const int base = row_ctx(ap, fj, ms) * ncell;
        arow[fj][fk] = a_text ? (fj < 4 ? tb_apm[base + fk] : (uint32_t)tb_apm1[base + fk]) : (uint32_t)tb_gen[base + fk];
      }
    }
    if (chain) {
      const int j = lane;
      if (a_text) {
        upd_idx = al_txt.idx;
        if (j < 4) { upd_v0 = apm_upd(al_txt.v0, y, a_limit); my_apm[upd_idx] = upd_v0; }
        else { upd_v0 = apm1_upd(al_txt.v0, y, j == 4 ? 7 : 6); upd_v1 = apm1_upd(al_txt.v1, y, j == 4 ? 7 : 6); my_apm1[upd_idx] = (uint16_t)upd_v0; my_apm1[upd_idx + 1] = (uint16_t)upd_v1; }
        a_base = row_ctx(ap, j, ms) * (j < 4 ? 24 : 33);
      } else {
        upd_idx = al_gen.idx;
        upd_v0 = apm1_upd(al_gen.v0, y, 7); upd_v1 = apm1_upd(al_gen.v1, y, 7);
        my_gen[upd_idx] = (uint16_t)upd_v0; my_gen[upd_idx + 1] = (uint16_t)upd_v1;
        a_base = row_ctx(ap, j, ms) * 33;
      }
    }
  };
  auto chain_patch = [&](int a_text) {
    const int j = lane;
    const bool one = a_text && j < 4;
    const int ncell = one ? 24 : 33, off = upd_idx - a_base;
    if (off >= 0 && off < ncell) arow[j][off] = upd_v0;
    if (!one && off + 1 >= 0 && off + 1 < ncell) arow[j][off + 1] = upd_v1;
  };
  bool pend26 = false, again26 = false;
  { uint32_t v0_ = 0, v1_ = 0; ring_load(t0, v0_); ring_load(t0 + 1, v1_); ring_store(t0, v0_); ring_store(t0 + 1, v1_); }
  __syncthreads();
  if (t0 < nbits) {
    if (MAIN) {
      const int y0 = t0 ? (int)bits[t0 - 1] : last_y;
      misses += misses + (unsigned long long)((T->pr >> 11) != y0);   // Predictor::update's first line (:8250), for the first step
      chain_fetch(t0, y0, misses);
    }
    load_early(w, row, t0);
    if (MAIN) load_late(w, row, t0, T->pr);
    stage_x(t0, t0 & 1);
  }
  for (int t = t0; t < nbits; ++t) {
    const int nx = t < first ? nx_first : P8_NX;
    uint32_t ring_v = 0;
    ring_load(t + 2, ring_v);   // two steps ahead; lands in LDS at the end of this step
    const int buf = t & 1;
    mx_lds_barrier();   // B1: xs[buf], the ring entries of this step (and, workgroup 0, arow / fin_s of the previous step)
    int a_text = 0;
    const P8ApmRec* arp = reinterpret_cast<const P8ApmRec*>(ring_apm[t & 3]);
    if (MAIN) {
      a_text = arp->text == P8_APM_TEXT;
      if (chain) chain_patch(a_text);
      if (wave == 6 && pend26 && !again26) {   // set 26's row was requested after the previous step's chain: take it now
#pragma unroll
        for (int g = 0; g < 4; ++g) w[g] = wn[g];
      }
      for (int i = tid; i < nx; i += MX_THREADS) outs[i] = (float)p8s_squash(squash, reinterpret_cast<const int16_t*>(xs[buf])[i]) * cf;
    }
    // ---- first layer: this wavefront's set on the row in registers ----
    int my_pr;
    {
      uint32_t acc = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int grp = lane + 64 * g;
        const uint4 xv = grp < MX_GROUPS ? reinterpret_cast<const uint4*>(xs[buf])[grp] : make_uint4(0, 0, 0, 0);
        acc += pair_dot(xv.x, w[g].x) + pair_dot(xv.y, w[g].y) + pair_dot(xv.z, w[g].z) + pair_dot(xv.w, w[g].w);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      my_pr = p8s_squash(squash, (int32_t)(acc * 9u) >> 9);   // uniform in the wavefront
      if (lane == 0) {
        if (MAIN) pr_s[si] = my_pr;
        else __hip_atomic_store(&prx[(size_t)t * P8_NSEL + si], (mx4_tag(epoch, t) << 12) | (unsigned)my_pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (MAIN && wave == 4 && lane < P8_NSEL && (lane & 3) != QSEL) {   // the 21 outputs of the other workgroups (they run ahead: normally there already)
      unsigned spins = 0;
      for (;;) {
        const unsigned long long v = __hip_atomic_load(&prx[(size_t)t * P8_NSEL + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 12) == mx4_tag(epoch, t)) { pr_s[lane] = (int)(v & 4095u); break; }
        if ((++spins & 1023u) == 0 && (spins > MX4_SPIN || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pr_s[lane] = 2048; break;
        }
      }
    }
    // ---- one step ahead: the row (unless it needs this step's final probability) and the input row ----
    const bool more = t + 1 < nbits;
    if (more) {
      load_early(wn, rown, t + 1);
      if (t + 1 >= first && tid < MX_GROUPS) xn = reinterpret_cast<const uint4*>(x + (size_t)(t + 1) * P8_NX)[tid];
    }
    if (MAIN) {
      mx_lds_barrier();   // B2: pr_s, arow
      if (chain) __builtin_amdgcn_s_waitcnt(0);   // this lane's table updates have reached L2 before other lanes fetch rows again (after B4)
      if (wave == 0) {   // second layer
        const int a = lane < P8_NSEL ? stretch[pr_s[lane]] : 0;
        if (lane < P8_NSEL) outs[nx + lane] = (float)p8s_squash(squash, a) * cf;
        const int b = __shfl_down(a, 1);
        if ((lane & 1) == 0 && lane < 32) st_s[lane >> 1] = ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        uint32_t acc = 0;
        if (lane < 16) acc = pair_dot(st_s[lane], reinterpret_cast<const MX_GLOBAL uint32_t*>(wx2)[lane]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) p_s = p8s_squash(squash, (int32_t)acc >> 9);
      }
      mx_lds_barrier();   // B3: p_s
      if (wave == 1) {   // the chains on the fetched rows: group A (lanes 0..3), then group B (lanes 4..6), then the read-out
        const int p2 = p_s;
        auto look_apm = [&](int j, int pr) {   // APM::p's interpolation :704-710 on arow[j]
          const int s = (stretch[pr] + 2048) * 23;
          const int wt = s & 0xfff, lo = s >> 12;
          al_txt.idx = a_base + lo + (wt >> 11); al_txt.v0 = arow[j][lo + (wt >> 11)];
          return (int)(((arow[j][lo] >> 13) * (uint32_t)(4096 - wt) + (arow[j][lo + 1] >> 13) * (uint32_t)wt) >> 19);
        };
        auto look_apm1 = [&](int j, int pr) {   // APM1::pp's interpolation :614-619
          const int s = stretch[pr];
          const int wgt = s & 127, lo = (s + 2048) >> 7;
          MxApmLane& al = a_text ? al_txt : al_gen;
          al.idx = a_base + lo; al.v0 = arow[j][lo]; al.v1 = arow[j][lo + 1];
          return (int)((arow[j][lo] * (uint32_t)(128 - wgt) + arow[j][lo + 1] * (uint32_t)wgt) >> 11);
        };
        if (lane < 4) res_s[lane] = a_text ? look_apm(lane, p2) : look_apm1(lane, p2);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if (lane >= 4 && lane < 7) {
          const int avg = (p2 + res_s[1] + res_s[2] + res_s[3] + 2) >> 2;
          res_s[lane] = look_apm1(lane, a_text ? (lane == 4 ? avg : res_s[0]) : res_s[0]);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) fin_s = p8s_tail_c(arp, p2, res_s, outs + nx + P8_NSEL);
      }
      mx_lds_barrier();   // B4: outs complete, fin_s
      float* orow = out + (size_t)t * ld;
      for (int i = tid; i < P8_NOUT; i += MX_THREADS) orow[i] = outs[i];
    }
    // ---- training with the step's own bit; the row goes back to HBM, the next step's row becomes current ----
    const int yb = ring_bit[t & 3];
    if (MAIN) {
      if (more) load_late(wn, rown, t + 1, fin_s);   // the one row that needed this step's final probability
      if (more) misses += misses + (unsigned long long)((fin_s >> 11) != yb);   // the next step's
      if (more) chain_fetch(t + 1, yb, misses);
    }
    {
      const int err = (int)(int16_t)(((yb << 12) - my_pr) * 7);
      MX_GLOBAL int16_t* wr = wx + (size_t)row * P8_NX;
      const bool again = more && rown == row;
      const bool is26 = MAIN && si == P8_SEL_LASTPR;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int grp = lane + 64 * g;
        uint4 v = w[g];
        if (grp < MX_GROUPS && err) {
          const uint4 xv = reinterpret_cast<const uint4*>(xs[buf])[grp];
          v.x = pair_train(xv.x, v.x, err); v.y = pair_train(xv.y, v.y, err);
          v.z = pair_train(xv.z, v.z, err); v.w = pair_train(xv.w, v.w, err);
          mx_gstore4(wr, grp, v);
        }
        if (is26) w[g] = v;   // resolved at the top of the next step, when the late request has arrived
        else w[g] = again ? v : wn[g];
      }
      if (is26) { pend26 = more; again26 = again; }
      row = more ? rown : row;
    }
    if (MAIN && wave == 0 && lane < 16) {
      const int err2 = (int)(int16_t)(((yb << 12) - p_s) * 7);
      MX_GLOBAL uint32_t* w2 = reinterpret_cast<MX_GLOBAL uint32_t*>(wx2);
      if (err2) w2[lane] = pair_train(st_s[lane], w2[lane], err2);
    }
    ring_store(t + 2, ring_v);
    if (more) {   // the next step's inputs into the other LDS buffer
      if (t + 1 < first) stage_x(t + 1, buf ^ 1);
      else if (tid < MX_GROUPS) reinterpret_cast<uint4*>(xs[buf ^ 1])[tid] = xn;
    }
  }
#undef MX4_LOAD_ROW
  if (MAIN) {
    __syncthreads();
    if (chain) {
      const int j = lane;
      if (j < 4) T->apm_cxt[j] = al_txt.idx; else T->apm1_idx[j - 4] = al_txt.idx;
      T->gen_idx[j] = al_gen.idx;
    }
    for (int i = tid; i < P8_NOUT; i += MX_THREADS) T->out[i] = outs[i];
    if (tid == 0) { T->pr = fin_s; T->misses = misses; }
  }
}

// ---- the mixer for a DECODER (cmx_late.h) --------------------------------------------------------------------------------------
// The same four workgroups, the same arithmetic per step, nothing fetched ahead: a step's input row exists only when the six
// producing roles have counted it (they wait for the bit before the step themselves), its selectors / APM contexts are host records
// that arrive with that bit, and the step's own bit -- which trains the rows -- arrives after the row of outputs has gone out
// (counter LC_P8) and the mixing network and the arithmetic decoder have used it. Per step: wait (bit before the step, producers)
// -> records and inputs into LDS -> chains' cell updates + row fetch, weight rows (kept in registers when the selector repeats)
// -> dot products, second layer, chains, export, LC_P8 -> wait for the step's bit -> training.
template <int BLK>
__device__ __forceinline__ void mx4_late_body(const P8MixDev* M, P8TailDev* T, CmxLate B, const int16_t* x, const int32_t* sel, const P8ApmRec* apm,
                                              const uint8_t* order, float* out, size_t ld, int nbits, int t0, int first, unsigned long long* prx,
                                              unsigned epoch, unsigned* fail) {
  constexpr int QSEL = (BLK + 2) & 3;
  constexpr bool MAIN = BLK == 0;
  __shared__ __attribute__((aligned(16))) uint32_t xs[P8_NX / 2];
  __shared__ float outs[MAIN ? P8_NOUT : 1];
  __shared__ int pr_s[32], res_s[8];
  __shared__ uint32_t st_s[16];
  __shared__ uint32_t arow[MAIN ? 7 : 1][36];
  __shared__ int p_s, fin_s, late_y_s;
  __shared__ int32_t sel_s[P8_NSEL];
  __shared__ uint32_t apm_s[6];
  __shared__ int ord_s;
  __shared__ int16_t squash[4096], stretch[MAIN ? 4096 : 1];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int si = 4 * wave + QSEL;
  for (int i = threadIdx.x; i < 4096; i += MX_THREADS) { squash[i] = M->squash[i]; if (MAIN) stretch[i] = M->stretch[i]; }
  const float cf = (float)(1.0 / 4095);
  if (MAIN) for (int i = tid; i < P8_NOUT; i += MX_THREADS) outs[i] = T->out[i];
  if (MAIN && tid == 0) fin_s = T->pr;
  unsigned long long misses = MAIN ? T->misses : 0;
  MX_GLOBAL int16_t* const wx = (MX_GLOBAL int16_t*)M->wx; MX_GLOBAL int16_t* const wx2 = (MX_GLOBAL int16_t*)M->wx2;
  const int nx_first = M->nx_first;
  const bool chain = MAIN && wave == 1 && lane < 7;
  const int fl = tid - 64, fj = fl >= 0 && fl < 7 * 36 ? fl / 36 : (chain ? lane : 0), fk = fl >= 0 ? fl % 36 : 0;
  MX_GLOBAL uint32_t* const tb_apm = (MX_GLOBAL uint32_t*)(MAIN && fj < 4 ? T->apm[fj] : nullptr);
  MX_GLOBAL uint16_t* const tb_apm1 = (MX_GLOBAL uint16_t*)(MAIN && fj >= 4 ? T->apm1[fj - 4] : nullptr);
  MX_GLOBAL uint16_t* const tb_gen = (MX_GLOBAL uint16_t*)(MAIN ? T->gen[fj] : nullptr);
  MX_GLOBAL uint32_t* const my_apm = (MX_GLOBAL uint32_t*)(chain && lane < 4 ? T->apm[lane] : nullptr);
  MX_GLOBAL uint16_t* const my_apm1 = (MX_GLOBAL uint16_t*)(chain && lane >= 4 ? T->apm1[lane - 4] : nullptr);
  MX_GLOBAL uint16_t* const my_gen = (MX_GLOBAL uint16_t*)(chain ? T->gen[lane] : nullptr);
  MxApmLane al_txt = {0, 0, 0}, al_gen = {0, 0, 0};
  if (chain) {
    const int j = lane;
    if (j < 4) { al_txt.idx = T->apm_cxt[j]; al_txt.v0 = my_apm[al_txt.idx]; }
    else { al_txt.idx = T->apm1_idx[j - 4]; al_txt.v0 = my_apm1[al_txt.idx]; al_txt.v1 = my_apm1[al_txt.idx + 1]; }
    al_gen.idx = T->gen_idx[j]; al_gen.v0 = my_gen[al_gen.idx]; al_gen.v1 = my_gen[al_gen.idx + 1];
  }
  __syncthreads();
  if (MAIN && t0) {   // no step 0: the constructor's values are row 0
    for (int i = tid; i < P8_NOUT; i += MX_THREADS) out[i] = outs[i];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY wave's stores have landed before the barrier (s_barrier alone does not wait for them) -- then one thread counts the row
    __syncthreads();
    if (tid == 0) late_publish(B, LC_P8, 1u);
  }
  uint4 w[4];
  int row = -1;
  int a_base = 0, upd_idx = -1; uint32_t upd_v0 = 0, upd_v1 = 0;
  auto row_ctx = [&](const P8ApmRec* a, int j, unsigned long long ms) {
    if (a->text == P8_APM_TEXT) return j == 0 ? (a->c[0] | (int)((ms & 0xF) << 4)) : j == 1 ? (int)a->c[1 + (int)(ms & 3)] : (int)a->c[3 + j];
    return j == 0 ? (a->c[0] | (int)(ms & 7)) : j < 4 ? (int)a->c[j] : j == 4 ? (int)a->c[4] : (int)a->c[j - 3];
  };
  for (int t = t0; t < nbits; ++t) {
    const int nx = t < first ? nx_first : P8_NX;
    // ---- the bit before the step (with it: the step's host records), the six producers' rows ----
    if (wave == 0) {   // lane j < 6: producer j's counter; lane 6: the bit (one memory round trip for all seven, not seven in a row)
      int yv = 0;
      bool ok = true;
      if (lane < 6) ok = late_wait_cnt(B, LC_CM2_0 + lane, (uint32_t)(t + 1));
      else if (lane == 6) { yv = late_y(B, t); ok = yv >= 0; }
      const bool all_ok = __ballot(!ok) == 0;
      yv = __shfl(yv, 6);
      if (lane == 0) late_y_s = all_ok ? yv : -1;
    }
    __syncthreads();
    const int y = late_y_s;
    if (y < 0) return;
    if (wave == 5) {
      if (lane < P8_NSEL) sel_s[lane] = *(volatile const int32_t*)&sel[(size_t)t * P8_NSEL + lane];
      else if (lane == 28) ord_s = *(volatile const uint8_t*)&order[t];
      else if (MAIN && lane >= 32 && lane < 38) apm_s[lane - 32] = reinterpret_cast<volatile const uint32_t*>(apm + t)[lane - 32];
    }
    {
      const int16_t* xr = x + (size_t)t * P8_NX;
      if (t < first) {
        int16_t* xh = reinterpret_cast<int16_t*>(xs);
        for (int i = tid; i < P8_NX; i += MX_THREADS) xh[i] = i < nx_first ? *(volatile const int16_t*)&xr[M->first_map[i]] : (int16_t)0;
      } else if (tid < MX_GROUPS) {
        const volatile uint32_t* xq = reinterpret_cast<const volatile uint32_t*>(xr) + 4 * tid;
        reinterpret_cast<uint4*>(xs)[tid] = make_uint4(xq[0], xq[1], xq[2], xq[3]);
      }
    }
    __syncthreads();
    const P8ApmRec* arp = reinterpret_cast<const P8ApmRec*>(apm_s);
    int a_text = 0;
    if (MAIN) {
      // Predictor::update's first line (:8250), then the chains' cell updates with the bit and the fetch of the step's rows
      misses += misses + (unsigned long long)((fin_s >> 11) != y);
      a_text = arp->text == P8_APM_TEXT;
      const int a_limit = arp->limit;
      if (fl >= 0 && fl < 7 * 36) {
        const int ncell = (a_text && fj < 4) ? 24 : 33;
        if (fk < ncell) {
          const int base = row_ctx(arp, fj, misses) * ncell;
          arow[fj][fk] = a_text ? (fj < 4 ? tb_apm[base + fk] : (uint32_t)tb_apm1[base + fk]) : (uint32_t)tb_gen[base + fk];
        }
      }
      if (chain) {
        const int j = lane;
        if (a_text) {
          upd_idx = al_txt.idx;
          if (j < 4) { upd_v0 = apm_upd(al_txt.v0, y, a_limit); my_apm[upd_idx] = upd_v0; }
          else { upd_v0 = apm1_upd(al_txt.v0, y, j == 4 ? 7 : 6); upd_v1 = apm1_upd(al_txt.v1, y, j == 4 ? 7 : 6); my_apm1[upd_idx] = (uint16_t)upd_v0; my_apm1[upd_idx + 1] = (uint16_t)upd_v1; }
          a_base = row_ctx(arp, j, misses) * (j < 4 ? 24 : 33);
        } else {
          upd_idx = al_gen.idx;
          upd_v0 = apm1_upd(al_gen.v0, y, 7); upd_v1 = apm1_upd(al_gen.v1, y, 7);
          my_gen[upd_idx] = (uint16_t)upd_v0; my_gen[upd_idx + 1] = (uint16_t)upd_v1;
          a_base = row_ctx(arp, j, misses) * 33;
        }
      }
    }
    {   // this wavefront's weight row of the step: a row the previous step trained and this one selects again stays in the registers
      const int r = p8s_sel(si, sel_s[si], ord_s, MAIN ? fin_s : 0);
      if (r != row) {
        row = r;
        const MX_GLOBAL int16_t* wr_ = wx + (size_t)r * P8_NX;
#pragma unroll
        for (int g = 0; g < 4; ++g) { const int grp = lane + 64 * g; w[g] = grp < MX_GROUPS ? mx_gload4(wr_, grp) : make_uint4(0, 0, 0, 0); }
      }
    }
    mx_lds_barrier();   // arow
    if (MAIN) {
      if (chain) {
        const int j = lane;
        const bool one = a_text && j < 4;
        const int ncell = one ? 24 : 33, off = upd_idx - a_base;
        if (off >= 0 && off < ncell) arow[j][off] = upd_v0;
        if (!one && off + 1 >= 0 && off + 1 < ncell) arow[j][off + 1] = upd_v1;
      }
      for (int i = tid; i < nx; i += MX_THREADS) outs[i] = (float)p8s_squash(squash, reinterpret_cast<const int16_t*>(xs)[i]) * cf;
    }
    // ---- first layer ----
    int my_pr;
    {
      uint32_t acc = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int grp = lane + 64 * g;
        const uint4 xv = grp < MX_GROUPS ? reinterpret_cast<const uint4*>(xs)[grp] : make_uint4(0, 0, 0, 0);
        acc += pair_dot(xv.x, w[g].x) + pair_dot(xv.y, w[g].y) + pair_dot(xv.z, w[g].z) + pair_dot(xv.w, w[g].w);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      my_pr = p8s_squash(squash, (int32_t)(acc * 9u) >> 9);
      if (lane == 0) {
        if (MAIN) pr_s[si] = my_pr;
        else __hip_atomic_store(&prx[(size_t)t * P8_NSEL + si], (mx4_tag(epoch, t) << 12) | (unsigned)my_pr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    if (MAIN && wave == 4 && lane < P8_NSEL && (lane & 3) != QSEL) {
      unsigned spins = 0;
      for (;;) {
        const unsigned long long v = __hip_atomic_load(&prx[(size_t)t * P8_NSEL + lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((v >> 12) == mx4_tag(epoch, t)) { pr_s[lane] = (int)(v & 4095u); break; }
        if ((++spins & 1023u) == 0 && (spins > MX4_SPIN || late_ld(&B.box->abort) || __hip_atomic_load(fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) {
          __hip_atomic_store(fail, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); pr_s[lane] = 2048; break;
        }
      }
    }
    if (MAIN) {
      mx_lds_barrier();   // pr_s, arow
      if (chain) __builtin_amdgcn_s_waitcnt(0);   // this lane's table updates have reached L2 before other lanes fetch rows again (next step)
      if (wave == 0) {   // second layer
        const int a = lane < P8_NSEL ? stretch[pr_s[lane]] : 0;
        if (lane < P8_NSEL) outs[nx + lane] = (float)p8s_squash(squash, a) * cf;
        const int b = __shfl_down(a, 1);
        if ((lane & 1) == 0 && lane < 32) st_s[lane >> 1] = ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        uint32_t acc = 0;
        if (lane < 16) acc = pair_dot(st_s[lane], reinterpret_cast<const MX_GLOBAL uint32_t*>(wx2)[lane]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        if (lane == 0) p_s = p8s_squash(squash, (int32_t)acc >> 9);
      }
      mx_lds_barrier();   // p_s
      if (wave == 1) {
        const int p2 = p_s;
        auto look_apm = [&](int j, int pr) {   // APM::p's interpolation :704-710 on arow[j]
          const int s_ = (stretch[pr] + 2048) * 23;
          const int wt = s_ & 0xfff, lo = s_ >> 12;
          al_txt.idx = a_base + lo + (wt >> 11); al_txt.v0 = arow[j][lo + (wt >> 11)];
          return (int)(((arow[j][lo] >> 13) * (uint32_t)(4096 - wt) + (arow[j][lo + 1] >> 13) * (uint32_t)wt) >> 19);
        };
        auto look_apm1 = [&](int j, int pr) {   // APM1::pp's interpolation :614-619
          const int s_ = stretch[pr];
          const int wgt = s_ & 127, lo = (s_ + 2048) >> 7;
          MxApmLane& al = a_text ? al_txt : al_gen;
          al.idx = a_base + lo; al.v0 = arow[j][lo]; al.v1 = arow[j][lo + 1];
          return (int)((arow[j][lo] * (uint32_t)(128 - wgt) + arow[j][lo + 1] * (uint32_t)wgt) >> 11);
        };
        if (lane < 4) res_s[lane] = a_text ? look_apm(lane, p2) : look_apm1(lane, p2);
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if (lane >= 4 && lane < 7) {
          const int avg = (p2 + res_s[1] + res_s[2] + res_s[3] + 2) >> 2;
          res_s[lane] = look_apm1(lane, a_text ? (lane == 4 ? avg : res_s[0]) : res_s[0]);
        }
        __builtin_amdgcn_s_waitcnt(0);
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) fin_s = p8s_tail_c(arp, p2, res_s, outs + nx + P8_NSEL);
      }
      mx_lds_barrier();   // outs complete, fin_s
      float* orow = out + (size_t)t * ld;
      for (int i = tid; i < P8_NOUT; i += MX_THREADS) orow[i] = outs[i];
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // EVERY wave's stores have landed before the barrier (s_barrier alone does not wait for them) -- then one thread counts the row
      __syncthreads();
      if (tid == 0) late_publish(B, LC_P8, (uint32_t)(t + 1));
    }
    // ---- the step's own bit, then training ----
    if (tid == 0) late_y_s = late_y(B, t + 1);
    __syncthreads();
    const int yb = late_y_s;
    if (yb < 0) return;
    {
      const int err = (int)(int16_t)(((yb << 12) - my_pr) * 7);
      MX_GLOBAL int16_t* wr = wx + (size_t)row * P8_NX;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int grp = lane + 64 * g;
        if (grp < MX_GROUPS && err) {
          const uint4 xv = reinterpret_cast<const uint4*>(xs)[grp];
          uint4 v = w[g];
          v.x = pair_train(xv.x, v.x, err); v.y = pair_train(xv.y, v.y, err);
          v.z = pair_train(xv.z, v.z, err); v.w = pair_train(xv.w, v.w, err);
          mx_gstore4(wr, grp, v);
          w[g] = v;
        }
      }
    }
    if (MAIN && wave == 0 && lane < 16) {
      const int err2 = (int)(int16_t)(((yb << 12) - p_s) * 7);
      MX_GLOBAL uint32_t* w2 = reinterpret_cast<MX_GLOBAL uint32_t*>(wx2);
      if (err2) w2[lane] = pair_train(st_s[lane], w2[lane], err2);
    }
    __syncthreads();   // xs, late_y_s and the second-layer weights are free for the next step
  }
  if (MAIN) {
    __syncthreads();
    if (chain) {
      const int j = lane;
      if (j < 4) T->apm_cxt[j] = al_txt.idx; else T->apm1_idx[j - 4] = al_txt.idx;
      T->gen_idx[j] = al_gen.idx;
    }
    for (int i = tid; i < P8_NOUT; i += MX_THREADS) T->out[i] = outs[i];
    if (tid == 0) { T->pr = fin_s; T->misses = misses; }
  }
}
__global__ __launch_bounds__(MX_THREADS) void cmx_p8s_mix4_late_kernel(const P8MixDev* M, P8TailDev* T, CmxLate B, const int16_t* x, const int32_t* sel, const P8ApmRec* apm,
                                                                      const uint8_t* order, float* out, size_t ld, int nbits, int t0, int first, unsigned long long* prx,
                                                                      unsigned epoch, unsigned* fail) {
  switch (blockIdx.x) {
    case 0: mx4_late_body<0>(M, T, B, x, sel, apm, order, out, ld, nbits, t0, first, prx, epoch, fail); break;
    case 1: mx4_late_body<1>(M, T, B, x, sel, apm, order, out, ld, nbits, t0, first, prx, epoch, fail); break;
    case 2: mx4_late_body<2>(M, T, B, x, sel, apm, order, out, ld, nbits, t0, first, prx, epoch, fail); break;
    default: mx4_late_body<3>(M, T, B, x, sel, apm, order, out, ld, nbits, t0, first, prx, epoch, fail); break;
  }
}

// prx: [>= nbits][28] exchange words, (launch number, step + 1) | 12-bit output: never cleared, a word of an earlier launch does not
// match (the launch number wraps after 2^24 launches, 64 GB of input in 4 KB chunks; every launch rewrites the words of its own steps); fail: sticky time-out flag of workgroup 0's wait (host-mapped)
__global__ __launch_bounds__(MX_THREADS) void cmx_p8s_mix4_kernel(const P8MixDev* M, P8TailDev* T, const int16_t* __restrict__ x, const int32_t* __restrict__ sel,
                                                                 const P8ApmRec* __restrict__ apm, const uint8_t* __restrict__ order,
                                                                 const uint8_t* __restrict__ bits, float* __restrict__ out, size_t ld, int nbits, int t0, int first,
                                                                 int last_y, unsigned long long* prx, unsigned epoch, unsigned* fail) {
  switch (blockIdx.x) {
    case 0: mx4_body<0>(M, T, x, sel, apm, order, bits, out, ld, nbits, t0, first, last_y, prx, epoch, fail); break;
    case 1: mx4_body<1>(M, T, x, sel, apm, order, bits, out, ld, nbits, t0, first, last_y, prx, epoch, fail); break;
    case 2: mx4_body<2>(M, T, x, sel, apm, order, bits, out, ld, nbits, t0, first, last_y, prx, epoch, fail); break;
    default: mx4_body<3>(M, T, x, sel, apm, order, bits, out, ld, nbits, t0, first, last_y, prx, epoch, fail); break;
  }
}

// ---------------------------------------------------------------- the image models' role kernels (p8_rec.h P8XLayout)
// A chunk that holds bytes of an image model is rare and runs its roles one after the other on one stream (cmx_p8stage_run): these
// kernels are plain -- correctness first; an image byte still costs far less than on the reference's CPU path.
//
// The model's one ContextMap: the first design's four phases (p8cm_dev.h: buckets touched / overlap check + draw ranks / the generator's
// values / every context with "its" draw, or the reference's serial walk on lane 0 when two contexts meet in a bucket), one lane per context.
// last_y / c1: the bit and the whole byte before the segment (the host knows them); the generator arrives in d->rnd (cmx_p8s_rnd_copy_kernel).
__global__ __launch_bounds__(64) void cmx_p8s_xfam_kernel(P8CmDev* d, const uint32_t* xctx, const uint16_t* xchk, const uint8_t* bits, int16_t* x, int nbits, int last_y, int c1_in) {
  __shared__ P8CmShared sh;
  const int s = threadIdx.x, S = d->nslots;
  if (s < S) { sh.r.cp[s] = d->regs.cp[s]; sh.r.cp0[s] = d->regs.cp0[s]; sh.r.runp[s] = d->regs.runp[s]; sh.r.sm_cxt[s] = d->regs.sm_cxt[s]; }
  sh.rnd.table[s] = d->rnd.table[s];
  if (s == 0) sh.rnd.i = d->rnd.i;
  __syncthreads();
  int y = last_y, c1 = c1_in, c0 = 1;
  for (int t = 0; t < nbits; t++) {
    const int bp = t & 7;
    if (bp == 0) c0 = 1;
    P8CmBit u;
    u.y = y; u.bp = bp; u.c0 = c0; u.c1 = c1; u.order = 0;
    u.ctx = xctx + (size_t)(t >> 3) * P8_XL_MAXS; u.chk = xchk + (size_t)(t >> 3) * P8_XL_MAXS; u.out = x + (size_t)t * P8_NX;
    const int A0 = (int)u.ctx[P8_XL_MAXS - 2], A = (int)u.ctx[P8_XL_MAXS - 1];   // the slots whose map is called with a context this byte (the row's last two cells): the others are not touched
    if (s == 0) { sh.act_lo = A0; sh.act_hi = A; }
    __syncthreads();
    const bool act = s >= A0 && s < A;
    if (act) p8d_cm_touch(d, &sh, u, s);
    __syncthreads();
    if (act) p8d_cm_check(d, &sh, s);
    __syncthreads();
    if (act) p8d_cm_draw(d, &sh, s);
    __syncthreads();
    if (act) p8d_cm_run(d, &sh, u, s);
    __syncthreads();
    const int bit = bits[t];
    y = bit; c0 = c0 * 2 + bit;
    if (bp == 7) c1 = c0 & 0xff;
  }
  if (s < S) { d->regs.cp[s] = sh.r.cp[s]; d->regs.cp0[s] = sh.r.cp0[s]; d->regs.runp[s] = sh.r.runp[s]; d->regs.sm_cxt[s] = sh.r.sm_cxt[s]; }
  d->rnd.table[s] = sh.rnd.table[s];
  if (s == 0) d->rnd.i = sh.rnd.i;
}
// the process-wide rnd() stream (:152-165) changes hands between the generic family and an image model's ContextMap
__global__ __launch_bounds__(64) void cmx_p8s_rnd_copy_kernel(P8CmDev* dst, const P8CmDev* src) {
  dst->rnd.table[threadIdx.x] = src->rnd.table[threadIdx.x];
  if (threadIdx.x == 0) dst->rnd.i = src->rnd.i;
}
// a model's family takes over / gives back the per-context state of the generic instances it contains (p8stage_build.h P8ViewMap): one thread per context
__global__ __launch_bounds__(64) void cmx_p8s_view_in_kernel(P8CmDev* view, const P8CmDev* gen, const P8FamHome* gh, P8ViewMap V) {
  for (int g = 0; g < V.n; g++)
    if ((int)threadIdx.x < V.count[g]) p8v_slot_in(gh, gen->sm, &view->regs, view->sm, V.gen_first[g] + (int)threadIdx.x, V.view_first[g] + (int)threadIdx.x);
}
__global__ __launch_bounds__(64) void cmx_p8s_view_out_kernel(const P8CmDev* view, P8CmDev* gen, P8FamHome* gh, P8ViewMap V) {
  for (int g = 0; g < V.n; g++)
    if ((int)threadIdx.x < V.count[g]) {
      const int vs = V.view_first[g] + (int)threadIdx.x;
      p8v_slot_out(gh, gen->sm, &view->regs, view->sm, view->inst[view->slot_inst[vs]].table, V.gen_first[g] + (int)threadIdx.x, vs);
    }
}
// the generic family after bytes it was not called for: the bit and the whole byte before its next step are the stream's (ContextMap::mix
// reads the globals y and buf(1), :1072-1145)
__global__ void cmx_p8s_fam_resume_kernel(P8CmDev* d, int last_y, int c1) { d->last_y = last_y; d->c1 = c1; }

// the model's small maps: one lane each, in the steps of its bytes
__global__ __launch_bounds__(P8_XL_NLANE) void cmx_p8s_xlanes_kernel(P8XLanesDev* d, const uint32_t* xops, const uint8_t* bits, const uint8_t* order, int16_t* x,
                                                                     const uint8_t* model, int nbits, int t0, int last_y) {
  const int l = threadIdx.x;
  if (l >= d->nlanes) return;
  P8LaneRegs r = d->regs[l];
  const P8LaneTabs tb = {d->nex, d->stretch, nullptr};
  const int mine = d->model;
  for (int t = t0; t < nbits; t++) {
    if (model[t >> 3] != mine) continue;
    const int y = t ? bits[t - 1] : last_y;
    const uint32_t op = xops[(size_t)t * P8_XL_NLANE + l];
    if (d->lane[l].q.kind == P8L_JPG) p8s_lane_jpg(&d->lane[l], &tb, d->squash, xops + (size_t)t * P8_XL_NLANE + l, y, x + (size_t)t * P8_NX);
    else if (d->lane[l].q.kind == P8L_HT16) p8s_lane_ht16(&d->lane[l], &tb, xops + (size_t)t * P8_XL_NLANE + l, y, t & 7, x + (size_t)t * P8_NX);
    else if (d->lane[l].q.kind == P8L_PIC2) p8s_lane_pic2(&d->lane[l], &tb, &r, op, xops[(size_t)t * P8_XL_NLANE + l + 1], y, x + (size_t)t * P8_NX);
    else if (op & P8OP_MIX) p8s_lane_step_t(&d->lane[l], &tb, &r, op, y, order[t], x + (size_t)t * P8_NX, P8_NX);   // (a map the step does not call writes nothing: its positions may be the model's other face's)
  }
  d->regs[l] = r;
}

// The mixer of an image model's steps (one segment of consecutive such bytes): the step's nx inputs (P8ApmRec.m[1]; zeros behind them, which
// neither the dot products nor the training see), its nsel weight sets (m[2]; absolute rows, no device terms) one per wavefront, the second layer,
// the model's APM chain on one lane (p8s_tail_image), the export: nx + nsel + 10 values back to back, the rest of the 1591 as they were
// (AddPrediction() counts on, :504-510). Same packed arithmetic as cmx_p8s_mix4_kernel. T: the state the generic mixer leaves and takes over.
constexpr int XMX_THREADS = 1024;
struct P8XMixMap { int16_t map[P8_NX]; int opt_lo, opt_n; int exp_n; int16_t exp[P8_NX]; };   // a model's inputs in add() order -> positions in the 1552-vector; its export order (P8XLayout)
__global__ __launch_bounds__(XMX_THREADS) void cmx_p8s_xmix_kernel(const P8MixDev* M, P8TailDev* T, const P8XMixMap* maps, const int16_t* x, const int32_t* sel, const P8ApmRec* apm,
                                                                 const uint8_t* bits, float* out, size_t ld, int nbits, int last_y) {
  __shared__ __attribute__((aligned(16))) uint32_t xs[P8_NX / 2];
  __shared__ float outs[P8_NOUT];
  __shared__ int pr_s[16], p_s, fin_s;
  __shared__ uint32_t st_s[16];
  __shared__ int16_t squash[4096], stretch[4096];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int i = tid; i < 4096; i += XMX_THREADS) { squash[i] = M->squash[i]; stretch[i] = M->stretch[i]; }
  for (int i = tid; i < P8_NOUT; i += XMX_THREADS) outs[i] = T->out[i];
  const float cf = (float)(1.0 / 4095);
  unsigned long long misses = T->misses;
  int lastpr = T->pr;
  MX_GLOBAL int16_t* const wx = (MX_GLOBAL int16_t*)M->wx; MX_GLOBAL int16_t* const wx2 = (MX_GLOBAL int16_t*)M->wx2;
  __syncthreads();
  for (int t = 0; t < nbits; t++) {
    const P8ApmRec* a = apm + t;
    const int nx = a->m[1], nsel = a->m[2];
    const int y = t ? (int)bits[t - 1] : last_y;
    misses += misses + (unsigned long long)((lastpr >> 11) != y);   // Predictor::update's first line (:8250)
    const int16_t* xr = x + (size_t)t * P8_NX;
    const P8XMixMap* mp = maps + (a->model - 1);
    const int skp = a->m[0] == 1 ? mp->opt_n : 0, olo = mp->opt_lo;   // the model's own ContextMap is silent this byte: its inputs are not there
    const int jc = a->model == P8_MODEL_JPEG ? (int)a->m[3] : 0;        // a stuffed / restart step of the JPEG model: its one constant input
    const int ne = a->m[0] == 2 ? mp->exp_n : nx;                       // exported values in front of the second layer's
    for (int i = tid; i < P8_NX / 2; i += XMX_THREADS) {
      const int i0 = 2 * i, i1 = 2 * i + 1;
      const uint32_t lo = i0 < nx ? (uint32_t)(uint16_t)xr[mp->map[(skp && i0 >= olo) ? i0 + skp : i0]] : 0u;
      uint32_t hi = i1 < nx ? (uint32_t)(uint16_t)xr[mp->map[(skp && i1 >= olo) ? i1 + skp : i1]] : 0u;
      uint32_t lo2 = lo;
      if (jc && i0 == nx - 1) lo2 = (uint32_t)(uint16_t)jc;
      if (jc && i1 == nx - 1) hi = (uint32_t)(uint16_t)jc;
      xs[i] = lo2 | (hi << 16);
    }
    __syncthreads();
    if (a->m[0] == 2) { for (int i = tid; i < ne; i += XMX_THREADS) outs[i] = (float)p8s_squash(squash, xr[mp->exp[i]]) * cf; }
    else for (int i = tid; i < nx; i += XMX_THREADS) outs[i] = (float)p8s_squash(squash, reinterpret_cast<const int16_t*>(xs)[i]) * cf;
    uint4 w[4];
    int my_pr = 2048, row = 0;
    if (wave < nsel) {
      row = sel[(size_t)t * P8_NSEL + wave];
      const MX_GLOBAL int16_t* wr = wx + (size_t)row * P8_NX;
      uint32_t acc = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int grp = lane + 64 * g;
        w[g] = grp < MX_GROUPS ? mx_gload4(wr, grp) : make_uint4(0, 0, 0, 0);
        const uint4 xv = grp < MX_GROUPS ? reinterpret_cast<const uint4*>(xs)[grp] : make_uint4(0, 0, 0, 0);
        acc += pair_dot(xv.x, w[g].x) + pair_dot(xv.y, w[g].y) + pair_dot(xv.z, w[g].z) + pair_dot(xv.w, w[g].w);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      my_pr = p8s_squash(squash, (int32_t)(acc * 9u) >> 9);
      if (lane == 0) pr_s[wave] = my_pr;
    }
    __syncthreads();
    if (wave == 0) {   // second layer (its row is 32 wide; inputs behind nsel are zero)
      const int av = lane < nsel ? stretch[pr_s[lane]] : 0;
      if (lane < nsel) outs[ne + lane] = (float)p8s_squash(squash, av) * cf;
      const int bv = __shfl_down(av, 1);
      if ((lane & 1) == 0 && lane < 32) st_s[lane >> 1] = ((uint32_t)av & 0xffffu) | ((uint32_t)bv << 16);
      __builtin_amdgcn_s_waitcnt(0);
      __builtin_amdgcn_wave_barrier();
      uint32_t acc = 0;
      if (lane < 16) acc = pair_dot(st_s[lane], reinterpret_cast<const MX_GLOBAL uint32_t*>(wx2)[lane]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) p_s = p8s_squash(squash, (int32_t)acc >> 9);
    }
    __syncthreads();
    if (tid == 0) {   // the model's chain, serial on the tables in HBM
      T->misses = misses;
      fin_s = p8s_tail_image(T, a, y, p_s, outs + ne + nsel);
    }
    __syncthreads();
    float* orow = out + (size_t)t * ld;
    for (int i = tid; i < P8_NOUT; i += XMX_THREADS) orow[i] = outs[i];
    lastpr = fin_s;
    const int yb = bits[t];
    if (wave < nsel) {
      const int err = (int)(int16_t)(((yb << 12) - my_pr) * 7);
      MX_GLOBAL int16_t* wr = wx + (size_t)row * P8_NX;
      if (err) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int grp = lane + 64 * g;
          if (grp < MX_GROUPS) {
            const uint4 xv = reinterpret_cast<const uint4*>(xs)[grp];
            uint4 v = w[g];
            v.x = pair_train(xv.x, v.x, err); v.y = pair_train(xv.y, v.y, err); v.z = pair_train(xv.z, v.z, err); v.w = pair_train(xv.w, v.w, err);
            mx_gstore4(wr, grp, v);
          }
        }
      }
    }
    if (wave == 0 && lane < 16) {
      const int err2 = (int)(int16_t)(((yb << 12) - p_s) * 7);
      MX_GLOBAL uint32_t* w2 = reinterpret_cast<MX_GLOBAL uint32_t*>(wx2);
      if (err2) w2[lane] = pair_train(st_s[lane], w2[lane], err2);
    }
    __syncthreads();   // (drains the rows' stores before the next step's loads; xs / st_s / p_s are free again)
  }
  for (int i = tid; i < P8_NOUT; i += XMX_THREADS) T->out[i] = outs[i];
  if (tid == 0) { T->pr = lastpr; T->misses = misses; }
}

// ---------------------------------------------------------------- host side
namespace {
struct DevPolicy {
  std::vector<void*> blocks;
  bool ok = true;
  void* zalloc(size_t bytes) {
    void* p = nullptr;
    if (!ok || hipMalloc(&p, bytes + 64) != hipSuccess || hipMemset(p, 0, bytes + 64) != hipSuccess) { ok = false; return nullptr; }
    blocks.push_back(p);
    return p;
  }
  void upload(void* dst, const void* src, size_t bytes) { if (dst && hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) ok = false; }
  void fill16(int16_t* dst, int16_t v, size_t n) { if (dst && hipMemsetD16((hipDeviceptr_t)dst, (unsigned short)v, n) != hipSuccess) ok = false; }
};
enum { P8S_BUFS = CMX_PIPELINE_SLOTS, P8S_XBUFS = 3 };   // staging buffers; input-row buffers (maps -> family -> mixer: three chunks deep)
struct Staging {   // one chunk's records: page-locked host arrays and their device twins
  size_t cap = 0;  // bytes of input
  char* h = nullptr; char* d = nullptr;
  size_t o_fctx, o_fchk, o_cctx[P8_NCM2], o_cchk[P8_NCM2], o_ops, o_sel, o_apm, o_bits, o_model, total;
  char* hx = nullptr; char* dx = nullptr;   // the image models' records (xops, xfam_ctx, xfam_chk): uploaded only for a chunk that holds such bytes
  size_t ox_ops, ox_fctx, ox_fchk, xtotal;
  hipEvent_t done = nullptr;
  hipEvent_t t0[7] = {}, t1[7] = {};   // HIP-event brackets of the role kernels of the chunk: family, mixer, cm2[0..2], lanes, DMC
  bool used = false, timed = false;
};
template <class Tp> Tp* dev_copy(const Tp& host, DevPolicy& pol) {
  Tp* p = (Tp*)pol.zalloc(sizeof(Tp));
  pol.upload(p, &host, sizeof(Tp));
  return p;
}
}  // namespace

struct cmx_p8stage {
  int device = 0;
  P8Front* front = nullptr;
  P8Layout L;
  DevPolicy pol;
  P8CmDev* d_fam = nullptr; P8FamHome* d_fam_home = nullptr; size_t fam_lds = 0; P8Cm2Dev* d_cm2[P8_NCM2] = {}; P8LanesDev* d_lanes = nullptr; P8DmcDev* d_dmc = nullptr;
  P8TailDev* d_tail = nullptr; P8MixDev* d_mix = nullptr;
  P8CmDev* d_xfam[P8_NMODEL - 1] = {}; P8XLanesDev* d_xlanes[P8_NMODEL - 1] = {};   // the image models (p8_rec.h P8XLayout)
  P8ViewMap xview[P8_NMODEL - 1]; P8XMixMap* d_xmaps = nullptr;
  int last_byte = 0;                    // the last whole byte of the stream
  uint64_t image_chunks = 0;
  Staging st[P8S_BUFS];
  int next = 0;
  int16_t* d_x[P8S_XBUFS] = {}; uint8_t* d_order[P8S_XBUFS] = {}; size_t x_cap = 0;   // two chunks' input rows / order values: the mixer of chunk c runs under the tables of chunk c + 1
  hipStream_t s_a = nullptr, s_b = nullptr, s_c = nullptr, s_d = nullptr, s_e = nullptr, s_m = nullptr, s_f = nullptr;   // s_f: the DMC forest (needs the bits only)
  hipStream_t s_up = nullptr; bool own_up = false;   // the chunk's records go up on a stream that never has a kernel in front of a copy (cmx_p8stage_set_upload_stream)
  hipEvent_t ev_up = nullptr, ev_ord = nullptr, ev_a = nullptr, ev_b = nullptr, ev_c = nullptr, ev_e = nullptr, ev_f = nullptr, ev_mix[P8S_XBUFS] = {};
  bool mix_used[P8S_XBUFS] = {};
  unsigned long long* d_prof = nullptr;   // CMX_P8FAM_PROFILE=1: the family kernel's clocks by bit position and phase (cmx_p8stage_profile)
  unsigned long long* d_prx = nullptr; size_t prx_steps = 0; unsigned mix_epoch = 0;   // the mixer's exchange rows (cmx_p8s_mix4_kernel)
  unsigned* h_mixfail = nullptr;         // host-mapped: the mixer's workgroup 0 gave up waiting for another workgroup
  uint64_t chunks = 0;
  uint64_t steps = 0;
  int last_bit = 0;
  bool failed = false;
  float ms_front = 0;   // host time of the last front-end pass
  // the decoder's form (cmx_late.h): three chunk slots of host-coherent records and rows
  struct Late { size_t cap = 0; char* rec = nullptr; char* d_rec = nullptr; size_t o_fctx, o_fchk, o_cctx[P8_NCM2], o_cchk[P8_NCM2], o_ops, o_sel, o_apm, total; int16_t* x = nullptr; uint8_t* order = nullptr; P8Chunk c = P8Chunk(); } late[3];   // rec: the front end's records (host-coherent), d_rec: their device mirror (the relay copies them over step by step), same layout
  double role_ms[7] = {0, 0, 0, 0, 0, 0, 0}; uint64_t role_chunks = 0;   // summed over the chunks collected so far
};
static void p8s_collect(cmx_p8stage* h, Staging& b) {   // the chunk that used b is complete
  if (!b.timed) return;
  for (int i = 0; i < 7; i++) { float f = 0; if (hipEventElapsedTime(&f, b.t0[i], b.t1[i]) == hipSuccess) h->role_ms[i] += f; }
  h->role_chunks++;
  b.timed = false;
}

extern "C" {

void cmx_p8stage_destroy(cmx_p8stage_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->pol.blocks) (void)hipFree(p);
  for (auto& s : h->st) {
    if (s.h) (void)hipHostFree(s.h);
    if (s.d) (void)hipFree(s.d);
    if (s.hx) (void)hipHostFree(s.hx);
    if (s.dx) (void)hipFree(s.dx);
    if (s.done) (void)hipEventDestroy(s.done);
    for (hipEvent_t e : s.t0) if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : s.t1) if (e) (void)hipEventDestroy(e);
  }
  for (int i = 0; i < P8S_XBUFS; i++) { if (h->d_x[i]) (void)hipFree(h->d_x[i]); if (h->d_order[i]) (void)hipFree(h->d_order[i]); }
  {
    hipStream_t seen[7] = {}; int ns = 0;   // in compact modes several roles share a stream
    for (hipStream_t q : {h->s_a, h->s_b, h->s_c, h->s_d, h->s_e, h->s_m, h->s_f}) {
      bool dup = !q;
      for (int i = 0; i < ns; i++) dup = dup || seen[i] == q;
      if (!dup) { seen[ns++] = q; (void)hipStreamDestroy(q); }
    }
  }
  if (h->d_prx) (void)hipFree(h->d_prx);
  for (auto& b : h->late) { cmx_late_free(b.rec); cmx_late_free_dev(b.d_rec); cmx_late_free_dev(b.x); cmx_late_free_dev(b.order); }
  if (h->h_mixfail) (void)hipHostFree(h->h_mixfail);
  if (h->own_up && h->s_up) (void)hipStreamDestroy(h->s_up);
  for (hipEvent_t e : {h->ev_up, h->ev_ord, h->ev_a, h->ev_b, h->ev_c, h->ev_e, h->ev_f}) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_mix) if (e) (void)hipEventDestroy(e);
  if (h->front) p8f_front_free(h->front);
  delete h;
}

cmx_p8stage_t* cmx_p8stage_create(int device) {
  if (cmx_device_count() <= 0) { cmx_set_err("cmx_p8stage_create: no HIP device visible (a gfx950 GPU is required)"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return nullptr; }
  cmx_p8stage_t* h = new cmx_p8stage();
  h->device = device;
  h->front = p8f_front_new(11);   // cmix runs paq8 at level 11 (reference src/predictor.cpp:85)
  if (!h->front) { cmx_set_err("cmx_p8stage_create: front end construction failed"); delete h; return nullptr; }
  h->L = *p8f_front_layout(h->front);
  P8StageState* S = new P8StageState();
  bool ok = p8b::build_stage(*S, h->pol, h->L, 11, p8f_state_table(), p8f_stretch_table(), p8f_squash_table(), p8f_ilog_table()) && h->pol.ok;
  if (ok) {
    const char* serial = getenv("CMX_P8CM_SERIAL");   // A/B switch: the reference's serial walk on lane 0
    if (serial && serial[0] == '1') { S->fam.slot_parallel = 0; for (auto& c : S->cm2) c.slot_parallel = 0; }
    if (serial && serial[0] == '2') S->fam.slot_parallel = 2;   // test switch: the family's narrowed walk treats every second visit as unlisted (its fall-back path)
    h->d_fam = dev_copy(S->fam, h->pol);
    h->d_fam_home = S->fam_home;
    h->fam_lds = sizeof(P8FamShared) + (size_t)S->fam.nslots * 512;
    for (int k = 0; k < P8_NCM2; k++) h->d_cm2[k] = dev_copy(S->cm2[k], h->pol);
    h->d_lanes = dev_copy(S->lanes, h->pol);
    h->d_dmc = dev_copy(S->dmc, h->pol);
    h->d_tail = dev_copy(S->tail, h->pol);
    h->d_mix = dev_copy(S->mix, h->pol);
    for (int m = 0; m < P8_NMODEL - 1; m++)
      if (h->L.xl[m].nx) { h->d_xfam[m] = dev_copy(S->xfam[m], h->pol); h->d_xlanes[m] = dev_copy(S->xlanes[m], h->pol); h->xview[m] = S->xview[m]; }
    {
      std::vector<P8XMixMap> mm(P8_NMODEL - 1);
      for (int m = 0; m < P8_NMODEL - 1; m++) {
        memcpy(mm[m].map, h->L.xl[m].map, sizeof mm[m].map); mm[m].opt_lo = h->L.xl[m].opt_lo; mm[m].opt_n = h->L.xl[m].opt_n;
        memcpy(mm[m].exp, h->L.xl[m].exp, sizeof mm[m].exp); mm[m].exp_n = h->L.xl[m].exp_n;
      }
      h->d_xmaps = (P8XMixMap*)h->pol.zalloc(mm.size() * sizeof(P8XMixMap));
      h->pol.upload(h->d_xmaps, mm.data(), mm.size() * sizeof(P8XMixMap));
    }
    ok = h->pol.ok;
  }
  delete S;
  ok = ok && hipFuncSetAttribute((const void*)cmx_p8s_fam2_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->fam_lds) == hipSuccess;
  {
    // CMX_PIPELINE_STREAMS (throughput mode, several streams per GPU; every HIP stream is a hardware queue and the scheduler
    // time-slices queues past ~16-24): unset = one stream per role (6); 2 = the three ContextMap2 instances share one (4);
    // 1 = they also share it with the small lanes / DMC (3). The roles of a chunk then run one after the other on the shared
    // stream: the stage's period grows to their sum, the stream count is what many engines per GPU need.
    const char* ns = getenv("CMX_PIPELINE_STREAMS");
    const int mode = ns && ns[0] == '2' ? 2 : ns && ns[0] == '1' ? 1 : 0;
    for (hipStream_t* q : {&h->s_a, &h->s_d, &h->s_m}) ok = ok && cmx_make_stream(q, 0) == 0;
    if (mode == 0) for (hipStream_t* q : {&h->s_b, &h->s_e}) ok = ok && cmx_make_stream(q, 0) == 0;
    else { h->s_b = h->s_d; h->s_e = h->s_d; }
    if (mode == 1) h->s_c = h->s_d;
    else ok = ok && cmx_make_stream(&h->s_c, 0) == 0;
    if (mode == 0) ok = ok && cmx_make_stream(&h->s_f, 0) == 0;   // the DMC forest beside the small learners
    else h->s_f = h->s_c;
  }
  for (hipEvent_t* e : {&h->ev_up, &h->ev_ord, &h->ev_a, &h->ev_b, &h->ev_c, &h->ev_e, &h->ev_f}) ok = ok && hipEventCreateWithFlags(e, hipEventDisableTiming) == hipSuccess;
  for (hipEvent_t& e : h->ev_mix) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  for (auto& s : h->st) {
    ok = ok && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess;
    for (int i = 0; i < 7; i++) ok = ok && hipEventCreate(&s.t0[i]) == hipSuccess && hipEventCreate(&s.t1[i]) == hipSuccess;
  }
  ok = ok && hipHostMalloc((void**)&h->h_mixfail, 64, hipHostMallocMapped) == hipSuccess;
  if (ok) *h->h_mixfail = 0;
  if (ok && getenv("CMX_P8FAM_PROFILE")) ok = hipMalloc((void**)&h->d_prof, 128 * 8) == hipSuccess && hipMemset(h->d_prof, 0, 128 * 8) == hipSuccess;
  ok = ok && hipDeviceSynchronize() == hipSuccess;
  if (!ok) { cmx_set_err("cmx_p8stage_create: allocation / init failed (the stage needs ~9 GB of HBM)"); cmx_p8stage_destroy(h); return nullptr; }
  return h;
}

// The next nbytes bytes of the stream (host memory): front end on the calling thread, then the role kernels, ordered
// behind `stream`. d_out: device matrix, row t (ld floats apart) receives the 1591 values before bit t of this chunk.
int cmx_p8stage_run(cmx_p8stage_t* h, const uint8_t* bytes, size_t nbytes, float* d_out, size_t ld, void* stream) {
  if (!h) { cmx_set_err("cmx_p8stage_run: null handle"); return 1; }
  if (h->failed) { cmx_set_err("cmx_p8stage_run: the stage failed earlier on this stream"); return 1; }
  if (nbytes == 0) return 0;
  if (!bytes || !d_out || ld < P8_NOUT || nbytes > (1u << 22)) { cmx_set_err("cmx_p8stage_run: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  const P8Layout& L = h->L;
  Staging& b = h->st[h->next];
  h->next = (h->next + 1) % P8S_BUFS;
  if (b.used && hipEventSynchronize(b.done) != hipSuccess) { cmx_set_err("cmx_p8stage_run: staging buffer wait failed"); h->failed = true; return 1; }
  p8s_collect(h, b);
  const size_t n = nbytes, T = 8 * n;
  if (b.cap < n) {
    if (b.h) (void)hipHostFree(b.h);
    if (b.d) (void)hipFree(b.d);
    if (b.hx) (void)hipHostFree(b.hx);
    if (b.dx) (void)hipFree(b.dx);
    b.h = b.d = b.hx = b.dx = nullptr; b.cap = 0;
    size_t o = 0;
    auto take = [&](size_t bytes_) { const size_t at = o; o += (bytes_ + 255) & ~(size_t)255; return at; };
    b.o_fctx = take(n * L.fam_slots * 4); b.o_fchk = take(n * L.fam_slots * 2);
    for (int k = 0; k < P8_NCM2; k++) { b.o_cctx[k] = take(n * L.cm2_count[k] * 4); b.o_cchk[k] = take(n * L.cm2_count[k] * 2); }
    b.o_ops = take(T * P8_NLANE * 4); b.o_sel = take(T * P8_NSEL * 4); b.o_apm = take(T * sizeof(P8ApmRec)); b.o_bits = take(T); b.o_model = take(n);
    b.total = o;
    o = 0;
    b.ox_ops = take(T * P8_XL_NLANE * 4); b.ox_fctx = take(n * P8_XL_MAXS * 4); b.ox_fchk = take(n * P8_XL_MAXS * 2);
    b.xtotal = o;
    if (hipHostMalloc((void**)&b.h, b.total, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&b.d, b.total) != hipSuccess ||
        hipHostMalloc((void**)&b.hx, b.xtotal, hipHostMallocDefault) != hipSuccess || hipMalloc((void**)&b.dx, b.xtotal) != hipSuccess) {
      cmx_set_err("cmx_p8stage_run: staging allocation failed"); h->failed = true; return 1;
    }
    b.cap = n;
  }
  P8Chunk c = P8Chunk();
  c.fam_ctx = (uint32_t*)(b.h + b.o_fctx); c.fam_chk = (uint16_t*)(b.h + b.o_fchk);
  for (int k = 0; k < P8_NCM2; k++) { c.cm2_ctx[k] = (uint32_t*)(b.h + b.o_cctx[k]); c.cm2_chk[k] = (uint16_t*)(b.h + b.o_cchk[k]); }
  c.ops = (uint32_t*)(b.h + b.o_ops); c.sel = (int32_t*)(b.h + b.o_sel); c.apm = (P8ApmRec*)(b.h + b.o_apm);
  c.model = (uint8_t*)(b.h + b.o_model); c.xops = (uint32_t*)(b.hx + b.ox_ops); c.xfam_ctx = (uint32_t*)(b.hx + b.ox_fctx); c.xfam_chk = (uint16_t*)(b.hx + b.ox_fchk);
  memset(c.model, 0, n);
  const int rc = p8f_front_run(h->front, bytes, n, &c);
  if (rc) { cmx_set_err(std::string("cmx_p8stage_run: ") + p8f_strerror(rc)); h->failed = true; return 1; }
  bool any_image = false;
  for (size_t i = 0; i < n && !any_image; i++) any_image = c.model[i] != 0;
  uint8_t* hb = (uint8_t*)(b.h + b.o_bits);
  for (size_t i = 0; i < T; i++) hb[i] = (bytes[i >> 3] >> (7 - (i & 7))) & 1;
  hipStream_t s = (hipStream_t)stream;
  bool ok = true;
  if (h->x_cap < n) {   // grown between chunks only when nothing is in flight on them
    ok = hipDeviceSynchronize() == hipSuccess;
    for (int i = 0; i < P8S_XBUFS; i++) {
      if (h->d_x[i]) (void)hipFree(h->d_x[i]);
      if (h->d_order[i]) (void)hipFree(h->d_order[i]);
      h->d_x[i] = nullptr; h->d_order[i] = nullptr;
      ok = ok && hipMalloc((void**)&h->d_x[i], T * P8_NX * 2) == hipSuccess && hipMalloc((void**)&h->d_order[i], T) == hipSuccess;
    }
    if (h->d_prx) (void)hipFree(h->d_prx);
    h->d_prx = nullptr;
    ok = ok && hipMalloc((void**)&h->d_prx, T * P8_NSEL * 8) == hipSuccess && hipMemset(h->d_prx, 0, T * P8_NSEL * 8) == hipSuccess;
    h->prx_steps = ok ? T : 0;
    h->x_cap = ok ? n : 0;
  }
  if (h->h_mixfail && *h->h_mixfail) { cmx_set_err("cmx_p8stage_run: the mixer kernel's workgroup 0 timed out waiting for another workgroup's outputs (are four workgroups co-resident?)"); h->failed = true; return 1; }
  // The stage's own streams: d = upload, order-N map; a = family (needs the order-N map's `order` of the same chunk, so it
  // runs one kernel behind it -- on its own stream the family of chunk c overlaps the order-N map of chunk c + 1);
  // b = TextModel's and exeModel's maps, c = small lanes -> DMC, m = mixer + chains. Input rows are double-buffered by chunk parity, so the mixer of chunk c runs under the tables of
  // chunk c + 1; the caller's stream only waits for this chunk's mixer at the end (the input is host memory: nothing of
  // the caller's earlier work is needed, d_out must simply not be in use).
  const int par = (int)(h->chunks % P8S_XBUFS);
  int16_t* dx = h->d_x[par]; uint8_t* dord = h->d_order[par];
  if (h->mix_used[par]) for (hipStream_t q : {h->s_a, h->s_b, h->s_c, h->s_d, h->s_e, h->s_f}) ok = ok && hipStreamWaitEvent(q, h->ev_mix[par], 0) == hipSuccess;   // the mixer that last read these rows
  // upload on the upload stream: on s_d the copy would sit behind the previous chunk's order-N kernel, and a host-to-device
  // copy that waits in stream order holds up every later copy of the process
  if (!h->s_up) { ok = ok && hipStreamCreateWithFlags(&h->s_up, hipStreamNonBlocking) == hipSuccess; h->own_up = true; }
  ok = ok && hipMemcpyAsync(b.d, b.h, b.total, hipMemcpyHostToDevice, h->s_up) == hipSuccess;
  ok = ok && hipEventRecord(h->ev_up, h->s_up) == hipSuccess;
  ok = ok && hipStreamWaitEvent(h->s_d, h->ev_up, 0) == hipSuccess;
  const int nbits = (int)T;
  const int skip = h->steps >= 8 ? 0 : (int)(8 - h->steps), t0 = h->steps == 0 ? 1 : 0;
  const uint8_t* d_bits = (const uint8_t*)(b.d + b.o_bits);
  const uint8_t* d_model = (const uint8_t*)(b.d + b.o_model);
  auto cm2 = [&](int k, hipStream_t q, uint8_t* ord, const uint8_t* mdl) {
    (void)hipEventRecord(b.t0[2 + k], q);
    hipLaunchKernelGGL(cmx_p8s_cm2v2_kernel, dim3(1), dim3(P8CM2_MAXC), 0, q, h->d_cm2[k], (const uint32_t*)(b.d + b.o_cctx[k]),
                       (const uint16_t*)(b.d + b.o_cchk[k]), d_bits, dx, ord, nbits, skip, mdl);
    (void)hipEventRecord(b.t1[2 + k], q);
  };
  if (ok && any_image) {
    // ---- a chunk that holds bytes of an image model (p8_rec.h P8XLayout): every role of the chunk on ONE stream (s_m), behind everything the
    // stage has in flight, in dependency order; the ContextMap family and the mixer by segments of consecutive bytes of one kind -- the
    // generic kernels on the generic bytes, the image model's on its own, the rnd() stream handed over at every switch. Correctness first:
    // such chunks are rare, and an image byte still costs far less here than on the reference's CPU path.
    hipStream_t q = h->s_m;
    { hipStream_t rs[6] = {h->s_a, h->s_b, h->s_c, h->s_d, h->s_e, h->s_f}; hipEvent_t re[6] = {h->ev_a, h->ev_b, h->ev_c, h->ev_ord, h->ev_e, h->ev_f};
      for (int i = 0; i < 6; i++) ok = ok && hipEventRecord(re[i], rs[i]) == hipSuccess && hipStreamWaitEvent(q, re[i], 0) == hipSuccess; }
    ok = ok && hipMemcpyAsync(b.dx, b.hx, b.xtotal, hipMemcpyHostToDevice, h->s_up) == hipSuccess;
    ok = ok && hipEventRecord(h->ev_up, h->s_up) == hipSuccess;
    ok = ok && hipStreamWaitEvent(q, h->ev_up, 0) == hipSuccess;
    for (int i = 0; i < 7; i++) (void)hipEventRecord(b.t0[i], q);
    cm2(0, q, dord, nullptr);
    cm2(1, q, nullptr, d_model);
    cm2(2, q, nullptr, d_model);
    hipLaunchKernelGGL(cmx_p8s_lanes_kernel, dim3(1), dim3(P8LANES_THREADS), 0, q, h->d_lanes, (const uint32_t*)(b.d + b.o_ops), d_bits, (const uint8_t*)dord, dx, nbits, t0, d_model);
    hipLaunchKernelGGL(cmx_p8s_dmc_kernel, dim3(1), dim3(P8DMC_THREADS), 0, q, h->d_dmc, d_bits, dx, (int)L.dmc_off, nbits, t0, d_model);
    for (int m = 0; m < P8_NMODEL - 1; m++)
      if (h->d_xlanes[m])
        hipLaunchKernelGGL(cmx_p8s_xlanes_kernel, dim3(1), dim3(P8_XL_NLANE), 0, q, h->d_xlanes[m], (const uint32_t*)(b.dx + b.ox_ops), d_bits, (const uint8_t*)dord, dx, d_model, nbits, t0,
                           h->last_bit);
    // segments: [b0, b1) bytes of one model
    int owner = 0;   // whose copy of the generator is current (0: the generic family's -- the state between chunks)
    for (int pass = 0; pass < 2 && ok; pass++) {   // pass 0: the ContextMaps, pass 1: the mixers (they need every input row of their steps)
      size_t b0 = 0;
      while (b0 < n) {
        const int md = c.model[b0];
        size_t b1 = b0 + 1;
        while (b1 < n && c.model[b1] == md) ++b1;
        const size_t s0 = 8 * b0;
        const int sbits = (int)(8 * (b1 - b0));
        const int ly = s0 ? (int)hb[s0 - 1] : h->last_bit, lc1 = b0 ? (int)bytes[b0 - 1] : h->last_byte;
        if (pass == 0) {
          if (md == 0) {
            if (owner) {
              if (h->xview[owner - 1].n) hipLaunchKernelGGL(cmx_p8s_view_out_kernel, dim3(1), dim3(64), 0, q, (const P8CmDev*)h->d_xfam[owner - 1], h->d_fam, h->d_fam_home, h->xview[owner - 1]);
              hipLaunchKernelGGL(cmx_p8s_rnd_copy_kernel, dim3(1), dim3(64), 0, q, h->d_fam, (const P8CmDev*)h->d_xfam[owner - 1]);
              owner = 0;
            }
            if (b0) hipLaunchKernelGGL(cmx_p8s_fam_resume_kernel, dim3(1), dim3(1), 0, q, h->d_fam, ly, lc1);   // (a chunk's first segment: the family's own registers are current)
            hipLaunchKernelGGL(cmx_p8s_fam2_kernel, dim3(1), dim3(P8FAM_THREADS), h->fam_lds, q, h->d_fam, h->d_fam_home, (const uint32_t*)(b.d + b.o_fctx) + b0 * L.fam_slots,
                               (const uint16_t*)(b.d + b.o_fchk) + b0 * L.fam_slots, d_bits + s0, dx + s0 * P8_NX, (const uint8_t*)dord + s0, sbits, b0 == 0 ? skip : 0, h->d_prof);
          } else {
            if (owner != md) {
              if (owner && h->xview[owner - 1].n) hipLaunchKernelGGL(cmx_p8s_view_out_kernel, dim3(1), dim3(64), 0, q, (const P8CmDev*)h->d_xfam[owner - 1], h->d_fam, h->d_fam_home, h->xview[owner - 1]);
              hipLaunchKernelGGL(cmx_p8s_rnd_copy_kernel, dim3(1), dim3(64), 0, q, h->d_xfam[md - 1], (const P8CmDev*)(owner ? h->d_xfam[owner - 1] : h->d_fam));
              if (h->xview[md - 1].n) hipLaunchKernelGGL(cmx_p8s_view_in_kernel, dim3(1), dim3(64), 0, q, h->d_xfam[md - 1], (const P8CmDev*)h->d_fam, (const P8FamHome*)h->d_fam_home, h->xview[md - 1]);
              owner = md;
            }
            if (h->L.xl[md - 1].nslots)   // (a model without ContextMaps -- im1bitModel -- only holds the generator for the moment)
            hipLaunchKernelGGL(cmx_p8s_xfam_kernel, dim3(1), dim3(64), 0, q, h->d_xfam[md - 1], (const uint32_t*)(b.dx + b.ox_fctx) + b0 * P8_XL_MAXS,
                               (const uint16_t*)(b.dx + b.ox_fchk) + b0 * P8_XL_MAXS, d_bits + s0, dx + s0 * P8_NX, sbits, ly, lc1);
          }
        } else {
          if (md == 0) {
            ++h->mix_epoch;
            hipLaunchKernelGGL(cmx_p8s_mix4_kernel, dim3(4), dim3(MX_THREADS), 0, q, (const P8MixDev*)h->d_mix, h->d_tail, (const int16_t*)dx + s0 * P8_NX, (const int32_t*)(b.d + b.o_sel) + s0 * P8_NSEL,
                               (const P8ApmRec*)(b.d + b.o_apm) + s0, (const uint8_t*)dord + s0, d_bits + s0, d_out + s0 * ld, ld, sbits, b0 == 0 ? t0 : 0, b0 == 0 ? skip : 0, ly, h->d_prx,
                               h->mix_epoch, h->h_mixfail);
          } else {
            hipLaunchKernelGGL(cmx_p8s_xmix_kernel, dim3(1), dim3(XMX_THREADS), 0, q, (const P8MixDev*)h->d_mix, h->d_tail, (const P8XMixMap*)h->d_xmaps, (const int16_t*)dx + s0 * P8_NX, (const int32_t*)(b.d + b.o_sel) + s0 * P8_NSEL,
                               (const P8ApmRec*)(b.d + b.o_apm) + s0, d_bits + s0, d_out + s0 * ld, ld, sbits, ly);
          }
        }
        b0 = b1;
      }
      if (pass == 0 && owner) {   // between chunks the generator -- and everything a model's family shares with it -- is the generic family's, and its registers follow the stream
        if (h->xview[owner - 1].n) hipLaunchKernelGGL(cmx_p8s_view_out_kernel, dim3(1), dim3(64), 0, q, (const P8CmDev*)h->d_xfam[owner - 1], h->d_fam, h->d_fam_home, h->xview[owner - 1]);
        hipLaunchKernelGGL(cmx_p8s_rnd_copy_kernel, dim3(1), dim3(64), 0, q, h->d_fam, (const P8CmDev*)h->d_xfam[owner - 1]);
        owner = 0;
      }
      if (pass == 0 && c.model[n - 1] != 0) hipLaunchKernelGGL(cmx_p8s_fam_resume_kernel, dim3(1), dim3(1), 0, q, h->d_fam, (int)hb[T - 1], (int)bytes[n - 1]);
    }
    ok = ok && hipGetLastError() == hipSuccess;
    for (int i = 0; i < 7; i++) (void)hipEventRecord(b.t1[i], q);
    b.timed = true;
    ok = ok && hipEventRecord(h->ev_mix[par], q) == hipSuccess;
    ok = ok && hipEventRecord(b.done, q) == hipSuccess;
    for (hipStream_t r : {h->s_a, h->s_b, h->s_c, h->s_d, h->s_e, h->s_f}) ok = ok && hipStreamWaitEvent(r, h->ev_mix[par], 0) == hipSuccess;   // the next chunk's roles start behind this one
    ok = ok && hipStreamWaitEvent(s, h->ev_mix[par], 0) == hipSuccess;
    h->mix_used[par] = true;
    h->chunks++;
    h->image_chunks++;
  } else if (ok) {
    cm2(0, h->s_d, dord, nullptr);
    ok = hipEventRecord(h->ev_ord, h->s_d) == hipSuccess;
    ok = ok && hipStreamWaitEvent(h->s_a, h->ev_ord, 0) == hipSuccess;   // (implies the upload)
    (void)hipEventRecord(b.t0[0], h->s_a);
    hipLaunchKernelGGL(cmx_p8s_fam2_kernel, dim3(1), dim3(P8FAM_THREADS), h->fam_lds, h->s_a, h->d_fam, h->d_fam_home, (const uint32_t*)(b.d + b.o_fctx),
                       (const uint16_t*)(b.d + b.o_fchk), d_bits, dx, (const uint8_t*)dord, nbits, skip, h->d_prof);
    (void)hipEventRecord(b.t1[0], h->s_a);
    ok = ok && hipEventRecord(h->ev_a, h->s_a) == hipSuccess;
    ok = ok && hipStreamWaitEvent(h->s_b, h->ev_up, 0) == hipSuccess;
    cm2(1, h->s_b, nullptr, nullptr);
    ok = ok && hipEventRecord(h->ev_b, h->s_b) == hipSuccess;
    ok = ok && hipStreamWaitEvent(h->s_e, h->ev_up, 0) == hipSuccess;
    cm2(2, h->s_e, nullptr, nullptr);
    ok = ok && hipEventRecord(h->ev_e, h->s_e) == hipSuccess;
    ok = ok && hipStreamWaitEvent(h->s_c, h->ev_ord, 0) == hipSuccess;
    (void)hipEventRecord(b.t0[5], h->s_c);
    hipLaunchKernelGGL(cmx_p8s_lanes_kernel, dim3(1), dim3(P8LANES_THREADS), 0, h->s_c, h->d_lanes, (const uint32_t*)(b.d + b.o_ops), d_bits, (const uint8_t*)dord, dx, nbits, t0,
                       (const uint8_t*)nullptr);
    (void)hipEventRecord(b.t1[5], h->s_c);
    ok = ok && hipEventRecord(h->ev_c, h->s_c) == hipSuccess;
    // the DMC forest reads the coded bits only: on a stream of its own it runs beside the small learners (4.0 + 2.8 us/bit in a row before)
    if (h->s_f != h->s_c) ok = ok && hipStreamWaitEvent(h->s_f, h->ev_up, 0) == hipSuccess;
    (void)hipEventRecord(b.t0[6], h->s_f);
    hipLaunchKernelGGL(cmx_p8s_dmc_kernel, dim3(1), dim3(P8DMC_THREADS), 0, h->s_f, h->d_dmc, d_bits, dx, (int)L.dmc_off, nbits, t0, (const uint8_t*)nullptr);
    (void)hipEventRecord(b.t1[6], h->s_f);
    ok = ok && hipEventRecord(h->ev_f, h->s_f) == hipSuccess;
    for (hipEvent_t e : {h->ev_a, h->ev_b, h->ev_c, h->ev_e, h->ev_f}) ok = ok && hipStreamWaitEvent(h->s_m, e, 0) == hipSuccess;
    (void)hipEventRecord(b.t0[1], h->s_m);
    {
      ++h->mix_epoch;
      hipLaunchKernelGGL(cmx_p8s_mix4_kernel, dim3(4), dim3(MX_THREADS), 0, h->s_m, (const P8MixDev*)h->d_mix, h->d_tail, (const int16_t*)dx, (const int32_t*)(b.d + b.o_sel),
                         (const P8ApmRec*)(b.d + b.o_apm), (const uint8_t*)dord, d_bits, d_out, ld, nbits, t0, skip, h->last_bit, h->d_prx, h->mix_epoch, h->h_mixfail);
    }
    ok = ok && hipGetLastError() == hipSuccess;
    (void)hipEventRecord(b.t1[1], h->s_m);
    b.timed = true;
    ok = ok && hipEventRecord(h->ev_mix[par], h->s_m) == hipSuccess;
    ok = ok && hipEventRecord(b.done, h->s_m) == hipSuccess;
    ok = ok && hipStreamWaitEvent(s, h->ev_mix[par], 0) == hipSuccess;   // what the caller enqueues next sees this chunk's rows of d_out
    h->mix_used[par] = true;
    h->chunks++;
  }
  if (!ok) { cmx_set_err(std::string("cmx_p8stage_run: launch failed: ") + hipGetErrorString(hipGetLastError())); h->failed = true; return 1; }
  b.used = true;
  h->steps += T;
  h->last_bit = hb[T - 1];
  h->last_byte = bytes[n - 1];
  return 0;
}

// ---- the decoder's form of a chunk (cmx_late.h) -------------------------------------------------------------------------------
// Launches the role kernels of the next nbytes bytes; their bits, and with each bit the host records of the step after it, arrive
// through `box` while they run. slot 0..2: which of the stage's three sets of host-coherent buffers the chunk uses (a chunk is
// launched while its predecessor is still being decoded, and its predecessor's buffers are still read when it starts).
// Everything the decoder's form allocates, for chunks of up to nbytes bytes: before the first chunk's kernels are launched (an
// allocation that maps memory into the device while persistent kernels wait for the host would wait for them).
int cmx_p8stage_late_prepare(cmx_p8stage_t* h, size_t nbytes) {
  if (!h || nbytes == 0 || nbytes > (1u << 16)) { cmx_set_err("cmx_p8stage_late_prepare: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  const P8Layout& L = h->L;
  const size_t n = nbytes, T = 8 * n;
  for (cmx_p8stage::Late& b : h->late) {
    if (b.cap >= n) continue;
    if (b.cap) { cmx_set_err("cmx_p8stage_late_prepare: the chunk size may not grow"); return 1; }
    size_t o = 0;
    auto take = [&](size_t bytes_) { const size_t at = o; o += (bytes_ + 255) & ~(size_t)255; return at; };
    b.o_fctx = take(n * L.fam_slots * 4); b.o_fchk = take(n * L.fam_slots * 2);
    for (int k = 0; k < P8_NCM2; k++) { b.o_cctx[k] = take(n * L.cm2_count[k] * 4); b.o_cchk[k] = take(n * L.cm2_count[k] * 2); }
    b.o_ops = take(T * P8_NLANE * 4); b.o_sel = take(T * P8_NSEL * 4); b.o_apm = take(T * sizeof(P8ApmRec));
    b.total = o;
    b.rec = (char*)cmx_late_alloc(o);
    b.d_rec = (char*)cmx_late_alloc_dev(h->device, o);
    b.x = (int16_t*)cmx_late_alloc_dev(h->device, T * P8_NX * 2);   // rows the role kernels hand to the mixer while all of them run
    b.order = (uint8_t*)cmx_late_alloc_dev(h->device, T);
    if (!b.rec || !b.d_rec || !b.x || !b.order) { cmx_set_err("cmx_p8stage_late_prepare: buffer allocation failed"); h->failed = true; return 1; }
    b.cap = n;
    b.c.fam_ctx = (uint32_t*)(b.rec + b.o_fctx); b.c.fam_chk = (uint16_t*)(b.rec + b.o_fchk);
    for (int k = 0; k < P8_NCM2; k++) { b.c.cm2_ctx[k] = (uint32_t*)(b.rec + b.o_cctx[k]); b.c.cm2_chk[k] = (uint16_t*)(b.rec + b.o_cchk[k]); }
    b.c.ops = (uint32_t*)(b.rec + b.o_ops); b.c.sel = (int32_t*)(b.rec + b.o_sel); b.c.apm = (P8ApmRec*)(b.rec + b.o_apm);
  }
  if (h->prx_steps < T) {
    bool ok = hipDeviceSynchronize() == hipSuccess;
    if (h->d_prx) (void)hipFree(h->d_prx);
    h->d_prx = nullptr;
    ok = ok && hipMalloc((void**)&h->d_prx, T * P8_NSEL * 8) == hipSuccess && hipMemset(h->d_prx, 0, T * P8_NSEL * 8) == hipSuccess;
    h->prx_steps = ok ? T : 0;
    if (!ok) { cmx_set_err("cmx_p8stage_late_prepare: exchange rows allocation failed"); return 1; }
  }
  if (hipFuncSetAttribute((const void*)cmx_p8s_fam2_late_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->fam_lds) != hipSuccess) { cmx_set_err("cmx_p8stage_late_prepare: kernel attribute failed"); return 1; }
  return 0;
}
int cmx_p8stage_run_late(cmx_p8stage_t* h, void* box_, size_t nbytes, float* d_out, size_t ld, int slot) {
  if (!h || !box_ || !d_out || nbytes == 0 || nbytes > (1u << 16) || ld < P8_NOUT || slot < 0 || slot > 2) { cmx_set_err("cmx_p8stage_run_late: bad argument"); return 1; }
  if (h->failed) { cmx_set_err("cmx_p8stage_run_late: the stage failed earlier on this stream"); return 1; }
  if (h->s_b == h->s_d || h->s_c == h->s_d || h->s_f == h->s_c) { cmx_set_err("cmx_p8stage_run_late: the roles share HIP streams (CMX_PIPELINE_STREAMS): a decoder needs every role kernel running at once"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  const CmxLate B = *(const CmxLate*)box_;
  const P8Layout& L = h->L;
  const size_t n = nbytes, T = 8 * n;
  cmx_p8stage::Late& b = h->late[slot];
  if (b.cap < n || h->prx_steps < T) { cmx_set_err("cmx_p8stage_run_late: call cmx_p8stage_late_prepare first (nothing may be allocated while the stream's kernels run)"); return 1; }
  bool ok = true;
  const int nbits = (int)T;
  const int skip = h->steps >= 8 ? 0 : (int)(8 - h->steps), t0 = h->steps == 0 ? 1 : 0;
  if (ok) {
    // every role on its own stream (all of them run at the same time, for the whole chunk)
    for (int k = 0; k < P8_NCM2; k++) {
      hipStream_t q = k == 0 ? h->s_d : k == 1 ? h->s_b : h->s_e;
      hipLaunchKernelGGL(cmx_p8s_cm2v2_late_kernel, dim3(1), dim3(P8CM2_MAXC), 0, q, h->d_cm2[k], B, (int)(LC_CM2_0 + k), (const uint32_t*)(b.d_rec + b.o_cctx[k]), (const uint16_t*)(b.d_rec + b.o_cchk[k]),
                         b.x, k == 0 ? b.order : (uint8_t*)nullptr, nbits, skip);
    }
    // (every record pointer below is the DEVICE mirror: the relay wave fills it step by step, cmx_p8stage_late_relay)
    hipLaunchKernelGGL(cmx_p8s_fam2_late_kernel, dim3(1), dim3(P8FAM_THREADS), h->fam_lds, h->s_a, h->d_fam, h->d_fam_home, B, (const uint32_t*)(b.d_rec + b.o_fctx), (const uint16_t*)(b.d_rec + b.o_fchk),
                       b.x, (const uint8_t*)b.order, nbits, skip);
    hipLaunchKernelGGL(cmx_p8s_lanes_late_kernel, dim3(1), dim3(P8LANES_THREADS), 0, h->s_c, h->d_lanes, B, (const uint32_t*)(b.d_rec + b.o_ops), (const uint8_t*)b.order, b.x, nbits, t0);
    hipLaunchKernelGGL(cmx_p8s_dmc_late_kernel, dim3(1), dim3(P8DMC_THREADS), 0, h->s_f, h->d_dmc, B, b.x, (int)L.dmc_off, nbits, t0);
    ++h->mix_epoch;
    hipLaunchKernelGGL(cmx_p8s_mix4_late_kernel, dim3(4), dim3(MX_THREADS), 0, h->s_m, (const P8MixDev*)h->d_mix, h->d_tail, B, (const int16_t*)b.x, (const int32_t*)(b.d_rec + b.o_sel),
                       (const P8ApmRec*)(b.d_rec + b.o_apm), (const uint8_t*)b.order, d_out, ld, nbits, t0, skip, h->d_prx, h->mix_epoch, h->h_mixfail);
    ok = hipGetLastError() == hipSuccess;
  }
  if (!ok) { cmx_set_err(std::string("cmx_p8stage_run_late: launch failed: ") + hipGetErrorString(hipGetLastError())); h->failed = true; return 1; }
  h->chunks++;
  h->steps += T;
  return 0;
}
// What the stream's relay wave has to bring over for this stage (cmx_late.h): every record array of the chunk slot -- host source, device
// mirror, bytes per row, when a row is written. Returns the number of entries (<= max), -1 on error.
int cmx_p8stage_late_relay(cmx_p8stage_t* h, int slot, void* out_, int max) {
  if (!h || slot < 0 || slot > 2 || !h->late[slot].cap || !out_) { cmx_set_err("cmx_p8stage_late_relay: bad argument (prepare first)"); return -1; }
  cmx_late_relay_t* out = (cmx_late_relay_t*)out_;
  const cmx_p8stage::Late& b = h->late[slot];
  const P8Layout& L = h->L;
  int n = 0;
  auto add = [&](size_t off, size_t stride, int kind) { if (n < max) { out[n].src = b.rec + off; out[n].dst = b.d_rec + off; out[n].stride = (uint32_t)stride; out[n].kind = kind; } ++n; };
  add(b.o_ops, P8_NLANE * 4, 0); add(b.o_sel, P8_NSEL * 4, 0); add(b.o_apm, sizeof(P8ApmRec), 0);
  add(b.o_fctx, (size_t)L.fam_slots * 4, 1); add(b.o_fchk, (size_t)L.fam_slots * 2, 1);
  for (int k = 0; k < P8_NCM2; k++) { add(b.o_cctx[k], (size_t)L.cm2_count[k] * 4, 1); add(b.o_cchk[k], (size_t)L.cm2_count[k] * 2, 1); }
  if (n > max) { cmx_set_err("cmx_p8stage_late_relay: table too small"); return -1; }
  return n;
}
// the host half of the decoder's form: the records of chunk-local step `step` of the chunk in `slot` (every bit before that step has
// been handed to cmx_p8stage_late_bit), then the bit that was decoded with them
int cmx_p8stage_late_emit(cmx_p8stage_t* h, int slot, size_t step) {
  if (!h || slot < 0 || slot > 2 || !h->late[slot].cap || step >= 8 * h->late[slot].cap) { cmx_set_err("cmx_p8stage_late_emit: bad argument"); return 1; }
  const int rc = p8f_front_emit_step(h->front, &h->late[slot].c, step);
  if (rc) { cmx_set_err(std::string("cmx_p8stage_late_emit: ") + p8f_strerror(rc)); h->failed = true; return 1; }
  return 0;
}
int cmx_p8stage_late_bit(cmx_p8stage_t* h, int bit) {
  if (!h) { cmx_set_err("cmx_p8stage_late_bit: null handle"); return 1; }
  p8f_front_set_bit(h->front, bit);
  h->last_bit = bit ? 1 : 0;
  return 0;
}
// test hook: place the counter of the ContextMap family's shared generator (Random::i, paq8.cpp:154) before the first byte. The generator's VALUES do not
// depend on the counter, only on counter mod 64 (the 64 table words keep their places for a multiple of 64): tests put it shortly before 2^31 / 2^32 to
// check in seconds what a stream reaches after 4 / 8 MB (tests/test_zgpu_p8stage.py, round 5's 8 MiB finding).
int cmx_p8stage_set_generator_counter(cmx_p8stage_t* h, uint32_t counter) {
  if (!h || !h->d_fam) { cmx_set_err("cmx_p8stage_set_generator_counter: null handle"); return 1; }
  if (counter & 63u) { cmx_set_err("cmx_p8stage_set_generator_counter: a multiple of 64 (the table words stay where they are)"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess || hipDeviceSynchronize() != hipSuccess) { cmx_set_err("cmx_p8stage_set_generator_counter: device error"); return 1; }
  const int v = (int)counter;
  if (hipMemcpy((char*)h->d_fam + offsetof(P8CmDev, rnd) + offsetof(P8Rnd, i), &v, sizeof v, hipMemcpyHostToDevice) != hipSuccess) { cmx_set_err("cmx_p8stage_set_generator_counter: copy failed"); return 1; }
  return 0;
}
// Test hook (state injection, round 6): the front end's byte position -- the index into paq8's 2^30-byte history ring (paq8.cpp:167-186), which a stream reaches
// after 1 GB; tests/golden/paq8_cols_pos_1g_6k.npz holds the unmodified paq8::Predictor's values across it. Before the first chunk only.
int cmx_p8stage_debug_set_pos(cmx_p8stage_t* h, int pos) {
  if (!h || !h->front) { cmx_set_err("cmx_p8stage_debug_set_pos: null handle"); return 1; }
  p8f_front_set_pos(h->front, pos);
  return 0;
}
int cmx_p8stage_mixfail(cmx_p8stage_t* h) { return h && h->h_mixfail && *h->h_mixfail ? 1 : 0; }

int cmx_p8stage_set_upload_stream(cmx_p8stage_t* h, void* stream) {
  if (!h) { cmx_set_err("cmx_p8stage_set_upload_stream: null handle"); return 1; }
  if (h->own_up && h->s_up) (void)hipStreamDestroy(h->s_up);
  h->s_up = (hipStream_t)stream; h->own_up = false;
  return 0;
}

// diagnostics (CMX_P8FAM_PROFILE=1 at create time): thread 0 of the family kernel, clocks summed over all steps so far: out[8 bp + k], k = 0 per-step
// values, 1 phase 1, 2 barrier, 3 run (common path) or rounds, 4 rest; out[64 + bp] steps that took the rounds path, out[72 + bp] instances walked,
// out[80 + bp] steps
int cmx_p8stage_profile(cmx_p8stage_t* h, unsigned long long out128[128]) {
  if (!h || !h->d_prof) return 1;
  return hipMemcpy(out128, h->d_prof, 128 * 8, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}

int cmx_p8stage_sync(cmx_p8stage_t* h) {
  if (!h) { cmx_set_err("cmx_p8stage_sync: null handle"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_p8stage_sync: ") + hipGetErrorString(e)); h->failed = true; return 1; }
  for (auto& b : h->st) p8s_collect(h, b);
  if (h->h_mixfail && *h->h_mixfail) { cmx_set_err("cmx_p8stage_sync: the mixer kernel's workgroup 0 timed out waiting for another workgroup's outputs"); h->failed = true; return 1; }
  return 0;
}

// HIP-event time of the role kernels, summed over the chunks completed and collected so far (a chunk is collected when its
// staging buffer comes round again, or by cmx_p8stage_sync): ms[0] family, [1] mixer + APM chains, [2..4] the three
// ContextMap2 instances, [5] small lanes + DMC. They run on streams of their own: the stage's period per chunk is the
// largest of them (the family after the order-N map of the same chunk, the mixer after all), not their sum.
int cmx_p8stage_role_ms(cmx_p8stage_t* h, double ms[7], uint64_t* chunks, int reset) {
  if (!h || !ms || !chunks) return 1;
  for (int i = 0; i < 7; i++) ms[i] = h->role_ms[i];
  *chunks = h->role_chunks;
  if (reset) { for (double& v : h->role_ms) v = 0; h->role_chunks = 0; }
  return 0;
}

}  // extern "C"
