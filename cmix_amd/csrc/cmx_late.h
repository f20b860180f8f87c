// cmx_late.h -- the LATE-BIT protocol: the stage kernels of the chunk pipeline driven by a decoder.
//
// Decoder::Decode (reference src/coder/decoder.cpp:20-39) calls Predictor::Predict() and only then learns the bit it hands to
// Perceive(): bit t of the stream is an OUTPUT of the arithmetic decoder that needs p(t). The stage kernels are the same persistent
// per-chunk kernels a compressor runs, but every place where they read a coded bit, a host-stage record or another
// stage's row now WAITS for it:
//
//   host (decoder thread)                               device (every stage kernel of the chunk, all co-resident)
//   ---------------------------------------------       -----------------------------------------------------------
//   poll p_word until its tag is bit t        <-------   mixing network: row t complete (all producers' counters) -> p(t)
//   bit = Decoder::Decode(p)
//   front ends for the step after the bit
//   (paq8 per step; PPMd, fxcm parser, LSTM
//   launch per byte) into host-mapped records
//   bit[t] = bit; nknown = t + 1              ------->   every kernel: learns bit t, produces row t + 1, bumps its row counter
//
// ONE counter carries both facts: nknown > t means "bit t is known AND every host record of step t + 1 is in place" (the host
// writes the records first). The box and the host stages' records live in host-coherent pinned memory; every buffer that one kernel
// writes and another reads WHILE BOTH RUN (rows, selectors, distributions, hints) and the producers' row counters live in UNCACHED
// DEVICE memory (no stale line in an XCD's L2), written with plain stores followed by `s_waitcnt vmcnt(0)` and the producer's counter,
// read after the consumer has seen the counter.
// Every wait is bounded by wall-clock time (30 s without progress by default; CMX_LATE_TIMEOUT_S=seconds when the stream starts puts another bound into every
// chunk's box -- a caller that stalls between predict() and perceive(), e.g. under a debugger, needs a longer one) and by the box's abort word: a decoder that
// stops mid-chunk (cmx_destroy) unwinds the kernels instead of leaving them spinning.
#ifndef CMX_LATE_H
#define CMX_LATE_H
#include <stdint.h>

enum {
  LC_CTX = 0,     // rows whose columns 1, 2, 2025..2075 and 47 selectors the context stage has written
  LC_BM0,         // rows with column 0 (Bracket's ByteModel::Predict)
  LC_BM1,         // rows with column 2076 (PPMd's)
  LC_BM2,         // rows with column 2077 (the LSTM byte mixer's) and the fxcm hints of the update before them
  LC_FX,          // rows with columns 3..433 (fxcm stage, role X)
  LC_P8,          // rows with columns 434..2024 (paq8 stage, mixer workgroup 0)
  LC_CM2_0,       // paq8: mixer-input rows (and `order`) of the order-N ContextMap2
  LC_CM2_1, LC_CM2_2,
  LC_FAM, LC_LANES, LC_DMC,
  LC_BRK,         // bytes whose Bracket distribution (after the byte) the context stage has written
  LC_LSTM,        // bytes whose LSTM distribution (after the byte) is in place (bumped behind the per-byte LSTM launch)
  LC_KNOWN,       // steps whose inputs are in DEVICE memory: the bit before the step and the host stages' records of the step (the relay wave)
  LC_N = 16
};
#define CMX_LATE_P_RING 8

struct CmxLateBox {   // HOST-coherent pinned memory: what the decoder thread and the kernels exchange
  uint32_t nknown;   // host -> device
  uint32_t start;    // host -> device: 1 = the chunk before this one is complete (last_y, the step-0 records and row-0 carry-overs are valid)
  uint32_t last_y;   // the bit before the chunk's first
  uint32_t abort;    // host -> device: leave
  uint32_t fail;     // device -> host, sticky: a wait ran out of time
  uint32_t nbits;    // bits of this chunk (information only)
  uint32_t timeout_s;   // host -> device: seconds a wait may last without progress (0: the default, CMX_LATE_TIMEOUT_TICKS); read on the slow path of a wait only
  uint32_t pad0[1];
  unsigned long long kb;   // host -> device, ONE 8-byte store: (bit t << 32) | (t + 1) -- what nknown and bit[t] say, in one PCIe read for the relay
  uint32_t pad1[6];
  unsigned long long p_word[CMX_LATE_P_RING];           // device -> host: ((t + 1) << 32) | bits of p(t), slot t % ring -- ONE self-validating word
  uint8_t bit[8];    // [nbits] follows (allocated behind the struct)
};
// What a late kernel is launched with (BY VALUE: the fields come out of the kernel-argument segment, not over PCIe). The row counters
// that kernels hand to each other live in UNCACHED DEVICE memory, like every buffer one kernel writes and another reads while both
// run: a producer's `data stores; s_waitcnt vmcnt(0); counter store` is only an ordering once the stores have reached the device's
// memory -- through host memory the counter was seen before the data (measured: MI355X, round 4). A counter holds base | rows, base =
// chunk number << 16: the three sets of buffers are reused every third chunk and never cleared, comparisons are wrap-safe differences.
#define CMX_LATE_CNT_STRIDE 16   /* u32 per counter: one 64-byte line each */
struct CmxLate {
  CmxLateBox* box;
  uint32_t* cnt;     // [LC_N][CMX_LATE_CNT_STRIDE]
  uint32_t base;     // (chunk number & 0xFFFF) << 16
  uint32_t pad;      // 1 = HOST PUSH (round 6): the decoder thread stores every step's bit and records into the device mirrors itself and counts the step (LC_KNOWN) --
                     //   posted writes through the PCIe BAR (hipDeviceAttributeIsLargeBar) instead of the relay wave's polls and copies, two non-posted read round
                     //   trips per bit (scripts/ubench/host_push.hip: 2.95 us against 7.89 us per host -> kernel -> host round trip with 1 KB of records); the relay stays idle
  const uint8_t* dbit0;   // DEVICE mirror of the bits: dbit0[t] = bit t of the chunk, dbit0[-1] = the bit before it (the relay wave fills it)
};
// ONE wavefront per stream talks to the host (the relay, wave 3 of cmx_bytemodel_late_kernel): it polls the box, copies every newly
// published bit and the host records that came with it into uncached device memory, and counts the step (LC_KNOWN). Every stage kernel
// waits on that device counter and reads device memory only: forty wavefronts polling host memory across PCIe -- and then fetching
// their records across it line by line -- cost 140 us per bit (measured), most of it queueing.
// kind 0: one row per step (row s, s < T); 1: one row per byte, written with the byte's first step (row s / 8 at s % 8 == 0, s < T);
// 2: one row per COMPLETED byte (row s / 8 - 1 at s % 8 == 0, 8 <= s <= T). src: host-coherent memory; dst: its device mirror (same layout).
typedef struct { const void* src; void* dst; uint32_t stride; int32_t kind; } cmx_late_relay_t;
#define CMX_LATE_RELAY_MAX 24

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#ifndef CMX_LATE_TIMEOUT_TICKS
#define CMX_LATE_TIMEOUT_TICKS (30ull * 100000000ull)   // 30 s of the 100 MHz constant clock without the awaited value
#endif
__device__ __forceinline__ uint32_t late_ld(const uint32_t* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
__device__ __forceinline__ void late_st(uint32_t* p, uint32_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }
// the bound of a wait, in ticks of the 100 MHz clock: the box's (CMX_LATE_TIMEOUT_S at cmx_pipeline_late_start) or the default
__device__ __forceinline__ unsigned long long late_timeout_ticks(CmxLateBox* B) {
  const uint32_t s = late_ld(&B->timeout_s);
  return s ? (unsigned long long)s * 100000000ull : CMX_LATE_TIMEOUT_TICKS;
}
// Bounded wait until *p has reached `want` (wrap-safe: a counter is base | rows). Callable by one lane or by a whole wavefront on a
// uniform address. false: aborted / timed out.
__device__ __forceinline__ bool late_wait_ge(CmxLateBox* B, const uint32_t* p, uint32_t want) {
  if ((int32_t)(late_ld(p) - want) >= 0) return true;
  unsigned spins = 0;
  unsigned long long t0 = 0;
  for (;;) {
    __builtin_amdgcn_s_sleep(1);
    if ((int32_t)(late_ld(p) - want) >= 0) return true;
    if ((++spins & 255u) == 0) {
      if (late_ld(&B->abort) || late_ld(&B->fail)) return false;
      const unsigned long long now = wall_clock64();
      if (!t0) t0 = now;
      else if (now - t0 > late_timeout_ticks(B)) { late_st(&B->fail, 1u); return false; }
    }
  }
}
// the bit before step t of the chunk (t == 0: the previous chunk's last), once the relay has brought it (and the step's records) over; -1: aborted
__device__ __forceinline__ int late_y(const CmxLate& L, int t) {
  if (!late_wait_ge(L.box, L.cnt + LC_KNOWN * CMX_LATE_CNT_STRIDE, L.base | (uint32_t)(t + 1))) return -1;
  asm volatile("" ::: "memory");
  return (int)*(volatile const uint8_t*)(L.dbit0 + (t - 1));
}
// have the inputs of step t arrived? (for waits that are not followed by a bit read)
__device__ __forceinline__ bool late_wait_step(const CmxLate& L, int t) {
  const bool ok = late_wait_ge(L.box, L.cnt + LC_KNOWN * CMX_LATE_CNT_STRIDE, L.base | (uint32_t)(t + 1));
  asm volatile("" ::: "memory");
  return ok;
}
// publish: every store of this wavefront issued so far has reached the device's memory before the counter moves (call with one lane;
// the others' stores are covered when the caller puts a wave / workgroup barrier with s_waitcnt vmcnt(0) in front)
__device__ __forceinline__ void late_publish(const CmxLate& L, int which, uint32_t v) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  late_st(L.cnt + which * CMX_LATE_CNT_STRIDE, L.base | v);
  L.cnt[which * CMX_LATE_CNT_STRIDE + 1] = (uint32_t)wall_clock64();   // diagnostics: when (100 MHz clock), cmx_pipeline_late_debug_times
}
// diagnostics: a time stamp of the mixing network's progress on the current bit (slots 1.. of the spare counter line 15)
__device__ __forceinline__ void late_stamp(const CmxLate& L, int slot) { L.cnt[15 * CMX_LATE_CNT_STRIDE + slot] = (uint32_t)wall_clock64(); }
__device__ __forceinline__ bool late_wait_cnt(const CmxLate& L, int which, uint32_t want) {
  const bool ok = late_wait_ge(L.box, L.cnt + which * CMX_LATE_CNT_STRIDE, L.base | want);
  asm volatile("" ::: "memory");
  return ok;
}
// The relay (one wavefront): for s = 0 .. nbits -- wait until the host has published step s (s == 0: `start`; else nknown >= s), copy
// bit s - 1 and the records of step s from host memory to their device mirrors, count the step. Loads from host memory are atomic loads
// (performed where they stand, not cached); the stores end in uncached device memory.
// rw / nrw: this wavefront's number among the relay's wavefronts (the entries are dealt out round robin: a copy is one PCIe round trip, and
// the round trips of different wavefronts overlap); done: LDS word the relay's wavefronts count their finished steps in (rw 0 publishes)
// (only wavefront 0 polls the host; the others follow its LDS word `seen` = steps the host has published, 0xFFFFFFFF = leave)
__device__ __forceinline__ void late_relay(const CmxLate& L, uint8_t* dbit0, const cmx_late_relay_t* ent, int nent, int nbits, int lane, int rw, int nrw, unsigned* done, unsigned* seen) {
  CmxLateBox* const B = L.box;
  for (int s = 0; s <= nbits; ++s) {
    int ybit = -1;   // wavefront 0: the bit before step s when it came with the count (-1: read it from the array)
    if (rw == 0) {
      bool ok;
      if (s == 0) ok = late_wait_ge(B, &B->start, 1u);
      else {   // poll the 8-byte word: count and bit in one read
        unsigned spins = 0;
        unsigned long long t0 = 0, v;
        ok = true;
        while ((uint32_t)(v = __hip_atomic_load(&B->kb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)) < (uint32_t)s) {
          __builtin_amdgcn_s_sleep(1);
          if ((++spins & 255u) == 0) {
            if (late_ld(&B->abort) || late_ld(&B->fail)) { ok = false; break; }
            const unsigned long long now = wall_clock64();
            if (!t0) t0 = now;
            else if (now - t0 > late_timeout_ticks(B)) { late_st(&B->fail, 1u); ok = false; break; }
          }
        }
        if (ok && (uint32_t)v == (uint32_t)s) ybit = (int)(v >> 32) & 1;
      }
      if (lane == 0) __hip_atomic_store(seen, ok ? (unsigned)(s + 1) : 0xFFFFFFFFu, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
      if (!ok) return;
    } else {
      unsigned v;
      while ((v = __hip_atomic_load(seen, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)) < (unsigned)(s + 1)) __builtin_amdgcn_s_sleep(1);
      if (v == 0xFFFFFFFFu) return;
    }
    asm volatile("" ::: "memory");
    if (rw == 0 && lane == 0) dbit0[s - 1] = s == 0 ? (uint8_t)late_ld(&B->last_y) : ybit >= 0 ? (uint8_t)ybit : *(volatile const uint8_t*)(B->bit + (s - 1));
    // wavefront 0 only watches the host and publishes; the copies are the others' (nrw > 1)
    for (int e = (nrw > 1 ? rw - 1 : 0); e >= 0 && e < nent; e += (nrw > 1 ? nrw - 1 : 1)) {
      const int kind = ent[e].kind;
      long r;
      if (kind == 0) { if (s >= nbits) continue; r = s; }
      else if (kind == 1) { if ((s & 7) || s >= nbits) continue; r = s >> 3; }
      else { if ((s & 7) || s < 8) continue; r = (s >> 3) - 1; }
      const size_t off = (size_t)r * ent[e].stride;
      const unsigned n = ent[e].stride;
      if (((off | n) & 3) == 0) {
        const uint32_t* sp = reinterpret_cast<const uint32_t*>(static_cast<const char*>(ent[e].src) + off);
        uint32_t* dp = reinterpret_cast<uint32_t*>(static_cast<char*>(ent[e].dst) + off);
        for (unsigned i = lane; i < n / 4; i += 64) dp[i] = __hip_atomic_load(sp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      } else {
        const uint16_t* sp = reinterpret_cast<const uint16_t*>(static_cast<const char*>(ent[e].src) + off);
        uint16_t* dp = reinterpret_cast<uint16_t*>(static_cast<char*>(ent[e].dst) + off);
        for (unsigned i = lane; i < n / 2; i += 64) dp[i] = __hip_atomic_load(sp + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    if (rw != 0) { if (lane == 0) __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); continue; }
    if (lane == 0) {
      // the other wavefronts' copies of this step have landed: `done` counts (step, wavefront) pairs
      unsigned spins = 0;
      while (__hip_atomic_load(done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < (unsigned)(s + 1) * (unsigned)(nrw - 1))
        if ((++spins & 1023u) == 0 && (late_ld(&B->abort) || late_ld(&B->fail))) break;
      late_st(L.cnt + LC_KNOWN * CMX_LATE_CNT_STRIDE, L.base | (uint32_t)(s + 1));
      L.cnt[LC_KNOWN * CMX_LATE_CNT_STRIDE + 1] = (uint32_t)wall_clock64();
    }
    __builtin_amdgcn_wave_barrier();
  }
}
#endif

#endif
