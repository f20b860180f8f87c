// mixnet_state.h -- HBM-resident state of one stream's final mixing network.
// Shared between the kernels (mixnet_kernels.hip) and the host C-ABI layer.
#ifndef CMX_MIXNET_STATE_H
#define CMX_MIXNET_STATE_H
#include <stdint.h>

#define CMX_IN0 2078
#define CMX_IN1 29
#define CMX_IN2 49
#define CMX_MIX0 26
#define CMX_MIX1 20
#define CMX_MIXERS 47
#define CMX_AUX 12

// Mixer::GetContextData caps a mixer at 10000 rows + 1 shared overflow row
// (reference src/mixer/mixer.cpp:16-36).
#define CMX_ROW_LIMIT 10000u
#define CMX_ROWS_PER_MIXER 10001u
#define CMX_MAP_SLOTS 32768u  // open-addressing table per mixer (load <= 0.31)

// Row layouts (floats). Layer 0: 2078 weights, pad to 2080, then 26 extra
// weights (only the first k used by mixer k), padded to 33 * 64.
#define CMX_ROW0_STRIDE 2112
#define CMX_ROW0_EXTRA 2080
// Layer 1: 29 weights, pad to 32, then 20 extra weights -> 64 floats (256 B).
#define CMX_ROW1_STRIDE 64
#define CMX_ROW1_EXTRA 32
#define CMX_ROW2_STRIDE 64

// SSE table volumes (reference src/mixer/sse.cpp:197-200); entries padded from
// 7 to 8 u16 so that an entry never straddles a 16-byte boundary.
#define CMX_SM6_VOL (3 * 128 * 256 * 256)
#define CMX_MIX1_VOL (4 * 256 * 8 * 79)
#define CMX_SM7_VOL (3 * 32 * 256 * 255)
#define CMX_MIX2_VOL (3 * 2 * 256 * 256)

struct MixState {
  // read-only tables, built on the host with the host libm and uploaded
  const float* logit_lut;   // [100001]  Sigmoid::Sigmoid, sigmoid.cpp:5-10
  const uint16_t* t_st;     // [32768]   Init_ST_SQ, sse.cpp:112-135
  const uint16_t* t_sq;     // [32768]
  float stretch_min, stretch_max;  // Logit(0), Logit(1): mixer-input.cpp:3-5
  float lr[CMX_MIXERS];     // learning rates in construction order, predictor.cpp:199-356

  // mixer weight rows
  float* rows0;             // [26][10001][2112]
  float* rows1;             // [20][10001][64]
  float* rows2;             // [1][64]
  uint64_t* row_steps;      // [47][10001]   ContextData::steps
  uint32_t* map_keys;       // [47][32768]
  uint32_t* map_vals;       // [47][32768]   row index + 1, 0 = empty
  uint32_t n_rows[CMX_MIXERS + 1];   // context_map_.size()
  uint64_t max_steps[CMX_MIXERS + 1];  // Mixer::max_steps_ (starts at 1)
  uint64_t steps;           // Mixer::steps_ (same value in all 47 mixers)

  // SSE
  uint16_t* s6;             // [SM6_VOL][8]
  uint16_t* s7;             // [SM7_VOL][8]
  int* x1;                  // [MIX1_VOL]
  int* x2;                  // [MIX2_VOL]
  uint32_t sse_j, sse_pc, sse_ffl;

  // forward results kept for a separate Perceive launch (bit-synchronous mode)
  float fwd_p[CMX_MIXERS + 1];
  float fwd_out0[32];
  float fwd_in2[64];

  int error;                // set by a kernel whose bounded spin timed out
  int pad_;
  uint64_t prof[16];        // phase timers (shader clocks), see PROF() in mixnet_kernels.hip
};

// ---- cmx_mixnet_spec_kernel (mixnet_chunk.hip): the layer-0 dot products on 26 helper workgroups -----------------------------
// Global hand-off area between the main workgroup (scout / gather / tail) and the helpers, one per handle. Everything in it is
// written and read with agent-scope atomics (the workgroups sit on different compute units, possibly different XCDs / L2s).
#define CMX_SPEC_RING 8        /* bits the scout may publish ahead (it is held to 2 ahead of the gather wave) */
#define CMX_SPEC_XS 2112       /* stretched inputs of a bit, zero padded (2078 used) */
#define CMX_SPEC_HELPERS CMX_MIX0
#define CMX_SPEC_THREADS 512      /* main: gather, tail a, select, tail b, 4 stretch waves; helpers use the first four */
#define CMX_MIXNET_XCD_DEFAULT (-1)   /* the XCD the mixing network's 27 workgroups are placed on (-1: as dispatched); CMX_MIXNET_XCD overrides */
struct SpecXfer {
  unsigned scout_epoch;        // bits whose inputs / rows the scout has published
  unsigned fail;               // sticky: a bounded in-launch wait ran out
  unsigned sel_epoch;          // a decoder's chunk (round 6): bits whose 46 context-keyed row selections are published -- ahead of scout_epoch, which also covers the
                               //   auxiliary-context mixer's (its key comes out of three of the bit's inputs)
  unsigned pad0[13];
  unsigned xcc[32];            // XCD placement (CMX_MIXNET_XCD): 1 + HW_REG_XCC_ID of the workgroup with role r, written at launch; all equal => the u / sum words
                               //   are exchanged through that XCD's L2 (plain stores, L1-bypassing loads), otherwise through the fabric (agent-scope stores)
  unsigned long long u[32];    // by the gather wave after bit t: ((2 (t + 1) + decay flag) << 32) | bits of u = decay * lr * err   (mixer.cpp:56-64)
  unsigned long long sum[32];  // by helper m after bit t: ((t + 1) << 32) | bits of the 2078-term ordered sum                     (mixer.cpp:40-43)
  unsigned rowidx[CMX_SPEC_RING][32];
  unsigned changed[CMX_SPEC_RING][32];
  float xs[CMX_SPEC_RING][CMX_SPEC_XS];
  unsigned long long stat[8];  // [0] speculative segments run, [1] of them resolved from a candidate lane, [2..4] misses of segment 1..3
};
#define CMX_SPEC_HEADER_BYTES (64 + 128 + 2 * 32 * 8)   /* what the host clears ahead of every launch (epochs and tags restart at 0) */
#define CMX_SPEC_LDS_BYTES 147456  /* main workgroup: 128 KB of SSE tables + records; a helper uses 9 KB of it */

// dynamic LDS of cmx_mixnet_chunk_kernel (see the carve-up in mixnet_chunk.hip)
#define CMX_CHUNK_LDS_BYTES 163840   /* the whole 160 KB LDS of a gfx950 CU: one workgroup per CU */
#define CMX_CHUNK_THREADS 768

#endif
