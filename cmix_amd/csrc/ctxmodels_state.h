// ctxmodels_state.h -- HBM-resident state of one stream's context plumbing and 54 small native
// models (reference src/context-manager.*, src/contexts/*, src/states/*,
// src/models/{direct,direct-hash,indirect,match,bracket,byte-model}.*).
//
// Lane l of the stage's single wavefront owns byte context l (0..53) AND small model l (0..53, in
// layer-0 column order 0,1,2,2025..2075) AND mixer selector l (0..46); CtxLane is that lane's
// read-only configuration, built on the host in the reference's construction order
// (predictor.cpp:90-178,199-356).
#ifndef CMX_CTXMODELS_STATE_H
#define CMX_CTXMODELS_STATE_H
#include <stdint.h>

#define CTX_N 54                       // byte contexts
#define CTX_NM 54                      // small models
#define CTX_NSEL 47                    // mixer selectors
#define CTX_HISTORY 100000000ull       // context-manager.cpp:3
#define CTX_SHARED 2048000000ull       // 256*8000000, context-manager.cpp:4
#define CTX_N_INDIRECT 31
#define CTX_N_MATCH 16
#define CTX_PRED_STRIDE 257            // LDS row stride of the 256-entry per-model tables (bank spread)

enum { CT_BRACKET = 0, CT_SPARSE, CT_HASH, CT_INDIRECT, CT_INTERVAL, CT_INTERVALHASH, CT_COMBINED, CT_NONE };
enum { MT_NONE = 0, MT_BRACKET, MT_DIRECT, MT_DIRECTHASH, MT_INDIRECT, MT_MATCH };
enum { SZ_POW2 = 0, SZ_10M, SZ_20M, SZ_500K, SZ_100K };  // table sizes that need a real modulo
enum { SEL_PLAIN = 0, SEL_BITCTX, SEL_LBC, SEL_ZERO };

// indices into the LDS register file regs[] (u64): manager registers, then the 54 contexts
#define R_ZERO 0
#define R_LINE_BREAK 1
#define R_LONGEST_MATCH 2
#define R_WRT_CONTEXT 3
#define R_RECENT 4   // ..11
#define R_WORDS 12   // ..19
#define R_CTX 20     // ..73
#define R_COUNT 76

struct CtxLane {
  // ---- context l ----
  int ctype;
  unsigned hash_size, hash_size1;      // bits shifted in per byte
  unsigned long long mask;             // size_-1 (every size_ is a power of two)
  unsigned mask1;                      // IndirectHash: size1_-1;  IntervalHash: interval mask
  unsigned shift;                      // Interval / IntervalHash
  int map_id;                          // byte-class map 0..4
  unsigned orders;                     // Sparse: count | order_i << (4+4i);  Combined: r1 | r2<<4
  uint32_t* ihash;                     // IndirectHash::hashes_ (values < 2^32 for every instance)
  // ---- model l ----
  int mtype, mctx, col, slot;          // slot: row in the LDS ipred / mpred tables
  float divisor;                       // Indirect: 1/delta; Direct/Match: 1/(limit+delta)
  int limit, divtab;                   // divtab: offset of this model's 1/(count+delta) table
  int size_kind;                       // SZ_*
  unsigned long long size;             // DirectHash rows / Match map entries
  unsigned long long offset;           // Indirect::map_offset_
  float* pred; uint8_t* cnt; unsigned long long* chk;  // Direct / DirectHash tables
  uint32_t* map;                       // Match::map_
  int run_map;                         // Indirect state machine: 0 nonstationary, 1 run map
  // ---- selector l ----
  int sel_kind, sel_src;
};

// Everything small and mutable: lives in LDS/registers while a chunk kernel runs, in HBM between
// launches.
struct CtxPersist {
  unsigned long long regs[R_COUNT];
  unsigned long long ctx1[64];         // IndirectHash::context1_ / IntervalHash::interval_
  unsigned long long history_pos, bytes_done;
  unsigned wrt_state;
  // BracketContext stack (bracket-context.h:21); entries below the top live in bstack[]
  unsigned bc_n, bc_top_active, bc_top_dist;
  // Bracket model (bracket.h:17-22): stack <= 10, stats for the 6 bracket characters
  unsigned br_n, br_active[16], br_dist[16];
  unsigned br_stats[6][200][2];
  float br_probs[256];
  // per-model learned tables
  float ipred[CTX_N_INDIRECT][256];
  float mpred[CTX_N_MATCH][256];
  int mcnt[CTX_N_MATCH][256];
  // per-lane dynamic model state
  unsigned long long map_index[64];    // Indirect::map_index_ (byte base) / DirectHash::index_ / Match map slot
  unsigned long long m_history_pos[64], cur_match[64];
  unsigned cur_byte[64], match_length[64];
  // statistics (test introspection): bytes that took the serial path [0] committed, [1] in dry passes
  unsigned long long slow_bytes[2];
};

struct CtxDev {                        // passed to the kernel by value
  const CtxLane* lanes;                // [64]
  CtxPersist* persist;
  uint8_t* history;                    // [CTX_HISTORY]
  uint8_t* shared_map;                 // [CTX_SHARED]
  uint16_t* bstack; unsigned bstack_cap;
  const float* divtabs;                // [31 + 201] 1/(count+delta): Direct (limit 30, delta 0), Match (200, .5)
  int* err;                            // device error word (0 = ok)
  unsigned char vocab[256];
};

#endif
