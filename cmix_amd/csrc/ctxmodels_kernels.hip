// ctxmodels_kernels.hip -- gfx950 kernel of the context / small-model stage.
//
// Reference: ContextManager::UpdateContexts (src/context-manager.cpp:69-94) and the Context::Update
// family (src/contexts/*.cpp); Direct / DirectHash / Indirect / Match / Bracket Predict, Perceive and
// ByteUpdate (src/models/*.cpp); their call order in Predictor::Predict/Perceive
// (src/predictor.cpp:361-369,421-446,468). Produces, per coded bit, layer-0 columns 1,2,2025..2075,
// the 47 mixer selectors (predictor.cpp:199-356) and, per byte, the Bracket model's byte
// distribution (column 0 is then formed by cmx_bytemodel_bits, the ByteModel::Predict kernel).
//
// Shape of the work: HBM-latency-bound integer/byte probing, not arithmetic. One wavefront per
// stream; lane l owns byte context l, small model l and selector l. In compression every bit of the
// chunk is known, so the wave walks BYTES, not bits: at the top of a byte all 8 bit-contexts of every
// model are known, the 8 probes of each of the 53 table models (shared_map_ state bytes, Direct
// probability/count cells) are issued together and their latency is paid once per byte; the 8
// sequential Predict/Perceive steps then run out of registers and LDS (the 256-entry adaptive tables
// of the 31 Indirect and 16 Match models live in LDS for the whole chunk).
//
// Exactness hazards handled here:
//  * all Indirect models share one 2 GB state map; two models alias when their byte bases are closer
//    than 256 (always in the very first byte: every base is 0). Such a byte takes the serial path,
//    which replays the reference order bit by bit, model by model (indirect.cpp:16-27).
//  * a Match model may read back the history byte / map slot written in the same byte update.
//  Global-memory accesses of ONE wavefront are performed in execution order, so plain program order
//  is sufficient; wave_mem_sync() additionally drains outstanding accesses at those points.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "cmx_ref_tables.h"
#include "ctxmodels_state.h"
#include "cmx_late.h"

namespace {

typedef unsigned long long u64;

__device__ __forceinline__ void wave_mem_sync() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); }

__device__ __forceinline__ int close4(unsigned c) {  // bracket-context.cpp:6
  return c == '(' ? ')' : c == '{' ? '}' : c == '[' ? ']' : c == '<' ? '>' : -1;
}
__device__ __forceinline__ int close6(unsigned c) {  // bracket.cpp:9-10
  return c == '(' ? ')' : c == '{' ? '}' : c == '[' ? ']' : c == '<' ? '>' : c == '\'' ? '\'' : c == '"' ? '"' : -1;
}
__device__ __forceinline__ int id6(unsigned c) {
  return c == '(' ? 0 : c == '{' ? 1 : c == '[' ? 2 : c == '<' ? 3 : c == '\'' ? 4 : 5;
}

__device__ __forceinline__ u64 modsz(u64 x, int kind, u64 size) {
  switch (kind) {
    case SZ_10M: return x % 10000000ull;
    case SZ_20M: return x % 20000000ull;
    case SZ_500K: return x % 500000ull;
    case SZ_100K: return x % 100000ull;
    default: return x & (size - 1);
  }
}

struct LdsMap {  // carve of the dynamic LDS region (byte offsets, all multiples of 16)
  static constexpr int regs = 0;                                             // u64[R_COUNT]
  static constexpr int ipred = regs + R_COUNT * 8;                           // f32[31][257]
  static constexpr int mpred = ipred + ((CTX_N_INDIRECT * CTX_PRED_STRIDE * 4 + 15) & ~15);
  static constexpr int mcnt = mpred + ((CTX_N_MATCH * CTX_PRED_STRIDE * 4 + 15) & ~15);  // u8[16][256]
  static constexpr int divtab = mcnt + CTX_N_MATCH * 256;                    // f32[232]
  static constexpr int nonstat = divtab + 240 * 4;                           // u8[512]
  static constexpr int runmap = nonstat + 512;                               // u8[512]
  static constexpr int cmap = runmap + 512;                                  // u8[5][256]
  static constexpr int stats = cmap + 5 * 256;                               // u32[6][200][2]
  static constexpr int brstk = stats + 6 * 200 * 2 * 4;                      // u32[16] active, u32[16] dist
  static constexpr int brout = brstk + 32 * 4;                               // mode, p, q, sym
  static constexpr int total = brout + 16;
};

}  // namespace

extern "C" unsigned cmx_ctxmodels_lds_bytes() { return LdsMap::total; }

namespace {
template <bool dry, bool late = false>
__device__ __forceinline__ void ctxmodels_body(const CtxDev& D, const uint8_t* __restrict__ bytes, size_t nbytes,
                                               float* probs, size_t pstride, uint32_t* sel, float* bracket_dist, CmxLate LB = CmxLate()) {
  // late (the decoder's form, cmx_late.h): the byte is not known when its first bit is predicted. The wave walks BITS: Predict of
  // bit j from the bits decoded so far (one probe per table model), row t and the selectors published (counter LC_CTX), then it
  // waits for the decoder's bit and runs the Perceive of every model -- the reference's own order (predictor.cpp:361-369, 421-446).
  // The byte-boundary part below is the chunk kernel's, on the byte the 8 bits spell.
  // dry != 0 (bit-synchronous mode, one byte): the 8 Predict/Perceive steps run on a byte whose low bits are
  // still unknown (zeros); outputs and selectors of bit j depend only on bits < j, so row j is exact once j bits
  // are known. Nothing of the pass survives: HBM writes are suppressed (or rolled back, overlapping Indirect
  // maps), the LDS-resident tables are never stored, and the kernel returns before the byte boundary.
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  u64* regs = (u64*)(smem + LdsMap::regs);
  float* ipred = (float*)(smem + LdsMap::ipred);
  float* mpred = (float*)(smem + LdsMap::mpred);
  uint8_t* mcnt = smem + LdsMap::mcnt;
  float* divtab = (float*)(smem + LdsMap::divtab);
  uint8_t* nonstat = smem + LdsMap::nonstat;
  uint8_t* runmap = smem + LdsMap::runmap;
  uint8_t* cmap = smem + LdsMap::cmap;
  unsigned* stats = (unsigned*)(smem + LdsMap::stats);
  unsigned* br_active = (unsigned*)(smem + LdsMap::brstk);
  unsigned* br_dist = br_active + 16;
  float* brout = (float*)(smem + LdsMap::brout);

  const int lane = threadIdx.x;
  CtxPersist* P = D.persist;
  const CtxLane L = D.lanes[lane];

  // ---- chunk prologue: small state HBM -> LDS / registers ----
  for (int i = lane; i < R_COUNT; i += 64) regs[i] = P->regs[i];
  for (int i = lane; i < CTX_N_INDIRECT * 256; i += 64) ipred[(i >> 8) * CTX_PRED_STRIDE + (i & 255)] = (&P->ipred[0][0])[i];
  for (int i = lane; i < CTX_N_MATCH * 256; i += 64) {
    mpred[(i >> 8) * CTX_PRED_STRIDE + (i & 255)] = (&P->mpred[0][0])[i];
    mcnt[i] = (uint8_t)(&P->mcnt[0][0])[i];
  }
  for (int i = lane; i < 232; i += 64) divtab[i] = D.divtabs[i];
  for (int i = lane; i < 512; i += 64) {
    nonstat[i] = CMX_NONSTATIONARY[i];
    int state = i / 2;  // run-map.cpp:3-15
    if (i % 2 == 0) { if (state < 127) ++state; else if (state >= 128) state = 0; }
    else { if (state < 128) state = 128; else if (state < 255) ++state; }
    runmap[i] = (uint8_t)state;
  }
  for (int i = lane; i < 256; i += 64) {
    cmap[0 * 256 + i] = (i < 1) + (i < 32) + (i < 64) + (i < 128) + (i < 255) + (i < 142) + (i < 138) + (i < 140) +
                        (i < 137) + (i < 97);                                            // predictor.cpp:222-225
    cmap[1 * 256 + i] = (i < 41) + (i < 92) + (i < 124) + (i < 58) + (i < 11) + (i < 46) + (i < 36) + (i < 47) +
                        (i < 64) + (i < 4) + (i < 61) + (i < 97) + (i < 125) + (i < 45) + (i < 48);  // :230-235
    cmap[2 * 256 + i] = ((i >= 'a' && i <= 'z') || (i >= 'A' && i <= 'Z') || (i >= '0' && i <= '9') || i >= 0x80);
    cmap[3 * 256 + i] = CMX_INTERVAL_WRT2B[i];
    cmap[4 * 256 + i] = CMX_INTERVAL_WRT3B[i];
  }
  for (int i = lane; i < 6 * 200 * 2; i += 64) stats[i] = (&P->br_stats[0][0][0])[i];
  if (lane < 16) { br_active[lane] = P->br_active[lane]; br_dist[lane] = P->br_dist[lane]; }
  u64 ctxv = lane < CTX_N ? P->regs[R_CTX + lane] : 0;
  u64 ctx1 = P->ctx1[lane];
  u64 mbase = P->map_index[lane];
  u64 mhp = P->m_history_pos[lane];
  u64 cur_match = P->cur_match[lane];
  unsigned cur_byte = P->cur_byte[lane], ml = P->match_length[lane];
  u64 history_pos = P->history_pos;
  const u64 bytes_done0 = P->bytes_done;
  unsigned wrt_state = P->wrt_state;
  unsigned bc_n = P->bc_n, bc_top_active = P->bc_top_active, bc_top_dist = P->bc_top_dist;
  unsigned br_n = P->br_n;
  __syncthreads();

  const bool is_ind = L.mtype == MT_INDIRECT;
  const bool is_dir = L.mtype == MT_DIRECT || L.mtype == MT_DIRECTHASH;
  const bool is_match = L.mtype == MT_MATCH;
  float* const ip = ipred + L.slot * CTX_PRED_STRIDE;
  float* const mp = mpred + L.slot * CTX_PRED_STRIDE;
  uint8_t* const mc = mcnt + L.slot * 256;
  const uint8_t* const trans = L.run_map ? runmap : nonstat;

  for (size_t n = 0; n < nbytes; ++n) {
    unsigned B = late ? 0u : (unsigned)bytes[n];
    const size_t t0 = 8 * n;

    // ---- the 47 selectors Mixer::Mix reads at each of the 8 Predict() calls ----
    if (!late && sel && lane < CTX_NSEL) {
      const u64 base = L.sel_kind == SEL_ZERO ? 0 : regs[L.sel_src];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned lbc = j == 0 ? 1u : ((1u << j) | (B >> (8 - j)));  // long_bit_context_
        u64 v = L.sel_kind == SEL_BITCTX ? (base << 8) + lbc : L.sel_kind == SEL_LBC ? (u64)lbc : base;
        if (L.sel_kind == SEL_BITCTX && j == 0 && n == 0 && bytes_done0 == 0) v = 0;  // BitContext ctor: context_ = 0
        sel[(t0 + j) * CTX_NSEL + lane] = (uint32_t)v;
      }
    }

    // ---- do two Indirect models overlap in the shared map during this byte? ----
    bool conf = false;
    {
      u64 m = __ballot(is_ind);
      while (m) {
        const int k = __ffsll((long long)m) - 1;
        m &= m - 1;
        const u64 bk = __shfl(mbase, k);
        const long long d = (long long)(mbase - bk);
        if (is_ind && lane != k && d > -256 && d < 256) conf = true;
      }
    }
    const bool slow = __any(conf);
    if (slow && lane == 0) ++P->slow_bytes[dry ? 1 : 0];

    if (late) {
      // ---- 8 x (Predict -> row out -> wait for the bit -> Perceive), bit by bit ----
      float* const pr = is_dir ? L.pred + (L.mtype == MT_DIRECT ? regs[R_CTX + L.mctx] : mbase) * 256 : nullptr;
      uint8_t* const cn = is_dir ? L.cnt + (L.mtype == MT_DIRECT ? regs[R_CTX + L.mctx] : mbase) * 256 : nullptr;
      uint8_t* const mp8 = D.shared_map + mbase;
      const u64 selbase = (lane < CTX_NSEL && L.sel_kind != SEL_ZERO) ? regs[L.sel_src] : 0;
      unsigned part = 0, m = ml;
#pragma unroll 1
      for (int j = 0; j < 8; ++j) {
        const size_t t = t0 + j;
        const unsigned bc = (1u << j) | part;
        if (sel && lane < CTX_NSEL) {
          const unsigned lbc = j == 0 ? 1u : bc;  // long_bit_context_
          u64 v = L.sel_kind == SEL_BITCTX ? (selbase << 8) + lbc : L.sel_kind == SEL_LBC ? (u64)lbc : selbase;
          if (L.sel_kind == SEL_BITCTX && j == 0 && n == 0 && bytes_done0 == 0) v = 0;  // BitContext ctor: context_ = 0
          sel[t * CTX_NSEL + lane] = (uint32_t)v;
        }
        float o = 0.5f, pv = 0.5f;
        unsigned cv = 0, s = 0;
        if (is_dir) { pv = pr[bc]; cv = cn[bc]; o = pv; }
        else if (is_ind) {
          if (slow) wave_mem_sync();
          s = mp8[bc];
          o = ip[s];
        } else if (is_match) {
          const int expected = (cur_byte >> (7 - j)) & 1;
          const float p = mp[m];
          o = expected ? p : 1.0f - p;
        }
        if (probs && lane >= 1 && lane < CTX_NM) probs[t * pstride + L.col] = o;
        wave_mem_sync();   // the row's stores (and, on the serial path, every Predict() probe) are complete
        if (lane == 0) late_publish(LB, LC_CTX, (uint32_t)(t + 1));
        const int bit = late_y(LB, (int)t + 1);   // uniform: the whole wave leaves on abort
        if (bit < 0) return;
        if (is_dir) {  // direct.cpp:22-28
          float div = L.divisor;
          if ((int)cv < L.limit) {
            const unsigned c = cv + 1;
            cn[bc] = (uint8_t)c;
            div = divtab[L.divtab + c];
          }
          pr[bc] = pv + ((float)bit - pv) * div;
        } else if (is_ind && !slow) {  // indirect.cpp:22-27
          const float p = ip[s];
          ip[s] = p + ((float)bit - p) * L.divisor;
          mp8[bc] = trans[s * 2 + bit];
        } else if (is_match) {  // match.cpp:25-46
          const int expected = (cur_byte >> (7 - j)) & 1;
          const float p = mp[m];
          const int match = bit == expected;
          float div = L.divisor;
          const unsigned c = mc[m];
          if ((int)c < L.limit) {
            mc[m] = (uint8_t)(c + 1);
            div = divtab[L.divtab + c + 1];
          }
          mp[m] = p + ((float)match - p) * div;
          m = match ? (m < 255 ? m + 1 : m) : 0;
        }
        if (slow) {  // overlapping Indirect maps: Perceive model by model, in the reference's order
          u64 mk = __ballot(is_ind);
          while (mk) {
            const int k = __ffsll((long long)mk) - 1;
            mk &= mk - 1;
            if (lane == k) {
              const unsigned s2 = D.shared_map[mbase + bc];
              const float p = ip[s2];
              ip[s2] = p + ((float)bit - p) * L.divisor;
              D.shared_map[mbase + bc] = trans[s2 * 2 + bit];
            }
            wave_mem_sync();
          }
        }
        part = (part << 1) | (unsigned)bit;
      }
      B = part;
      if (is_match) {
        ml = m;
        L.map[mbase] = (uint32_t)mhp;  // map_[byte_context_ % map_.size()] = history_pos_
        ++mhp;
      }
    }
    // ---- 8 x (Predict, Perceive) of the table models ----
    float out[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) out[j] = 0.5f;
    if (late) {
    } else if (is_dir) {  // direct.cpp:15-28, direct-hash.cpp:16-29
      const u64 row = L.mtype == MT_DIRECT ? regs[R_CTX + L.mctx] : mbase;
      float* const pr = L.pred + row * 256;
      uint8_t* const cn = L.cnt + row * 256;
      float pv[8];
      unsigned cv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned bc = (1u << j) | (B >> (8 - j));
        pv[j] = pr[bc];
        cv[j] = cn[bc];
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned bc = (1u << j) | (B >> (8 - j));
        const int bit = (B >> (7 - j)) & 1;
        out[j] = pv[j];
        float div = L.divisor;
        if (dry) continue;
        if ((int)cv[j] < L.limit) {
          const unsigned c = cv[j] + 1;
          cn[bc] = (uint8_t)c;
          div = divtab[L.divtab + c];
        }
        pr[bc] = pv[j] + ((float)bit - pv[j]) * div;
      }
    } else if (is_ind) {  // indirect.cpp:16-27
      if (!slow) {
        uint8_t* const mp8 = D.shared_map + mbase;
        unsigned sv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) sv[j] = mp8[(1u << j) | (B >> (8 - j))];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int bit = (B >> (7 - j)) & 1;
          const unsigned s = sv[j];
          const float p = ip[s];
          out[j] = p;
          ip[s] = p + ((float)bit - p) * L.divisor;
          if (!dry) mp8[(1u << j) | (B >> (8 - j))] = trans[s * 2 + bit];
        }
      }
    } else if (is_match) {  // match.cpp:17-46
      unsigned m = ml;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int bit = (B >> (7 - j)) & 1;
        const int expected = (cur_byte >> (7 - j)) & 1;
        const float p = mp[m];
        out[j] = expected ? p : 1.0f - p;
        const int match = bit == expected;
        float div = L.divisor;
        const unsigned c = mc[m];
        if ((int)c < L.limit) {
          mc[m] = (uint8_t)(c + 1);
          div = divtab[L.divtab + c + 1];
        }
        mp[m] = p + ((float)match - p) * div;
        m = match ? (m < 255 ? m + 1 : m) : 0;
      }
      ml = m;
      if (!dry) L.map[mbase] = (uint32_t)mhp;  // map_[byte_context_ % map_.size()] = history_pos_
      ++mhp;
    }
    unsigned undo[8];  // dry pass over overlapping maps: the states this lane replaced, restored below
    if (slow && !late) {  // reference order, bit by bit and model by model
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const unsigned bc = (1u << j) | (B >> (8 - j));
        const int bit = (B >> (7 - j)) & 1;
        wave_mem_sync();
        if (is_ind) out[j] = ip[D.shared_map[mbase + bc]];  // every Predict() precedes every Perceive()
        wave_mem_sync();
        u64 m = __ballot(is_ind);
        while (m) {
          const int k = __ffsll((long long)m) - 1;
          m &= m - 1;
          if (lane == k) {
            const unsigned s = D.shared_map[mbase + bc];
            const float p = ip[s];
            ip[s] = p + ((float)bit - p) * L.divisor;
            D.shared_map[mbase + bc] = trans[s * 2 + bit];
            undo[j] = s;
          }
          wave_mem_sync();
        }
      }
      if (dry) {  // exact rollback: newest write first (bits 7..0, lanes high..low)
#pragma unroll
        for (int j = 7; j >= 0; --j) {
          const unsigned bc = (1u << j) | (B >> (8 - j));
          u64 m = __ballot(is_ind);
          while (m) {
            const int k = 63 - __clzll((long long)m);
            m &= ~(1ull << k);
            if (lane == k) D.shared_map[mbase + bc] = (uint8_t)undo[j];
            wave_mem_sync();
          }
        }
      }
    }
    if (!late && probs && lane >= 1 && lane < CTX_NM) {
#pragma unroll
      for (int j = 0; j < 8; ++j) probs[(t0 + j) * pstride + L.col] = out[j];
    }
    if (dry) return;  // uniform: the whole wave leaves before the byte boundary and the epilogue

    // ---- ContextManager::UpdateContexts at the byte boundary (context-manager.cpp:69-94) ----
    __syncthreads();
    if (lane == 0) {
      u64 lb = regs[R_LINE_BREAK];
      if (B == '\n') lb = 0; else if (lb < 99) ++lb;
      regs[R_LINE_BREAK] = lb;
      D.history[history_pos] = (uint8_t)B;  // UpdateHistory
      unsigned c = B;                       // UpdateWords
      u64 w7 = regs[R_WORDS + 7];
      if ((c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || c >= 0x80) w7 = w7 * 997 * 16 + c; else w7 = 0;
      regs[R_WORDS + 7] = w7;
      if (c >= 'A' && c <= 'Z') c += 'a' - 'A';
      if ((c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || c == 8 || c == 6 || c >= 0x80) {
        regs[R_WORDS + 0] = (regs[R_WORDS + 0] * 997 * 16 + c) & 0xfffffff;
        regs[R_WORDS + 1] = regs[R_WORDS + 1] * 263 * 32 + c;
      } else {
        for (int i = 6; i >= 2; --i) regs[R_WORDS + i] = regs[R_WORDS + i - 1];
        regs[R_WORDS + 1] = 0;
      }
      for (int i = 7; i >= 1; --i) regs[R_RECENT + i] = regs[R_RECENT + i - 1];  // UpdateRecentBytes
      regs[R_RECENT] = B;
      u64 wc = regs[R_WRT_CONTEXT];  // UpdateWRTContext
      if (B < 0x80) wrt_state = 0;
      else {
        if (wrt_state == 0) wc = 0;
        wrt_state = 1;
        wc = (wc << 8) + B;
        if (wc > 0xFFEFCF) wc = 0;
      }
      regs[R_WRT_CONTEXT] = wc;
    }
    if (++history_pos == CTX_HISTORY) history_pos = 0;
    __syncthreads();

    u64 nc = ctxv;
    switch (L.ctype) {
      case CT_BRACKET: {  // bracket-context.cpp:11-34 (distance_limit_ 256; the stack never shrinks by limit)
        if (bc_n) {
          if (close4(bc_top_active) == (int)B || bc_top_dist >= 255) {
            if (--bc_n) { const unsigned e = D.bstack[bc_n - 1]; bc_top_active = e >> 8; bc_top_dist = e & 255; }
          } else ++bc_top_dist;
        }
        if (close4(B) >= 0) {
          if (bc_n >= D.bstack_cap) *D.err = 1;
          else {
            if (bc_n) D.bstack[bc_n - 1] = (uint16_t)((bc_top_active << 8) | bc_top_dist);
            ++bc_n; bc_top_active = B; bc_top_dist = 0;
          }
        }
        nc = bc_n ? (u64)256 * (bc_top_active + 1) + bc_top_dist : 0;
        break;
      }
      case CT_SPARSE: {  // sparse.cpp:17-22
        const unsigned cnt = L.orders & 15;
        nc = regs[R_WORDS + ((L.orders >> 4) & 15)];
        u64 f = 256;  // factors_: 1, 256, 29*31, *37, *41, *43
        for (unsigned i = 1; i < cnt; ++i) {
          nc += f * regs[R_WORDS + ((L.orders >> (4 + 4 * i)) & 15)];
          f = i == 1 ? 29 * 31 : i == 2 ? 29 * 31 * 37 : i == 3 ? 29 * 31 * 37 * 41 : 29ull * 31 * 37 * 41 * 43;
        }
        break;
      }
      case CT_HASH:  // context-hash.cpp:9-11
        nc = ((ctxv << L.hash_size) + B) & L.mask;
        break;
      case CT_INDIRECT: {  // indirect-hash.cpp:13-17
        L.ihash[ctx1] = (uint32_t)(((ctxv << L.hash_size) + B) & L.mask);
        ctx1 = ((ctx1 << L.hash_size1) + B) & L.mask1;
        nc = L.ihash[ctx1];
        break;
      }
      case CT_INTERVAL:  // interval.cpp:17-19
        nc = L.mask & ((ctxv << L.shift) + cmap[L.map_id * 256 + B]);
        break;
      case CT_INTERVALHASH: {  // interval-hash.cpp:18-21
        const unsigned iv = L.mask1 & (((unsigned)ctx1 << L.shift) + cmap[L.map_id * 256 + B]);
        ctx1 = iv;
        nc = ((ctxv << L.hash_size) + iv) & L.mask;
        break;
      }
      case CT_COMBINED:  // combined-context.cpp:13-15 (shift_ = 8)
        nc = (regs[R_RECENT + ((L.orders >> 4) & 15)] << 8) + regs[R_RECENT + (L.orders & 15)];
        break;
      default: break;
    }
    ctxv = nc;
    if (lane < CTX_N) regs[R_CTX + lane] = nc;
    __syncthreads();
    wave_mem_sync();  // history_/hashes_ stores above are complete before the models probe

    // ---- Model::ByteUpdate (predictor.cpp:443-446) ----
    const u64 mctxv = (lane < CTX_NM && L.mtype >= MT_DIRECT) ? regs[R_CTX + L.mctx] : 0;
    bool need_reset = false;
    if (is_ind) {  // indirect.cpp:29-31
      mbase = (257 * mctxv + L.offset) % (CTX_SHARED - 257);
    } else if (L.mtype == MT_DIRECTHASH) {  // direct-hash.cpp:31-48
      u64 idx = modsz(mctxv, L.size_kind, L.size);
      u64 ck[20];
#pragma unroll
      for (int k = 0; k < 20; ++k) {
        u64 pos = idx + k;
        if (pos >= L.size) pos -= L.size;
        ck[k] = L.chk[pos];
      }
      int found = 19;
      bool claim = false;
#pragma unroll
      for (int k = 19; k >= 0; --k) {  // first k with an empty or matching slot wins
        if (ck[k] == 0) { found = k; claim = true; }
        else if (ck[k] == mctxv) { found = k; claim = false; }
      }
      if (found == 19 && ck[19] != 0 && ck[19] != mctxv) { need_reset = true; claim = true; }
      u64 pos = idx + found;
      if (pos >= L.size) pos -= L.size;
      if (claim) L.chk[pos] = mctxv;
      mbase = pos;
    } else if (is_match) {  // match.cpp:48-60
      mbase = modsz(mctxv, L.size_kind, L.size);
      if (ml < 8) cur_match = L.map[mbase]; else ++cur_match;
      cur_match %= CTX_HISTORY;
      cur_byte = D.history[cur_match];
    }
    {  // a DirectHash row that lost its 20 probes is re-initialised (direct-hash.cpp:38-42): whole wave fills it
      u64 rm = __ballot(need_reset);
      while (rm) {
        const int k = __ffsll((long long)rm) - 1;
        rm &= rm - 1;
        float* pr = (float*)__shfl((u64)L.pred, k) + __shfl(mbase, k) * 256;
        uint8_t* cn = (uint8_t*)__shfl((u64)L.cnt, k) + __shfl(mbase, k) * 256;
        *(float4*)(pr + 4 * lane) = make_float4(0.5f, 0.5f, 0.5f, 0.5f);
        *(uint32_t*)(cn + 4 * lane) = 0;
      }
    }
    {  // longest_match_ = max over Match models of match_length_/32 (match.cpp:58-59; reset at :73)
      unsigned lm = 0;
#pragma unroll
      for (unsigned v = 1; v < 8; ++v)
        if (__ballot(is_match && (ml >> 5) >= v)) lm = v;
      if (lane == 0) regs[R_LONGEST_MATCH] = lm;
    }

    // ---- Bracket::ByteUpdate (bracket.cpp:13-60) -> byte distribution ----
    if (lane == 0) {
      float p = 0.0f;
      int mode = 0, sym = 0;
      const int top = br_n ? (int)br_active[br_n - 1] : -1;
      const int cB = close6(B);
      if (br_n == 0 || (cB >= 0 && !(top == (int)B && cB == (int)B))) {
        if (cB >= 0) {
          br_active[br_n] = B; br_dist[br_n] = 0; ++br_n;
          if (br_n > 10) {  // stack_limit_
            for (unsigned i = 0; i + 1 < br_n; ++i) { br_active[i] = br_active[i + 1]; br_dist[i] = br_dist[i + 1]; }
            --br_n;
          }
          const unsigned* st = stats + (id6(B) * 200 + 0) * 2;
          p = (float)((1. * st[0]) / st[1]);
          mode = 1; sym = cB;
        }
      } else {
        const unsigned active = br_active[br_n - 1];
        unsigned distance = br_dist[br_n - 1];
        unsigned* st = stats + (id6(active) * 200 + distance) * 2;
        ++st[1];
        if (close6(active) == (int)B) ++st[0];
        if (st[1] > 100000) { st[0] /= 2; st[1] /= 2; }  // stats_limit_
        if (close6(active) == (int)B || distance >= 200 - 1) {
          --br_n;
          if (br_n) {
            const unsigned a = br_active[br_n - 1], d = br_dist[br_n - 1];
            const unsigned* s2 = stats + (id6(a) * 200 + d) * 2;
            p = (float)((1. * s2[0]) / s2[1]);
            mode = 1; sym = close6(a);
          }
        } else {
          ++br_dist[br_n - 1];
          ++distance;
          const unsigned* s2 = stats + (id6(active) * 200 + distance) * 2;
          p = (float)((1. * s2[0]) / s2[1]);
          mode = 1; sym = close6(active);
        }
      }
      brout[0] = (float)mode;
      brout[1] = p;
      brout[2] = (1.0f - p) / 255.0f;
      brout[3] = (float)sym;
    }
    __syncthreads();
    {
      const int mode = (int)brout[0], sym = (int)brout[3];
      const float p = brout[1], q = brout[2];
      float v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int i = 4 * lane + k;
        float x = mode ? (i == sym ? p : q) : (float)(1. / 256);
        if (!D.vocab[i]) x = 0.0f;  // ByteModel::ByteUpdate, byte-model.cpp:39-45
        v[k] = x;
      }
      *(float4*)(bracket_dist + n * 256 + 4 * lane) = make_float4(v[0], v[1], v[2], v[3]);
    }
    if (late) wave_mem_sync();   // (this wave's stores of the distribution have landed before the barrier: a barrier alone does not wait for them)
    __syncthreads();
    if (late) {   // the Bracket model's distribution after byte n is in place (the ByteModel kernel of the late pipeline reads it)
      if (lane == 0) late_publish(LB, LC_BRK, (uint32_t)(n + 1));
    }
  }

  // ---- chunk epilogue: LDS / registers -> HBM ----
  __syncthreads();
  for (int i = lane; i < R_COUNT; i += 64) P->regs[i] = regs[i];
  for (int i = lane; i < CTX_N_INDIRECT * 256; i += 64) (&P->ipred[0][0])[i] = ipred[(i >> 8) * CTX_PRED_STRIDE + (i & 255)];
  for (int i = lane; i < CTX_N_MATCH * 256; i += 64) {
    (&P->mpred[0][0])[i] = mpred[(i >> 8) * CTX_PRED_STRIDE + (i & 255)];
    (&P->mcnt[0][0])[i] = mcnt[i];
  }
  for (int i = lane; i < 6 * 200 * 2; i += 64) (&P->br_stats[0][0][0])[i] = stats[i];
  if (lane < 16) { P->br_active[lane] = br_active[lane]; P->br_dist[lane] = br_dist[lane]; }
  P->ctx1[lane] = ctx1;
  P->map_index[lane] = mbase;
  P->m_history_pos[lane] = mhp;
  P->cur_match[lane] = cur_match;
  P->cur_byte[lane] = cur_byte;
  P->match_length[lane] = ml;
  if (nbytes) {
    const float4 last = *(const float4*)(bracket_dist + (nbytes - 1) * 256 + 4 * lane);
    *(float4*)(P->br_probs + 4 * lane) = last;
  }
  if (lane == 0) {
    P->history_pos = history_pos;
    P->bytes_done = bytes_done0 + nbytes;
    P->wrt_state = wrt_state;
    P->bc_n = bc_n; P->bc_top_active = bc_top_active; P->bc_top_dist = bc_top_dist;
    P->br_n = br_n;
  }
}
}  // namespace

// chunk mode (and Pretrain): committing passes over known bytes
extern "C" __global__ void __launch_bounds__(64)
cmx_ctxmodels_kernel(const CtxDev D, const uint8_t* __restrict__ bytes, size_t nbytes, float* probs, size_t pstride,
                     uint32_t* sel, float* bracket_dist) {
  ctxmodels_body<false>(D, bytes, nbytes, probs, pstride, sel, bracket_dist);
}

// the decoder's form (cmx_late.h): bits arrive through the box as the arithmetic decoder produces them
extern "C" __global__ void __launch_bounds__(64)
cmx_ctxmodels_late_kernel(const CtxDev D, CmxLate box, size_t nbytes, float* probs, size_t pstride, uint32_t* sel,
                          float* bracket_dist) {
  ctxmodels_body<false, true>(D, nullptr, nbytes, probs, pstride, sel, bracket_dist, box);
}

// bit-synchronous mode: one dry pass over a partially known byte (see the comment at the top of the body)
extern "C" __global__ void __launch_bounds__(64)
cmx_ctxmodels_peek_kernel(const CtxDev D, const uint8_t* __restrict__ bytes, float* probs, size_t pstride,
                          uint32_t* sel) {
  ctxmodels_body<true>(D, bytes, 1, probs, pstride, sel, nullptr);
}
