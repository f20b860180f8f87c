// ppmd_host.cpp -- HOST stage: the order-25 PPMd byte model that feeds the LSTM byte mixer and
// layer-0 column 2076 (C ABI: cmx_ppmd_* in include/cmix_amd.h).
//
// Reference: PPMD::PPMD / PPMD::ByteUpdate (src/models/ppmd.cpp:1322-1338), ppmd_UpdateByte (:1282),
// ppmd_PrepareByte (:1256), ConvertSQ (:1130), the symbol coders (:958-1107, :1156-1230),
// UpdateModel / CreateSuccessors / ReduceOrder (:729-955), rescale (:493-560), StartModelRare (:646),
// the sub-allocator (:56-290), constructed with (order 25, 14000 MB) at src/predictor.cpp:101.
//
// Why a host stage (note 2): PPMd is a suffix-linked context tree in a private unit
// allocator -- byte-rate pointer chasing, 0.4 % of the reference's CPU time, and a pure function of the
// byte stream, so it runs ahead of the device pipeline on one host core and ships 1 KB per byte.
// There is no device or fallback twin of this code; it IS the product's PPMd.
//
// Own design, same arithmetic: contexts and symbol-state arrays live in one lazily committed arena of
// 12-byte units addressed by 32-bit handles; a handle below `text_limit` is a position in the raw-text
// area (a successor that has not been turned into a context yet), anything above is a unit number.
// Unit bookkeeping mirrors the reference's size classes and free-list discipline so that the arena
// fills up at exactly the same moment; what happens after that moment (the reference's cut-off /
// restore path, ppmd.cpp:562-640,686-727 -- in which the reference build itself segfaults, checked with a
// 1 MB arena) is NOT implemented: the model reports the condition and
// stops (cmx_ppmd_run fails loudly) instead of drifting away from the reference. With the reference's
// 14000 MB arena this is beyond ~1.7 GB of input text.
#include <stddef.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <sys/mman.h>
#include <string>

#include "../../include/cmix_amd.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip

namespace {

typedef uint8_t u8;
typedef uint16_t u16;
typedef uint32_t u32;
typedef uint64_t u64;

constexpr int kUnit = 12;
constexpr int kClasses = 4 + 4 + 4 + (128 + 3 - 1 * 4 - 2 * 4 - 3 * 4) / 4;  // 38 size classes (ppmd.cpp:38-42)
constexpr int kMaxFreq = 124, kIntBits = 7, kPeriodBits = 7, kBinScale = 1 << (kIntBits + kPeriodBits);
constexpr int kInterval = 1 << kIntBits, kScale = 1 << 15;

#pragma pack(push, 1)
struct Sym {           // one symbol of a context: 6 bytes, two per unit
  u8 symbol, freq;
  u32 succ;            // handle: text position (< text_limit) or context unit
};
struct Ctx {           // 12 bytes = one unit
  u8 nstats;           // number of symbols - 1; 0 = binary context, its single Sym overlays the next 6 bytes
  u8 flags;
  u16 summ;
  u32 stats;           // handle of the Sym array
  u32 suffix;
  Sym& one() { return *reinterpret_cast<Sym*>(&summ); }
};
struct FreeBlk { u32 stamp, next, nu; };  // header written into a free block
#pragma pack(pop)

struct See { u16 summ; u8 shift, count; };

struct Ppmd {
  // ---- arena ----
  u8* heap = nullptr;
  u64 heap_bytes = 0;
  u8 *text, *units_start, *lo, *hi;     // raw text cursor; [lo,hi) = untouched units
  u32 text_limit = 0;
  struct { u32 count, next; } flist[kClasses + 1];
  u8 idx2units[kClasses], units2idx[128];
  // ---- model ----
  u8 ns2bs[256], qtable[260];
  int max_order = 25, order_fall = 0, run_length = 0, init_rl = 0, prev_success = 0, bsumm = 0, num_masked = 0;
  u32 esc_count = 1, char_mask[256];
  Ctx* max_ctx = nullptr;
  Sym* found = nullptr;
  u16 bin_summ[25][64];
  See see[23][32], dummy_see;
  struct { u16 sym, freq, total; } sq[1024];
  u32 sq_n = 0, sqp[256];
  bool exhausted = false;
  u8 vocab[256];

  // handles <-> pointers (ppmd.cpp:46-57)
  u32 handle(const void* p) const {
    const u64 a = (const u8*)p - heap;
    return a >= text_limit ? (u32)((a - text_limit) / kUnit + text_limit) : (u32)a;
  }
  u8* ptr(u32 h) const { return heap + (h >= text_limit ? (u64)(h - text_limit) * kUnit + text_limit : (u64)h); }
  Ctx* ctx(u32 h) const { return (Ctx*)ptr(h); }
  Sym* syms(const Ctx* c) const { return (Sym*)ptr(c->stats); }
  bool is_text(u32 h) const { return h < text_limit; }

  // ---- unit allocator: size classes and LIFO free lists as in the reference (ppmd.cpp:59-118,214-262) ----
  void fl_push(int k, void* p, u32 nu) {
    FreeBlk* b = (FreeBlk*)p;
    b->next = flist[k].next; b->stamp = ~0u; b->nu = nu;
    flist[k].next = handle(p); flist[k].count++;
  }
  void* fl_pop(int k) {
    FreeBlk* b = (FreeBlk*)ptr(flist[k].next);
    flist[k].next = b->next; flist[k].count--;
    return b;
  }
  void split(void* pv, int old_k, int new_k) {  // hand the tail of a block back (ppmd.cpp:198-209)
    u32 diff = idx2units[old_k] - idx2units[new_k];
    u8* p = (u8*)pv + (u32)idx2units[new_k] * kUnit;
    int i = units2idx[diff - 1];
    if (idx2units[i] != diff) {
      const u32 k = idx2units[--i];
      fl_push(i, p, k);
      p += k * kUnit; diff -= k;
    }
    fl_push(units2idx[diff - 1], p, diff);
  }
  void* alloc_units(u32 nu) {
    const int k = units2idx[nu - 1];
    if (flist[k].next) return fl_pop(k);
    void* r = lo;
    lo += (u32)idx2units[k] * kUnit;
    if (lo <= hi) return r;
    lo -= (u32)idx2units[k] * kUnit;
    exhausted = true;  // the reference would now scavenge larger classes / glue / eat the text area
    return nullptr;
  }
  Ctx* alloc_ctx() {
    if (hi != lo) return (Ctx*)(hi -= kUnit);
    exhausted = true;
    return nullptr;
  }
  void free_units(void* p, u32 nu) { const int k = units2idx[nu - 1]; fl_push(k, p, idx2units[k]); }
  void* expand_units(void* old, u32 old_nu) {
    const int k0 = units2idx[old_nu - 1], k1 = units2idx[old_nu];
    if (k0 == k1) return old;
    void* p = alloc_units(old_nu + 1);
    if (p) { memcpy(p, old, (size_t)old_nu * kUnit); fl_push(k0, old, old_nu); }
    return p;
  }
  void* shrink_units(void* old, u32 old_nu, u32 new_nu) {
    const int k0 = units2idx[old_nu - 1], k1 = units2idx[new_nu - 1];
    if (k0 == k1) return old;
    if (flist[k1].next) {
      void* p = fl_pop(k1);
      memcpy(p, old, (size_t)new_nu * kUnit);
      fl_push(k0, old, idx2units[k0]);
      return p;
    }
    split(old, k0, k1);
    return old;
  }

  bool init(int order, u64 memory_mb) {
    max_order = order;
    heap_bytes = memory_mb << 20;   // 14000 MB at predictor.cpp:101
    void* m = mmap(nullptr, heap_bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
    if (m == MAP_FAILED) return false;
    heap = (u8*)m;
    // size classes: 1,2,3,4, 6,8,10,12, 15,18,21,24, 28,32,...,128 units (ppmd.cpp:379-388)
    int i = 0, k = 1;
    for (; i < 4; ++i, k += 1) idx2units[i] = k;
    for (k++; i < 8; ++i, k += 2) idx2units[i] = k;
    for (k++; i < 12; ++i, k += 3) idx2units[i] = k;
    for (k++; i < kClasses; ++i, k += 4) idx2units[i] = k;
    for (k = 0, i = 0; k < 128; ++k) { i += idx2units[i] < k + 1; units2idx[k] = i; }
    ns2bs[0] = 0; ns2bs[1] = ns2bs[2] = 2;
    memset(ns2bs + 3, 4, 26); memset(ns2bs + 29, 6, 256 - 29);
    for (i = 0; i < 5; ++i) qtable[i] = i;
    int m2 = 5, step = 1;
    for (i = 5, k = 1; i < 260; ++i) { qtable[i] = m2; if (!--k) { k = ++step; m2++; } }
    restart();
    return true;
  }

  void restart() {  // StartModelRare for max_order >= 2 (ppmd.cpp:646-684)
    memset(char_mask, 0, sizeof char_mask);
    esc_count = 1;
    order_fall = max_order;
    memset(flist, 0, sizeof flist);
    hi = heap + heap_bytes; text = heap;
    const u64 diff = heap_bytes / 8 / kUnit * 7 * kUnit;
    lo = units_start = hi - diff;
    text_limit = (u32)(units_start - heap);
    init_rl = -(max_order < 13 ? max_order : 13);
    run_length = init_rl;
    max_ctx = alloc_ctx();
    max_ctx->nstats = 255; max_ctx->summ = 257; max_ctx->flags = 0; max_ctx->suffix = 0;
    Sym* s = (Sym*)alloc_units(128);
    max_ctx->stats = handle(s);
    prev_success = 0;
    for (int i = 0; i < 256; ++i) { s[i].symbol = i; s[i].freq = 1; s[i].succ = 0; }
    static const signed char esc_coef[12] = {16, -10, 1, 51, 14, 89, 23, 35, 64, 26, -42, 43};
    u8 i2f[25];
    for (int k = 0, i = 0; i < 25; i2f[i++] = k + 1) while (qtable[k] == i) k++;
    for (int k = 0; k < 64; ++k) {
      int sum = 0;
      for (int i = 0; i < 6; ++i) sum += esc_coef[2 * i + ((k >> i) & 1)];
      sum = 128 * (sum < 32 ? 32 : sum > 224 ? 224 : sum);
      for (int i = 0; i < 25; ++i) bin_summ[i][k] = (u16)(kBinScale - sum / i2f[i]);
    }
    for (int i = 0; i < 23; ++i)
      for (int k = 0; k < 32; ++k) { See& e = see[i][k]; e.shift = kPeriodBits - 4; e.summ = (u16)((8 * i + 5) << e.shift); e.count = 7; }
  }

  // ---- SEE (ppmd.cpp:463-489) ----
  static void see_update(See& e) {
    if (--e.count == 0) {
      u32 i = e.summ >> e.shift;
      i = kPeriodBits - (i > 40) - (i > 280) - (i > 1020);
      if (i < e.shift) { e.summ >>= 1; e.shift--; } else if (i > e.shift) { e.summ <<= 1; e.shift++; }
      e.count = (u8)(5 << e.shift);
    }
  }
  See* see_for(Ctx* q, int& see_freq) {  // ppmd.cpp:1052-1062
    const int cnum = q->nstats;
    if (cnum == 0xFF) { see_freq = 1; return &dummy_see; }
    See* e = see[qtable[cnum + 3] - 4];
    e += q->summ > 10 * (cnum + 1);
    e += 2 * (2 * cnum < ctx(q->suffix)->nstats + num_masked) + q->flags;
    see_freq = (e->summ >> e->shift) + 1;
    return e;
  }
  u16& bin_cell(Ctx* q) {  // ppmd.cpp:961-963
    Sym& rs = q->one();
    const int i = ns2bs[ctx(q->suffix)->nstats] + prev_success + q->flags + ((run_length >> 26) & 0x20);
    return bin_summ[qtable[rs.freq - 1]][i];
  }

  static void swap_sym(Sym& a, Sym& b) { Sym t = a; a = b; b = t; }

  // ---- frequency halving when a symbol outgrows MAX_FREQ (ppmd.cpp:493-560) ----
  Sym* rescale(Ctx* q, Sym* fs) {
    q->flags &= 0x14;
    Sym* base = syms(q);
    Sym tmp = *fs;
    for (Sym* p = fs; p != base; --p) p[0] = p[-1];
    base[0] = tmp;
    const int of = order_fall != 0;
    Sym* p = base;
    const int f0 = p->freq, sf0 = q->summ;
    int esc = sf0 - f0, sf = sf0;
    q->summ = p->freq = (u8)((f0 + of) >> 1);
    for (int i = 0; i < q->nstats; ++i) {
      ++p;
      int a = p->freq;
      esc -= a;
      a = (a + of) >> 1;
      p->freq = (u8)a;
      q->summ += a;
      if (a) q->flags |= 0x08 * (p->symbol >= 0x40);
      if (a > p[-1].freq) {
        tmp = *p;
        Sym* p1 = p;
        for (; tmp.freq > p1[-1].freq; --p1) p1[0] = p1[-1];
        *p1 = tmp;
      }
    }
    if (p->freq == 0) {
      int i = 0;
      for (; p->freq == 0; ++i, --p) {}
      esc += i;
      const int a = (q->nstats + 2) >> 1;
      if ((q->nstats -= i) == 0) {
        tmp = syms(q)[0];
        const int v = (2 * tmp.freq + esc - 1) / esc;
        tmp.freq = (u8)(v < kMaxFreq / 3 ? v : kMaxFreq / 3);
        q->flags &= 0x18;
        free_units(syms(q), a);
        q->one() = tmp;
        return &q->one();
      }
      q->stats = handle(shrink_units(syms(q), a, (q->nstats + 2) >> 1));
    }
    q->summ += (esc + 1) >> 1;
    int a;
    if (order_fall || (q->flags & 0x04) == 0) {
      a = (sf -= esc) - f0;
      const u32 v = (u32)((f0 * q->summ - sf * syms(q)->freq + a - 1) / a);
      a = v < 2u ? 2 : v > kMaxFreq / 2u - 18u ? (int)(kMaxFreq / 2u - 18u) : (int)v;
    } else a = 2;
    Sym* r = syms(q);
    r->freq += a;
    q->summ += a;
    q->flags |= 0x04;
    return r;
  }

  // ---- coding one known symbol (the <0> instantiations, ppmd.cpp:958-1107) ----
  void code_binary(Ctx* q, int symbol) {
    Sym& rs = q->one();
    u16& bs = bin_cell(q);
    bsumm = bs;
    bs -= (bsumm + 64) >> kPeriodBits;
    if (rs.symbol != symbol) {
      char_mask[rs.symbol] = esc_count;
      num_masked = 0; prev_success = 0; found = nullptr;
    } else {
      bs += kInterval;
      rs.freq += rs.freq < 196;
      run_length++;
      prev_success = 1; found = &rs;
    }
  }
  void code_first(Ctx* q, int symbol) {
    Sym* p = syms(q);
    const int cnum = q->nstats;
    prev_success = 0;
    if (p[0].symbol == symbol) {
      p[0].freq += 4; q->summ += 4;
    } else {
      int i = 1;
      for (; i <= cnum && p[i].symbol != symbol; ++i) {}
      if (i <= cnum) {
        p[i].freq += 4; q->summ += 4;
        if (p[i].freq > p[i - 1].freq) { swap_sym(p[i], p[i - 1]); --i; }
        p = &p[i];
      } else {
        num_masked = cnum;
        for (i = 0; i <= cnum; ++i) char_mask[p[i].symbol] = esc_count;
        p = nullptr;
      }
    }
    found = p;
    if (p && p->freq > kMaxFreq) found = rescale(q, found);
  }
  void code_masked(Ctx* q, int symbol) {
    Sym* p = syms(q);
    const int cnum = q->nstats;
    int see_freq;
    See* e = see_for(q, see_freq);
    int low = 0, hit = -1;
    for (int i = 0; i <= cnum; ++i) {
      const int c = p[i].symbol;
      if (char_mask[c] != esc_count) {
        char_mask[c] = esc_count;
        low += p[i].freq;
        if (c == symbol) hit = i;
      }
    }
    const int total = see_freq + low;
    if (hit >= 0) {
      p += hit;
      if (see_freq > 2) e->summ -= see_freq;
      see_update(*e);
      found = p;
      p->freq += 4; q->summ += 4;
      if (p->freq > kMaxFreq) found = rescale(q, found);
      run_length = init_rl;
      esc_count++;
    } else {
      num_masked = cnum;
      e->summ += total - see_freq;
    }
  }

  // ---- listing the whole next-byte distribution (the _T variants, ppmd.cpp:1156-1230) ----
  void sq_put(u32 s, u32 f, u32 t) { sq[sq_n].sym = (u16)s; sq[sq_n].freq = (u16)f; sq[sq_n].total = (u16)t; ++sq_n; }
  void list_binary(Ctx* q) {
    Sym& rs = q->one();
    bsumm = bin_cell(q);
    sq_put(rs.symbol, bsumm + bsumm, kScale);
    sq_put(256, kScale - bsumm - bsumm, kScale);
    char_mask[rs.symbol] = esc_count;
    num_masked = 0;
  }
  void list_first(Ctx* q) {
    Sym* p = syms(q);
    const int cnum = q->nstats, total = q->summ;
    int low = 0;
    for (int i = 0; i <= cnum; ++i) { sq_put(p[i].symbol, p[i].freq, total); low += p[i].freq; }
    num_masked = cnum;
    for (int i = 0; i <= cnum; ++i) char_mask[p[i].symbol] = esc_count;
    sq_put(256, total - low, total);
  }
  void list_masked(Ctx* q) {
    Sym* p = syms(q);
    const int cnum = q->nstats;
    int see_freq;
    see_for(q, see_freq);
    int low = 0;
    for (int i = 0; i <= cnum; ++i) if (char_mask[p[i].symbol] != esc_count) low += p[i].freq;
    const int total = see_freq + low;
    for (int i = 0; i <= cnum; ++i) {
      const int c = p[i].symbol;
      if (char_mask[c] != esc_count) { sq_put(c, p[i].freq, total); char_mask[c] = esc_count; }
    }
    sq_put(256, see_freq, total);
    num_masked = cnum;
  }
  void prepare_byte() {  // ppmd_PrepareByte + ConvertSQ (ppmd.cpp:1256-1280, 1130-1154)
    sq_n = 0; num_masked = 0;
    const int saved_of = order_fall;
    Ctx* mc = max_ctx;
    if (mc->nstats) list_first(mc); else list_binary(mc);
    for (;;) {
      bool root = false;
      do {
        if (!mc->suffix) { root = true; break; }
        order_fall++;
        mc = ctx(mc->suffix);
      } while (mc->nstats == num_masked);
      if (root) break;
      list_masked(mc);
    }
    esc_count++; num_masked = 0; order_fall = saved_of;
    u32 cum = 0xFFFFFF00u;
    memset(sqp, 0, sizeof sqp);
    for (u32 i = 0; i < sq_n; ++i) {
      const u32 prob = (u32)((u64)cum * sq[i].freq / sq[i].total);
      if (sq[i].sym < 256) sqp[sq[i].sym] = prob + 1; else cum = prob;
    }
  }

  // ---- model growth (ppmd.cpp:729-955) ----
  u32 create_successors(bool skip, Sym* p, Ctx* pc) {
    Sym* stack[256];
    int n = 0;
    u8 sym = found->symbol;
    const u32 up = found->succ;
    bool enter = false;
    if (!skip) {
      stack[n++] = found;
      if (!pc->suffix) goto build;
    }
    if (p) { pc = ctx(pc->suffix); enter = true; }
    do {
      if (!enter) {
        pc = ctx(pc->suffix);
        if (pc->nstats) {
          for (p = syms(pc); p->symbol != sym; ++p) {}
          const u8 t = 2 * (p->freq < kMaxFreq - 1);
          p->freq += t; pc->summ += t;
        } else {
          p = &pc->one();
          p->freq += (!ctx(pc->suffix)->nstats) & (p->freq < 16);
        }
      }
      enter = false;
      if (p->succ != up) { pc = ctx(p->succ); break; }
      stack[n++] = p;
    } while (pc->suffix);
  build:
    if (n == 0) return handle(pc);
    Ctx ct;
    ct.nstats = 0;
    ct.flags = 0x10 * (sym >= 0x40);
    sym = *ptr(up);                       // the byte that followed in the raw text
    ct.one().succ = up + 1;               // (both are text positions: handle == offset)
    ct.one().symbol = sym;
    ct.flags |= 0x08 * (sym >= 0x40);
    if (pc->nstats) {
      for (p = syms(pc); p->symbol != sym; ++p) {}
      u32 cf = p->freq - 1;
      const u32 s0 = pc->summ - pc->nstats - cf;
      cf = 1 + ((2 * cf < s0) ? (12 * cf > s0) : 2 + cf / s0);
      ct.one().freq = (u8)(cf < 7 ? cf : 7);
    } else ct.one().freq = pc->one().freq;
    do {
      Ctx* c1 = alloc_ctx();
      if (!c1) return 0;
      memcpy(c1, &ct, 8);
      c1->suffix = handle(pc);
      pc = c1;
      stack[--n]->succ = handle(pc);
    } while (n);
    return handle(pc);
  }

  u32 reduce_order(Sym* p, Ctx* pc) {
    Ctx* const pc1 = pc;
    found->succ = handle(text);
    const u8 sym = found->symbol;
    const u32 up = found->succ;
    order_fall++;
    bool enter = p != nullptr;
    if (enter) pc = ctx(pc->suffix);
    for (;;) {
      if (!enter) {
        if (!pc->suffix) return handle(pc);
        pc = ctx(pc->suffix);
        if (pc->nstats) {
          for (p = syms(pc); p->symbol != sym; ++p) {}
          const u8 t = 2 * (p->freq < kMaxFreq - 3);
          p->freq += t; pc->summ += t;
        } else {
          p = &pc->one();
          p->freq += p->freq < 11;
        }
      }
      enter = false;
      if (p->succ) break;
      p->succ = up;
      order_fall++;
    }
    if (p->succ <= up) {
      Sym* keep = found;
      found = p;
      p->succ = create_successors(false, nullptr, pc);
      found = keep;
    }
    if (order_fall == 1 && pc1 == max_ctx) {
      found->succ = p->succ;
      text--;
    }
    return p->succ;
  }

  Ctx* update_model(Ctx* min_ctx) {
    static const u8 exp_escape[16] = {51, 43, 18, 12, 11, 9, 8, 7, 6, 5, 4, 3, 3, 2, 2, 2};
    const u8 fsym = found->symbol;
    const u32 ffreq = found->freq;
    u32 fsucc = found->succ;
    Sym* p = nullptr;
    Ctx* pc;
    if (min_ctx->suffix) {
      pc = ctx(min_ctx->suffix);
      if (pc->nstats) {
        p = syms(pc);
        if (p->symbol != fsym) {
          for (++p; p->symbol != fsym; ++p) {}
          if (p[0].freq >= p[-1].freq) { swap_sym(p[0], p[-1]); --p; }
        }
        if (p->freq < kMaxFreq - 3) {
          const u32 cf = 2 + (ffreq < 28);
          p->freq += cf; pc->summ += cf;
        }
      } else {
        p = &pc->one();
        p->freq += p->freq < 14;
      }
    }
    if (!order_fall && fsucc) {
      found->succ = create_successors(true, p, min_ctx);
      if (!found->succ) return nullptr;
      max_ctx = ctx(found->succ);
      return max_ctx;
    }
    *text++ = fsym;
    u32 succ = handle(text);
    if (text >= units_start) { exhausted = true; return nullptr; }
    if (fsucc) {
      if (is_text(fsucc)) fsucc = create_successors(false, p, min_ctx);
    } else fsucc = reduce_order(p, min_ctx);
    if (!fsucc) return nullptr;
    if (!--order_fall) {
      succ = fsucc;
      text -= max_ctx != min_ctx;
    }
    const u32 s0 = min_ctx->summ - ffreq, ns = min_ctx->nstats;
    const u8 flag = 0x08 * (fsym >= 0x40);
    for (pc = max_ctx; pc != min_ctx; pc = ctx(pc->suffix)) {
      const u32 ns1 = pc->nstats;
      if (ns1) {
        if (ns1 & 1) {
          p = (Sym*)expand_units(syms(pc), (ns1 + 1) >> 1);
          if (!p) return nullptr;
          pc->stats = handle(p);
        }
        pc->summ += qtable[ns + 4] >> 3;
      } else {
        p = (Sym*)alloc_units(1);
        if (!p) return nullptr;
        p[0] = pc->one();
        pc->stats = handle(p);
        p[0].freq = p[0].freq <= kMaxFreq / 3 ? 2 * p[0].freq - 1 : kMaxFreq - 15;
        pc->summ = p[0].freq + (ns > 1) + exp_escape[qtable[bsumm >> 8]];
      }
      u32 cf = (ffreq - 1) * (5 + pc->summ);
      const u32 sf = s0 + pc->summ;
      if (cf <= 3 * sf) {
        cf = 1 + (2 * cf > sf) + (2 * cf > 3 * sf);
        pc->summ += 4;
      } else {
        cf = 5 + (cf > 5 * sf) + (cf > 6 * sf) + (cf > 8 * sf) + (cf > 10 * sf) + (cf > 12 * sf);
        pc->summ += cf;
      }
      p = syms(pc) + (++pc->nstats);
      p->succ = succ; p->symbol = fsym; p->freq = (u8)cf;
      pc->flags |= flag;
    }
    max_ctx = ctx(fsucc);
    return max_ctx;
  }

  bool update_byte(int c) {  // ppmd_UpdateByte (ppmd.cpp:1282-1318)
    Ctx* mc = max_ctx;
    if (mc->nstats) code_first(mc, c); else code_binary(mc, c);
    while (!found) {
      do {
        order_fall++;
        mc = ctx(mc->suffix);
      } while (mc->nstats == num_masked);
      code_masked(mc, c);
    }
    if (order_fall != 0 || is_text(found->succ)) {
      Ctx* p = update_model(mc);
      if (!p) return false;  // the reference's RestoreModelRare point
      max_ctx = p;
    } else max_ctx = ctx(found->succ);
    return true;
  }
};

}  // namespace

struct cmx_ppmd {
  Ppmd m;
};

extern "C" {

cmx_ppmd_t* cmx_ppmd_create(const uint8_t vocab[256]) { return cmx_ppmd_create_ex(vocab, 25, 14000); }

// Test hook: other order / arena size (the reference's own constructor arguments, ppmd.cpp:1322-1326).
cmx_ppmd_t* cmx_ppmd_create_ex(const uint8_t vocab[256], int order, int memory_mb) {
  if (order < 2 || order > 64 || memory_mb < 1) { cmx_set_err("cmx_ppmd_create_ex: bad argument"); return nullptr; }
  cmx_ppmd_t* h = new cmx_ppmd();
  memcpy(h->m.vocab, vocab, 256);
  if (!h->m.init(order, (uint64_t)memory_mb)) {
    cmx_set_err("cmx_ppmd_create: cannot reserve the PPMd arena (mmap failed)");
    delete h;
    return nullptr;
  }
  return h;
}

void cmx_ppmd_destroy(cmx_ppmd_t* h) {
  if (!h) return;
  if (h->m.heap) munmap(h->m.heap, h->m.heap_bytes);
  delete h;
}

// diagnostics: bytes of the arena in use -- raw text from the bottom, units from the top; the model restarts (not implemented: the call
// above fails) when the untouched gap [lo, hi) between them is used up
int cmx_ppmd_arena(cmx_ppmd_t* h, uint64_t out3[3]) {
  if (!h || !out3) return 1;
  const Ppmd& m = h->m;
  out3[0] = m.heap_bytes;
  out3[1] = (uint64_t)(m.hi - m.lo);                 // untouched
  out3[2] = m.heap_bytes - (uint64_t)(m.hi - m.lo);  // in use (text + units, incl. free-listed units)
  return 0;
}

int cmx_ppmd_run(cmx_ppmd_t* h, const uint8_t* bytes, size_t nbytes, float* out_probs) {
  if (!h || (nbytes && (!bytes || !out_probs))) { cmx_set_err("cmx_ppmd_run: bad argument"); return 1; }
  Ppmd& m = h->m;
  for (size_t n = 0; n < nbytes; ++n) {
    if (m.exhausted || !m.update_byte(bytes[n])) {
      m.exhausted = true;
      cmx_set_err("cmx_ppmd_run: the PPMd arena is exhausted; the reference's cut-off/restore path "
                  "(ppmd.cpp:686-727) is not implemented (the reference build itself crashes there)");
      return 1;
    }
    m.prepare_byte();
    // PPMD::ByteUpdate tail (ppmd.cpp:1331-1337): floor at 1, vocabulary mask, normalise by the forward sum
    float* pr = out_probs + n * 256;
    for (int i = 0; i < 256; ++i) {
      float v = (float)m.sqp[i];
      if (v < 1) v = 1;
      pr[i] = m.vocab[i] ? v : 0.0f;
    }
    float sum = 0.0f;
    for (int i = 0; i < 256; ++i) sum = sum + pr[i];
    for (int i = 0; i < 256; ++i) pr[i] = pr[i] / sum;
  }
  return 0;
}

}  // extern "C"
