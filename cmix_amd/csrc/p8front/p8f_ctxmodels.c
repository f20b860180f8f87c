/* p8front/p8f_ctxmodels.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for three of paq8's context models that read only the byte history and plain globals -- nestModel
 * (reference src/models/paq8.cpp:4107-4181: bracket / quote / vowel-class nesting state), distanceModel (:4598-4612:
 * distance to the last 0x00 / space / line end) and indirectModel (:7548-7599: byte-history-indexed second-order
 * contexts) -- each feeding a ContextMap (p8f_emit.c). They show the shape of the rest of paq8's front end: a
 * few dozen integer state updates per byte ending in ContextMap::set(hash(...)) calls, then ContextMap::mix() per bit.
 * Parity: tests/test_p8stage_host.py (stage vs columns 434..2024 of reference traces).
 *
 * Inputs per bit: the coded bit, bpos, c0 and -- used at bpos == 0 only -- c4 (last four bytes), f4, pos, and
 * last[i-1] = buf(i) for i = 1..8. */
#include <ctype.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct CM1 CM1;
CM1* p8f_cm_new(uint64_t size_bytes, int count);
int p8f_cm_step(CM1* c, int y1, int bp, int c0, int c1, const uint64_t* ctx, int nset, int16_t* out, int* nout);
void p8f_cm_order_slot(CM1* c, int idx, uint64_t seed);

#define PHI64 0x9E3779B97F4A7C15ull
static const uint64_t MUL[8] = {PHI64, 0x993DDEFFB1462949ull, 0xE9C91DC159AB0D2Dull, 0x83D6A14F1B0CED73ull,
                                0xA14F1B0CED5A841Full, 0xC0E51314A614F4EFull, 0xDA9CC2600AE45A27ull, 0x826797AA04A65737ull};
static uint64_t hashn(int n, const uint64_t* x) {  /* hash(x0 .. x(n-1)) :742-773 */
  uint64_t h = 0;
  for (int i = 0; i < n; ++i) h += (x[i] + 1) * MUL[i];
  return h;
}
#define H2(a, b) hashn(2, (const uint64_t[]){(uint64_t)(a), (uint64_t)(b)})
#define H3(a, b, c) hashn(3, (const uint64_t[]){(uint64_t)(a), (uint64_t)(b), (uint64_t)(c)})
#define H4(a, b, c, d) hashn(4, (const uint64_t[]){(uint64_t)(a), (uint64_t)(b), (uint64_t)(c), (uint64_t)(d)})
#define H6(a, b, c, d, e, f) hashn(6, (const uint64_t[]){(uint64_t)(a), (uint64_t)(b), (uint64_t)(c), (uint64_t)(d), (uint64_t)(e), (uint64_t)(f)})

static unsigned ilog2u(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }
static uint64_t mem_of(int level) { return 0x10000ull << level; }

typedef struct {
  int which, level;
  CM1* cm;
  /* nestModel :4109-4110 */
  int ic, bc, pc, qc, lvc, ac, ec, uc, sense1, sense2, w;
  unsigned vc, wc;
  /* distanceModel :4601 */
  int pos00, pos20, posnl;
  /* indirectModel :7550-7554 */
  uint32_t t1[256];
  uint16_t t2[0x10000], t3[0x8000], t4[0x8000];
  uint32_t ictx_data[1 << 16];
  uint32_t ictx_cur;  /* index into ictx_data */
} P8CtxModel;

P8CtxModel* p8f_ctxmodel_new(int which, int level) {
  P8CtxModel* m = (P8CtxModel*)calloc(1, sizeof *m);
  m->which = which; m->level = level;
  m->cm = which == 0 ? p8f_cm_new(mem_of(level) / 2, 12) : which == 1 ? p8f_cm_new(mem_of(level), 3)
                                                                         : p8f_cm_new(mem_of(level), 15);
  return m;
}

static int nest_contexts(P8CtxModel* m, uint32_t c4, uint32_t f4, const uint8_t* last, uint64_t* cx) {
  int c = c4 & 255, matched = 1, vv;
  m->w *= ((m->vc & 7) > 0 && (m->vc & 7) < 3);
  if (c & 0x80) m->w = m->w * 11 * 32 + c;
  const int lc = (c >= 'A' && c <= 'Z' ? c + 'a' - 'A' : c);
  if (lc == 'a' || lc == 'e' || lc == 'i' || lc == 'o' || lc == 'u') { vv = 1; m->w = m->w * 997 * 8 + (lc / 4 - 22); }
  else if (lc >= 'a' && lc <= 'z') { vv = 2; m->w = m->w * 271 * 32 + lc - 97; }
  else if (lc == ' ' || lc == '.' || lc == ',' || lc == '!' || lc == '?' || lc == '\n') vv = 3;
  else if (lc >= '0' && lc <= '9') vv = 4;
  else if (lc == 'y') vv = 5;
  else if (lc == '\'') vv = 6;
  else vv = (c & 32) ? 7 : 0;
  m->vc = (m->vc << 3) | (unsigned)vv;
  if (vv != m->lvc) { m->wc = (m->wc << 3) | (unsigned)vv; m->lvc = vv; }
  switch (c) {
    case ' ': m->qc = 0; break;
    case '(': m->ic += 31; break;
    case ')': m->ic -= 31; break;
    case '[': m->ic += 11; break;
    case ']': m->ic -= 11; break;
    case '<': m->ic += 23; m->qc += 34; break;
    case '>': m->ic -= 23; m->qc /= 5; break;
    case ':': m->pc = 20; break;
    case '{': m->ic += 17; break;
    case '}': m->ic -= 17; break;
    case '|': m->pc += 223; break;
    case '"': m->pc += 0x40; break;
    case '\'': m->pc += 0x42; if (c != (uint8_t)(c4 >> 8)) m->sense2 ^= 1; else m->ac += (2 * m->sense2 - 1); break;
    case '\n': m->pc = m->qc = 0; break;
    case '.': case '!': case '?': m->pc = 0; break;
    case '#': m->pc += 0x08; break;
    case '%': m->pc += 0x76; break;
    case '$': m->pc += 0x45; break;
    case '*': m->pc += 0x35; break;
    case '-': m->pc += 0x3; break;
    case '@': m->pc += 0x72; break;
    case '&': m->qc += 0x12; break;
    case ';': m->qc /= 3; break;
    case '\\': m->pc += 0x29; break;
    case '/': m->pc += 0x11; if (last[0] == '<') m->qc += 74; break;  /* buf(1) is the byte itself here: never '<' */
    case '=': m->pc += 87; if (c != (uint8_t)(c4 >> 8)) m->sense1 ^= 1; else m->ec += (2 * m->sense1 - 1); break;
    default: matched = 0;
  }
  if (c4 == 0x266C743B) m->uc = m->uc + 1 < 7 ? m->uc + 1 : 7;
  else if (c4 == 0x2667743B) m->uc -= (m->uc > 0);
  if (matched) m->bc = 0; else m->bc += 1;
  if (m->bc > 300) m->bc = m->ic = m->pc = m->qc = m->uc = 0;
  uint64_t i = 0;
  const unsigned vc = m->vc, wc = m->wc;
  const int ic = m->ic, pc = m->pc, qc = m->qc, bc = m->bc;
  ++i; cx[0] = H6(i, (vv > 0 && vv < 3) ? 0 : (lc | 0x100), ic & 0x3FF, m->ec & 0x7, m->ac & 0x7, m->uc);
  ++i; cx[1] = H4(i, ic, m->w, ilog2u((unsigned)(bc + 1)));
  ++i; cx[2] = H2(i, (3 * vc + 77 * pc + 373 * ic + qc) & 0xffff);
  ++i; cx[3] = H2(i, (31 * vc + 27 * pc + 281 * qc) & 0xffff);
  ++i; cx[4] = H2(i, (13 * vc + 271 * ic + qc + bc) & 0xffff);
  ++i; cx[5] = H2(i, (17 * pc + 7 * ic) & 0xffff);
  ++i; cx[6] = H2(i, (13 * vc + ic) & 0xffff);
  ++i; cx[7] = H2(i, (vc / 3 + pc) & 0xffff);
  ++i; cx[8] = H2(i, (7 * wc + qc) & 0xffff);
  ++i; cx[9] = H3(i, vc & 0xffff, f4 & 0xf);
  ++i; cx[10] = H3(i, (3 * pc) & 0xffff, f4 & 0xf);
  ++i; cx[11] = H3(i, ic & 0xffff, f4 & 0xf);
  return 12;
}

static int distance_contexts(P8CtxModel* m, uint32_t c4, int pos, uint64_t* cx) {
  const int c = c4 & 0xff;
  if (c == 0x00) m->pos00 = pos;
  if (c == 0x20) m->pos20 = pos;
  if (c == 0xff || c == '\r' || c == '\n') m->posnl = pos;
  const int a = pos - m->pos00 < 255 ? pos - m->pos00 : 255, b = pos - m->pos20 < 255 ? pos - m->pos20 : 255,
            d = pos - m->posnl < 255 ? pos - m->posnl : 255;
  cx[0] = H2(1, a | c << 8);
  cx[1] = H2(2, b | c << 8);
  cx[2] = H2(3, d | c << 8);
  return 3;
}

static int indirect_contexts(P8CtxModel* m, uint32_t c4, const uint8_t* last, uint64_t* cx) {
  uint32_t d = c4 & 0xffff, c = d & 255;
  const uint32_t d2 = (last[0] & 31) + 32 * (last[1] & 31) + 1024 * (last[2] & 31);
  const uint32_t d3 = (last[0] >> 3 & 31) + 32 * (last[2] >> 3 & 31) + 1024 * (last[3] >> 3 & 31);
  uint32_t* r1 = &m->t1[d >> 8]; *r1 = *r1 << 8 | c;
  uint16_t* r2 = &m->t2[c4 >> 8 & 0xffff]; *r2 = (uint16_t)(*r2 << 8 | c);
  uint16_t* r3 = &m->t3[(last[1] & 31) + 32 * (last[2] & 31) + 1024 * (last[3] & 31)]; *r3 = (uint16_t)(*r3 << 8 | c);
  uint16_t* r4 = &m->t4[(last[1] >> 3 & 31) + 32 * (last[3] >> 3 & 31) + 1024 * (last[4] >> 3 & 31)]; *r4 = (uint16_t)(*r4 << 8 | c);
  const uint32_t t = c | m->t1[c] << 8;
  const uint32_t t0 = d | (uint32_t)m->t2[d] << 16;
  const uint32_t ta = d2 | (uint32_t)m->t3[d2] << 16;
  const uint32_t tc = d3 | (uint32_t)m->t4[d3] << 16;
  const uint8_t pc = (uint8_t)tolower((uint8_t)(c4 >> 8));
  c = (uint32_t)tolower((int)c);
  m->ictx_data[m->ictx_cur] = (m->ictx_data[m->ictx_cur] << 8) | (c & 0xff);   /* iCtx += c  (:1484-1487) */
  m->ictx_cur = (((uint32_t)pc << 8) | c) & 0xffff;                              /* iCtx = (pc << 8) | c */
  const uint32_t ctx0 = m->ictx_data[m->ictx_cur];
  const uint32_t mask = ((uint8_t)m->t1[c] == (uint8_t)m->t2[d]) | (((uint8_t)m->t1[c] == (uint8_t)m->t3[d2]) << 1) |
                        (((uint8_t)m->t1[c] == (uint8_t)m->t4[d3]) << 2) | (((uint8_t)m->t1[c] == (uint8_t)ctx0) << 3);
  uint64_t i = 0;
  ++i; cx[0] = H2(i, t);
  ++i; cx[1] = H2(i, t0);
  ++i; cx[2] = H2(i, ta);
  ++i; cx[3] = H2(i, tc);
  ++i; cx[4] = H3(i, t & 0xff00, mask);
  ++i; cx[5] = H2(i, t0 & 0xff0000);
  ++i; cx[6] = H2(i, ta & 0xff0000);
  ++i; cx[7] = H2(i, tc & 0xff0000);
  ++i; cx[8] = H2(i, t & 0xffff);
  ++i; cx[9] = H2(i, t0 & 0xffffff);
  ++i; cx[10] = H2(i, ta & 0xffffff);
  ++i; cx[11] = H2(i, tc & 0xffffff);
  ++i; cx[12] = H3(i, ctx0 & 0xff, c);
  ++i; cx[13] = H2(i, ctx0 & 0xffff);
  ++i; cx[14] = H2(i, ctx0 & 0x7f7fff);
  return 15;
}

int p8f_ctxmodel_step(P8CtxModel* m, int y, int bpos, int c0, uint32_t c4, uint32_t f4, int pos, const uint8_t* last,
                         int16_t* out) {
  uint64_t cx[16];
  int nset = 0, nout = 0;
  if (bpos == 0)
    nset = m->which == 0 ? nest_contexts(m, c4, f4, last, cx) : m->which == 1 ? distance_contexts(m, c4, pos, cx)
                                                                              : indirect_contexts(m, c4, last, cx);
  p8f_cm_step(m->cm, y, bpos, c0, last[0], cx, nset, out, &nout);
  return nout;
}

/* ---- sparseModel (:4504-4535): 42 skip / masked contexts of the last bytes; sparseModel1 (:4539-4596): 31 contexts
 * mixing byte history with the word-level globals, plus seven SmallStationaryContextMaps. g[] = c4, f4, x4, w4, tt,
 * words, spaces, frstchar, spafdo (plain globals maintained elsewhere in paq8); last[i-1] = buf(i), i = 1..10. ---- */
typedef struct DMap DMap;
DMap* p8f_dmap_new(int kind, int bits_of_context, int bits_per_context, int rate);
void p8f_dmap_set_direct(DMap* m, uint32_t ctx);
void p8f_dmap_set_order(DMap* m);
int p8f_dmap_mix(DMap* m, int y, int a, int mul, int div, int16_t* out);

typedef struct { int which; CM1* cm; DMap* scm[7]; } P8Sparse;
P8Sparse* p8f_sparse_new(int which, int level) {
  P8Sparse* m = (P8Sparse*)calloc(1, sizeof *m);
  m->which = which;
  if (which == 0) m->cm = p8f_cm_new(mem_of(level) * 2, 40 + 2);
  else {
    m->cm = p8f_cm_new(mem_of(level) * 4, 31);
    static const int bits[7] = {7, 8, 4, 6, 4, 4, 7};  /* scm1 .. scm6, scma */
    for (int i = 0; i < 7; ++i) m->scm[i] = p8f_dmap_new(0, bits[i], 8, 0);
  }
  return m;
}
#define BUF(i) ((uint32_t)last[(i) - 1])
int p8f_sparse_step(P8Sparse* m, int y, int bpos, int c0, const uint32_t* g, int seenbefore, int howmany,
                       const uint8_t* last, int16_t* out) {
  const uint32_t c4 = g[0], f4 = g[1], x4 = g[2], w4 = g[3], tt = g[4], words = g[5], spaces = g[6], frstchar = g[7],
                 spafdo = g[8];
  uint64_t cx[48];
  int n = 0, nout = 0;
  if (bpos == 0 && m->which == 0) {
    uint64_t i = 0;
    ++i; cx[n++] = H2(i, seenbefore);
    ++i; p8f_cm_order_slot(m->cm, n, i); cx[n++] = 0;   /* hash(i, howmany): howmany is the order-N map's return value, known on the device only */
    ++i; cx[n++] = H2(i, BUF(1) | BUF(5) << 8);
    ++i; cx[n++] = H2(i, BUF(1) | BUF(6) << 8);
    ++i; cx[n++] = H2(i, BUF(3) | BUF(6) << 8);
    ++i; cx[n++] = H2(i, BUF(4) | BUF(8) << 8);
    ++i; cx[n++] = H2(i, BUF(1) | BUF(3) << 8 | BUF(5) << 16);
    ++i; cx[n++] = H2(i, BUF(2) | BUF(4) << 8 | BUF(6) << 16);
    static const uint32_t masks1[5] = {0x00f0f0ff, 0x00ff00ff, 0xff0000ff, 0x00f8f8f8, 0xf8f8f8f8};
    for (int k = 0; k < 5; ++k) { ++i; cx[n++] = H2(i, c4 & masks1[k]); }
    ++i; cx[n++] = H2(i, f4 & 0x00000fff);
    ++i; cx[n++] = H2(i, f4);
    static const uint32_t masks2[6] = {0x00e0e0e0, 0xe0e0e0e0, 0x810000c1, 0xC3CCC38C, 0x0081CC81, 0x00c10081};
    for (int k = 0; k < 6; ++k) { ++i; cx[n++] = H2(i, c4 & masks2[k]); }
    for (int j = 1; j < 8; ++j) {
      ++i; cx[n++] = H2(i, (uint32_t)seenbefore | BUF(j) << 8);
      ++i; cx[n++] = H2(i, (BUF(j + 2) << 8) | BUF(j + 1));
      ++i; cx[n++] = H2(i, (BUF(j + 3) << 8) | BUF(j + 1));
    }
  } else if (bpos == 0) {
    p8f_dmap_set_direct(m->scm[4], (uint32_t)seenbefore);  /* scm5 */
    p8f_dmap_set_order(m->scm[5]);                           /* scm6.set(howmany): device */
    uint32_t h = x4 << 6;
    cx[n++] = BUF(1) + (h & 0xffffff00);
    cx[n++] = BUF(1) + (h & 0x00ffff00);
    cx[n++] = BUF(1) + (h & 0x0000ff00);
    uint32_t d = c4 & 0xffff;
    h <<= 6;
    cx[n++] = d + (h & 0xffff0000);
    cx[n++] = d + (h & 0x00ff0000);
    h <<= 6; d = c4 & 0xffffff;
    cx[n++] = d + (h & 0xff000000);
    for (int i = 1; i < 5; ++i) {
      cx[n++] = (uint32_t)seenbefore | BUF(i) << 8;
      cx[n++] = (BUF(i + 3) << 8) | BUF(i + 1);
    }
    cx[n++] = spaces & 0x7fff;
    cx[n++] = spaces & 0xff;
    cx[n++] = words & 0x1ffff;
    cx[n++] = f4 & 0x000fffff;
    cx[n++] = tt & 0x00000fff;
    h = w4 << 6;
    cx[n++] = BUF(1) + (h & 0xffffff00);
    cx[n++] = BUF(1) + (h & 0x00ffff00);
    cx[n++] = BUF(1) + (h & 0x0000ff00);
    d = c4 & 0xffff;
    h <<= 6;
    cx[n++] = d + (h & 0xffff0000);
    cx[n++] = d + (h & 0x00ff0000);
    h <<= 6; d = c4 & 0xffffff;
    cx[n++] = d + (h & 0xff000000);
    cx[n++] = w4 & 0xf0f0f0ff;
    cx[n++] = (w4 & 63) * 128 + (5 << 17);
    cx[n++] = (f4 & 0xffff) << 11 | frstchar;
    cx[n++] = spafdo * 8 * ((w4 & 3) == 1);
    p8f_dmap_set_direct(m->scm[0], words & 127);
    p8f_dmap_set_direct(m->scm[1], (words & 12) * 16 + (w4 & 12) * 4 + (BUF(1) >> 4));
    p8f_dmap_set_direct(m->scm[2], w4 & 15);
    p8f_dmap_set_direct(m->scm[3], spafdo * ((w4 & 3) == 1));
    p8f_dmap_set_direct(m->scm[6], frstchar);
  }
  p8f_cm_step(m->cm, y, bpos, c0, last[0], cx, n, out, &nout);
  if (m->which == 1) {
    static const int order[7] = {0, 1, 2, 3, 4, 5, 6};  /* scm1, scm2, scm3, scm4, scm5, scm6, scma */
    for (int k = 0; k < 7; ++k) nout += p8f_dmap_mix(m->scm[order[k]], y, 7, 1, 4, out + nout);
  }
  return nout;
}


/* ---- picModel (:3844-3864): three bit-history contexts over the bits 215 / 431 / 647 bytes back (it runs on every
 * file type); recordModel1 (:4435-4474): five small ContextMaps over byte / word distances. hist[] = the reference's
 * ring buffer (bmask + 1 bytes), pos = bytes so far. ---- */
#include "p8f_tables.h"
typedef struct P8fPic P8fPic;
P8fPic* p8f_pic_new(void);
void p8f_pic_emit(P8fPic* p, int i, int cxt, int first, int16_t* out);
typedef struct {
  int which;
  uint32_t r0, r1, r2, r3;
  P8fPic* maps;   /* the bit-history bytes t[0x10200] and the three u16 StateMaps: device */
  int cxt[3], started;
  int cpos1[256], wpos1[0x10000];
  CM1 *cm, *cn, *co, *cp, *cq;
} P8Small;
P8Small* p8f_small_new(int which) {
  P8Small* m = (P8Small*)calloc(1, sizeof *m);
  m->which = which;
  if (which == 0) m->maps = p8f_pic_new();
  else {
    m->cm = p8f_cm_new(32768, 2); m->cn = p8f_cm_new(32768 / 2, 4 + 1); m->co = p8f_cm_new(32768 * 4, 4);
    m->cp = p8f_cm_new(32768 * 2, 3); m->cq = p8f_cm_new(32768 * 2, 3);
  }
  return m;
}
static int llog_u(uint32_t x) {  /* llog :268-275 */
  int p8f_ilog(int);
  if (x >= 0x1000000) return 256 + p8f_ilog((int)(x >> 16));
  if (x >= 0x10000) return 128 + p8f_ilog((int)(x >> 8));
  return p8f_ilog((int)x);
}
int p8f_small_step(P8Small* m, int y, int bpos, int c0, uint32_t c4, uint32_t f4, uint32_t w5, const uint8_t* hist,
                      uint32_t bmask, int pos, int16_t* out) {
#define RB(i) ((uint32_t)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
  int n = 0;
  if (m->which == 0) {
    m->r0 += m->r0 + (uint32_t)y;
    m->r1 += m->r1 + ((RB(215) >> (7 - bpos)) & 1);
    m->r2 += m->r2 + ((RB(431) >> (7 - bpos)) & 1);
    m->r3 += m->r3 + ((RB(647) >> (7 - bpos)) & 1);
    m->cxt[0] = (int)((m->r0 & 0x7) | ((m->r1 >> 4) & 0x38) | ((m->r2 >> 3) & 0xc0));
    m->cxt[1] = (int)(0x100 + ((m->r0 & 1) | ((m->r1 >> 4) & 0x3e) | ((m->r2 >> 2) & 0x40) | ((m->r3 >> 1) & 0x80)));
    m->cxt[2] = (int)(0x200 + ((m->r0 & 0x3f) ^ (m->r1 & 0x3ffe) ^ ((m->r2 << 2) & 0x7f00) ^ ((m->r3 << 5) & 0xf800)));
    /* t[old cxt] = nex(t[old cxt], y), then stretch(sm.p(t[cxt])) -- on the device, one lane per map. All three old
     * contexts are cell 0 at the very first step (the only time the maps share a cell): flagged so lane 0 applies it thrice */
    for (int i = 0; i < 3; ++i) p8f_pic_emit(m->maps, i, m->cxt[i], !m->started, out + n++);
    m->started = 1;
    return n;
  }
  uint64_t a[2], b[5], c3[4], d3[3], e3[3];
  int na = 0, nb = 0, nc = 0, nd = 0, ne = 0;
  if (bpos == 0) {
    const int w = c4 & 0xffff, c = w & 255, d = w & 0xf0ff, e = c4 & 0xffffff;
    const int dist = pos - m->cpos1[c];
    a[na++] = (uint64_t)(c << 8 | ((dist < 255 ? dist : 255) / 4));
    a[na++] = (uint64_t)(int64_t)(w << 9 | llog_u((uint32_t)(pos - m->wpos1[w])) >> 2);
    b[nb++] = (uint64_t)w; b[nb++] = (uint64_t)(d << 8); b[nb++] = (uint64_t)(c << 16); b[nb++] = f4 & 0xfffff;
    b[nb++] = (uint64_t)((pos & 3) | 2 << 12);
    c3[nc++] = (uint64_t)c; c3[nc++] = (uint64_t)(w << 8); c3[nc++] = w5 & 0x3ffff; c3[nc++] = (uint64_t)(int64_t)(e << 3);
    d3[nd++] = (uint64_t)d; d3[nd++] = (uint64_t)(c << 8); d3[nd++] = (uint64_t)(int64_t)(w << 16);
    e3[ne++] = (uint64_t)(w << 3); e3[ne++] = (uint64_t)(c << 19); e3[ne++] = (uint64_t)e;
    m->cpos1[c] = pos;
    m->wpos1[w] = pos;
  }
  int k = 0;
  const int c1 = (int)RB(1);
  p8f_cm_step(m->cm, y, bpos, c0, c1, a, na, out + n, &k); n += k;
  p8f_cm_step(m->cn, y, bpos, c0, c1, b, nb, out + n, &k); n += k;
  p8f_cm_step(m->co, y, bpos, c0, c1, c3, nc, out + n, &k); n += k;
  p8f_cm_step(m->cq, y, bpos, c0, c1, e3, ne, out + n, &k); n += k;
  p8f_cm_step(m->cp, y, bpos, c0, c1, d3, nd, out + n, &k); n += k;
  return n;
#undef RB
}

