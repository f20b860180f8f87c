/* p8front/p8f_record.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's recordModel (reference src/models/paq8.cpp:4204-4433): detection of a fixed record length
 * from byte-recurrence distances (two candidates with counters, dBASE headers), column / row-neighbour contexts into
 * four ContextMaps, six StationaryMaps, three IndirectMaps, three SmallStationaryContextMaps, five IndirectContexts,
 * and three mixer weight-set selectors. It runs on every block type, text included. Pinned against the reference's own
 * function in tests/test_oracle_paq8core.py. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct CM1 CM1;
CM1* p8f_cm_new(uint64_t size_bytes, int count);
int p8f_cm_step(CM1* c, int y1, int bp, int c0, int c1, const uint64_t* ctx, int nset, int16_t* out, int* nout);
typedef struct DMap DMap;
DMap* p8f_dmap_new(int kind, int bits_of_context, int bits_per_context, int rate);
void p8f_dmap_set_direct(DMap* m, uint32_t ctx);
int p8f_dmap_mix(DMap* m, int y, int a, int mul, int div, int16_t* out);
int p8f_ilog(int x);
uint32_t p8f_finalize64(uint64_t h, int bits);
uint64_t p8f_hash2(uint64_t a, uint64_t b);
uint64_t p8f_hash3(uint64_t a, uint64_t b, uint64_t c);
uint64_t p8f_hash5(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e);
uint64_t p8f_hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d);

static unsigned ilog2u(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }
static int llog_u(uint32_t x) {
  if (x >= 0x1000000) return 256 + p8f_ilog((int)(x >> 16));
  if (x >= 0x10000) return 128 + p8f_ilog((int)(x >> 8));
  return p8f_ilog((int)x);
}
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }
static uint8_t clip8(int v) { return (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }

typedef struct { uint16_t* data; uint32_t cur, mask; int inbits; } ICtx16;  /* IndirectContext<U16> :1470-1494 */
static void ic_init(ICtx16* c, int bits, int inbits) { c->data = (uint16_t*)calloc((size_t)1 << bits, 2); c->mask = (1u << bits) - 1; c->inbits = inbits; c->cur = 0; }
static void ic_add(ICtx16* c, uint32_t v) { c->data[c->cur] = (uint16_t)((c->data[c->cur] << c->inbits) | (v & ((1u << c->inbits) - 1))); }
static void ic_set(ICtx16* c, uint32_t v) { c->cur = v & c->mask; }
static uint32_t ic_get(const ICtx16* c) { return c->data[c->cur]; }

typedef struct {
  int cpos1[256], cpos2[256], cpos3[256], cpos4[256], wpos1[0x10000];
  int rlen[3], rcount[2];
  uint8_t padding, N, NN, NNN, NNNN, WxNW;
  int prevTransition, nTransition, col, mxCtx, x, maybe24;
  CM1 *cm, *cn, *co, *cp;
  DMap *maps[6], *smap[3], *imap[3];
  struct { uint8_t Version; uint32_t nRecords; uint16_t RecordLength, HeaderLength; int Start, End; } dbase;
  ICtx16 ic[5];
} Record;

Record* p8f_record_new(int level) {
  Record* r = (Record*)calloc(1, sizeof *r);
  r->rlen[0] = 2; r->rlen[1] = 3; r->rlen[2] = 4;
  r->cm = p8f_cm_new(32768, 3); r->cn = p8f_cm_new(32768 / 2, 3); r->co = p8f_cm_new(32768 * 2, 3);
  r->cp = p8f_cm_new(0x10000ull << level, 16);
  static const int mb[6][2] = {{10, 8}, {10, 8}, {8, 8}, {8, 8}, {8, 8}, {11, 1}};
  for (int i = 0; i < 6; ++i) r->maps[i] = p8f_dmap_new(1, mb[i][0], mb[i][1], 0);
  r->smap[0] = p8f_dmap_new(0, 11, 1, 0); r->smap[1] = p8f_dmap_new(0, 3, 1, 0); r->smap[2] = p8f_dmap_new(0, 19, 1, 0);
  for (int i = 0; i < 3; ++i) r->imap[i] = p8f_dmap_new(2, 8, 8, 0);
  ic_init(&r->ic[0], 16, 8); ic_init(&r->ic[1], 16, 8); ic_init(&r->ic[2], 16, 8); ic_init(&r->ic[3], 20, 8); ic_init(&r->ic[4], 11, 1);
  return r;
}

/* io: [0] blpos, [1] grp0, [2] filetype (0 DEFAULT, 4 TEXT), [3] Stats.Record (in/out), [4] Match.length, [5] Match.expectedByte */
int p8f_record_step(Record* r, int y, int bpos, int c0, uint32_t c4, uint32_t* io, const uint8_t* hist, uint32_t bmask,
                       int pos, int16_t* out, int* sets) {
#define RB(i) ((int)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
  const int blpos = (int)io[0], grp0 = (int)io[1], filetype = (int)io[2];
  uint64_t a[3], b[3], c3[3], d[16];
  int na = 0, nb = 0, nc = 0, nd = 0;
  if (!bpos) {
    const int w = c4 & 0xffff, c = w & 255, dd = w >> 8;
    if (io[3] && (io[3] >> 16) != (uint32_t)r->rlen[0]) {
      r->rlen[0] = (int)(io[3] >> 16);
      r->rcount[0] = r->rcount[1] = 0;
    } else {
      if (blpos == 0 || (r->dbase.Version > 0 && blpos >= r->dbase.End)) r->dbase.Version = 0;
      else if (r->dbase.Version == 0 && (filetype == 0 || filetype == 4) && blpos >= 31) {
        uint8_t bb = (uint8_t)RB(32);
        int ok = ((bb & 7) == 3 || (bb & 7) == 4 || (bb >> 4) == 3 || bb == 0xF5);
        ok = ok && ((bb = (uint8_t)RB(30)) > 0 && bb < 13);
        ok = ok && ((bb = (uint8_t)RB(29)) > 0 && bb < 32);
        ok = ok && ((r->dbase.nRecords = (uint32_t)(RB(28) | (RB(27) << 8) | (RB(26) << 16) | (RB(25) << 24))) > 0 && r->dbase.nRecords < 0xFFFFF);
        if (ok) {
          r->dbase.HeaderLength = (uint16_t)(RB(24) | (RB(23) << 8));
          ok = r->dbase.HeaderLength > 32 &&
               (((r->dbase.HeaderLength - 32 - 1) % 32) == 0 ||
                (r->dbase.HeaderLength > 255 + 8 && (((r->dbase.HeaderLength = (uint16_t)(r->dbase.HeaderLength - (255 + 8))) - 32 - 1) % 32) == 0));
        }
        ok = ok && ((r->dbase.RecordLength = (uint16_t)(RB(22) | (RB(21) << 8))) > 8);
        ok = ok && (RB(20) == 0 && RB(19) == 0 && RB(17) <= 1 && RB(16) <= 1);
        if (ok) {
          bb = (uint8_t)RB(32);
          r->dbase.Version = (uint8_t)(((bb >> 4) == 3) ? 3 : bb & 7);
          r->dbase.Start = blpos - 32 + r->dbase.HeaderLength;
          r->dbase.End = r->dbase.Start + (int)(r->dbase.nRecords * r->dbase.RecordLength);
          if (r->dbase.Version == 3) { r->rlen[0] = 32; r->rcount[0] = r->rcount[1] = 0; }
        }
      } else if (r->dbase.Version > 0 && blpos == r->dbase.Start) {
        r->rlen[0] = r->dbase.RecordLength;
        r->rcount[0] = r->rcount[1] = 0;
      }
      const int rr = pos - r->cpos1[c];
      if (rr > 1 && rr == r->cpos1[c] - r->cpos2[c] && rr == r->cpos2[c] - r->cpos3[c] && (rr > 32 || rr == r->cpos3[c] - r->cpos4[c]) &&
          (rr > 10 || ((c == RB(rr * 5 + 1)) && c == RB(rr * 6 + 1)))) {
        if (rr == r->rlen[1]) ++r->rcount[0];
        else if (rr == r->rlen[2]) ++r->rcount[1];
        else if (r->rcount[0] > r->rcount[1]) { r->rlen[2] = rr; r->rcount[1] = 1; }
        else { r->rlen[1] = rr; r->rcount[0] = 1; }
      }
      for (int i = 0; i < 2; i++) {
        if (r->rcount[i] > imax(0, 12 - (int)ilog2u((unsigned)r->rlen[i + 1]))) {
          if (r->rlen[0] != r->rlen[i + 1]) {
            if (r->maybe24 && r->rlen[i + 1] == 3) { r->rcount[0] >>= 1; r->rcount[1] >>= 1; continue; }
            else if ((r->rlen[i + 1] > r->rlen[0]) && (r->rlen[i + 1] % r->rlen[0] == 0)) {
              if ((r->rlen[0] > 32) && (r->rlen[i + 1] == r->rlen[0] * 2)) { r->rcount[0] >>= 1; r->rcount[1] >>= 1; continue; }
            }
            r->rlen[0] = r->rlen[i + 1];
            r->rcount[i] = 0;
            r->maybe24 = (r->rlen[0] > 30 && (r->rlen[0] % 3) == 0);
            r->nTransition = 0;
          } else r->rcount[i] >>= 2;
          if (r->rlen[i + 1] << 4 > r->rlen[1 + (i ^ 1)]) r->rcount[i ^ 1] = 0;
        }
      }
    }
    const int R = r->rlen[0];
    r->col = pos % R;
    r->x = imin(0x1F, r->col / imax(1, R / 32));
    r->N = (uint8_t)RB(R); r->NN = (uint8_t)RB(R * 2); r->NNN = (uint8_t)RB(R * 3); r->NNNN = (uint8_t)RB(R * 4);
    for (int i = 0; i < 4; i++) ic_add(&r->ic[i], (uint32_t)c);
    ic_set(&r->ic[0], (uint32_t)((c << 8) | r->N));
    ic_set(&r->ic[1], (uint32_t)((RB(R - 1) << 8) | r->N));
    ic_set(&r->ic[2], (uint32_t)((c << 8) | RB(R - 1)));
    ic_set(&r->ic[3], p8f_finalize64(p8f_hash3((uint64_t)c, r->N, (uint64_t)RB(R + 1)), 20));
    if (!r->col) r->nTransition = 0;
    if ((((c4 >> 8) == 0x20u * 0x010101) && (c != 0x20)) ||
        (!(c4 >> 8) && c && ((r->padding != 0x20) || (pos - r->prevTransition > R)))) {
      r->prevTransition = pos;
      r->nTransition += (r->nTransition < 31);
      r->padding = (uint8_t)dd;
    }
    uint64_t i = 0;
    const int N = r->N, NN = r->NN, NNN = r->NNN, NNNN = r->NNNN, col = r->col;
    ++i; a[na++] = p8f_hash2(i, (uint64_t)(int64_t)(c << 8 | (imin(255, pos - r->cpos1[c]) >> 2)));
    ++i; a[na++] = p8f_hash2(i, (uint64_t)(int64_t)(w << 9 | llog_u((uint32_t)(pos - r->wpos1[w])) >> 2));
    ++i; a[na++] = p8f_hash2(i, (uint64_t)(int64_t)(R | N << 10 | NN << 18));
    ++i; b[nb++] = p8f_hash2(i, (uint64_t)(int64_t)(w | R << 16));
    ++i; b[nb++] = p8f_hash2(i, (uint64_t)(int64_t)(dd | R << 8));
    ++i; b[nb++] = p8f_hash2(i, (uint64_t)(int64_t)(c | R << 8));
    ++i; c3[nc++] = p8f_hash2(i, (uint64_t)(int64_t)(c << 8 | imin(255, pos - r->cpos1[c])));
    ++i; c3[nc++] = p8f_hash2(i, (uint64_t)(int64_t)(c << 17 | dd << 9 | llog_u((uint32_t)(pos - r->wpos1[w])) >> 2));
    ++i; c3[nc++] = p8f_hash2(i, (uint64_t)(int64_t)(c << 8 | N));
    ++i; d[nd++] = p8f_hash2(i, (uint64_t)(int64_t)(R | N << 10 | col << 18));
    ++i; d[nd++] = p8f_hash2(i, (uint64_t)(int64_t)(R | c << 10 | col << 18));
    ++i; d[nd++] = p8f_hash2(i, (uint64_t)(int64_t)(col | R << 12));
    if (R > 8) {
      ++i; d[nd++] = p8f_hash5(i, (uint64_t)(int64_t)imin(imin(0xFF, R), pos - r->prevTransition), (uint64_t)(int64_t)imin(0x3FF, col),
                                  (uint64_t)(int64_t)((w & 0xF0F0) | (w == ((r->padding << 8) | r->padding))), (uint64_t)(int64_t)r->nTransition);
      ++i; d[nd++] = p8f_hash4(i, (uint64_t)(int64_t)w, (uint64_t)(RB(R + 1) == r->padding && N == r->padding), (uint64_t)(int64_t)(col / imax(1, R / 32)));
    } else { d[nd++] = 0; d[nd++] = 0; }
    ++i; d[nd++] = p8f_hash2(i, (uint64_t)(int64_t)(N | ((NN & 0xF0) << 4) | ((NNN & 0xE0) << 7) | ((NNNN & 0xE0) << 10) | ((col / imax(1, R / 16)) << 18)));
    ++i; d[nd++] = p8f_hash2(i, (uint64_t)(int64_t)((N & 0xF8) | ((NN & 0xF8) << 8) | (col << 16)));
    ++i; d[nd++] = p8f_hash3(i, (uint64_t)N, (uint64_t)NN);
    ++i; d[nd++] = p8f_hash3(i, (uint64_t)(int64_t)col, ic_get(&r->ic[0]));
    ++i; d[nd++] = p8f_hash3(i, (uint64_t)(int64_t)col, ic_get(&r->ic[1]));
    ++i; d[nd++] = p8f_hash4(i, (uint64_t)(int64_t)col, ic_get(&r->ic[0]) & 0xFF, ic_get(&r->ic[1]) & 0xFF);
    ++i; d[nd++] = p8f_hash2(i, ic_get(&r->ic[2]));
    ++i; d[nd++] = p8f_hash2(i, ic_get(&r->ic[3]));
    ++i; d[nd++] = p8f_hash3(i, ic_get(&r->ic[1]) & 0xFF, ic_get(&r->ic[3]) & 0xFF);
    r->WxNW = (uint8_t)(c ^ RB(R + 1));
    ++i; d[nd++] = p8f_hash3(i, (uint64_t)N, (uint64_t)r->WxNW);
    ++i; d[nd++] = p8f_hash4(i, io[4] > 0 ? (uint64_t)io[5] : (uint64_t)(0x100 | (uint8_t)ic_get(&r->ic[1])), (uint64_t)N, (uint64_t)r->WxNW);
    int k = 0x300;
    if (r->maybe24) {
      k = (col % 3) << 8;
      p8f_dmap_set_direct(r->maps[0], (uint32_t)(clip8((int)((uint8_t)(c4 >> 16)) + c - (int)(c4 >> 24)) | k));
    } else p8f_dmap_set_direct(r->maps[0], (uint32_t)(clip8(c * 2 - dd) | k));
    p8f_dmap_set_direct(r->maps[1], (uint32_t)(clip8(c + N - RB(R + 1)) | k));
    p8f_dmap_set_direct(r->maps[2], clip8(N + NN - NNN));
    p8f_dmap_set_direct(r->maps[3], clip8(N * 2 - NN));
    p8f_dmap_set_direct(r->maps[4], clip8(N * 3 - NN * 3 + NNN));
    p8f_dmap_set_direct(r->imap[0], (uint32_t)(N + NN - NNN));
    p8f_dmap_set_direct(r->imap[1], (uint32_t)(N * 2 - NN));
    p8f_dmap_set_direct(r->imap[2], (uint32_t)(N * 3 - NN * 3 + NNN));
    r->cpos4[c] = r->cpos3[c]; r->cpos3[c] = r->cpos2[c]; r->cpos2[c] = r->cpos1[c]; r->cpos1[c] = pos;
    r->wpos1[w] = pos;
    r->mxCtx = (R > 128) ? imin(0x7F, col / imax(1, R / 128)) : col;
  }
  const uint8_t B = (uint8_t)(c0 << (8 - bpos));
  const uint32_t ctx = (uint32_t)(r->N ^ B) | ((uint32_t)bpos << 8);
  ic_add(&r->ic[4], (uint32_t)y);
  ic_set(&r->ic[4], ctx);
  p8f_dmap_set_direct(r->maps[5], ctx);
  p8f_dmap_set_direct(r->smap[0], ctx);
  p8f_dmap_set_direct(r->smap[1], ic_get(&r->ic[4]));
  p8f_dmap_set_direct(r->smap[2], (ctx << 8) | r->WxNW);
  int n = 0, k = 0;
  const int c1 = RB(1);
  p8f_cm_step(r->cm, y, bpos, c0, c1, a, na, out + n, &k); n += k;
  p8f_cm_step(r->cn, y, bpos, c0, c1, b, nb, out + n, &k); n += k;
  p8f_cm_step(r->co, y, bpos, c0, c1, c3, nc, out + n, &k); n += k;
  p8f_cm_step(r->cp, y, bpos, c0, c1, d, nd, out + n, &k); n += k;
  for (int i = 0; i < 6; i++) n += p8f_dmap_mix(r->maps[i], y, 1023, 1, 3, out + n);
  for (int i = 0; i < 3; i++) n += p8f_dmap_mix(r->imap[i], y, 255, 1, 3, out + n);
  n += p8f_dmap_mix(r->smap[0], y, 6, 1, 3, out + n);
  n += p8f_dmap_mix(r->smap[1], y, 6, 1, 3, out + n);
  n += p8f_dmap_mix(r->smap[2], y, 5, 1, 2, out + n);
  sets[0] = (r->rlen[0] > 2) * ((bpos << 7) | r->mxCtx);
  sets[1] = 1024 + (((r->N ^ B) >> 4) | (r->x << 4));
  sets[2] = 1024 + 512 + ((grp0 << 5) | r->x);
  io[3] = ((uint32_t)imin(0xFFFF, r->rlen[0]) << 16) | (uint32_t)imin(0xFFFF, r->col);
  return n;
#undef RB
}
