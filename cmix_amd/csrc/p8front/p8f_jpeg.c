/* p8front/p8f_jpeg.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's JPEG model (reference src/models/paq8.cpp:5871-6597, jpegModel): the marker parser (SOI, APPx / COM skipping with
 * embedded thumbnails, DQT, DHT, SOF0/1, SOS, RSTx, EOI), the Huffman decoder that follows the coded bits one by one (user or default tables),
 * the rotating buffers of decoded coefficients, the neighbourhood predictors of the coefficient being coded (adv_pred / run_pred / lcp / prev_coef,
 * from the blocks above and to the left, the other components and the quantisation tables) and the 32 context hashes built from them every third
 * bit of a Huffman code. Everything that learns -- the BH<9> table of bit histories, 32 StateMaps, the model's own 33-input mixer, two APM
 * stages -- is the device's (p8stage_dev.h P8L_JPG); this file hands it the hashed contexts, the selectors and the APM contexts per step.
 * A step of the model is one of: silent (the parser is between markers: no input at all), stuffed (a 0xFF byte: one constant input),
 * restart (the restart marker's known bits: one constant input), coded (64 + 6 inputs; the 33 inputs and 3 outputs of the model's own mixer
 * are exported in between, in call order).
 * Parity: tests/test_p8stage_host.py (stage vs per-step hashes of the unmodified reference's 1591 values on a JPEG stream). */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct P8fJpg P8fJpg;
P8fJpg* p8f_jpg_new(uint64_t table_items);
int p8f_jpg_step(P8fJpg* j, int hbcount, int hc_low, const uint64_t* cxt, const int* m1sel, int a1ctx, int a2ctx, int16_t* out);
int p8f_ilog(int x);

#define PHI64 0x9E3779B97F4A7C15ull
static const uint64_t MUL[8] = {PHI64, 0x993DDEFFB1462949ull, 0xE9C91DC159AB0D2Dull, 0x83D6A14F1B0CED73ull,
                                0xA14F1B0CED5A841Full, 0xC0E51314A614F4EFull, 0xDA9CC2600AE45A27ull, 0x826797AA04A65737ull};
static uint64_t hashn(int n, const int64_t* x) {  /* hash(x0 .. x(n-1)) :742-773; int arguments widen with their sign */
  uint64_t h = 0;
  for (int i = 0; i < n; ++i) h += ((uint64_t)x[i] + 1) * MUL[i];
  return h;
}
#define HN(n, ...) hashn((n), (const int64_t[]){__VA_ARGS__})
static int imin(int a, int b) { return a < b ? a : b; }
static unsigned ilog2u(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }

enum { SOF0 = 0xc0, DHT = 0xc4, RST0 = 0xd0, SOI = 0xd8, EOI = 0xd9, SOS = 0xda, DQT = 0xdb, FF = 0xff };
typedef struct { uint32_t min, max; int val; } Huf;
typedef struct { int offset, jpeg, next_jpeg, app, sof, sos, data, htsize; int ht[8]; uint8_t qtab[256]; int qmap[10]; } JImage;

typedef struct Jpeg {
  P8fJpg* dev;
  JImage images[4];
  int idx, last_pos;
  uint32_t huffcode; int huffbits, huffsize, rs, mcupos;
  Huf huf[128];
  int mcusize, hufsel[2][10];
  uint8_t hbuf[2048];
  int color[10], pred[4], dc, width, row, column;
  uint8_t cbuf[0x20000];
  int cpos, rs1, rstpos, rstlen, ssum, ssum1, ssum2, ssum3;
  int cbuf2[0x20000];
  int adv_pred[4], sumu[8], sumv[8], run_pred[6];
  int prev_coef, prev_coef2, prev_coef_rs;
  int ls[10], blockW[10], blockN[10], sampling[4], lcp[7], zpos[64];
  int dqt_state, dqt_end, qnum;
  int hbcount;
  uint64_t cxt[32];
} Jpeg;

static const uint8_t zzu[64] = {0, 1, 0, 0, 1, 2, 3, 2, 1, 0, 0, 1, 2, 3, 4, 5, 4, 3, 2, 1, 0, 0, 1, 2, 3, 4, 5, 6, 7, 6, 5, 4,
                                3, 2, 1, 0, 1, 2, 3, 4, 5, 6, 7, 7, 6, 5, 4, 3, 2, 3, 4, 5, 6, 7, 7, 6, 5, 4, 5, 6, 7, 7, 6, 7};
static const uint8_t zzv[64] = {0, 0, 1, 2, 1, 0, 0, 1, 2, 3, 4, 3, 2, 1, 0, 0, 1, 2, 3, 4, 5, 6, 5, 4, 3, 2, 1, 0, 0, 1, 2, 3,
                                4, 5, 6, 7, 7, 6, 5, 4, 3, 2, 1, 2, 3, 4, 5, 6, 7, 7, 6, 5, 4, 3, 4, 5, 6, 7, 7, 6, 5, 6, 7, 7};
/* the standard Huffman tables (JPEG annex K.3), used when a stream brings none (:5981-6046) */
static const uint8_t bits_dc_lum[16] = {0, 1, 5, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0, 0, 0}, bits_dc_chr[16] = {0, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0, 0};
static const uint8_t val_dc[12] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11};
static const uint8_t bits_ac_lum[16] = {0, 2, 1, 3, 3, 2, 4, 3, 5, 5, 4, 4, 0, 0, 1, 0x7d}, bits_ac_chr[16] = {0, 2, 1, 2, 4, 4, 3, 4, 7, 5, 4, 4, 0, 1, 2, 0x77};
static const uint8_t val_ac_lum[162] = {
    0x01, 0x02, 0x03, 0x00, 0x04, 0x11, 0x05, 0x12, 0x21, 0x31, 0x41, 0x06, 0x13, 0x51, 0x61, 0x07, 0x22, 0x71, 0x14, 0x32, 0x81, 0x91, 0xa1, 0x08, 0x23, 0x42, 0xb1,
    0xc1, 0x15, 0x52, 0xd1, 0xf0, 0x24, 0x33, 0x62, 0x72, 0x82, 0x09, 0x0a, 0x16, 0x17, 0x18, 0x19, 0x1a, 0x25, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x34, 0x35, 0x36, 0x37,
    0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69, 0x6a,
    0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a, 0xa2, 0xa3,
    0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca, 0xd2, 0xd3,
    0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe1, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf1, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};
static const uint8_t val_ac_chr[162] = {
    0x00, 0x01, 0x02, 0x03, 0x11, 0x04, 0x05, 0x21, 0x31, 0x06, 0x12, 0x41, 0x51, 0x07, 0x61, 0x71, 0x13, 0x22, 0x32, 0x81, 0x08, 0x14, 0x42, 0x91, 0xa1, 0xb1, 0xc1,
    0x09, 0x23, 0x33, 0x52, 0xf0, 0x15, 0x62, 0x72, 0xd1, 0x0a, 0x16, 0x24, 0x34, 0xe1, 0x25, 0xf1, 0x17, 0x18, 0x19, 0x1a, 0x26, 0x27, 0x28, 0x29, 0x2a, 0x35, 0x36,
    0x37, 0x38, 0x39, 0x3a, 0x43, 0x44, 0x45, 0x46, 0x47, 0x48, 0x49, 0x4a, 0x53, 0x54, 0x55, 0x56, 0x57, 0x58, 0x59, 0x5a, 0x63, 0x64, 0x65, 0x66, 0x67, 0x68, 0x69,
    0x6a, 0x73, 0x74, 0x75, 0x76, 0x77, 0x78, 0x79, 0x7a, 0x82, 0x83, 0x84, 0x85, 0x86, 0x87, 0x88, 0x89, 0x8a, 0x92, 0x93, 0x94, 0x95, 0x96, 0x97, 0x98, 0x99, 0x9a,
    0xa2, 0xa3, 0xa4, 0xa5, 0xa6, 0xa7, 0xa8, 0xa9, 0xaa, 0xb2, 0xb3, 0xb4, 0xb5, 0xb6, 0xb7, 0xb8, 0xb9, 0xba, 0xc2, 0xc3, 0xc4, 0xc5, 0xc6, 0xc7, 0xc8, 0xc9, 0xca,
    0xd2, 0xd3, 0xd4, 0xd5, 0xd6, 0xd7, 0xd8, 0xd9, 0xda, 0xe2, 0xe3, 0xe4, 0xe5, 0xe6, 0xe7, 0xe8, 0xe9, 0xea, 0xf2, 0xf3, 0xf4, 0xf5, 0xf6, 0xf7, 0xf8, 0xf9, 0xfa};

Jpeg* p8f_jpeg_new(int level) {
  Jpeg* j = (Jpeg*)calloc(1, sizeof *j);
  j->dev = p8f_jpg_new(0x10000ull << level);   /* BH<9> t(MEM()) :6484 */
  j->idx = -1; j->rs = -1; j->dqt_state = -1; j->hbcount = 2;
  return j;
}

/* finish(success) :5880-5888 */
static void jfinish(Jpeg* j, int pos) {
  const int length = pos - j->images[j->idx].offset;
  memset(&j->images[j->idx], 0, sizeof(JImage));
  j->mcusize = 0; j->dqt_state = -1;
  j->idx -= (j->idx > 0);
  j->images[j->idx].app -= length;
  if (j->images[j->idx].app < 0) j->images[j->idx].app = 0;
}
/* jassert(x) :5890-5895: a failed check leaves the model (an embedded image is dropped, the outermost one switched off) */
#define JASSERT(x) do { if (!(x)) { if (j->idx > 0) jfinish(j, pos); else j->images[j->idx].jpeg = 0; *kind = 0; return j->images[j->idx].next_jpeg; } } while (0)

/* One step of jpegModel. hist / bmask / pos: the byte history (buf(i) = hist[(pos - i) & bmask], buf[p] = hist[p & bmask]).
 * Returns jpegModel's return value (non-zero: the step is the JPEG model's). *kind: 0 silent (no input), 1 stuffed byte / 2 restart marker (one input, in
 * out[0]), 3 coded (70 inputs through the device's maps). sets[3] / ranges[3] for kinds 1..3. */
int p8f_jpeg_step(Jpeg* j, int y, int bpos, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* sets, int* ranges, int* kind) {
#define BUF(i) ((int)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
#define BAT(p) ((int)hist[(uint32_t)(p) & bmask])
  *kind = 0;
  if (j->idx < 0) { memset(j->images, 0, sizeof j->images); j->idx = 0; j->last_pos = pos; }
  JImage* im = &j->images[j->idx];
  if (!bpos) im->next_jpeg = im->jpeg > 1;
  if (bpos && !im->jpeg) return im->next_jpeg;
  if (!bpos && im->app > 0) {
    --im->app;
    if (j->idx < 3 && BUF(4) == FF && BUF(3) == SOI && BUF(2) == FF && ((BUF(1) & 0xFE) == 0xC0 || BUF(1) == 0xC4 || (BUF(1) >= 0xDB && BUF(1) <= 0xFE))) {
      memset(&j->images[++j->idx], 0, sizeof(JImage));
      im = &j->images[j->idx];
    }
  }
  if (im->app > 0) return im->next_jpeg;
  if (!bpos) {
    if (!im->jpeg && BUF(4) == FF && BUF(3) == SOI && BUF(2) == FF && ((BUF(1) & 0xFE) == 0xC0 || BUF(1) == 0xC4 || (BUF(1) >= 0xDB && BUF(1) <= 0xFE))) {   /* SOI + a valid marker :6098-6107 */
      im->jpeg = 1;
      im->offset = pos - 4;
      im->sos = im->sof = im->htsize = im->data = 0; im->app = (BUF(1) >> 4 == 0xE) * 2;
      j->mcusize = 0; j->huffcode = 0; j->huffbits = j->huffsize = j->mcupos = j->cpos = 0; j->rs = -1;
      memset(j->huf, 0, sizeof j->huf);
      memset(j->pred, 0, sizeof j->pred);
      j->rstpos = j->rstlen = 0;
    }
    if (im->jpeg && im->data && ((BUF(2) == FF && BUF(1) && (BUF(1) & 0xf8) != RST0) || (pos - j->last_pos > 1))) {   /* a marker other than RSTx inside the data: the end :6111-6114 */
      JASSERT((BUF(1) == EOI) || (pos - j->last_pos > 1));
      jfinish(j, pos);
      im = &j->images[j->idx];
    }
    j->last_pos = pos;
    if (!im->jpeg) return im->next_jpeg;
    if (!im->data && !im->app && BUF(4) == FF && (((BUF(3) > 0xC1) && (BUF(3) <= 0xCF) && (BUF(3) != DHT)) || ((BUF(3) >= 0xDC) && (BUF(3) <= 0xFE)))) {   /* a segment to skip */
      im->app = BUF(2) * 256 + BUF(1) + 2;
      if (j->idx > 0) JASSERT(pos + im->app < im->offset + j->images[j->idx - 1].app);
    }
    if (BUF(5) == FF && BUF(4) == SOS) {
      const int len = BUF(3) * 256 + BUF(2);
      if (len == 6 + 2 * BUF(1) && BUF(1) && BUF(1) <= 4) { im->sos = pos - 5; im->data = im->sos + len + 2; im->jpeg = 2; }
    }
    if (BUF(4) == FF && BUF(3) == DHT && im->htsize < 8) im->ht[im->htsize++] = pos - 4;
    if (BUF(4) == FF && (BUF(3) & 0xFE) == SOF0) im->sof = pos - 4;
    if (BUF(4) == FF && BUF(3) == DQT) { j->dqt_end = pos + BUF(2) * 256 + BUF(1) - 1; j->dqt_state = 0; }   /* quantisation tables :6136-6151 */
    else if (j->dqt_state >= 0) {
      if (pos >= j->dqt_end) j->dqt_state = -1;
      else {
        if (j->dqt_state % 65 == 0) j->qnum = BUF(1);
        else {
          JASSERT(BUF(1) > 0);
          JASSERT(j->qnum >= 0 && j->qnum < 4);
          im->qtab[j->qnum * 64 + ((j->dqt_state % 65) - 1)] = (uint8_t)(BUF(1) - 1);
        }
        j->dqt_state++;
      }
    }
    if (BUF(2) == FF && (BUF(1) & 0xf8) == RST0) {   /* restart :6154-6159 */
      j->huffcode = 0; j->huffbits = j->huffsize = j->mcupos = 0; j->rs = -1;
      memset(j->pred, 0, sizeof j->pred);
      j->rstlen = j->column + j->row * j->width - j->rstpos;
      j->rstpos = j->column + j->row * j->width;
    }
  }
  /* the first data bit: build the Huffman tables, the MCU layout, the block geometry :6162-6297 */
  if (pos == im->data && bpos == 1) {
    int i;
    for (i = 0; i < im->htsize; ++i) {
      int p = im->ht[i] + 4;
      const int end = p + BAT(p - 2) * 256 + BAT(p - 1) - 2;
      int count = 0;
      while (p < end && end < pos && end < p + 2100 && ++count < 10) {
        const int tc = BAT(p) >> 4, th = BAT(p) & 15;
        if (tc >= 2 || th >= 4) break;
        Huf* h = &j->huf[tc * 64 + th * 16];
        int val = p + 17, hval = tc * 1024 + th * 256, k;
        for (k = 0; k < 256; ++k) j->hbuf[hval + k] = (uint8_t)BAT(val + k);
        int code = 0;
        for (k = 0; k < 16; ++k) {
          h[k].min = (uint32_t)code;
          h[k].max = (uint32_t)(code += BAT(p + k + 1));
          h[k].val = hval;
          val += BAT(p + k + 1);
          hval += BAT(p + k + 1);
          code *= 2;
        }
        p = val;
        JASSERT(hval >= 0 && hval < 2048);
      }
      JASSERT(p == end);
    }
    j->huffcode = 0; j->huffbits = j->huffsize = 0; j->rs = -1;
    if (!im->htsize) {   /* no DHT: the standard tables */
      for (int tc = 0; tc < 2; tc++)
        for (int th = 0; th < 2; th++) {
          Huf* h = &j->huf[tc * 64 + th * 16];
          int hval = tc * 1024 + th * 256, code = 0, c = 0, x = 0;
          const uint8_t* bits = tc * 2 + th == 0 ? bits_dc_lum : tc * 2 + th == 1 ? bits_dc_chr : tc * 2 + th == 2 ? bits_ac_lum : bits_ac_chr;
          const uint8_t* vals = tc * 2 + th == 0 ? val_dc : tc * 2 + th == 1 ? val_dc : tc * 2 + th == 2 ? val_ac_lum : val_ac_chr;
          for (i = 0; i < 16; i++) {
            x = bits[i];
            h[i].min = (uint32_t)code;
            h[i].max = (uint32_t)(code += x);
            h[i].val = hval;
            hval += x;
            code += code;
            c += x;
          }
          hval = tc * 1024 + th * 256;
          c--;
          while (c >= 0) { j->hbuf[hval + c] = vals[c]; c--; }
        }
      im->htsize = 4;
    }
    if (!im->sof && im->sos) return im->next_jpeg;
    const int ns = BAT(im->sos + 4), nf = BAT(im->sof + 9);
    JASSERT(ns <= 4 && nf <= 4);
    j->mcusize = 0;
    int hmax = 0;
    for (i = 0; i < ns; ++i)
      for (int k = 0; k < nf; ++k)
        if (BAT(im->sos + 2 * i + 5) == BAT(im->sof + 3 * k + 10)) {
          int hv = BAT(im->sof + 3 * k + 11);
          j->sampling[k] = hv;
          if (hv >> 4 > hmax) hmax = hv >> 4;
          hv = (hv & 15) * (hv >> 4);
          JASSERT(hv >= 1 && hv + j->mcusize <= 10);
          while (hv) {
            JASSERT(j->mcusize < 10);
            j->hufsel[0][j->mcusize] = BAT(im->sos + 2 * i + 6) >> 4 & 15;
            j->hufsel[1][j->mcusize] = BAT(im->sos + 2 * i + 6) & 15;
            JASSERT(j->hufsel[0][j->mcusize] < 4 && j->hufsel[1][j->mcusize] < 4);
            j->color[j->mcusize] = i;
            const int tq = BAT(im->sof + 3 * k + 12);
            JASSERT(tq >= 0 && tq < 4);
            im->qmap[j->mcusize] = tq;
            --hv;
            ++j->mcusize;
          }
        }
    JASSERT(hmax >= 1 && hmax <= 10);
    int k;
    for (k = 0; k < j->mcusize; ++k) {
      j->ls[k] = 0;
      for (i = 1; i < j->mcusize; ++i) if (j->color[(k + i) % j->mcusize] == j->color[k]) j->ls[k] = i;
      j->ls[k] = (j->mcusize - j->ls[k]) << 6;
    }
    for (k = 0; k < 64; ++k) j->zpos[zzu[k] + 8 * zzv[k]] = k;
    j->width = BAT(im->sof + 7) * 256 + BAT(im->sof + 8);
    j->width = (j->width - 1) / (hmax * 8) + 1;
    JASSERT(j->width > 0);
    j->mcusize *= 64;
    j->row = j->column = 0;
    int x = 0, yy = 0;
    for (k = 0; k < (j->mcusize >> 6); k++) {
      const int c = j->color[k];
      const int w = j->sampling[c] >> 4, h = j->sampling[c] & 0xf;
      j->blockW[k] = x == 0 ? j->mcusize - 64 * (w - 1) : 64;
      j->blockN[k] = yy == 0 ? j->mcusize * j->width - 64 * w * (h - 1) : w * 64;
      x++;
      if (x >= w) { x = 0; yy++; }
      if (yy >= h) { x = 0; yy = 0; }
    }
  }
  /* the Huffman decoder: one coded bit :6300-6461 */
  if (j->mcusize && BUF(1 + (!bpos)) != FF) {
    JASSERT(j->huffbits <= 32);
    j->huffcode += j->huffcode + (uint32_t)y;
    ++j->huffbits;
    if (j->rs < 0) {
      JASSERT(j->huffbits >= 1 && j->huffbits <= 16);
      const int ac = (j->mcupos & 63) > 0;
      JASSERT(j->mcupos >= 0 && (j->mcupos >> 6) < 10);
      const int sel = j->hufsel[ac][j->mcupos >> 6];
      JASSERT(sel >= 0 && sel < 4);
      const int i = j->huffbits - 1;
      const Huf* h = &j->huf[ac * 64 + sel * 16];
      JASSERT(h[i].min <= h[i].max && h[i].val < 2048 && j->huffbits > 0);
      if (j->huffcode < h[i].max) {
        JASSERT(j->huffcode >= h[i].min);
        const int k = h[i].val + (int)(j->huffcode - h[i].min);
        JASSERT(k >= 0 && k < 2048);
        j->rs = j->hbuf[k];
        j->huffsize = j->huffbits;
      }
    }
    if (j->rs >= 0 && j->huffsize + (j->rs & 15) == j->huffbits) {   /* a whole code with its extra bits */
      int rs = j->rs;
      j->rs1 = rs;
      int x = 0;
      if (j->mcupos & 63) {   /* AC */
        if (rs == 0) {   /* end of block */
          j->mcupos = (j->mcupos + 63) & -64;
          JASSERT(j->mcupos >= 0 && j->mcupos <= j->mcusize && j->mcupos <= 640);
          while (j->cpos & 63) {
            j->cbuf2[j->cpos & 0x1ffff] = 0;
            j->cbuf[j->cpos & 0x1ffff] = (uint8_t)((!rs) ? 0 : (63 - (j->cpos & 63)) << 4); j->cpos++; rs++;
          }
        } else {
          JASSERT((rs & 15) <= 10);
          const int r = rs >> 4, s = rs & 15;
          JASSERT(j->mcupos >> 6 == (j->mcupos + r) >> 6);
          j->mcupos += r + 1;
          x = (int)(j->huffcode & ((1u << s) - 1));
          if (s && !(x >> (s - 1))) x -= (1 << s) - 1;
          for (int i = r; i >= 1; --i) { j->cbuf2[j->cpos & 0x1ffff] = 0; j->cbuf[j->cpos++ & 0x1ffff] = (uint8_t)(i << 4 | s); }
          j->cbuf2[j->cpos & 0x1ffff] = x;
          j->cbuf[j->cpos++ & 0x1ffff] = (uint8_t)((s << 4) | (j->huffcode << 2 >> s & 3) | 12);
          j->ssum += s;
        }
      } else {   /* DC */
        JASSERT(rs < 12);
        ++j->mcupos;
        x = (int)(j->huffcode & ((1u << rs) - 1));
        if (rs && !(x >> (rs - 1))) x -= (1 << rs) - 1;
        JASSERT(j->mcupos >= 0 && j->mcupos >> 6 < 10);
        const int comp = j->color[j->mcupos >> 6];
        JASSERT(comp >= 0 && comp < 4);
        j->dc = j->pred[comp] += x;
        JASSERT((j->cpos & 63) == 0);
        j->cbuf2[j->cpos & 0x1ffff] = j->dc;
        j->cbuf[j->cpos++ & 0x1ffff] = (uint8_t)((j->dc + 1023) >> 3);
        if ((j->mcupos >> 6) == 0) { j->ssum1 = 0; j->ssum2 = j->ssum3; }
        else {
          if (j->color[(j->mcupos >> 6) - 1] == j->color[0]) j->ssum1 += (j->ssum3 = j->ssum);
          j->ssum2 = j->ssum1;
        }
        j->ssum = rs;
      }
      JASSERT(j->mcupos >= 0 && j->mcupos <= j->mcusize);
      if (j->mcupos >= j->mcusize) {
        j->mcupos = 0;
        if (++j->column == j->width) { j->column = 0; ++j->row; }
      }
      j->huffcode = 0; j->huffsize = j->huffbits = 0; j->rs = -1;
      /* the predictors of the coefficient that comes next :6376-6457 */
      {
#define CB2(i) j->cbuf2[(uint32_t)(i) & 0x1ffff]
#define CB(i) ((int)j->cbuf[(uint32_t)(i) & 0x1ffff])
        const int cpos = j->cpos, mcupos = j->mcupos;
        const int acomp = mcupos >> 6, q = 64 * im->qmap[acomp];
        const int zz = mcupos & 63, cpos_dc = cpos - zz;
        const int norst = j->rstpos != j->column + j->row * j->width;
        int* sumu = j->sumu; int* sumv = j->sumv; int* adv_pred = j->adv_pred; int* run_pred = j->run_pred; int* lcp = j->lcp;
        const uint8_t* qt = im->qtab;
        if (zz == 0) {
          for (int i = 0; i < 8; ++i) sumu[i] = sumv[i] = 0;
          const int off_w = cpos_dc - j->blockW[acomp], off_n = cpos_dc - j->blockN[acomp];
          for (int i = 0; i < 64; ++i) {
            sumu[zzu[i]] += (zzv[i] & 1 ? -1 : 1) * (zzv[i] ? 16 * (16 + zzv[i]) : 185) * (qt[q + i] + 1) * CB2(off_n + i);
            sumv[zzv[i]] += (zzu[i] & 1 ? -1 : 1) * (zzu[i] ? 16 * (16 + zzu[i]) : 185) * (qt[q + i] + 1) * CB2(off_w + i);
          }
        } else {
          sumu[zzu[zz - 1]] -= (zzv[zz - 1] ? 16 * (16 + zzv[zz - 1]) : 185) * (qt[q + zz - 1] + 1) * CB2(cpos - 1);
          sumv[zzv[zz - 1]] -= (zzu[zz - 1] ? 16 * (16 + zzu[zz - 1]) : 185) * (qt[q + zz - 1] + 1) * CB2(cpos - 1);
        }
        for (int i = 0; i < 3; ++i) {
          run_pred[i] = run_pred[i + 3] = 0;
          for (int st = 0; st < 10 && zz + st < 64; ++st) {
            const int zz2 = zz + st;
            int p = sumu[zzu[zz2]] * i + sumv[zzv[zz2]] * (2 - i);
            p /= (qt[q + zz2] + 1) * 185 * (16 + zzv[zz2]) * (16 + zzu[zz2]) / 128;
            if (zz2 == 0 && (norst || j->ls[acomp] == 64)) p -= CB2(cpos_dc - j->ls[acomp]);
            p = (p < 0 ? -1 : +1) * p8f_ilog(abs(p) + 1);
            if (st == 0) adv_pred[i] = p;
            else if (abs(p) > abs(adv_pred[i]) + 2 && abs(adv_pred[i]) < 210) {
              if (run_pred[i] == 0) run_pred[i] = st * 2 + (p > 0);
              if (abs(p) > abs(adv_pred[i]) + 21 && run_pred[i + 3] == 0) run_pred[i + 3] = st * 2 + (p > 0);
            }
          }
        }
        x = 0;
        for (int i = 0; i < 8; ++i) x += (zzu[zz] < i) * sumu[i] + (zzv[zz] < i) * sumv[i];
        x = (sumu[zzu[zz]] * (2 + zzu[zz]) + sumv[zzv[zz]] * (2 + zzv[zz]) - x * 2) * 4 / (zzu[zz] + zzv[zz] + 16);
        x /= (qt[q + zz] + 1) * 185;
        if (zz == 0 && (norst || j->ls[acomp] == 64)) x -= CB2(cpos_dc - j->ls[acomp]);
        adv_pred[3] = (x < 0 ? -1 : +1) * p8f_ilog(abs(x) + 1);
        for (int i = 0; i < 4; ++i) {
          const int a = (i & 1 ? zzv[zz] : zzu[zz]), b = (i & 2 ? 2 : 1);
          if (a < b) x = 65535;
          else {
            const int zz2 = j->zpos[zzu[zz] + 8 * zzv[zz] - (i & 1 ? 8 : 1) * b];
            x = (qt[q + zz2] + 1) * CB2(cpos_dc + zz2) / (qt[q + zz] + 1);
            x = (x < 0 ? -1 : +1) * (p8f_ilog(abs(x) + 1) + (x != 0 ? 17 : 0));
          }
          lcp[i] = x;
        }
        if ((zzu[zz] * zzv[zz]) != 0) {
          int zz2 = j->zpos[zzu[zz] + 8 * zzv[zz] - 9];
          x = (qt[q + zz2] + 1) * CB2(cpos_dc + zz2) / (qt[q + zz] + 1);
          lcp[4] = (x < 0 ? -1 : +1) * (p8f_ilog(abs(x) + 1) + (x != 0 ? 17 : 0));
          zz2 = j->zpos[8 * zzv[zz]];
          x = (qt[q + zz2] + 1) * CB2(cpos_dc + zz2) / (qt[q + zz] + 1);
          lcp[5] = (x < 0 ? -1 : +1) * (p8f_ilog(abs(x) + 1) + (x != 0 ? 17 : 0));
          zz2 = j->zpos[zzu[zz]];
          x = (qt[q + zz2] + 1) * CB2(cpos_dc + zz2) / (qt[q + zz] + 1);
          lcp[6] = (x < 0 ? -1 : +1) * (p8f_ilog(abs(x) + 1) + (x != 0 ? 17 : 0));
        } else lcp[4] = lcp[5] = lcp[6] = 65535;
        int prev1 = 0, prev2 = 0, cnt1 = 0, cnt2 = 0, r = 0, s = 0;
        j->prev_coef_rs = CB(cpos - 64);
        for (int i = 0; i < acomp; i++) {
          x = 0;
          x += CB2(cpos - (acomp - i) * 64);
          if (zz == 0 && (norst || j->ls[i] == 64)) x -= CB2(cpos_dc - (acomp - i) * 64 - j->ls[i]);
          if (j->color[i] == j->color[acomp] - 1) { prev1 += x; cnt1++; r += CB(cpos - (acomp - i) * 64) >> 4; s += CB(cpos - (acomp - i) * 64) & 0xF; }
          if (j->color[acomp] > 1 && j->color[i] == j->color[0]) { prev2 += x; cnt2++; }
        }
        if (cnt1 > 0) { prev1 /= cnt1; r /= cnt1; s /= cnt1; j->prev_coef_rs = (r << 4) | s; }
        if (cnt2 > 0) prev2 /= cnt2;
        j->prev_coef = (prev1 < 0 ? -1 : +1) * p8f_ilog(11 * abs(prev1) + 1) + (cnt1 << 20);
        j->prev_coef2 = (prev2 < 0 ? -1 : +1) * p8f_ilog(11 * abs(prev2) + 1);
        if (j->column == 0 && j->blockW[acomp] > 64 * acomp) { run_pred[1] = run_pred[2]; run_pred[0] = 0; adv_pred[1] = adv_pred[2]; adv_pred[0] = 0; }
        if (j->row == 0 && j->blockN[acomp] > 64 * acomp) { run_pred[1] = run_pred[0]; run_pred[2] = 0; adv_pred[1] = adv_pred[0]; adv_pred[2] = 0; }
      }
    }
  }
  /* the step's inputs :6464-6596 */
  if (!im->jpeg || !im->data) return im->next_jpeg;
  if (BUF(1 + (!bpos)) == FF) {   /* a stuffed byte follows 0xFF */
    out[0] = 128; *kind = 1;
    sets[0] = 0; ranges[0] = 9; sets[1] = 0; ranges[1] = 1025; sets[2] = BUF(1); ranges[2] = 1024;
    return 1;
  }
  if (j->rstlen > 0 && j->rstlen == j->column + j->row * j->width - j->rstpos && j->mcupos == 0 && (int)j->huffcode == (1 << j->huffbits) - 1) {   /* the bits of a restart marker */
    out[0] = 4095; *kind = 2;
    sets[0] = 0; ranges[0] = 9; sets[1] = 0; ranges[1] = 1025; sets[2] = BUF(1); ranges[2] = 1024;
    return 1;
  }
  const int mcupos = j->mcupos, cpos = j->cpos;
  const int comp = j->color[mcupos >> 6];
  const int coef = (mcupos & 63) | comp << 6;
  const int hc = (int)((j->huffcode * 4 + ((mcupos & 63) == 0) * 2 + (comp == 0)) | 1u << (j->huffbits + 2));
  const int firstcol = j->column == 0 && j->blockW[mcupos >> 6] > mcupos;
  if (++j->hbcount > 2 || j->huffbits == 0) j->hbcount = 0;
  JASSERT(coef >= 0 && coef < 256);
  const int zu = zzu[mcupos & 63], zv = zzv[mcupos & 63];
  if (j->hbcount == 0) {
    const int* adv_pred = j->adv_pred; const int* run_pred = j->run_pred; const int* lcp = j->lcp;
    const int prev_coef = j->prev_coef, prev_coef2 = j->prev_coef2, rs1 = j->rs1, ssum = j->ssum, ssum2 = j->ssum2;
    const int cbN = CB(cpos - j->blockN[mcupos >> 6]), cbW = CB(cpos - j->blockW[mcupos >> 6]);
    int64_t n = (int64_t)(int32_t)((uint32_t)hc * 32u);   /* U64 n = hc * 32: the product is an int's (it wraps for the longest codes), then widens */
    uint64_t* cxt = j->cxt;
    cxt[0] = HN(5, ++n, coef, adv_pred[2] / 12 + (run_pred[2] << 8), ssum2 >> 6, prev_coef / 72);
    cxt[1] = HN(5, ++n, coef, adv_pred[0] / 12 + (run_pred[0] << 8), ssum2 >> 6, prev_coef / 72);
    cxt[2] = HN(4, ++n, coef, adv_pred[1] / 11 + (run_pred[1] << 8), ssum2 >> 6);
    cxt[3] = HN(5, ++n, rs1, adv_pred[2] / 7, run_pred[5] / 2, prev_coef / 10);
    cxt[4] = HN(5, ++n, rs1, adv_pred[0] / 7, run_pred[3] / 2, prev_coef / 10);
    cxt[5] = HN(4, ++n, rs1, adv_pred[1] / 11, run_pred[4]);
    cxt[6] = HN(5, ++n, adv_pred[2] / 14, run_pred[2], adv_pred[0] / 14, run_pred[0]);
    cxt[7] = HN(5, ++n, cbN >> 4, adv_pred[3] / 17, run_pred[1], run_pred[5]);
    cxt[8] = HN(5, ++n, cbW >> 4, adv_pred[3] / 17, run_pred[1], run_pred[3]);
    cxt[9] = HN(5, ++n, lcp[0] / 22, lcp[1] / 22, adv_pred[1] / 7, run_pred[1]);
    cxt[10] = HN(5, ++n, lcp[0] / 22, lcp[1] / 22, mcupos & 63, lcp[4] / 30);
    cxt[11] = HN(5, ++n, zu / 2, lcp[0] / 13, lcp[2] / 30, prev_coef / 40 + ((prev_coef2 / 28) << 20));
    cxt[12] = HN(5, ++n, zv / 2, lcp[1] / 13, lcp[3] / 30, prev_coef / 40 + ((prev_coef2 / 28) << 20));
    cxt[13] = HN(8, ++n, rs1, prev_coef / 42, prev_coef2 / 34, lcp[0] / 60, lcp[2] / 14, lcp[1] / 60, lcp[3] / 14);
    cxt[14] = HN(3, ++n, mcupos & 63, j->column >> 1);
    cxt[15] = HN(7, ++n, j->column >> 3, imin(5 + 2 * (!comp), zu + zv), lcp[0] / 10, lcp[2] / 40, lcp[1] / 10, lcp[3] / 40);
    cxt[16] = HN(3, ++n, ssum >> 3, mcupos & 63);
    cxt[17] = HN(4, ++n, rs1, mcupos & 63, run_pred[1]);
    {
      ++n;
      const uint64_t inner = comp ? HN(2, prev_coef / 22, prev_coef2 / 50) : (uint64_t)(int64_t)(ssum / ((mcupos & 0x3F) + 1));
      uint64_t h = ((uint64_t)n + 1) * MUL[0] + ((uint64_t)(int64_t)coef + 1) * MUL[1] + ((uint64_t)(int64_t)(ssum2 >> 5) + 1) * MUL[2] + ((uint64_t)(int64_t)(adv_pred[3] / 30) + 1) * MUL[3] + (inner + 1) * MUL[4];
      cxt[18] = h;
    }
    cxt[19] = HN(7, ++n, lcp[0] / 40, lcp[1] / 40, adv_pred[1] / 28, (comp) ? prev_coef / 40 + ((prev_coef2 / 40) << 20) : lcp[4] / 22, imin(7, zu + zv), ssum / (2 * (zu + zv) + 1));
    cxt[20] = HN(5, ++n, zv, cbN, adv_pred[2] / 28, run_pred[2]);
    cxt[21] = HN(5, ++n, zu, cbW, adv_pred[0] / 28, run_pred[0]);
    cxt[22] = HN(3, ++n, adv_pred[2] / 7, run_pred[2]);
    cxt[23] = HN(3, n, adv_pred[0] / 7, run_pred[0]);
    cxt[24] = HN(3, n, adv_pred[1] / 7, run_pred[1]);
    cxt[25] = HN(5, ++n, zv, lcp[1] / 14, adv_pred[2] / 16, run_pred[5]);
    cxt[26] = HN(5, ++n, zu, lcp[0] / 14, adv_pred[0] / 16, run_pred[3]);
    cxt[27] = HN(4, ++n, lcp[0] / 14, lcp[1] / 14, adv_pred[3] / 16);
    cxt[28] = HN(4, ++n, coef, prev_coef / 10, prev_coef2 / 20);
    cxt[29] = HN(4, ++n, coef, ssum >> 2, j->prev_coef_rs);
    cxt[30] = HN(6, ++n, coef, adv_pred[1] / 17, lcp[(zu < zv)] / 24, lcp[2] / 20, lcp[3] / 24);
    cxt[31] = HN(6, ++n, coef, adv_pred[3] / 11, lcp[(zu < zv)] / 50, lcp[2 + 3 * (zu * zv > 1)] / 50, lcp[3 + 3 * (zu * zv > 1)] / 50);
  }
  /* the device's part: bit histories, StateMaps, the model's own mixer (sets: firstcol / 2, coef + 256 min(3, huffbits) / 1024,
   * (hc & 0x1FE) * 2 + min(3, ilog2(zu + zv)) / 1024), two APM stages */
  const int m1sel[3] = {firstcol, 2 + coef + 256 * imin(3, j->huffbits), 2 + 1024 + (hc & 0x1FE) * 2 + imin(3, (int)ilog2u((unsigned)(zu + zv)))};
  const int a1ctx = (hc & 511) | (((j->adv_pred[1] / 16) & 63) << 9), a2ctx = (hc & 511) | (coef << 9);
  const int n = p8f_jpg_step(j->dev, j->hbcount, (int)(j->huffcode & 1), j->hbcount == 0 ? j->cxt : NULL, m1sel, a1ctx, a2ctx, out);
  *kind = 3;
  sets[0] = 1 + (zu + zv < 5) + (j->huffbits > 8) * 2 + firstcol * 4; ranges[0] = 9;
  sets[1] = 1 + (hc & 0xFF) + 256 * imin(3, (zu + zv) / 3); ranges[1] = 1025;
  sets[2] = coef + 256 * imin(3, j->huffbits / 2); ranges[2] = 1024;
  return n ? 1 : 1;
#undef CB
#undef CB2
#undef BAT
#undef BUF
}
