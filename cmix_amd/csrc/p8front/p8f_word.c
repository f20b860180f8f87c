/* p8front/p8f_word.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's wordModel (reference src/models/paq8.cpp:3873-4105): word / number / punctuation state
 * over the byte stream (six word hashes, hyphenation repair, first characters of lines, column context, wiki markup
 * words, the English stemmer on completed words) feeding 61 contexts into one ContextMap, and the word-level globals
 * other models read (spaces, words, wordlen, frstchar, spafdo, col ...). Parity: tests/test_p8stage_host.py (stage vs columns 434..2024 of reference traces). The reference's quirks are kept as they are (the `lastLetter = 3 && ...` assignments
 * inside the hyphenation test, which leave 0 or 1 in lastLetter). */
#include <ctype.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "p8f_stem.h"

typedef struct CM1 CM1;
CM1* p8f_cm_new(uint64_t size_bytes, int count);
int p8f_cm_step(CM1* c, int y1, int bp, int c0, int c1, const uint64_t* ctx, int nset, int16_t* out, int* nout);
int p8f_ilog(int x);
uint64_t p8f_combine64(uint64_t seed, uint64_t x);
uint64_t p8f_hash2(uint64_t a, uint64_t b);
uint64_t p8f_hash3(uint64_t a, uint64_t b, uint64_t c);
uint64_t p8f_hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d);
uint64_t p8f_hash5(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e);

static int llog_u(uint32_t x) {
  if (x >= 0x1000000) return 256 + p8f_ilog((int)(x >> 16));
  if (x >= 0x10000) return 128 + p8f_ilog((int)(x >> 8));
  return p8f_ilog((int)x);
}
#define U(x) ((uint64_t)(x))

typedef struct {
  CM1* cm;
  uint64_t word0, word1, word2, word3, word4, word5, xword0, xword1, xword2, cword0, ccword, number0, number1;
  uint32_t wrdhsh, text0, data0, type0, lastLetter, firstLetter, lastUpper, lastDigit, wordGap, mask, mask2;
  int nl1, nl, w;
  int wpos[0x10000];
  P8Word stem[4];
  int cWord, pWord, StemIndex;
  /* namespace-level globals the model maintains (:3870-3872) */
  uint32_t frstchar, spafdo, spaces, spacecount, words, wordcount, wordlen, wordlen1, col;
} WordM;

WordM* p8f_word_new(int level) {
  WordM* m = (WordM*)calloc(1, sizeof *m);
  m->cm = p8f_cm_new((0x10000ull << level) * 16, 61);
  m->nl1 = -3; m->nl = -2;
  m->cWord = 0; m->pWord = 3;
  return m;
}

int p8f_word_step(WordM* m, int y, int bpos, int c0, uint32_t c4, uint32_t f4, uint32_t b3, int blpos, const uint8_t* hist,
                     uint32_t bmask, int pos, int16_t* out, uint32_t* g_out) {
#define RB(i) ((uint32_t)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])
#define RA(i) ((uint32_t)hist[(uint32_t)(i) & bmask])
  uint64_t cx[64];
  int n = 0;
  if (bpos == 0) {
    int c = c4 & 255, pC = (uint8_t)(c4 >> 8), f = 0;
    if (m->spaces & 0x80000000) --m->spacecount;
    if (m->words & 0x80000000) --m->wordcount;
    m->spaces *= 2;
    m->words *= 2;
    m->lastUpper = m->lastUpper + 1 < 255 ? m->lastUpper + 1 : 255;
    m->lastLetter = m->lastLetter + 1 < 255 ? m->lastLetter + 1 : 255;
    m->mask2 <<= 2;
    if (c >= 'A' && c <= 'Z') { c += 'a' - 'A'; m->lastUpper = 0; }
    P8Word* cw = &m->stem[m->cWord];
    if ((c >= 'a' && c <= 'z') || c == '\'' || c == '-') p8w_add(cw, c);
    else if (p8w_len(cw) > 0) {
      p8_en_stem(cw);
      p8w_hashes(cw);
      m->StemIndex = (m->StemIndex + 1) & 3;
      m->pWord = m->cWord;
      m->cWord = m->StemIndex;
      p8w_init(&m->stem[m->cWord]);
    }
    if ((c >= 'a' && c <= 'z') || ((c >= 128 && (b3 != 3)) || (c > 0 && c < 4))) {
      if (!m->wordlen) {
        /* four tests, each ASSIGNING its truth value to lastLetter before it is used (reference :3910-3911) */
        int hyph = (m->lastLetter = (uint32_t)(3 && (c4 & 0xFFFF00) == 0x2B0A00 && RB(4) != 0x2B));
        if (!hyph) hyph = (m->lastLetter = (uint32_t)(4 && (c4 & 0xFFFFFF00) == 0x2B0D0A00 && RB(5) != 0x2B));
        if (!hyph) hyph = (m->lastLetter = (uint32_t)(3 && (c4 & 0xFFFF00) == 0x2D0A00 && RB(4) != 0x2D));
        if (!hyph) hyph = (m->lastLetter = (uint32_t)(4 && (c4 & 0xFFFFFF00) == 0x2D0D0A00 && RB(5) != 0x2D));
        if (hyph) {
          m->word0 = m->word1; m->word1 = m->word2; m->word2 = m->word3; m->word3 = m->word4; m->word4 = m->word5; m->word5 = 0;
          m->wordlen = m->wordlen1;
          if (c < 128) {
            m->StemIndex = (m->StemIndex - 1) & 3;
            m->cWord = m->pWord;
            m->pWord = (m->StemIndex - 1) & 3;
            p8w_init(&m->stem[m->cWord]);
            for (uint32_t i = 0; i <= m->wordlen; i++)
              p8w_add(&m->stem[m->cWord], tolower((int)RB(m->wordlen - i + 1 + 2 * (i != m->wordlen))));
          }
        } else {
          m->wordGap = m->lastLetter;
          m->firstLetter = (uint32_t)c;
          m->wrdhsh = 0;
        }
      }
      m->lastLetter = 0;
      ++m->words; ++m->wordcount;
      if (c > 4) m->word0 = p8f_combine64(m->word0, U(c));
      m->text0 = m->text0 * 997 * 16 + (uint32_t)c;
      m->wordlen++;
      m->wordlen = m->wordlen < 45 ? m->wordlen : 45;
      f = 0;
      m->w = (int)((uint32_t)m->word0 & 0xffff);
      if ((c == 'a' || c == 'e' || c == 'i' || c == 'o' || c == 'u') ||
          (c == 'y' && (m->wordlen > 0 && pC != 'a' && pC != 'e' && pC != 'i' && pC != 'o' && pC != 'u'))) {
        m->mask2++;
        m->wrdhsh = m->wrdhsh * 997 * 8 + (uint32_t)(c / 4 - 22);
      } else if (c >= 'b' && c <= 'z') {
        m->mask2 += 2;
        m->wrdhsh = m->wrdhsh * 271 * 32 + (uint32_t)(c - 97);
      } else m->wrdhsh = m->wrdhsh * 11 * 32 + (uint32_t)c;
    } else {
      if (m->word0) {
        m->type0 = (m->type0 << 2) | 1;
        m->word5 = m->word4; m->word4 = m->word3; m->word3 = m->word2; m->word2 = m->word1; m->word1 = m->word0;
        m->wordlen1 = m->wordlen;
        m->wpos[m->w] = blpos;
        if (c == ':' || c == '=') m->cword0 = m->word0;
        if (c == ']' && (m->frstchar != ':')) m->xword0 = m->word0;
        m->ccword = 0;
        m->word0 = 0;
        m->wordlen = 0;
        if ((c == '.' || c == '!' || c == '?' || c == '}' || c == ')') && RB(2) != 10) f = 1;
      }
      if (c == 32 || c == 10 || c == 5) { ++m->spaces; ++m->spacecount; if (c == 10 || c == 5) { m->nl1 = m->nl; m->nl = pos - 1; } }
      else if (c == '.' || c == '!' || c == '?' || c == ',' || c == ';' || c == ':') { m->spafdo = 0; m->ccword = U(c); m->mask2 += 3; }
      else { ++m->spafdo; m->spafdo = m->spafdo < 63 ? m->spafdo : 63; }
    }
    if ((c4 & 0xFFFF) == 0x3D3D && m->frstchar == 0x3d) m->xword1 = m->word1;
    if ((c4 & 0xFFFF) == 0x2727) m->xword2 = m->word1;
    m->lastDigit = m->lastDigit + 1 < 0xFF ? m->lastDigit + 1 : 0xFF;
    if (c >= '0' && c <= '9') {
      if (RB(3) >= '0' && RB(3) <= '9' && (RB(2) == '.') && m->number0 == 0) { m->number0 = m->number1; m->number1 = 0; }
      m->number0 = p8f_combine64(m->number0, U(c));
      m->lastDigit = 0;
    } else if (m->number0) {
      m->type0 = (m->type0 << 2) | 2;
      m->number1 = m->number0;
      m->number0 = 0; m->ccword = 0;
    }
    if (!((c >= 'a' && c <= 'z') || (c >= '0' && c <= '9') || (c >= 128))) m->data0 ^= (uint32_t)p8f_combine64(m->data0, U(c));
    else if (m->data0) { m->type0 = (m->type0 << 2) | 3; m->data0 = 0; }
    m->col = (uint32_t)(pos - m->nl < 255 ? pos - m->nl : 255);
    const int above = (int)RA((uint32_t)(m->nl1 + (int)m->col));
    if (m->col <= 2) m->frstchar = (m->col == 2 ? (uint32_t)(c < 96 ? c : 96) : 0);
    if (m->frstchar == '[' && c == 32) { if (RB(3) == ']' || RB(4) == ']') { m->frstchar = 96; m->xword0 = 0; } }
    const uint32_t col = m->col, frst = m->frstchar;
    cx[n++] = p8f_hash4(513, m->spafdo, m->spaces, m->ccword);
    cx[n++] = p8f_hash3(514, frst, U(c));
    cx[n++] = p8f_hash4(515, col, frst, U((m->lastUpper < col) * 4 + (m->mask2 & 3)));
    cx[n++] = p8f_hash3(516, m->spaces, (m->words & 255));
    cx[n++] = m->spaces & 0x7fff;
    cx[n++] = m->spaces & 0xff;
    cx[n++] = p8f_hash4(257, m->number0, m->word1, m->wordGap);
    cx[n++] = p8f_hash4(258, m->number1, U(c), m->ccword);
    cx[n++] = p8f_hash4(259, m->number0, m->number1, m->wordGap);
    cx[n++] = p8f_hash4(260, m->word0, m->number1, U(m->lastDigit < m->wordGap + m->wordlen));
    cx[n++] = p8f_hash3(274, m->number0, m->cword0);
    cx[n++] = p8f_hash3(518, m->wordlen1, col);
    cx[n++] = p8f_hash4(519, U(c), m->spacecount / 2, m->wordGap);
    uint32_t h = m->wordcount * 64 + m->spacecount;
    cx[n++] = p8f_hash4(520, U(c), h, m->ccword);
    cx[n++] = p8f_hash4(517, frst, h, m->lastLetter);
    cx[n++] = p8f_hash4(m->data0, m->word1, m->number1, m->type0 & 0xFFF);
    cx[n++] = p8f_hash3(521, h, m->spafdo);
    const uint32_t d = c4 & 0xf0ff;
    cx[n++] = p8f_hash4(522, d, frst, m->ccword);
    h = (uint32_t)(m->word0 * 271);
    h = h + RB(1);
    cx[n++] = p8f_hash3(262, h, 0);
    cx[n++] = p8f_hash2(m->number0 * 271 + RB(1), 0);
    cx[n++] = p8f_hash3(263, m->word0, 0);
    if (m->wrdhsh) cx[n++] = p8f_hash2(m->wrdhsh, RB((uint32_t)m->wpos[m->word1 & 0xffff])); else cx[n++] = 0;
    cx[n++] = p8f_hash3(264, h, m->word1);
    cx[n++] = p8f_hash3(265, m->word0, m->word1);
    cx[n++] = p8f_hash5(266, h, m->word1, m->word2, U(m->lastUpper < m->wordlen));
    cx[n++] = p8f_hash3(267, m->text0 & 0xffffff, 0);
    cx[n++] = m->text0 & 0xfffff;
    cx[n++] = p8f_hash3(269, m->word0, m->xword0);
    cx[n++] = p8f_hash3(270, h, m->xword1);
    cx[n++] = p8f_hash3(271, h, m->xword2);
    cx[n++] = p8f_hash3(272, frst, m->xword2);
    cx[n++] = p8f_hash3(273, m->word0, m->cword0);
    cx[n++] = p8f_hash3(275, h, m->word2);
    cx[n++] = p8f_hash3(276, h, m->word3);
    cx[n++] = p8f_hash3(277, h, m->word4);
    cx[n++] = p8f_hash3(278, h, m->word5);
    cx[n++] = p8f_hash4(279, h, m->word1, m->word3);
    cx[n++] = p8f_hash4(280, h, m->word2, m->word3);
    cx[n++] = RB(1) | RB(3) << 8 | RB(5) << 16;
    cx[n++] = RB(2) | RB(4) << 8 | RB(6) << 16;
    cx[n++] = RB(1) | RB(4) << 8 | RB(7) << 16;
    if (f) { m->word5 = m->word4; m->word4 = m->word3; m->word3 = m->word2; m->word2 = m->word1; m->word1 = '.'; }
    if (col < 255u) {
      cx[n++] = p8f_hash4(523, col, RB(1), U(above));
      cx[n++] = p8f_hash3(524, RB(1), U(above));
      cx[n++] = p8f_hash3(525, col, RB(1));
      cx[n++] = p8f_hash3(526, col, U(c == 32));
    } else { cx[n++] = 0; cx[n++] = 0; cx[n++] = 0; cx[n++] = 0; }
    if (m->wordlen) cx[n++] = p8f_hash3(281, m->word0, U(llog_u((uint32_t)(blpos - m->wpos[m->word1 & 0xffff])) >> 4));
    else cx[n++] = 0;
    cx[n++] = p8f_hash3(282, RB(1), U(llog_u((uint32_t)(blpos - m->wpos[m->word1 & 0xffff])) >> 2));
    cx[n++] = p8f_hash4(283, RB(1), m->word0, U(llog_u((uint32_t)(blpos - m->wpos[m->word2 & 0xffff])) >> 2));
    int fl = 0;
    const int cc = c4 & 0xff;
    if (cc != 0) {
      if (isalpha(cc)) fl = 1;
      else if (ispunct(cc)) fl = 2;
      else if (isspace(cc)) fl = 3;
      else if (cc == 0xff) fl = 4;
      else if (cc < 16) fl = 5;
      else if (cc < 64) fl = 6;
      else fl = 7;
    }
    m->mask = (m->mask << 3) | (uint32_t)fl;
    cx[n++] = p8f_hash3(528, m->mask, 0);
    cx[n++] = p8f_hash3(529, m->mask, RB(1));
    cx[n++] = p8f_hash3(530, m->mask & 0xff, col);
    cx[n++] = p8f_hash4(531, m->mask, RB(2), RB(3));
    cx[n++] = p8f_hash3(532, m->mask & 0x1ff, f4 & 0x00fff0);
    cx[n++] = p8f_hash5(h, U(llog_u(m->wordGap)), m->mask & 0x1FF,
                           U(((m->wordlen1 > 3) << 6) | ((m->wordlen > 0) << 5) | ((m->spafdo == m->wordlen + 2) << 4) |
                             ((m->spafdo == m->wordlen + m->wordlen1 + 3) << 3) | ((m->spafdo >= m->lastLetter + m->wordlen1 + m->wordGap) << 2) |
                             ((m->lastUpper < m->lastLetter + m->wordlen1) << 1) | (m->lastUpper < m->wordlen + m->wordlen1 + m->wordGap)),
                           m->type0 & 0xFFF);
    if (m->wordlen1) cx[n++] = p8f_hash4(col, m->wordlen1, U(above & 0x5F), c4 & 0x5F); else cx[n++] = 0;
    if (m->wrdhsh) cx[n++] = p8f_hash4(m->mask2 & 0x3F, m->wrdhsh & 0xFFF, U((0x100 | m->firstLetter) * (m->wordlen < 6)),
                                          U((m->wordGap > 4) * 2 + (m->wordlen1 > 5)));
    else cx[n++] = 0;
    if (m->lastLetter < 16) cx[n++] = p8f_hash2(m->stem[m->pWord].Hash[2], h); else cx[n++] = 0;
  }
  int nout = 0;
  p8f_cm_step(m->cm, y, bpos, c0, (int)RB(1), cx, n, out, &nout);
  g_out[0] = m->spaces; g_out[1] = m->spacecount; g_out[2] = m->words; g_out[3] = m->wordcount; g_out[4] = m->wordlen;
  g_out[5] = m->wordlen1; g_out[6] = m->frstchar; g_out[7] = m->spafdo; g_out[8] = m->col;
  return nout;
}
