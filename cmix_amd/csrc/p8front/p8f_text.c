/* p8front/p8f_text.c -- HOST FRONT END of the paq8 stage (product code; tables are recorded through p8f_emit.h, the device learns).
 *
 * Host front end for paq8's TextModel (reference src/models/paq8.cpp:3006-3518): a parser over the byte stream that
 * tracks words (stemmed in English, French and German at once, the language with the most recognised words among
 * the last 64 wins), segments, sentences and paragraphs, numbers and their differences, quotes, nesting, punctuation
 * and a 12-deep history of ASCII groups; 33 contexts go to one ContextMap2 and eight mixer weight-set selectors are
 * derived from the same state. Parity: tests/test_p8stage_host.py (stage vs columns 434..2024 of reference traces). */
#include <ctype.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "p8f_stem.h"
#include "p8f_tables.h"

typedef struct CM2 CM2;
CM2* p8f_cm2_new(uint64_t size_bytes, uint32_t count);
int p8f_cm2_step(CM2* c, int y_prev, int bpos, const uint64_t* ctx, int nset, int16_t* out, int* nout);
int p8f_ilog(int x);
uint64_t p8f_combine64(uint64_t seed, uint64_t x);
uint32_t p8f_finalize64(uint64_t h, int bits);
uint64_t p8f_hash2(uint64_t a, uint64_t b);
uint64_t p8f_hash3(uint64_t a, uint64_t b, uint64_t c);
uint64_t p8f_hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d);
uint64_t p8f_hash5(uint64_t a, uint64_t b, uint64_t c, uint64_t d, uint64_t e);

#define U(x) ((uint64_t)(x))
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a < b ? b : a; }
static unsigned ilog2u(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }
static int llog_u(uint32_t x) {
  if (x >= 0x1000000) return 256 + p8f_ilog((int)(x >> 16));
  if (x >= 0x10000) return 128 + p8f_ilog((int)(x >> 8));
  return p8f_ilog((int)x);
}

enum { ST_Unknown, ST_ReadingWord, ST_PossibleHyphenation, ST_WasAbbreviation, ST_AfterComma, ST_AfterQuote, ST_AfterAbbreviation, ST_ExpectDigit };
typedef struct { P8Word FirstWord; uint32_t WordCount, NumCount; } Segm;                 /* :1623-1629 */
typedef struct {                                                                         /* :1631-1646 */
  P8Word FirstWord; uint32_t WordCount, NumCount;
  int Type; uint32_t SegmentCount, VerbIndex, NounIndex, CapitalIndex;
  P8Word lastVerb, lastNoun, lastCapital;
} Sent;
typedef struct { uint32_t SentenceCount, TypeCount[3], TypeMask; } Para;                 /* :1648-1651 */

typedef struct {
  CM2* map;
  P8Word words[LANG_Count][8]; uint32_t wi[LANG_Count];   /* Cache<Word, 8> per language slot */
  Segm segments[4]; uint32_t si;
  Sent sentences[4]; uint32_t ni;
  Para paragraphs[2]; uint32_t pi;
  uint32_t WordPos[0x10000], BytePos[256];
  P8Word *cWord, *pWord;
  Segm* cSegment; Sent* cSentence; Para* cParagraph;
  int State, pState;
  uint32_t lang_count[LANG_Count - 1]; uint64_t lang_mask[LANG_Count - 1]; int lang_id, lang_pid;
  /* Info :3101-3134 */
  uint64_t numbers[2], numHashes[2]; uint8_t numLength[2];
  uint32_t numMask, numDiff, lastUpper, maskUpper, lastLetter, lastDigit, lastPunct, lastNewLine, prevNewLine, wordGap, spaces, spaceCount,
      commas, quoteLength, maskPunct, nestHash, lastNest;
  uint64_t asciiMask;
  uint32_t masks[5], wordLength[2];
  int UTF8Remaining;
  uint8_t firstLetter, firstChar, expectedDigit, prevPunct;
  P8Word TopicDescriptor;
  uint64_t ParseCtx;
} TextM;

#define WORDS(l, i) (&m->words[l][(m->wi[l] - (uint32_t)(i)) & 7])
#define SEGMENTS(i) (&m->segments[(m->si - (uint32_t)(i)) & 3])

static int (*const STEM[LANG_Count - 1])(P8Word*) = {p8_en_stem, p8_fr_stem, p8_de_stem};
static int (*const IS_VOWEL[LANG_Count - 1])(int) = {p8_en_is_vowel, p8_fr_is_vowel, p8_de_is_vowel};
static int is_abbreviation(int lang, const P8Word* w) {  /* English / French / German ::IsAbbreviation :1664-1727 */
  static const char* const en[] = {"mr", "mrs", "ms", "dr", "st", "jr"};
  static const char* const fr[] = {"m", "mm"};
  static const char* const de[] = {"fr", "hr", "hrn"};
  return lang == LANG_English ? p8w_matches_any(w, en, 6) : lang == LANG_French ? p8w_matches_any(w, fr, 2) : p8w_matches_any(w, de, 3);
}

TextM* p8f_text_new(uint32_t size_bytes) {
  TextM* m = (TextM*)calloc(1, sizeof *m);
  m->map = p8f_cm2_new(size_bytes, 33);
  m->cWord = WORDS(0, 0);
  m->pWord = WORDS(0, 1);
  m->cSegment = &m->segments[0];
  m->cSentence = &m->sentences[0];
  m->cParagraph = &m->paragraphs[0];
  return m;
}

#define RB(i) ((uint32_t)hist[((uint32_t)pos - (uint32_t)(i)) & bmask])

static void word_completed(TextM* m, int pos) {  /* :3249-3298: stem in every language, pick the language, file the word */
  if (m->lang_id != LANG_Unknown) memcpy(WORDS(LANG_Unknown, 0), m->cWord, sizeof(P8Word));
  for (int i = LANG_Count - 1; i > LANG_Unknown; i--) {
    m->lang_count[i - 1] -= (uint32_t)(m->lang_mask[i - 1] >> 63);
    m->lang_mask[i - 1] <<= 1;
    if (i != m->lang_id) memcpy(WORDS(i, 0), m->cWord, sizeof(P8Word));
    if (STEM[i - 1](WORDS(i, 0))) { m->lang_count[i - 1]++; m->lang_mask[i - 1] |= 1; }
  }
  m->lang_id = LANG_Unknown;
  uint32_t best = 4;  /* MIN_RECOGNIZED_WORDS */
  for (int i = LANG_Count - 1; i > LANG_Unknown; i--) {
    if (m->lang_count[i - 1] >= best) { best = m->lang_count[i - 1] + (uint32_t)(i == m->lang_pid); m->lang_id = i; }
    m->wi[i]++;
  }
  m->wi[LANG_Unknown]++;
  m->lang_pid = m->lang_id;
  m->pWord = WORDS(m->lang_id, 1);
  m->cWord = WORDS(m->lang_id, 0);
  p8w_init(m->cWord);
  m->WordPos[m->pWord->Hash[1] & 0xFFFF] = (uint32_t)pos;
  if (m->cSegment->WordCount == 0) memcpy(&m->cSegment->FirstWord, m->pWord, sizeof(P8Word));
  m->cSegment->WordCount++;
  if (m->cSentence->WordCount == 0) memcpy(&m->cSentence->FirstWord, m->pWord, sizeof(P8Word));
  m->cSentence->WordCount++;
  m->wordLength[1] = m->wordLength[0]; m->wordLength[0] = 0;
  m->quoteLength += (m->quoteLength > 0);
  if (m->quoteLength > 0x1F) m->quoteLength = 0;
  m->cSentence->VerbIndex++; m->cSentence->NounIndex++; m->cSentence->CapitalIndex++;
  if (m->pWord->Type & LANG_Verb) { m->cSentence->VerbIndex = 0; memcpy(&m->cSentence->lastVerb, m->pWord, sizeof(P8Word)); }
  if (m->pWord->Type & LANG_Noun) { m->cSentence->NounIndex = 0; memcpy(&m->cSentence->lastNoun, m->pWord, sizeof(P8Word)); }
  if (m->cSentence->WordCount > 1 && m->lastUpper < m->wordLength[1]) {
    m->cSentence->CapitalIndex = 0;
    memcpy(&m->cSentence->lastCapital, m->pWord, sizeof(P8Word));
  }
}

static void text_update(TextM* m, const uint8_t* hist, uint32_t bmask, int pos) {  /* Update :3188-3425 */
  m->lastUpper = (uint32_t)imin(0xFF, (int)(m->lastUpper + 1)); m->maskUpper <<= 1;
  m->lastLetter = (uint32_t)imin(0x1F, (int)(m->lastLetter + 1));
  m->lastDigit = (uint32_t)imin(0xFF, (int)(m->lastDigit + 1));
  m->lastPunct = (uint32_t)imin(0x3F, (int)(m->lastPunct + 1));
  m->lastNewLine++; m->prevNewLine++; m->lastNest++;
  m->spaceCount -= (m->spaces >> 31); m->spaces <<= 1;
  m->masks[0] <<= 2; m->masks[1] <<= 2; m->masks[2] <<= 4; m->masks[3] <<= 3;
  m->pState = m->State;

  uint8_t c = (uint8_t)RB(1), pC = (uint8_t)tolower(c);
  const uint8_t g = (c < 0x80) ? P8_ASCII_GROUP[c] : 31;
  if (!(g <= 4 && g == (m->asciiMask & 0x1f))) m->asciiMask = ((m->asciiMask << 5) | g) & ((1ull << 60) - 1);
  m->masks[4] = (uint32_t)(m->asciiMask & ((1u << 30) - 1));
  m->BytePos[c] = (uint32_t)pos;
  if (c != pC) { c = pC; m->lastUpper = 0; m->maskUpper |= 1; }
  pC = (uint8_t)RB(2);
  m->State = ST_Unknown;
  m->ParseCtx = p8f_hash5(ST_Unknown, m->pWord->Hash[1], c, U((ilog2u(m->lastNewLine) + 1) * (m->lastNewLine * 3 > m->prevNewLine)), m->masks[1] & 0xFC);

  if ((c >= 'a' && c <= 'z') || c == '\'' || c == '-' || c > 0x7F) {
    if (m->wordLength[0] == 0) {
      /* a word continued after "+\n" / "+\r\n": take the previous word back and extend it */
      if (pC == 10 && ((m->lastLetter == 3 && RB(3) == '+') || (m->lastLetter == 4 && RB(3) == 13 && RB(4) == '+'))) {
        m->wordLength[0] = m->wordLength[1];
        for (int i = LANG_Unknown; i < LANG_Count; i++) m->wi[i]--;
        m->cWord = m->pWord; m->pWord = WORDS(m->lang_pid, 1);
        p8w_init(m->cWord);
        for (uint32_t i = 0; i < m->wordLength[0]; i++) p8w_add(m->cWord, (int)RB(m->wordLength[0] - i + m->lastLetter));
        m->wordLength[1] = p8w_len(m->pWord);
        m->cSegment->WordCount--;
        m->cSentence->WordCount--;
      } else { m->wordGap = m->lastLetter; m->firstLetter = c; }
    }
    m->lastLetter = 0;
    m->wordLength[0]++;
    m->masks[0] += (m->lang_id != LANG_Unknown) ? 1u + (uint32_t)IS_VOWEL[m->lang_id - 1]((char)c) : 1u;
    m->masks[1]++;
    m->masks[3] += m->masks[0] & 3;
    if (c == '\'') {
      m->masks[2] += 12;
      if (m->wordLength[0] == 1) {
        if (m->quoteLength == 0 && pC == 32) m->quoteLength = 1;
        else if (m->quoteLength > 0 && m->lastPunct == 1) { m->quoteLength = 0; m->State = ST_AfterQuote; m->ParseCtx = p8f_hash2(ST_AfterQuote, pC); }
      }
    }
    p8w_add(m->cWord, (char)c);
    p8w_hashes(m->cWord);
    m->State = ST_ReadingWord;
    m->ParseCtx = p8f_hash2(ST_ReadingWord, m->cWord->Hash[1]);
  } else {
    if (p8w_len(m->cWord) > 0) word_completed(m, pos);
    const int sentence_end = (c == '.' || c == '?' || c == '!');
    if (c == '.' && m->lang_id != LANG_Unknown && m->lastUpper == m->wordLength[1] && is_abbreviation(m->lang_id, m->pWord)) {
      m->State = ST_WasAbbreviation;
      m->ParseCtx = p8f_hash2(ST_WasAbbreviation, m->pWord->Hash[1]);
    } else if (sentence_end || c == ',' || c == ';' || c == ':') {
      if (sentence_end) {
        Sent* s = m->cSentence;
        s->Type = (c == '.') ? 0 : (c == '?') ? 1 : 2;  /* Declarative, Interrogative, Exclamative */
        s->SegmentCount++;
        m->cParagraph->SentenceCount++;
        m->cParagraph->TypeCount[s->Type]++;
        m->cParagraph->TypeMask <<= 2; m->cParagraph->TypeMask |= (uint32_t)s->Type;
        m->ni++;
        m->cSentence = &m->sentences[m->ni & 3];
        memset(m->cSentence, 0, sizeof(Sent));
        m->masks[3] += 3;
      }
      if (c == ',') {
        m->commas++;
        m->State = ST_AfterComma;
        m->ParseCtx = p8f_hash4(ST_AfterComma, ilog2u(m->quoteLength + 1), ilog2u(m->lastNewLine), m->lastUpper < m->lastLetter + m->wordLength[1]);
      } else if (c == ':') memcpy(&m->TopicDescriptor, m->pWord, sizeof(P8Word));
      if (!sentence_end) { m->cSentence->SegmentCount++; m->masks[3] += 4; }
      m->lastPunct = 0; m->prevPunct = c;
      m->masks[0] += 3; m->masks[1] += 2; m->masks[2] += 15;
      m->si++;
      m->cSegment = &m->segments[m->si & 3];
      memset(m->cSegment, 0, sizeof(Segm));
    } else if (c == 10 || c == 9 || c == 13 || c == 32) {
      if (c == 10) {
        m->prevNewLine = m->lastNewLine; m->lastNewLine = 0;
        m->commas = 0;
        if (m->prevNewLine == 1 || (m->prevNewLine == 2 && pC == 13)) {
          m->pi++;
          m->cParagraph = &m->paragraphs[m->pi & 1];
          memset(m->cParagraph, 0, sizeof(Para));
        } else if ((m->lastLetter == 2 && pC == '+') || (m->lastLetter == 3 && pC == 13 && RB(3) == '+')) {
          m->ParseCtx = p8f_hash2(ST_ReadingWord, m->pWord->Hash[1]);
          m->State = ST_PossibleHyphenation;
        }
      }
      m->spaceCount++; m->spaces |= 1;
      m->masks[1] += 3; m->masks[3] += 5;
      if (c == 32 && m->pState == ST_WasAbbreviation) { m->State = ST_AfterAbbreviation; m->ParseCtx = p8f_hash2(ST_AfterAbbreviation, m->pWord->Hash[1]); }
    } else {
      static const struct { uint8_t ch, m2, m3; int8_t nest; } BR[] = {  /* brackets: masks[2] code, masks[3] code, nesting step */
          {'(', 1, 6, 31}, {'[', 2, 0, 11}, {'{', 3, 0, 17}, {'<', 4, 0, 23}, {')', 6, 0, -31}, {']', 7, 0, -11}, {'}', 8, 0, -17}, {'>', 9, 0, -23}};
      int done = 0;
      for (int k = 0; k < 8 && !done; k++)
        if (c == BR[k].ch) { m->masks[2] += BR[k].m2; m->masks[3] += BR[k].m3; m->nestHash += (uint32_t)(int32_t)BR[k].nest; m->lastNest = 0; done = 1; }
      if (done) {}
      else if (c == 0xAB) m->masks[2] += 5;
      else if (c == 0xBB) m->masks[2] += 10;
      else if (c == '"') {
        m->masks[2] += 11;
        if (m->quoteLength == 0) m->quoteLength = 1;
        else { m->quoteLength = 0; m->State = ST_AfterQuote; m->ParseCtx = p8f_hash2(ST_AfterQuote, 0x100 | pC); }
      } else if (c == '/' || c == '-' || c == '+' || c == '*' || c == '=' || c == '%') m->masks[2] += 13;
      else if (c == '\\' || c == '|' || c == '_' || c == '@' || c == '&' || c == '^') m->masks[2] += 14;
    }
    if (c >= '0' && c <= '9') {
      m->numbers[0] = m->numbers[0] * 10 + (c & 0xF);
      m->numLength[0] = (uint8_t)imin(19, m->numLength[0] + 1);
      m->numHashes[0] = p8f_combine64(m->numHashes[0], c);
      m->expectedDigit = 0xFF;
      if (m->numLength[0] < m->numLength[1] && (m->pState == ST_ExpectDigit || ((m->numDiff & 3) == 0 && m->numLength[0] <= 1))) {
        const uint64_t expected = m->numbers[1] + (m->numMask & 3) - 2;
        uint64_t place = 1;
        for (int i = 0; i < m->numLength[1] - m->numLength[0]; i++) place *= 10;
        if (expected / place == m->numbers[0]) {
          place /= 10;
          m->expectedDigit = (uint8_t)((expected / place) % 10);
          m->State = ST_ExpectDigit;
        }
      } else {
        const uint8_t d = (uint8_t)RB(m->numLength[0] + 2);
        if (m->numLength[0] < 3 && RB(m->numLength[0] + 1) == ',' && d >= '0' && d <= '9') m->State = ST_ExpectDigit;
      }
      m->lastDigit = 0;
      m->masks[3] += 7;
    } else if (m->numbers[0] > 0) {
      m->numMask <<= 2; m->numMask |= 1u + (m->numbers[0] >= m->numbers[1]) + (m->numbers[0] > m->numbers[1]);
      const int32_t diff = (int32_t)(uint32_t)(m->numbers[0] - m->numbers[1]);
      const uint32_t adiff = diff < 0 ? 0u - (uint32_t)diff : (uint32_t)diff;
      m->numDiff <<= 2; m->numDiff |= (uint32_t)imin(3, (int)ilog2u(adiff));
      m->numbers[1] = m->numbers[0]; m->numbers[0] = 0;
      m->numHashes[1] = m->numHashes[0]; m->numHashes[0] = 0;
      m->numLength[1] = m->numLength[0]; m->numLength[0] = 0;
      m->cSegment->NumCount++; m->cSentence->NumCount++;
    }
  }
  if (m->lastNewLine == 1) m->firstChar = (m->lang_id != LANG_Unknown) ? c : (uint8_t)imin(c, 96);
  if (m->lastNest > 512) m->nestHash = 0;
  int lead = 0;
  while (lead < 8 && ((c >> (7 - lead)) & 1) != 0) lead++;
  if (m->UTF8Remaining > 0 && lead == 1) m->UTF8Remaining--;
  else m->UTF8Remaining = (lead != 1) ? ((c != 0xC0 && c != 0xC1 && c < 0xF5) ? (lead - (lead > 0)) : -1) : 0;
  const uint32_t comma = m->BytePos[','];
  m->maskPunct = (comma > m->BytePos['.']) | ((comma > m->BytePos['!']) << 1) | ((comma > m->BytePos['?']) << 2) | ((comma > m->BytePos[':']) << 3) |
                 ((comma > m->BytePos[';']) << 4);
}

static int text_contexts(TextM* m, const uint8_t* hist, uint32_t bmask, int pos, uint64_t* cx) {  /* SetContexts :3427-3518 */
  const uint8_t c = (uint8_t)RB(1), lc = (uint8_t)tolower(c), m2 = m->masks[2] & 0xF, column = (uint8_t)imin(0xFF, (int)m->lastNewLine);
  const int reading = (m->State == ST_ReadingWord);
  const uint16_t w = (uint16_t)((reading ? m->cWord->Hash[1] : m->pWord->Hash[1]) & 0xFFFF);
  const uint32_t h = (uint32_t)((reading ? m->cWord->Hash[1] : m->pWord->Hash[2]) * 271 + c);
  const uint32_t wl0 = m->wordLength[0], wl1 = m->wordLength[1], gap = m->wordGap, up = m->lastUpper, let = m->lastLetter, dig = m->lastDigit,
                 pun = m->lastPunct, nlgap = m->prevNewLine - m->lastNewLine;
  const P8Word *cW = m->cWord, *pW = m->pWord, *w2 = WORDS(m->lang_pid, 2), *w3 = WORDS(m->lang_pid, 3);
  const Sent* sn = m->cSentence;
  uint64_t i = (uint64_t)m->State << 6;
  int n = 0;
  cx[n++] = m->ParseCtx;
  cx[n++] = p8f_hash4(i++, cW->Hash[0], pW->Hash[0], (up < wl0) | ((dig < wl0 + gap) << 1));
  cx[n++] = p8f_hash5(i++, cW->Hash[1], w2->Hash[1], U(imin(10, (int)ilog2u((uint32_t)m->numbers[0]))),
                         (up < let + wl1) | ((let > 3) << 1) | ((let > 0 && wl1 < 3) << 2));
  cx[n++] = p8f_hash5(i++, cW->Hash[1] & 0xFFF, m->masks[1] & 0x3FF, w3->Hash[2], (dig < wl0 + gap) | ((up < let + wl1) << 1) | ((m->spaces & 0x7F) << 2));
  cx[n++] = p8f_hash4(i++, cW->Hash[1], pW->Hash[3], w2->Hash[3]);
  cx[n++] = p8f_hash4(i++, h & 0x7FFF, w2->Hash[1] & 0xFFF, w3->Hash[1] & 0xFFF);
  cx[n++] = p8f_hash4(i++, cW->Hash[1], c, (sn->VerbIndex < sn->WordCount) ? sn->lastVerb.Hash[1] : 0);
  cx[n++] = p8f_hash5(i++, pW->Hash[2], m->masks[1] & 0xFC, lc, gap);
  cx[n++] = p8f_hash5(i++, (let == 0) ? cW->Hash[1] : pW->Hash[1], c, m->cSegment->FirstWord.Hash[2], U(imin(3, (int)ilog2u(m->cSegment->WordCount + 1))));
  cx[n++] = p8f_hash4(i++, cW->Hash[1], c, SEGMENTS(1)->FirstWord.Hash[3]);
  cx[n++] = p8f_hash5(i++, U(imax(31, lc)), m->masks[1] & 0xFFC, (m->spaces & 0xFE) | (pun < let),
                         (m->maskUpper & 0xFF) | (U((0x100 | m->firstLetter) * (wl0 > 1)) << 8));
  cx[n++] = p8f_hash4(i++, column, U(imin(7, (int)ilog2u(up + 1))), ilog2u(pun + 1));
  cx[n++] = U((uint32_t)(column & 0xF8) | (m->masks[1] & 3) | ((uint32_t)(nlgap > 63) << 2) | ((uint32_t)imin(3, (int)let) << 8) | ((uint32_t)m->firstChar << 10) |
              ((uint32_t)(m->commas > 4) << 18) | ((uint32_t)(m2 >= 1 && m2 <= 5) << 19) | ((uint32_t)(m2 >= 6 && m2 <= 10) << 20) |
              ((uint32_t)(m2 == 11 || m2 == 12) << 21) | ((uint32_t)(up < column) << 22) | ((uint32_t)(dig < column) << 23) | ((uint32_t)(column < nlgap) << 24));
  cx[n++] = p8f_hash5(U((2 * column) / 3), (uint32_t)imin(13, (int)pun) + (pun > 16) + (pun > 32) + m->maskPunct * 16, ilog2u(up + 1), ilog2u(nlgap),
                         ((m->masks[1] & 3) == 0) | ((m2 < 6) << 1) | ((m2 < 11) << 2));
  cx[n++] = p8f_hash3(i++, column >> 1, m->spaces & 0xF);
  cx[n++] = p8f_hash5(m->masks[3] & 0x3F, U(imin((imax((int)wl0, 3) - 2) * (wl0 < 8), 3)), U(m->firstLetter * (wl0 < 5)), w & 0x3FF,
                         (c == RB(2)) | ((m->masks[2] > 0) << 1) | ((pun < wl0 + gap) << 2) | ((up < wl0) << 3) | ((dig < wl0 + gap) << 4) |
                             ((pun < 2 + wl0 + gap + wl1) << 5));
  cx[n++] = p8f_hash4(i++, w, c, m->numHashes[1]);
  cx[n++] = p8f_hash4(i++, w, c, U(llog_u((uint32_t)pos - m->WordPos[w]) >> 1));
  cx[n++] = p8f_hash4(i++, w, c, m->TopicDescriptor.Hash[1] & 0x7FFF);
  cx[n++] = p8f_hash4(i++, m->numLength[0], c, m->TopicDescriptor.Hash[1] & 0x7FFF);
  cx[n++] = p8f_hash4(i++, (let > 0) ? c : 0x100, m->masks[1] & 0xFFC, m->nestHash & 0x7FF);
  cx[n++] = p8f_hash4(i++, U(w * 17 + c), m->masks[3] & 0x1FF,
                         ((sn->VerbIndex == 0 && p8w_len(&sn->lastVerb) > 0) << 6) | ((wl1 > 3) << 5) | ((m->cSegment->WordCount == 0) << 4) |
                             ((sn->SegmentCount == 0 && sn->WordCount < 2) << 3) | ((pun >= let + wl1 + gap) << 2) | ((up < let + wl1) << 1) |
                             (up < wl0 + gap + wl1));
  cx[n++] = p8f_hash5(i++, c, pW->Hash[2], U(m->firstLetter * (wl0 < 6)), ((pun < wl0 + gap) << 1) | (pun >= let + wl1 + gap));
  const P8Word* wk = WORDS(m->lang_pid, 1 + (wl0 == 0));
  cx[n++] = p8f_hash4(i++, U(w * 23 + c), wk->Letters[wk->Start], U(m->firstLetter * (wl0 < 7)));
  cx[n++] = p8f_hash4(i++, column, m->spaces & 7, m->nestHash & 0x7FF);
  cx[n++] = p8f_hash4(i++, cW->Hash[1], (up < column) | ((up < wl0) << 1), U(imin(5, (int)wl0)));
  cx[n++] = m->masks[4];                                                             /* last 6 ASCII groups */
  cx[n++] = p8f_hash2((uint32_t)m->asciiMask, (uint32_t)(m->asciiMask >> 32));  /* last 12 */
  cx[n++] = m->asciiMask & ((1u << 20) - 1);
  cx[n++] = m->asciiMask & ((1u << 10) - 1);
  cx[n++] = p8f_hash2((m->asciiMask >> 5) & ((1u << 30) - 1), RB(1));
  cx[n++] = p8f_hash3((m->asciiMask >> 10) & ((1u << 30) - 1), RB(1), RB(2));
  cx[n++] = p8f_hash4((m->asciiMask >> 15) & ((1u << 30) - 1), RB(1), RB(2), RB(3));
  return n;
}

/* One coded bit of TextModel::Predict :3157-3185. hist/bmask/pos: the byte history ring (pos = bytes seen). out: the
 * ContextMap2's stretch-domain inputs; sel[8]: the mixer weight-set selectors (ranges 2048, 2048, 4096, 4096, 2048,
 * 2048, 4096, 8192); stats[6]: what Update leaves in ModelStats::Text (state, lastPunct, wordLength, boolmask,
 * firstLetter, mask). Returns the number of inputs written. */
int p8f_text_step(TextM* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* sel, uint32_t* stats) {
  uint64_t cx[40];
  int n = 0;
  if (bpos == 0) {
    text_update(m, hist, bmask, pos);
    n = text_contexts(m, hist, bmask, pos, cx);
  }
  int nout = 0;
  p8f_cm2_step(m->map, y, bpos, cx, n, out, &nout);
  const uint32_t grp0 = (bpos > 0) ? P8_ASCII_GROUP_C0[(1 << bpos) - 2 + (c0 & ((1 << bpos) - 1))] : 0;  /* :8274 */
  const uint32_t wl0 = m->wordLength[0], wl1 = m->wordLength[1], gap = m->wordGap, up = m->lastUpper, let = m->lastLetter, pun = m->lastPunct;
  sel[0] = (int)p8f_finalize64(p8f_hash3((m->lang_id != LANG_Unknown) ? 1 + IS_VOWEL[m->lang_id - 1]((char)RB(1)) : 0, m->masks[1] & 0xFF, U(c0)), 11);
  sel[1] = (int)p8f_finalize64(p8f_hash3(ilog2u(wl0 + 1), U(c0), (m->lastDigit < wl0 + gap) | ((up < let + wl1) << 1) | ((pun < wl0 + gap) << 2) | ((up < wl0) << 3)), 11);
  sel[2] = (int)p8f_finalize64(p8f_hash4(m->masks[1] & 0x3FF, grp0, up < wl0, up < let + wl1), 12);
  sel[3] = (int)p8f_finalize64(p8f_hash3(m->spaces & 0x1FF, grp0,
                                               (up < wl0) | ((up < let + wl1) << 1) | ((pun < let) << 2) | ((pun < wl0 + gap) << 3) | ((pun < let + wl1 + gap) << 4)), 12);
  sel[4] = (int)p8f_finalize64(p8f_hash3(U(m->firstLetter * (wl0 < 4)), U(imin(6, (int)wl0)), U(c0)), 11);
  sel[5] = (int)p8f_finalize64(p8f_hash4(p8w_at(m->pWord, 0), p8w_back(m->pWord, 0), U(imin(4, (int)wl0)), pun < let), 11);
  sel[6] = (int)p8f_finalize64(p8f_hash4(U(imin(4, (int)wl0)), grp0, up < wl0,
                                               (m->nestHash > 0) ? m->nestHash & 0xFF : 0x100u | (uint32_t)(m->firstLetter * (wl0 > 0 && wl0 < 4))), 12);
  sel[7] = (int)p8f_finalize64(p8f_hash3(grp0, m->masks[4] & 0x1F, (m->masks[4] >> 5) & 0x1F), 13);
  stats[0] = (uint32_t)m->State & 7; stats[1] = pun < 0x1F ? pun : 0x1F; stats[2] = wl0 < 0xF ? wl0 : 0xF;
  stats[3] = (m->lastDigit < wl0 + gap) | ((up < let + wl1) << 1) | ((pun < wl0 + gap) << 2) | ((up < wl0) << 3);
  stats[4] = m->firstLetter; stats[5] = m->masks[1] & 0xFF;
  return nout;
}
