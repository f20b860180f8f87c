/* oracle/fxcm_model.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The vendored fxcm model as cmix drives it (reference src/models/fxcmv1.cpp), assembled from the restated blocks
 * (oracle/fxcm_core.c, fxcm_maps.c, fxcm_match.c, fxcm_stem.c): update1 (:4758-4833: per-byte rate / dead-zone
 * adaptation, the 12 mixers' training, the failure history, six chained APMs and the final blend) around
 * modelPrediction (:3798-4757: a text parser over the byte stream -- 2/3/4-bit quantised byte streams, words and
 * stems, sentence / paragraph / stream word lists, brackets, quotes, first characters of lines, wiki tables and
 * columns, numbers, indirect histories -- feeding 81 context slots in 32 hashed maps, 7 direct maps, two match
 * models and a run map; ten 512-input mixers selected by hand-written contexts, two final mixers). What
 * FXCM::Predict() hands to cmix is every value passed to AddPrediction in call order: 431 columns (layer-0 columns
 * 3..433 of the cmix predictor). Inputs besides the coded bits: the LSTM's per-bit hints lstmpr / lstmex
 * (predictor.cpp:462-465) and, optionally, cmix's WRT dictionary: codeword bytes (128..255) are decoded to words that
 * feed the stemmer and switch the <text> / <math> / <pre> / <nowiki> states.
 * Pinned against the reference's own fxcmv1::Predictor in tests/test_oracle_fxcmcore.py. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fxcm_core.h"
#include "fxcm_stem.h"

typedef struct FxCm FxCm;
FxCm* orc_fx_cm_new(int kind, uint32_t m, int c, int s3, int which_sta, int cs4, int k, int u, int which_st2);
void fx_cm_set(FxCm* x, uint32_t cx);
void fx_cm_skip(FxCm* x);
int fx_cm_mix(FxCm* x, FxSink* s, int y, int bpos, int c0, int c1);
int fx_cm_skipmask(const FxCm* x);
typedef struct FxSparseMatch FxSparseMatch;
FxSparseMatch* orc_fx_sparsematch_new(void);
int fx_sparsematch_p(FxSparseMatch* m, FxSink* s, int bpos, int c0, const uint8_t* hist, uint32_t mask, int pos);

enum { LF = 10, ESCAPE = 12, SPACE = 32, HTLINK = 31, HTML = 30, APOSTROPHE = 39, QUOTATION = 34, FIRSTUPPER = 64, UPPER = 7, TEXTDATA = 96,
       COLON = 'J', SEMICOLON = 'K', LESSTHAN = 'L', EQUALS = 'M', GREATERTHAN = 'N', QUESTION = 'O', SQUAREOPEN = 91, SQUARECLOSE = 93,
       CURLYOPENING = 'P', VERTICALBAR = 'Q', CURLYCLOSE = 'R', WIKIHEADER = GREATERTHAN, WIKITABLE = '-' };   /* :1852-1876, :2004-2005 (WRT-swapped alphabet) */
enum { T_Verb = 1, T_Noun = 2, T_Adjective = 4, T_Plural = 8, T_PresentParticiple = (1 << 4) | 1, T_AdverbOfManner = 1 << 8, T_Suffix = 1 << 9,
       T_Prefix = 1 << 10, T_Male = 1 << 11, T_Female = 1 << 13, T_Article = 1 << 14, T_Conjunction = 1 << 15, T_Adposition = 1 << 16,
       T_Number = 1 << 17, T_ConjunctiveAdverb = 1 << 19 };
enum { BMASK = 0xffffff, CBMASK = 0xfff, FX_OUTPUTS = 431, MAXLEN = 62, MINLEN_RM = 3, LEN1 = 5, LEN2 = 7, LEN3 = 9 };
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a < b ? b : a; }

/* ---- vec<T,S> :1882-1930: a stack that wraps at its capacity and never really erases ---- */
typedef struct { int cxt[512]; int size; } Vec512;
#define VPUSH(v, cap, e) do { (v)->cxt[(v)->size++] = (e); (v)->size &= (cap) - 1; } while (0)
#define VPOP(v) do { if ((v)->size > 0) { (v)->cxt[(v)->size] = 0; (v)->size--; } } while (0)
#define VRESET(v) do { (v)->cxt[0] = 0; (v)->size = 0; } while (0)
#define VTOP(v) ((v)->cxt[(v)->size - 1])

/* ---- BracketContext<T> :1932-1998 ---- */
typedef struct { uint32_t context; Vec512 active, distance; const uint16_t* element; int count, doPop, limit, bits; uint32_t cxt, dst; } Bracket;
static void br_init(Bracket* b, const uint16_t* el, int n, int pop, int bits, int limit) {
  memset(b, 0, sizeof *b);
  b->element = el; b->count = n; b->doPop = pop; b->bits = bits; b->limit = limit;
}
static void br_reset(Bracket* b) { VRESET(&b->active); VRESET(&b->distance); b->context = b->cxt = b->dst = 0; }
static void br_update(Bracket* b, int byte) {
  int pop = 0;
  if (b->active.size != 0) {
    int end = 0;
    for (int i = 0; i < b->count; i += 2) if (b->element[i] == VTOP(&b->active) && b->element[i + 1] == byte) end = 1;
    if (end || VTOP(&b->distance) >= b->limit) { VPOP(&b->active); VPOP(&b->distance); pop = b->doPop; }
    else VTOP(&b->distance)++;
  }
  if (!pop) {
    int found = 0;
    for (int i = 0; i < b->count; i += 2) if (b->element[i] == byte) { found = 1; break; }
    if (found) { VPUSH(&b->active, 512, byte); VPUSH(&b->distance, 512, 0); }
  }
  if (b->active.size != 0) {
    const uint32_t tmask = (1u << b->bits) - 1;
    b->cxt = (uint32_t)VTOP(&b->active) & tmask;
    b->dst = (uint32_t)imin(VTOP(&b->distance), (1 << b->bits) - 1) & tmask;
    b->context = (1u << b->bits) * b->cxt + b->dst;
  } else b->context = b->cxt = b->dst = 0;
}
static int br_last(const Bracket* b) { return b->active.size > 1 ? b->active.cxt[b->active.size - 2] : 0; }
static const uint16_t kBrackets[8] = {'(', ')', CURLYOPENING, CURLYCLOSE, '[', ']', LESSTHAN, GREATERTHAN};
static const uint16_t kQuotes[4] = {APOSTROPHE, APOSTROPHE, QUOTATION, QUOTATION};
static const uint16_t kFchar[20] = {FIRSTUPPER, LF, TEXTDATA, LF, COLON, LF, LESSTHAN, GREATERTHAN, EQUALS, LF, SQUAREOPEN, SQUARECLOSE, CURLYOPENING, CURLYCLOSE,
                                    '*', LF, VERTICALBAR, LF, HTLINK, LF};
static const uint16_t kHtml[2] = {'&' * 256 + 'L', '&' * 256 + 'N'};

/* ---- ColumnContext :2000-2155: the last four lines, and cell positions of wiki tables / page headers ---- */
typedef struct { uint32_t linepos; uint8_t fc; uint8_t bytes[2048]; int size; } Line;
typedef struct { uint32_t cxt[32]; int size; } CellRow;
typedef struct {
  Line col[4]; CellRow cell[4];
  int rows, cellCount, cells, abovecellpos, abovecellpos1, NL, isTemp, limit;
  uint8_t nlChar;
} Columns;
static int col_len(const Columns* c, int i, int l) { return imin(l ? l : c->limit, c->col[(c->rows - i) & 3].size + 1); }
static int col_lastfc(const Columns* c, int i) { return c->col[(c->rows - i) & 3].fc; }
static int col_b(const Columns* c, int i, int j) {  /* colb: index -1 (first byte of a line, j = 1) lands on a padding byte of the reference's struct: 0 */
  const int k = col_len(c, 0, 0) - (1 + j);
  return col_len(c, 0, 0) < col_len(c, i, 0) && k >= 0 ? c->col[(c->rows - i) & 3].bytes[k] : 0;
}
static int cells_count(const Columns* c) { return c->cell[(c->cells - 1) & 3].size; }
static int cell_pos(const Columns* c, int id) { return (int)c->cell[(c->cells - 1) & 3].cxt[imin(cells_count(c) - 1, id)]; }
static void cell_new_row(Columns* c, int blpos) {
  c->cells = (c->cells + 1) & 3;
  CellRow* r = &c->cell[c->cells];
  r->cxt[0] = 0; r->size = 0;
  r->cxt[r->size++] = (uint32_t)blpos; r->size &= 31;
  c->cellCount = c->abovecellpos = c->abovecellpos1 = 0;
}
static void cell_step(Columns* c, int newcell) {
  if (c->abovecellpos) { c->abovecellpos++; if (c->abovecellpos > c->abovecellpos1) c->abovecellpos = c->abovecellpos1 = 0; }
  if (newcell && cells_count(c) > 0) { c->abovecellpos = cell_pos(c, c->cellCount - 1); c->abovecellpos1 = cell_pos(c, c->cellCount); }
}
static void col_update(Columns* c, int byte, uint32_t b2, int blpos, int isPre) {
  if (b2 == ((CURLYOPENING << 16) + (CURLYOPENING << 8) + VERTICALBAR)) c->nlChar = WIKITABLE;
  else if (b2 == ((VERTICALBAR << 16) + (CURLYCLOSE << 8) + CURLYCLOSE)) { c->nlChar = LF; for (int i = 0; i < 4; i++) { c->cell[i].cxt[0] = 0; c->cell[i].size = 0; } }
  if (byte != CURLYOPENING && (b2 & 0xff00) == (CURLYOPENING << 8) && (b2 & 0xff0000) != (CURLYOPENING << 16)) c->isTemp = 1;
  else if (c->isTemp && byte == CURLYCLOSE) c->isTemp = 0;
  c->NL = 0;
  Line* ln = &c->col[c->rows];
  ln->bytes[ln->size++] = (uint8_t)byte; ln->size &= 2047;
  if (byte == LF) {
    c->rows = (c->rows + 1) & 3;
    ln = &c->col[c->rows];
    ln->bytes[0] = 0; ln->size = 0;
    ln->fc = 0;
    ln->linepos = (uint32_t)(blpos - 1);
  } else if (col_len(c, 0, 0) == 2) {
    ln->fc = (uint8_t)imin(byte, TEXTDATA);
    c->NL = 1;
    if (ln->fc == GREATERTHAN && !isPre) c->nlChar = WIKIHEADER;
    if (ln->fc == SQUAREOPEN && c->nlChar == WIKIHEADER) c->nlChar = LF;
  }
  if (c->nlChar == WIKITABLE) {  /* {| |- | || |} */
    if ((b2 & 0xffff) == (WIKITABLE + VERTICALBAR * 256)) cell_new_row(c, blpos);
    int newcell = 0;
    if ((b2 & 0xffff) == (VERTICALBAR + VERTICALBAR * 256) || (b2 & 0xffff00) == ((VERTICALBAR + LF * 256) * 256)) {
      CellRow* r = &c->cell[c->cells];
      r->cxt[r->size++] = (uint32_t)blpos; r->size &= 31;
      c->cellCount++; newcell = 1;
    }
    cell_step(c, newcell);
  }
  if (c->nlChar == WIKIHEADER) {  /* the header block of a filtered wiki page: one '>' per field */
    if ((b2 & 0xffff) == (WIKIHEADER + LF * 256)) cell_new_row(c, blpos);
    else {
      int newcell = 0;
      if ((b2 & 0xff) == WIKIHEADER) { CellRow* r = &c->cell[c->cells]; r->cxt[r->size++] = (uint32_t)blpos; r->size &= 31; c->cellCount++; newcell = 1; }
      cell_step(c, newcell);
    }
  }
}

/* ---- WordsContext :2157-2274: the words of the current sentence / paragraph / stream with their surroundings ---- */
typedef struct {
  uint16_t sbytes[256]; uint32_t type[256], stem[256]; uint8_t capital[256];
  int n_sbytes, n_type, n_stem, n_capital;
  uint32_t fword, ftype; uint8_t pbyte; int wordcount, upper, ref;
} Words;
static void wc_reset(Words* w) {
  w->sbytes[0] = 0; w->n_sbytes = 0; w->type[0] = 0; w->n_type = 0; w->stem[0] = 0; w->n_stem = 0; w->capital[0] = 0; w->n_capital = 0;
  w->fword = w->ftype = 0; w->pbyte = 0; w->wordcount = w->upper = w->ref = 0;
}
static void wc_set(Words* w, int b, int a) { w->pbyte = (uint8_t)b; w->upper = a; }
static void wc_update(Words* w, uint32_t word, int b, uint32_t t, uint32_t s) {
  if (w->fword == 0) w->fword = word;
  w->sbytes[w->n_sbytes++] = (uint16_t)(w->pbyte * 256 + b); w->n_sbytes &= 255;
  w->type[w->n_type++] = t; w->n_type &= 255;
  w->stem[w->n_stem++] = s; w->n_stem &= 255;
  w->capital[w->n_capital++] = (uint8_t)w->upper; w->n_capital &= 255;
  w->pbyte = 0; w->wordcount++;
  if (w->ftype == 0 && t) w->ftype = t;
}
static void wc_remove(Words* w) {
  if (w->n_stem) {
    if (w->n_sbytes > 0) { w->sbytes[w->n_sbytes] = 0; w->n_sbytes--; }
    if (w->n_type > 0) { w->type[w->n_type] = 0; w->n_type--; }
    if (w->n_stem > 0) { w->stem[w->n_stem] = 0; w->n_stem--; }
    if (w->n_capital > 0) { w->capital[w->n_capital] = 0; w->n_capital--; }
    w->wordcount--;
  }
}
static uint32_t wc_word(const Words* w, int i) { return w->n_stem >= i ? w->stem[w->n_stem - i] : 0; }
static uint32_t wc_sbytes(const Words* w, int i) { return w->n_sbytes >= i ? w->sbytes[(w->n_sbytes - i) & 255] : 0; }
static uint32_t wc_type(const Words* w, int i) { return w->n_type >= i ? w->type[w->n_type - i] : 0; }
static uint32_t wc_capital(const Words* w, int i) { return w->n_capital >= i ? w->capital[w->n_capital - i] : 0; }
static uint32_t wc_last(const Words* w, int j, uint32_t t, int or_zero) {  /* Last / LastIf */
  if (t == 0) return wc_word(w, j);
  if (w->n_type >= j)
    for (int i = j; i < w->n_type; i++) if (wc_type(w, i) & t) return wc_word(w, i);
  return or_zero ? 0 : wc_word(w, j);
}
static void wc_remove_words(Words* w, int len, int c, int d, int left) {  /* removeWordsL / removeWordsR */
#define SIDE(i) (left ? (wc_sbytes(w, i) >> 8) : (wc_sbytes(w, i) & 0xff))
  if ((wc_sbytes(w, 1) & 0xff) == (uint32_t)d)
    for (int i = 1; i < len; i++)
      if (SIDE(i) == (uint32_t)c) {
        while (SIDE(1) != (uint32_t)c) wc_remove(w);
        wc_remove(w);
        break;
      }
#undef SIDE
}

/* ---- MatchModel2 :3420-3676: up to four match candidates found through order-9 / 7 / 5 and last-word hashes ---- */
typedef struct { uint32_t length, index, lengthBak, indexBak; uint8_t expectedByte, delta; } MatchInfo;
typedef struct { uint32_t pos[4]; } MatchSlot;

typedef struct FxModel {
  /* BlockData x :203-219 */
  int y, c0, bpos, blpos; uint32_t c4;
  FxSink in1;                     /* mxInputs1: values persist between bits beyond the count */
  int16_t in2[32]; int n2;        /* mxInputs2 */
  /* globals :3222-3279 */
  uint32_t t[14];
  int c1, c2, c3;
  uint8_t words, spaces, numbers;
  uint32_t word0, word00, word1, word2, word3, wshift, x4, x5, isMatch, firstWord, linkword, senword;
  uint32_t number0, number1, numlen0, numlen1, mybenum;
  uint32_t FcIdx, BrFcIdx, AH1, AH2, fails, failz, failcount;
  int nl, nl1, col, fc;
  uint32_t t1[0x100], t2[0x10000];
  int wp[0x10000];
  uint16_t* ind3;
  uint32_t indirectBrByte, indirectByte, indirectWord0Pos, indirectWord, u8w, context1_ind3, cxtind3, lastWT;
  uint32_t o3bState, n3bState, stream3bR, stream3b, stream3bMask, stream3bMask1, stream3bRMask1, stream3bRMask2;
  uint32_t o2bState, n2bState, stream2bR, stream2b, stream2bMask, o4bState, n4bState, stream4bR, stream4b;
  int ordX, ordW;
  uint8_t* buffer; int pos;
  uint8_t cwbuf[0x1000]; int cwpos;
  FxWord StemWords[4]; int cWord, pWord, StemIndex;
  int dcw, dcwl;
  uint32_t sVerb;
  int lastArt, isNowiki, isText, isMath, isPre, isParagraph, utf8left, deccode;
  int pr, rate, sscmrate;
  int lstmpr, lstmex;
  /* WRT dictionary (:352-437): the decoded word of the last codeword and the one before the last ':' */
  char** dictW; int sizeDict, lastCW; const char *so, *colonstr;
  /* model components :3281-3311 */
  FxStateMap1 smA[3]; FxSscm scmA[7]; FxMixer* mxA[12];
  FxCm *cmC[6], *cmC1[8], *cmC2[18];
  FxApm* apm[6]; FxRcm rcmA;
  Bracket brcxt, qocxt, fccxt, htcxt;
  Columns colcxt;
  Words worcxt, worcxt1, worcxt2;
  FxSparseMatch* smatch;
  MatchInfo cand[4]; uint32_t nActive; MatchSlot* mhash; uint32_t mhashmask; uint32_t mctx[3];
} FxModel;

static uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) {  /* hash :2276-2279 (c defaults to 0xffffffff) */
  const uint32_t h = a * 110002499u + b * 30005491u + c * 50004239u;
  return h ^ h >> 9 ^ a >> 3 ^ b >> 3 ^ c >> 4;
}
static int char_swap(int c) {  /* charSwap :2281-2287: undo cmix's WRT character swap */
  if (c >= '{' && c < 127) c += 'P' - '{';
  else if (c >= 'P' && c < 'T') c -= 'P' - '{';
  else if ((c >= ':' && c <= '?') || (c >= 'J' && c <= 'O')) c ^= 0x70;
  if (c == 'X' || c == '`') c ^= 'X' ^ '`';
  return c;
}
static const uint8_t kFcy[128] = {['"'] = 5, ['\''] = 6, ['('] = 1, ['L'] = 4, ['P'] = 2, ['['] = 3};               /* :3680-3689 */
static const uint8_t kFcq[128] = {['*'] = 6, ['@'] = 1, ['J'] = 3, ['L'] = 4, ['M'] = 5, ['P'] = 2, ['Q'] = 7, ['['] = 2, ['`'] = 2};  /* :3691-3700 */
static const uint32_t kPrimes[14] = {0, 257, 251, 241, 239, 233, 229, 227, 223, 211, 199, 197, 193, 191};
#include "fxcm_tables.h"

#define BUF(i) ((int)m->buffer[((uint32_t)m->pos - (uint32_t)(i)) & BMASK])
#define BUFR(i) ((int)m->buffer[(uint32_t)(i) & BMASK])
#define BUFFER1(i) ((int)m->cwbuf[((uint32_t)m->cwpos - (uint32_t)(i)) & CBMASK])

static int get_wt(uint32_t t) {  /* getWT :3706-3722 */
  if (t & T_Verb) return 1;
  if (t & T_Noun) return 2;
  if (t & T_Adjective) return 3;
  if (t & T_Male) return 4;
  if (t & T_Female) return 5;
  if (t & T_Article) return 6;
  if (t & T_Conjunction) return 7;
  if (t & T_Adposition) return 8;
  if (t & T_ConjunctiveAdverb) return 9;
  if (t & T_AdverbOfManner) return 11;
  if (t & T_Suffix) return 12;
  if (t & T_Prefix) return 13;
  if (t & T_Plural) return 10;
  return t ? 14 : 15;
}
/* setbuf + setbufstem :3724-3772: the un-swapped text goes to a 4 KB buffer; letters build the current word, any other
 * character closes it: stem, classify, file it in the sentence / paragraph / stream word lists */
static void set_buf(FxModel* m, int ch) {
  const char c = (char)ch;
  m->cwbuf[m->cwpos & CBMASK] = (uint8_t)c;
  m->cwpos++;
  FxWord* cw = &m->StemWords[m->cWord];
  if ((c >= 'a' && c <= 'z') || (c == APOSTROPHE && m->c2 != APOSTROPHE) || (c == '-' && p8w_len(&cw->w) > 0)) { fxw_add(cw, c); return; }
  if (p8w_len(&cw->w) > 0 && c == SQUARECLOSE && m->fccxt.cxt != HTLINK && m->isParagraph) return;  /* [dog]s stays one word */
  if (p8w_len(&cw->w) == 0) return;
  fx_stem(cw, m->blpos);
  m->StemIndex = (m->StemIndex + 1) & 3;
  m->pWord = m->cWord;
  m->cWord = m->StemIndex;
  memset(&m->StemWords[m->cWord], 0, sizeof(FxWord));
  FxWord* pw = &m->StemWords[m->pWord];
  if (pw->Type & T_Verb) m->sVerb = pw->Hash;
  if (m->lastArt) pw->Type |= T_Noun;
  m->lastArt = (pw->Type == T_Article && BUFFER1(5) == SPACE && BUFFER1(4) == 't' && BUFFER1(3) == 'h' && BUFFER1(2) == 'e');
  uint32_t whash = m->isMath ? m->word0 : pw->Hash;
  m->lastWT = m->lastWT * 16 + (uint32_t)get_wt(pw->Type);
  if (pw->Type == T_Number && wc_type(&m->worcxt, 1) == T_Number) {  /* multi-word numbers become one entry */
    const uint32_t sb = wc_sbytes(&m->worcxt, 1);
    whash = whash + wc_word(&m->worcxt, 1);
    wc_remove(&m->worcxt);
    wc_set(&m->worcxt, (int)(sb >> 8), 0);
  }
  wc_update(&m->worcxt, m->word0, m->c1, pw->Type, whash);
  if ((pw->Type & (T_Conjunction + T_Article + T_Male + T_Female + T_Number + T_ConjunctiveAdverb)) == 0 && m->brcxt.cxt != LESSTHAN)
    wc_update(&m->worcxt1, m->word0, m->c1, pw->Type, whash);
  if ((pw->Type & (T_Conjunction + T_Article + T_Male + T_Female + T_Adposition + T_Number + T_AdverbOfManner + T_ConjunctiveAdverb)) == 0 &&
      m->brcxt.cxt != LESSTHAN && pw->Type)
    wc_update(&m->worcxt2, m->word0, m->c1, pw->Type, whash);
}

static const char kEmpty[1] = {0};
static int decode_codeword(int cw) {  /* decodeCodeWord :389-411: 1 to 3 codeword bytes (128..255) -> dictionary index */
  enum { d1 = 80, d2 = 32 };
#define SYM(c) ((c) >= 128 ? (c) - 128 : 0)   /* codeword2sym after dosym() :424-431 */
  int c = cw & 255, i;
  if (SYM(c) < d1) return SYM(c);
  i = d1 * (SYM(c) - d1);
  c = (cw >> 8) & 255;
  if (SYM(c) < d1) return i + SYM(c) + d1;
  i = (i - d1 * d2) * d2;
  i += d1 * (SYM(c) - d1);
  c = (cw >> 16) & 255;
  return i + SYM(c) + 80 * 49;
#undef SYM
}
static void set_buf(FxModel* m, int ch);
static void proc_word(FxModel* m) {  /* procWord :3782-3795: a finished codeword is decoded and its letters go through the word parser */
  if (m->dcwl > 0) {
    if (m->dcwl == 2) m->dcw = (m->dcw / 256) + (m->dcw & 255) * 256;
    if (m->dcwl == 3) m->dcw = ((m->dcw / 256) / 256) + (m->dcw & 0xff00) + (m->dcw & 255) * 256 * 256;
    if (m->dcwl > 3) return;
    if (m->dictW) {
      const int j = decode_codeword(m->dcw);
      if (j > 0 && j < m->sizeDict) { m->lastCW = j; m->so = m->dictW[j]; }
    }
    m->dcw = m->dcwl = 0;
    for (const char* p = m->so; *p; ++p) set_buf(m, *p);   /* an undecodable codeword replays the previous word (:3790-3794) */
  }
}
#define CM(k) fx_cm_set(m->cmC[k],
#define CM1(k) fx_cm_set(m->cmC1[k],
#define CM2(k) fx_cm_set(m->cmC2[k],
#define SKIP(map) fx_cm_skip(map)

/* the byte-boundary half of modelPrediction :3802-4600 */
static void byte_update(FxModel* m) {
  int i;
  uint32_t h = 0, j;
  uint32_t c4 = m->c4;
  m->c3 = m->c2; m->c2 = m->c1; m->c1 = (int)(c4 & 0xff);
  int c1 = m->c1, c2 = m->c2; const int c3 = m->c3;
  m->n2bState = FX_WRT_2B[c1]; m->n3bState = FX_WRT_3B[c1]; m->n4bState = FX_WRT_4B[c1];
  m->stream2b = m->stream2b * 4 + m->n2bState;
  m->stream4b = m->stream4b * 16 + m->n4bState;
  m->buffer[m->pos & BMASK] = (uint8_t)c1;
  m->pos++;
  if (c2 == GREATERTHAN && m->isText) {  /* the line after <text ...> starts a paragraph */
    m->isText = 0;
    if (c1 == APOSTROPHE || c1 == FIRSTUPPER) {
      col_update(&m->colcxt, LF, 0, m->blpos, m->isPre);
      wc_reset(&m->worcxt); wc_reset(&m->worcxt1);
      m->fc = m->isParagraph = 0; m->firstWord = 0;
      m->nl1 = m->nl; m->nl = m->pos - 2;
    }
  }
  col_update(&m->colcxt, c1, c4 & 0xffffff, m->blpos, m->isPre);
  if (c1 < 'a') br_update(&m->brcxt, c1);
  if (c1 == SPACE && c2 == LESSTHAN) br_update(&m->brcxt, GREATERTHAN);
  CM(4) (m->brcxt.context << 8) + (uint32_t)c1);
  br_update(&m->qocxt, c1);
  if (m->htcxt.cxt && c2 == 'L' && (c1 == SPACE || c1 == '!' || c1 < 128)) br_update(&m->htcxt, '&' * 256 + 'N');
  br_update(&m->htcxt, (int)(c4 & 0xffff));

  if (c1 == '$' || c1 == SQUARECLOSE || c1 == VERTICALBAR || c1 == ')' || c1 == SQUAREOPEN) {  /* these end an order-X context */
    if (c1 != c2) for (i = 13; i > 0; --i) m->t[i] = m->t[i - 1] * kPrimes[i];
    m->x4 = (m->x4 << 8) + (uint32_t)c2;
    m->stream2b = m->stream2b * 4 + m->n2bState;
    m->stream2bR = (m->stream2bR << 2) + m->n2bState;
    m->stream3bR = (m->stream3bR << 3) + m->n3bState;
  }
  m->x4 = (m->x4 << 8) + (uint32_t)c1;
  for (i = 13; i > 0; --i) m->t[i] = m->t[i - 1] * kPrimes[i] + (uint32_t)c1 + (uint32_t)i * 256;
  if (m->fc == SPACE && c1 == SPACE) { SKIP(m->cmC2[0]); SKIP(m->cmC2[0]); SKIP(m->cmC2[0]); }
  else for (i = 3; i < 6; ++i) CM2(0) m->t[i]);
  CM2(1) m->t[6]);
  CM2(2) m->t[8]);
  CM2(3) m->t[13]);

  m->words = (uint8_t)(m->words << 1); m->spaces = (uint8_t)(m->spaces << 1); m->numbers = (uint8_t)(m->numbers << 1);
  j = (uint32_t)c1;
  if ((j - 'a') <= ('z' - 'a') || (c1 > 127 && c2 != ESCAPE)) {   /* a letter (or a WRT codeword byte) */
    if (m->word0 == 0) {
      if (m->isMath && c2 == '/' && c3 == LESSTHAN) m->isMath = 0;
      int reChar = c2;
      if (c2 == FIRSTUPPER || c2 == UPPER) {
        if (c3 != APOSTROPHE) reChar = c3;
        else if (BUF(4) != APOSTROPHE) reChar = BUF(4);
        else if (BUF(5) != APOSTROPHE) reChar = BUF(5);
        else if (BUF(6) != APOSTROPHE) reChar = BUF(6);
        else reChar = c3;
      } else if (c2 == '/' && c3 == LESSTHAN) reChar = c3;
      wc_set(&m->worcxt, reChar & 255, c2 == FIRSTUPPER ? 1 : 0);
      wc_set(&m->worcxt1, reChar & 255, 0);
    }
    m->words |= 1;
    m->word0 = m->word0 * 2104 + j;
    m->word00 = m->word0;
    h = m->word0 * 271; m->u8w = 0;
    if (m->brcxt.cxt == SQUAREOPEN && m->fccxt.cxt != HTLINK && m->fc != HTML) m->linkword = m->linkword * 2104 + j;
    if (m->isParagraph && m->fccxt.cxt != HTLINK && !m->colcxt.isTemp) m->senword = m->senword * 2104 + j;
    const int word3bit = m->words & 7;
    if ((word3bit == 5 && c2 == APOSTROPHE) || (word3bit == 1 && c3 == SQUARECLOSE && c2 == APOSTROPHE) || (word3bit == 1 && (m->numbers & 4) && c2 == APOSTROPHE))
      br_update(&m->qocxt, (int)m->qocxt.cxt);
    if (c1 > 127) {   /* a codeword byte: try to decode what there is of it */
      m->dcw = (int)((uint32_t)m->dcw * 256 + (uint32_t)c1); m->dcwl++;
      if (m->blpos > 6) {
        int dcw2 = 0;
        if (m->dcwl == 2) dcw2 = (m->dcw / 256) + (m->dcw & 255) * 256;
        else if (m->dcwl == 3) dcw2 = ((m->dcw / 256) / 256) + (m->dcw & 0xff00) + (m->dcw & 255) * 256 * 256;
        const int k = m->dictW ? decode_codeword(dcw2) : 0;
        if (k > 0 && k < m->sizeDict) m->deccode = k;
      }
    } else if (m->dcw) { proc_word(m); if (m->blpos < 448131719) m->deccode = m->lastCW; }
    if (c1 == 10 || c1 == 9 || (c1 > 31 && c1 < 128)) set_buf(m, char_swap(c1));
  } else {
    if (m->word0) { proc_word(m); if (m->blpos < 448131719) m->deccode = m->lastCW; }
    else m->deccode = (int)(0x10000 + (m->stream2b & 0xffff));
    if (c1 == 10 || c1 == 9 || (c1 > 31 && c1 < 128)) set_buf(m, char_swap(c1));
    if (c1 >= '0' && c1 <= '9') {   /* numbers: (number), (number.number), (number,number) */
      m->numbers = (uint8_t)(m->numbers + 1);
      if ((m->numbers & 4) && c2 == ',') { m->number0 = m->number1; m->number1 = 0; m->numlen0 = m->numlen1; m->numlen1 = 0; }
      if (m->mybenum && m->numlen1 <= 2) { m->number0 = m->number1; m->number1 = 0; m->numlen0 = m->numlen1; m->numlen1 = 0; }
      m->number0 = m->number0 * 10 + (uint32_t)(c1 & 0x0f);
      m->numlen0 = (uint32_t)imin(19, (int)(m->numlen0 + 1)); m->mybenum = 0;
    } else {
      if (m->numlen0 || (m->numbers & 0xf) == 0) { m->number1 = m->number0; m->numlen1 = m->numlen0; m->number0 = m->numlen0 = 0; }
      if (m->numlen1 <= 2 && m->numlen1 && (m->numbers & 5) == 5 && m->numlen0 == 0 && c2 == '.') m->mybenum = 2;
      else if (m->numlen1 <= 2 && m->numlen1 && (m->numbers & 2) && m->numlen0 == 0 && c1 == '.') m->mybenum = 1;
      else if (m->mybenum == 1 && c1 != '.') m->mybenum = 0;
    }
    const int word3bit = m->words & 7;
    if ((word3bit == 4 && c1 == SPACE && c2 == APOSTROPHE) || (c1 == FIRSTUPPER && (m->numbers & 4) && c2 == APOSTROPHE) ||
        (word3bit == 4 && c1 == FIRSTUPPER && c2 == APOSTROPHE) || (word3bit == 4 && (m->numbers & 1) && c2 == APOSTROPHE))
      br_update(&m->qocxt, (int)m->qocxt.cxt);
    if (m->word00 && m->fccxt.cxt != SQUAREOPEN) m->word00 = 0;
    Words* wc = &m->worcxt;
    if (m->word0) {   /* a word just ended */
      if (m->blpos > 463139793 || (m->StemWords[m->pWord].Type & (T_ConjunctiveAdverb + T_Conjunction)) == 0) {
        m->word3 = m->word2 * 47; m->word2 = m->word1 * 53; m->word1 = m->word0 * 83;
      }
      if (wc_type(wc, 1) == T_Number) { m->stream3bR = (m->stream3bR << 7) + 1; m->stream3b = (m->stream3b << 7) + 1; }
      if (m->firstWord == 0 && m->fccxt.cxt != SQUAREOPEN) m->firstWord = m->word0;
      if (wc_type(wc, 1) & T_Conjunction) { m->stream3bR <<= 7; m->stream3b <<= 7; if (m->isParagraph) m->senword = 0; }
      if (wc_type(wc, 1) & T_Article) { m->stream3bR = (m->stream3bR << 7) + 2; m->stream3b = (m->stream3b << 7) + 2; }
      if ((wc_type(wc, 1) & T_Adposition) || (m->isParagraph && (wc_type(wc, 1) & T_PresentParticiple))) {
        m->stream2bR = (m->stream2bR << 2) + (m->stream2bR & 3);
        m->stream2b = (m->stream2b << 2) + (m->stream2b & 3);
      }
      if ((wc_type(wc, 1) & T_AdverbOfManner) && m->isParagraph) wc_remove(wc);
      if ((wc_type(wc, 1) & T_Noun) && (wc_type(wc, 2) & T_Article)) {   /* article + noun become one entry */
        m->stream3bR = (m->stream3bR << 6) + 1; m->stream3b = (m->stream3b << 6) + 1;
        const uint32_t sb = wc_sbytes(wc, 1), w = wc_word(wc, 1), t = wc_type(wc, 1), ca = wc_capital(wc, 1);
        wc_remove(wc); wc_remove(wc);
        wc_set(wc, (int)(sb >> 8), (int)ca);
        wc_update(wc, w, c1, t, w);
      }
      m->stream3bRMask2 = m->stream3bRMask1;
      m->stream3bMask1 = m->stream3bMask;
      m->stream3bMask = m->stream2bMask = m->stream3bRMask1 = 0;
    } else if (c1 == VERTICALBAR && m->colcxt.isTemp) {
      const uint32_t sb = wc_sbytes(wc, 1), w = wc_word(wc, 1), t = wc_type(wc, 1), ca = wc_capital(wc, 1);
      wc_remove(wc);
      wc_set(wc, (int)(sb >> 8), (int)ca);
      wc_update(wc, w, c1, t, w);
    }
    /* <text>, <nowiki>, <math>, <pre>, </page> boundaries, recognised through the last decoded dictionary word (:4028-4045) */
#define SO_IS(str) (strcmp(m->so, str) == 0)
    const int lt = char_swap(LESSTHAN);
    if (BUFFER1(6) == lt && BUFFER1(5) == 't' && !m->isText && c1 == SPACE && SO_IS("text")) { m->isText = 1; m->so = kEmpty; }
    if (BUFFER1(8) == lt && !m->isNowiki && SO_IS("nowiki")) m->isNowiki = 1;
    else if (BUFFER1(9) == '/' && c1 == GREATERTHAN && m->isNowiki && SO_IS("nowiki")) { m->isNowiki = m->isPre = 0; m->so = kEmpty; }
    if (m->isMath && ((c1 == SPACE && col_lastfc(&m->colcxt, 0) != COLON) || c1 == ',') && c2 == GREATERTHAN && SO_IS("math")) { m->isMath = 0; m->so = kEmpty; }
    if (m->isMath && c1 == '/' && c2 == LESSTHAN && c3 == GREATERTHAN && BUFFER1(4) == 'h') { m->isMath = 0; m->so = kEmpty; }
    if (!m->isNowiki && BUFFER1(6) == lt && BUFFER1(5) == 'm' && !m->isMath && c1 != '.' && BUFFER1(7) != '&' && BUFFER1(8) != '&' && SO_IS("math")) m->isMath = 1;
    else if (BUFFER1(6) == '/' && (c1 == GREATERTHAN || c1 == '&') && m->isMath && SO_IS("math")) { m->isMath = 0; m->so = kEmpty; }
    if (BUFFER1(5) == lt && c1 == GREATERTHAN && BUFFER1(4) == 'p' && !m->isPre && SO_IS("pre")) { m->isPre = 1; m->so = kEmpty; }
    else if (BUFFER1(5) == '/' && c1 == GREATERTHAN && BUFFER1(4) == 'p' && SO_IS("pre")) { m->isPre = 0; m->so = kEmpty; }
    if (BUFFER1(6) == '/' && c1 == GREATERTHAN && BUFFER1(5) == 'p' && SO_IS("page")) m->isPre = m->isMath = m->isNowiki = 0;
#undef SO_IS

    m->wp[m->word0 & 0xffff] = m->pos;
    m->word0 = h = 0;
    if (m->linkword && c1 == COLON) m->linkword = 0;
    if (c1 == '-' && c2 == SPACE) { wc_reset(&m->worcxt1); m->sVerb = 0; }
    if (c1 == SPACE) m->spaces++;
    else if (c1 == LF) {
      m->fc = m->isParagraph = 0; m->firstWord = 0; m->lastWT = 0;
      m->nl1 = m->nl; m->nl = m->pos - 1;
      m->stream3bR <<= 7;
      m->stream2b |= 0x3fc;
      m->words = 0xfc;
      wc_reset(&m->worcxt); wc_reset(&m->worcxt1);
      m->stream2bR <<= 2;
      m->stream4b |= 0xfff0;
      if (c2 == LF) m->isNowiki = 0;
    } else if (c1 == '.' || c1 == ')' || c1 == QUESTION) {
      m->lastWT *= 16;
      m->stream3bR <<= 7; m->stream3b <<= 7;
      m->words |= 0xfe;
      m->x5 = (m->x5 << 8) + (c4 & 0xff);
      m->stream2b |= 204;
      m->stream4b = ((m->stream4b & 0xffff0) << 8) + (m->stream4b & 0xf);
      m->stream2bR &= 0xffffffc0;
      if (c1 == '.') {
        m->wshift = 1;
        if (!(m->fccxt.cxt == SQUAREOPEN || m->fccxt.cxt == '(' || m->colcxt.nlChar == WIKITABLE || col_lastfc(&m->colcxt, 0) == '*')) wc_reset(&m->worcxt);
        m->senword = 0;
      }
      if (c1 == ')') m->senword = 0;
    } else if (c1 == ',') { m->words |= 0xfc; m->senword = 0; }
    else if (c1 == '(') m->senword = 0;
    else if (c1 == SEMICOLON) wc_reset(&m->worcxt);
    else if (c1 == COLON) {
      m->stream3b = (m->stream3b & 0xfffffff8) + 4;
      m->stream2b |= 12;
      m->x5 = (m->x5 << 8) + (c4 & 0xff);
      m->senword = 0;
    } else if (c1 == CURLYCLOSE || c1 == CURLYOPENING) {
      m->words |= 0xfc;
      m->stream3bR &= 0xffffffc0;
      m->x5 = (m->x5 << 8) + (c4 & 0xff);
      m->stream3b = (m->stream3b & 0xfffffff8) + 3;
    } else if (c1 == SQUARECLOSE) { m->stream3b = (m->stream3b & 0xfffffff8) + 3; m->linkword = 0; }
    else if (c1 == LESSTHAN || c2 == '&') m->words |= 0xfc;
    else if (c1 == '-' && col_lastfc(&m->colcxt, 0) == '*' && m->brcxt.cxt != SQUAREOPEN && m->isParagraph == 0) { m->isParagraph = 1; m->fc = FIRSTUPPER; }
    else if (c1 == EQUALS) {
      m->stream3b = (m->stream3b & 0xfffffff8) + 4;
      m->c2 = c2 = '.';
      m->words = (uint8_t)(m->words * 2);
    }
    if (c1 == '!' && c2 == '&') {   /* "&nbsp;" arrives as "&!" and counts as a space */
      m->c1 = c1 = SPACE;
      c4 = (c4 & 0xffffff00) + SPACE;
      m->stream2b = (m->stream2b & 0xfffffffc) + FX_WRT_2B[SPACE];
      m->stream3b = (m->stream3b & 0xfffffff8) + FX_WRT_3B[SPACE];
    } else if (col_lastfc(&m->colcxt, 0) == '*' && (c1 == ',' || c1 == SPACE) && c2 == SQUARECLOSE && m->isParagraph == 0) { m->isParagraph = 1; m->fc = FIRSTUPPER; }
  }

  m->x5 = (m->x5 << 8) + (c4 & 0xff);
  if (m->o2bState != m->n2bState) { m->stream2bR = (m->stream2bR << 2) + m->n2bState; m->o2bState = m->n2bState; }   /* non-repeating streams */
  m->stream2bMask = (m->stream2bMask << 2) + 3;
  if (m->o3bState != m->n3bState) {
    m->stream3bR = (m->stream3bR << 3) + m->n3bState;
    m->stream3bRMask1 = (m->stream3bRMask1 << 3) + 7;
    m->stream3bRMask2 = (m->stream3bRMask2 << 3) + 7;
    m->o3bState = m->n3bState;
  }
  m->stream3b = (m->stream3b << 3) + m->n3bState;
  m->stream3bMask = (m->stream3bMask << 3) + 7;
  m->stream3bMask1 = (m->stream3bMask1 << 3) + 7;
  const uint32_t brcontext = m->brcxt.cxt & 255;
  m->BrFcIdx = 0;
  if (m->brcxt.context) m->BrFcIdx = kFcy[brcontext & 127];
  if (m->brcxt.context == 0 && m->qocxt.context) m->BrFcIdx = kFcy[(m->qocxt.context >> 8) & 127];

  Columns* cc = &m->colcxt;
  m->col = col_len(cc, 0, 0);
  int above = m->buffer[(uint32_t)(m->nl1 + m->col) & BMASK], above1 = m->buffer[(uint32_t)(m->nl1 + m->col - 1) & BMASK];
  if (cc->nlChar == WIKIHEADER) { above = col_b(cc, 1, 0); above1 = col_b(cc, 1, 1); }
  if (cc->NL) {
    if ((int)(cc->col[cc->rows & 3].linepos + 2 - cc->col[(cc->rows - 1) & 3].linepos) < 4) {   /* two empty lines reset the nesting contexts */
      br_reset(&m->fccxt); br_reset(&m->brcxt); br_reset(&m->qocxt); br_reset(&m->htcxt);
    }
    m->fc = col_lastfc(cc, 0);
    if (m->fc == WIKIHEADER) br_reset(&m->fccxt);
    m->isParagraph = (m->fc == FIRSTUPPER);
    br_update(&m->fccxt, m->fc);
  }
  if (m->col > 2 && c1 > FIRSTUPPER && !m->isMath) {
    if (m->fccxt.cxt == VERTICALBAR && (c1 == SQUARECLOSE || c1 == CURLYCLOSE)) while (m->fccxt.cxt == VERTICALBAR) br_update(&m->fccxt, LF);
    if ((m->fccxt.cxt == COLON || m->fccxt.cxt == HTLINK) && c1 == SQUARECLOSE) while (m->fccxt.cxt == COLON || m->fccxt.cxt == HTLINK) br_update(&m->fccxt, LF);
    if (c1 < 128) br_update(&m->fccxt, c1);
  }
  if (c1 == COLON && (m->words & 2) == 2) m->colonstr = m->so;   /* the decoded dictionary word before ':' */
  if (c1 == SPACE && m->fccxt.cxt == COLON && col_lastfc(cc, 0) != COLON && cc->nlChar != WIKITABLE && strcmp(m->colonstr, "image") != 0)
    while (m->fccxt.cxt == COLON) br_update(&m->fccxt, LF);
  if (c1 == COLON && (strcmp(m->colonstr, "category") == 0 || strcmp(m->colonstr, "wikipedia") == 0)) { br_update(&m->fccxt, LF); wc_remove(&m->worcxt); }
  if (c1 == SPACE && c2 == LESSTHAN) br_update(&m->fccxt, GREATERTHAN);
  if (m->fccxt.cxt == COLON && c2 == '/' && c1 == '/') { br_update(&m->fccxt, LF); br_update(&m->fccxt, HTLINK); }
  if (col_lastfc(cc, 0) == SQUAREOPEN && c1 == SPACE && m->isParagraph == 0 && (c2 == SQUARECLOSE || c3 == SQUARECLOSE)) {
    m->fc = FIRSTUPPER; m->isParagraph = 1;
    br_reset(&m->fccxt); br_update(&m->fccxt, m->fc);
  }
  if (m->fc == SPACE && c1 != SPACE) {
    m->fc = imin(c1, TEXTDATA);
    m->isParagraph = (m->fc == FIRSTUPPER);
    br_update(&m->fccxt, m->fc);
  }
  const uint32_t fccontext = m->fccxt.cxt & 255;
  if (m->BrFcIdx == 0 && m->fccxt.context) m->BrFcIdx = kFcy[fccontext & 127];
  m->FcIdx = kFcq[fccontext & 127];
  CM(5) (m->fccxt.context & 0xff00) + (uint32_t)c1 + (m->stream2b & 12) * 256 + ((brcontext + (uint32_t)br_last(&m->brcxt)) << 24));

  if (m->fc == '*' && c1 != SPACE) m->fc = imin(c1, TEXTDATA);
  if (m->fc == '&' && c1 == LESSTHAN) m->fc = HTML;
  if (c2 == GREATERTHAN && m->fc == LESSTHAN && c1 == APOSTROPHE) m->fc = APOSTROPHE;
  if ((col_lastfc(cc, 0) == APOSTROPHE || (m->fc == APOSTROPHE && col_lastfc(cc, 0) != '*')) && c1 == SPACE && (c2 == APOSTROPHE || c3 == APOSTROPHE)) {
    m->fc = FIRSTUPPER; m->isParagraph = 1;
    br_reset(&m->fccxt); br_update(&m->fccxt, m->fc);
  }
  if (m->fc != FIRSTUPPER && (c4 & 0xffffff) == 0x4a2f2f) m->fc = HTLINK;
  wc_remove_words(&m->worcxt, 8, '(', ')', 1); wc_remove_words(&m->worcxt1, 8, '(', ')', 1);
  wc_remove_words(&m->worcxt, 8, SQUAREOPEN, VERTICALBAR, 1); wc_remove_words(&m->worcxt1, 8, SQUAREOPEN, VERTICALBAR, 1);
  wc_remove_words(&m->worcxt, 8, LESSTHAN, COLON, 1);
  if (cc->isTemp) wc_remove_words(&m->worcxt, 10, EQUALS, VERTICALBAR, 0);
  wc_remove_words(&m->worcxt, 8, LESSTHAN, GREATERTHAN, 1); wc_remove_words(&m->worcxt1, 8, LESSTHAN, GREATERTHAN, 1);

  /* indirect histories */
  m->indirectWord = (c4 >> 8) & 0xffff;
  m->t2[m->indirectWord] = (m->t2[m->indirectWord] << 8) | (uint32_t)c1;
  m->indirectWord = c4 & 0xffff;
  m->indirectWord = m->indirectWord | (m->t2[m->indirectWord] << 16);
  m->indirectByte = (c4 >> 8) & 0xff;
  m->t1[m->indirectByte] = (m->t1[m->indirectByte] << 8) | (uint32_t)c1;
  m->indirectByte = (uint32_t)c1 | (m->t1[c1] << 8);
  m->t1[brcontext] = (m->t1[brcontext] << 2) | (m->stream2b & 3);
  m->indirectBrByte = (m->stream3b & 7) | (m->t1[brcontext] << 3);
  m->indirectWord0Pos = (uint32_t)(m->pos - m->wp[m->word0 & 0xffff]);
  if (m->indirectWord0Pos > 255) m->indirectWord0Pos = 256 + ((uint32_t)c1 << 16);
  else m->indirectWord0Pos = m->indirectWord0Pos + ((uint32_t)BUF(m->indirectWord0Pos) << 8) + ((uint32_t)c1 << 16);
  m->ind3[m->context1_ind3] = (uint16_t)((m->cxtind3 * 32 + (uint32_t)c1) & (0x2000000 - 1));
  m->context1_ind3 = (m->context1_ind3 * 32 + (uint32_t)c1) & (0x2000000 - 1);
  m->cxtind3 = m->ind3[m->context1_ind3];
  if (c2 == 12) {   /* escaped UTF-8 */
    if (m->utf8left == 0) {
      if ((c1 >> 5) == 6) { m->utf8left = 1; m->u8w = m->u8w * 191 + (uint32_t)c1; }
      else if ((c1 >> 4) == 0xE) { m->utf8left = 2; m->u8w = m->u8w * 191 + (uint32_t)c1; }
      else if ((c1 >> 3) == 0x1E) { m->utf8left = 3; m->u8w = m->u8w * 191 + (uint32_t)c1; }
      else m->utf8left = 0;
    } else { m->utf8left--; if ((c1 >> 6) != 2) m->utf8left = 0; }
  }
  h = h + (uint32_t)c1;

  /* ---- the 81 context slots, in the order the maps receive them (:4323-4596) ---- */
  Words *wc = &m->worcxt, *wc1 = &m->worcxt1, *wc2 = &m->worcxt2;
  const uint32_t s2 = m->stream2b, s3 = m->stream3b, s3R = m->stream3bR, word0 = m->word0, word00 = m->word00, BrFc = m->BrFcIdx, uc1 = (uint32_t)c1;
  const uint32_t fc = (uint32_t)m->fc, col = (uint32_t)m->col;
  const int lastfc = col_lastfc(cc, 0), utf8left = m->utf8left, isPar = m->isParagraph;
  fx_rcm_set(&m->rcmA, m->word3 * 53 + uc1 + 193 * (s3 & 0x7fff), c1);
  if (m->col < 2 || m->fc == SPACE) { SKIP(m->cmC2[4]); SKIP(m->cmC2[4]); SKIP(m->cmC2[17]); }
  else {
    CM2(4) word00 + (m->number0 * 191 + m->numlen0) + m->u8w);
    if (lastfc == '&' || utf8left) SKIP(m->cmC2[4]); else CM2(4) h + m->word1);
    if (m->brcxt.cxt == LESSTHAN) SKIP(m->cmC2[17]); else CM2(17) wc_word(wc1, 1) * 53 + wc_word(wc1, 2) * 11 + h + (m->lastWT & 0xf));
  }
  if (c1 == ESCAPE || m->col < 2 || utf8left || m->fc == SPACE) SKIP(m->cmC2[5]); else CM2(5) h + m->word2 * 71);
  if (m->fc == SPACE || m->brcxt.cxt == LESSTHAN) { for (i = 0; i < 5; i++) SKIP(m->cmC2[5]); }
  else {
    CM2(5) wc_word(wc, 4) * 53 + wc_word(wc1, 1) + h + (s3 & 511));
    CM2(5) wc_last(wc, 4, wc_type(wc, 4) ^ T_Verb, 0) * 53 + m->sVerb + h + (s3R & 63));
    CM2(5) wc->fword * 53 + wc_word(wc1, 1) + h + (s3 & 63));
    CM2(5) wc_word(wc2, 1) + wc_word(wc2, 2) * 11 + word00 + uc1);
    const uint32_t lastParVerb = wc_last(wc2, 1, wc_type(wc, 1) & T_Verb, 1);
    if (lastParVerb) CM2(5) lastParVerb * 11 + word00 + uc1); else SKIP(m->cmC2[5]);
  }
  CM1(6) h + (wc_type(wc, 1) & 0x1FF) + wc_word(wc1, 1));
  CM2(6) ((s2 & 15) << 16) + (m->t[2] & 0xffff));
  if (c1 == ESCAPE || utf8left || fccontext == CURLYOPENING) CM2(7) 0); else CM2(7) m->indirectBrByte);
  CM2(8) (m->indirectBrByte & 0x7ff) * 32 + ((m->stream4b & 0xfff0) << 16) + BrFc);
  CM2(8) (s3R & 0x3fffffff) * 4 + (s2 & 3));
  CM2(8) fccontext * 4 + ((s3R & 0x3ffff) << 9) + BrFc);
  if (fccontext == HTLINK) SKIP(m->cmC2[8]); else CM2(8) (c4 & 0xffffff) + ((s2 << 18) & 0xff000000));
  CM1(0) (uint32_t)lastfc | (fccontext << 15) | ((s3 & 63) << 7) | (brcontext << 24));
  CM1(0) (uint32_t)lastfc | ((c4 & 0xffffff) << 8));
  CM1(1) (s2 & 3) + word00 * 11);
  CM1(1) c4 & 0xffff);
  CM1(1) ((fc << 11) | uc1) + ((s2 & 3) << 18));
  CM1(2) (s2 & 15) + ((s3 & 7) << 6));
  CM1(2) uc1 | ((col * (uint32_t)(c1 == SPACE)) << 8) | ((s2 & 15) << 16));
  CM1(2) isPar ? m->firstWord : (fc << 11));
  if (c1 == ESCAPE || m->fc == SPACE || utf8left) SKIP(m->cmC1[2]); else CM1(2) 91u * 83u * wc_word(wc, 1) + 89u * word0);
  if (m->fc == SPACE) SKIP(m->cmC1[4]); else CM1(4) uc1 + ((s3 & 0xe38) << 6));
  CM1(4) wc->fword * 11 + BrFc);
  CM1(4) uc1 + word0 + m->number0 * 191);
  CM1(4) ((c4 & 0xffff) << 16) | (fccontext << 8) | fc);
  CM1(4) ((s3R & 0xfff) << 8) + (s2 & 0xfc));
  if (c1 == ESCAPE) { for (i = 0; i < 6; i++) SKIP(m->cmC[0]); }
  else {
    if (isPar == 1) {
      CM(0) wc->fword * 3191 + (s2 & 3));
      CM(0) h + m->firstWord * 89);
      CM(0) word0 * 53 + uc1 + BrFc);
    } else {
      CM(0) (uint32_t)above | ((s3 & 0x3f) << 9) | ((uint32_t)col_len(cc, 0, 0) << 19) | ((s2 & 3) << 16));
      CM(0) h + m->firstWord * 89);
      CM(0) (uint32_t)above | (uc1 << 16) | ((col + m->numlen0 + BrFc) << 8) | ((uint32_t)above1 << 24));
    }
    const uint32_t cellb = (uint32_t)BUFR(cc->abovecellpos);
    if (lastfc == '*') {
      CM(0) (word0 + (fccontext << 8)) | (BrFc << 16));
      CM(0) uc1);
      CM(0) word0);
    } else {
      CM(0) FX_WRT_2B[cellb] | (fccontext << 8) | (BrFc << 16));
      CM(0) cellb | (uc1 << 8));
      CM(0) word0 + FX_WRT_2B[cellb]);
    }
  }
  CM(1) (s3 & 0x7fff) * word0 + BrFc);
  CM(1) (m->x4 & 0xff0000ff) | ((s3 & 0xe07) << 8));
  CM(1) (m->indirectBrByte & 0xffff) | ((s3 & 0x38) << 16));
  if (m->isMath) SKIP(m->cmC[0]); else CM(0) (m->indirectByte & 0xff00) + 257u * wc_word(wc, 1) * 53u + uc1);
  CM(2) (uc1 << 8) | (m->indirectByte >> 2) | (fc << 16));
  CM(2) (c4 & 0xffff) + (uint32_t)(c2 == c3 ? 1 : 0));
  CM1(3) ((s3 & m->stream3bMask) * 256) | (s2 & m->stream2bMask & 255));
  CM1(3) m->x4);
  CM2(9) 257u * m->StemWords[m->pWord].Hash + fccontext + 193u * (s3 & m->stream3bMask));
  CM2(9) fc | ((m->stream2bR & 0xfff) << 9) | (uc1 << 24));
  CM2(16) wc->fword * 83 + (s2 & 15) * 11 + brcontext);
  CM2(17) wc_last(wc, 1, T_Verb, 0) + wc_word(wc, 1) * 83 + h);
  CM2(9) (m->x4 & 0xffff00) + brcontext + (fccontext << 24));
  if (m->linkword) CM2(9) m->linkword);
  else if (m->isMath) SKIP(m->cmC2[9]);
  else if (m->senword) CM2(9) m->senword * 1471 + uc1);
  else if (m->fc == HTML || brcontext == LESSTHAN) SKIP(m->cmC2[9]);
  else CM2(9) 0);
  CM2(10) m->indirectByte);
  CM2(10) ((m->indirectByte & 0xffff00) >> 4) | (s2 & m->stream2bMask & 0xf) | ((s3 & 0xfff) << 20));
  CM2(10) (m->x4 >> 16) | ((s2 & 255) << 24));
  if (c1 > 127) CM2(10) ((((s2 & 12) * 256) + uc1) << 11) | ((m->indirectWord & 0xffffff) >> 16));
  else CM2(10) (uc1 << 11) | (BrFc << 8) | ((m->indirectWord & 0xffffff) >> 16));
  if (m->isMath) SKIP(m->cmC2[10]); else CM2(10) (fccontext * 4 + BrFc) | ((c4 & 0xffff) << 9) | ((s2 & 0xff) << 24));
  CM2(10) (m->indirectWord >> 16) | ((s2 & 0x3c) << 25) | ((s3 & 0x1ff) << 16));
  CM2(11) (uint32_t)m->words + ((uint32_t)m->spaces << 8) + ((s2 & 15) << 16) + (((s3R >> 3) & 511) << 21) + ((uint32_t)isPar << 30));
  CM2(11) uc1 + ((s3 << 5) & 0x1fffff00));
  CM2(11) m->stream2bR * 16 + BrFc);
  CM2(11) ((m->indirectByte & 0xffff) >> 8) + ((64 * m->stream2bR) & 0x3ffff00) + (brcontext << 25));
  if (fccontext == FIRSTUPPER && brcontext == SQUAREOPEN) SKIP(m->cmC2[11]); else CM2(11) m->indirectWord0Pos | ((m->indirectByte & 0xff00) << 16));
  CM2(12) (m->x4 & 0x80f00000) + ((m->x4 & 0x0000f0ff) << 12));
  if (isPar == 1) {
    if (c1 == ESCAPE || fccontext == HTLINK || fccontext == CURLYOPENING || m->isMath || m->isPre) SKIP(m->cmC2[12]);
    else CM2(12) h + wc_word(wc, 1) * 53u * 79u + wc_word(wc, 3) * 53u * 47u * 71u);
  } else {
    if (fccontext == HTLINK || brcontext == LESSTHAN || m->htcxt.cxt) SKIP(m->cmC2[12]);
    else if (m->col == 31) CM2(12) c4 << 16);
    else CM2(12) (uint32_t)above | ((c4 & 0xffff) << 16) | ((uint32_t)above1 << 8));
  }
  const int esc_word = (wc_sbytes(wc, 0) >> 8) == '\\';
  if (c1 == ESCAPE || utf8left || fccontext == CURLYOPENING || fccontext == HTLINK || m->fc == HTML || m->htcxt.cxt || m->fc == SPACE || m->isPre || c1 == '&' ||
      brcontext == LESSTHAN || m->isMath || m->col < 2 || esc_word) { SKIP(m->cmC2[13]); SKIP(m->cmC2[13]); }
  else {
    CM2(13) wc_word(wc, 1) * 83u * 1471u - word0 * 53 + wc_word(wc, 2));
    CM2(13) h + wc_word(wc, 2) * 53u * 79u + wc_word(wc, 3) * 53u * 47u * 71u);
  }
  CM(3) ((s3R & 7) << 10) + (s2 & 3) + fc * 4 + (BrFc << 24));
  CM(3) (m->linkword ? m->linkword : word0) * 3301 + m->number0 * 3191);
  if (c1 == ESCAPE || utf8left || fccontext == CURLYOPENING || fccontext == HTLINK || m->fc == SPACE || m->fc == HTML || brcontext == LESSTHAN || m->col < 2 ||
      m->isMath || esc_word) SKIP(m->cmC2[14]);
  else CM2(14) BrFc + wc_word(wc, 2) * (s3R & m->stream3bRMask2) + (wc_type(wc, 1) & 0x1ff));
  if (c1 == ESCAPE || utf8left || m->fc == SPACE) { for (i = 0; i < 4; i++) SKIP(m->cmC1[7]); }
  else {
    CM1(7) wc_word(wc1, 1) + word00);
    CM1(7) wc_word(wc, 2) + word0 * 191 + (s3R & 63));
    CM1(7) word0 * 191 + (s3R & 63));
    CM1(7) (m->indirectWord0Pos & 0xffff) * 191 + word0 + (s3R & 63));
  }
  fx_sscm_set(&m->scmA[0], uc1);
  fx_sscm_set(&m->scmA[1], (uint32_t)(c2 * isPar));
  fx_sscm_set(&m->scmA[2], (m->indirectWord & 0xffffff) >> 16);
  fx_sscm_set(&m->scmA[3], s3 & 0x1ff);
  fx_sscm_set(&m->scmA[4], s2 & 0xff);
  fx_sscm_set(&m->scmA[5], brcontext);
  fx_sscm_set(&m->scmA[6], (uint32_t)isPar + 2 * (s3R & 0x3f));
  if (m->wshift || c1 == LF) {
    m->word3 = m->word3 * 47; m->word2 = m->word2 * 53; m->word1 = m->word1 * 83;
    m->wshift = 0;
    if (c1 == LF) m->sVerb = 0;
  }
  CM2(15) (BrFc * 256) + fc + ((s3R & 0xFFF) << 16));
  m->AH1 = hash3(m->x5 & 255, (m->x5 >> 8) & 255, (m->x5 >> 16) & 0x80ff);
  m->AH2 = hash3(19, m->x5 & 0x80ffff, 0xffffffffu);
  m->mxA[8]->cxt = m->deccode;
}

/* ---- MatchModel2 (:3433-3676) ---- */
static int mi_no_match(const MatchInfo* c) { return c->length == 0 && !c->delta && c->lengthBak == 0; }
static int mi_recovering(const MatchInfo* c) { return c->length != 0 && c->lengthBak != 0; }
static uint32_t mi_prio(const MatchInfo* c) {
  return (uint32_t)(c->length != 0) << 31 | (uint32_t)c->delta << 30 | (c->delta ? (c->lengthBak >> 1) : (c->length >> 1)) << 24 | (c->index & 0x00ffffff);
}
static void mi_update(FxModel* m, MatchInfo* c) {
  if (c->length != 0) {
    const int expectedBit = (c->expectedByte >> ((8 - m->bpos) & 7)) & 1;
    if (m->y != expectedBit) {
      if (mi_recovering(c)) { c->lengthBak = 0; c->indexBak = 0; }
      else { c->lengthBak = c->length; c->indexBak = c->index; c->delta = 1; }
      c->length = 0;
    }
  }
  if (m->bpos == 0) {
    if (c->length == 0 && !c->delta && c->lengthBak != 0) {   /* one byte after the mismatch: try to pick the match up again */
      c->indexBak++;
      if (c->lengthBak < MAXLEN) c->lengthBak++;
      if (BUFR(c->indexBak) == m->c1) { c->length = c->lengthBak; c->index = c->indexBak; }
      else c->lengthBak = c->indexBak = 0;
    }
    if (c->length != 0) {
      c->index++;
      if (c->length < MAXLEN) c->length++;
      if (mi_recovering(c) && c->length - c->lengthBak >= MINLEN_RM) c->lengthBak = c->indexBak = 0;
    }
    c->delta = 0;
  }
}
static void add_candidates(FxModel* m, const MatchSlot* slot, uint32_t LEN) {
  for (uint32_t i = 0; m->nActive < 4 && i < 4; i++) {
    const uint32_t matchpos = slot->pos[i];
    if (matchpos == 0) break;
    int ok = 1;
    for (int length = 1; length <= (int)LEN; length++) if (BUF(length) != BUFR(matchpos - (uint32_t)length)) { ok = 0; break; }
    if (!ok) continue;
    int same = 0;
    for (uint32_t k = 0; k < m->nActive; k++) if (m->cand[k].index == matchpos) { same = 1; break; }
    if (!same) {
      MatchInfo* c = &m->cand[m->nActive++];
      c->length = LEN - LEN1 + 1; c->index = matchpos; c->lengthBak = c->indexBak = 0; c->expectedByte = 0; c->delta = 0;
    }
  }
}
static void slot_add(MatchSlot* s, uint32_t pos) { memmove(&s->pos[1], &s->pos[0], 3 * sizeof(uint32_t)); s->pos[0] = pos; }
static int match_model2_mix(FxModel* m) {
  const uint32_t n = (uint32_t)imax((int)m->nActive, 1);
  for (uint32_t i = 0; i < n; i++) {
    MatchInfo* c = &m->cand[i];
    mi_update(m, c);
    if (m->nActive != 0 && mi_no_match(c)) {
      m->nActive--;
      if (m->nActive == i) break;
      memmove(&m->cand[i], &m->cand[i + 1], (m->nActive - i) * sizeof(MatchInfo));
      i--;
    }
  }
  if (m->bpos == 0) {
    const uint32_t hashes[4] = {m->t[LEN3], m->t[LEN2], m->t[LEN1], wc_word(&m->worcxt, 1)};
    const uint32_t lens[4] = {LEN3, LEN2, LEN1, LEN1};
    for (int k = 0; k < 4; k++) {
      MatchSlot* slot = &m->mhash[hashes[k] & m->mhashmask];
      if (m->nActive < 4) add_candidates(m, slot, lens[k]);
      slot_add(slot, (uint32_t)m->pos);
    }
    for (uint32_t i = 0; i < m->nActive; i++) m->cand[i].expectedByte = (uint8_t)BUFR(m->cand[i].index);
  }
  uint32_t ctx[3] = {0, 0, 0};
  int best = 0;
  for (uint32_t i = 1; i < m->nActive; i++) if (mi_prio(&m->cand[i]) > mi_prio(&m->cand[best])) best = (int)i;
  const uint32_t length = m->cand[best].length;
  const uint32_t expectedByte = m->cand[best].expectedByte;
  const int delta = m->cand[best].delta;
  const int expectedBit = length != 0 ? (int)(expectedByte >> (7 - m->bpos)) & 1 : 0;
  if (length != 0) {
    const uint32_t dense = length <= 16 ? length - 1 : 12 + (length >> 2);
    ctx[0] = (dense << 4) | ((uint32_t)expectedBit << 3) | (uint32_t)m->bpos;
    ctx[1] = (expectedByte << 11) | ((uint32_t)m->bpos << 8) | (uint32_t)m->c1;
    fx_add(&m->in1, (2 * expectedBit - 1) * (int)(length << 5));
  } else fx_add(&m->in1, 0);
  if (delta) ctx[2] = (expectedByte << 8) | (uint32_t)m->c0;
  for (int i = 0; i < 3; i++) {
    if (ctx[i] != 0) {
      const int p1 = fx_statemap1_set(&m->smA[i], m->y, (int)ctx[i]);
      fx_add(&m->in1, fx_stretch(p1) >> 2);
      fx_add(&m->in1, (p1 - 2048) >> 3);
    } else { fx_add(&m->in1, 0); fx_add(&m->in1, 0); }
  }
  return (int)length;
}

static void export_value(FxModel* m, int v) { m->in1.exported[m->in1.pidx++] = (float)v * (float)(1.0 / 4095); }   /* AddPrediction :98-101 */
static void add2(FxModel* m, int p) { m->in2[m->n2++] = (int16_t)p; export_value(m, fx_squash(p)); }             /* mxInputs2.add */

/* modelPrediction :3798-4757 */
static int model_prediction(FxModel* m) {
  const int bpos = m->bpos, c0 = m->c0, y = m->y;
  if (bpos == 0) byte_update(m);
  const int c0b = c0 << (8 - bpos);
  for (int i = 0; i < 7; i++) fx_sscm_mix(&m->scmA[i], &m->in1, y, m->sscmrate);
  m->isMatch = (uint32_t)match_model2_mix(m);
  fx_sparsematch_p(m->smatch, &m->in1, bpos, c0, m->buffer, BMASK, m->pos);
#define MIX(map) fx_cm_mix(map, &m->in1, y, bpos, c0, (int)(m->c4 & 255))   /* the run model sees the real last byte (x.c4), not the parser's c1 */
  int ordX = 0;
  if (fx_cm_skipmask(m->cmC2[0])) ordX = 2;
  ordX += MIX(m->cmC2[0]);
  if (ordX == 3) ordX = 2;
  ordX += MIX(m->cmC2[1]); ordX += MIX(m->cmC2[2]); ordX += MIX(m->cmC2[3]);
  int ordW = MIX(m->cmC2[4]);
  ordW += MIX(m->cmC2[5]);
  if (ordW > 3) ordW = 3;
  MIX(m->cmC2[6]); MIX(m->cmC2[7]); MIX(m->cmC2[8]);
  MIX(m->cmC1[0]); MIX(m->cmC1[1]); MIX(m->cmC1[2]); MIX(m->cmC1[4]);
  MIX(m->cmC[0]); MIX(m->cmC[1]); MIX(m->cmC[2]);
  MIX(m->cmC1[3]);
  MIX(m->cmC2[9]); MIX(m->cmC2[10]); MIX(m->cmC2[11]); MIX(m->cmC2[12]);
  ordW += MIX(m->cmC2[13]);
  MIX(m->cmC[3]);
  ordW += MIX(m->cmC2[14]);
  MIX(m->cmC2[15]); MIX(m->cmC[4]); MIX(m->cmC[5]); MIX(m->cmC2[16]); MIX(m->cmC2[17]); MIX(m->cmC1[6]); MIX(m->cmC1[7]);
#undef MIX
  fx_rcm_mix(&m->rcmA, &m->in1, bpos, c0);
  export_value(m, fx_squash(64));

  /* the ten mixer selectors :4616-4737 */
  const uint32_t s2 = m->stream2b, s3 = m->stream3b, s3R = m->stream3bR, BrFc = m->BrFcIdx, words = m->words;
  int c;
  if (bpos == 0) m->mxA[0]->cxt = (int)((s2 & 255) * 8 + (s3 & 7));
  else if (bpos > 3) m->mxA[0]->cxt = (int)((((s2 << 2) & 255) + FX_WRT_2B[c0b & 255]) * 8 + BrFc);
  else m->mxA[0]->cxt = (int)((s2 & 255) * 8 + BrFc);
  if (bpos) {
    c = c0b;
    if (bpos == 1) c = c + 16 * (int)(words * 2 & 4);
    else if (bpos > 3) c = FX_WRT_2B[c0b & 255] * 64;
    c = imin(bpos, 5) * 256 + (int)(s3R & 7) + (int)m->FcIdx * 8 + (c & 192);
  } else c = (int)((words & 12) * 16 + (s3R & 7) + BrFc * 8);
  m->mxA[1]->cxt = c;
  m->mxA[2]->cxt = (int)(((4 * words) & 0xf0) * 4 + (uint32_t)ordX * 256 * 4 + (s2 & 63));
  m->mxA[6]->cxt = (int)((s3R & 0xff8) * 4 + ((2 * words) & 0x1c) + (s2 & 3));
  c = c0b;
  m->mxA[3]->cxt = bpos * 256 + (int)((((((uint32_t)m->numbers | words) << bpos) & 255) >> bpos) | ((uint32_t)c & 255));
  m->mxA[10]->cxt = (int)(((uint32_t)ordX * 8 + (BrFc ? 1u : 0u) * 4 + (s2 & 3)) * 2 + (words & 1));
  if (bpos) {
    if (bpos == 1) c = c + 16 * (int)(s3 & 7);
    else if (bpos == 2) c = c + 16 * (int)(s2 & 3);
    else if (bpos == 3) c = c + 16 * (int)(words & 1);
    else c = bpos + (c & 0xf0);
    if (bpos < 5) c = bpos + (c & 0xf0);
  } else c = 16 * (int)(s2 & 0xf);
  ordX = ordX - 1;
  if (ordX < 0) ordX = 0;
  if (m->isMatch) ordX = ordX + 1;
  m->ordX = ordX; m->ordW = ordW;
  m->mxA[4]->cxt = c + ordX * 256 + 8 * m->isParagraph;
  m->mxA[5]->cxt = (int)(((uint32_t)ordW * 256 + (s2 & 0xf0) + ((s3 & 0x38) >> 2)) * 4 + m->FcIdx);
  if (bpos > 2) m->mxA[7]->cxt = (int)(((s3 & 7) * 8 + FX_WRT_3B[c0b & 255]) * 256 + BrFc * 32 + (words & 7) * 4 + (uint32_t)m->isParagraph + (m->isMatch ? 2u : 0u));
  else m->mxA[7]->cxt = (int)(((s3 & 63) * 256 + BrFc * 16 + (words & 7) * 2 + (uint32_t)m->isParagraph) | (m->isMatch ? 128u : 0u));
  m->mxA[9]->cxt = (bpos << 8) * 4 + (int)(m->fails & 3) * 256 + m->lstmex;

  fx_add(&m->in1, fx_stretch(m->lstmpr)); fx_unexport(&m->in1);
  m->n2 = 0;
  for (int i = 0; i < 10; i++) add2(m, fx_mixer_p1(m->mxA[i]));
  add2(m, fx_stretch(m->lstmpr) / 2); m->in1.pidx--;
  return fx_squash((fx_mixer_p1(m->mxA[10]) * 7 + fx_mixer_p1(m->mxA[11]) + 4) >> 3);
}

/* update1 :4758-4833 = FXCM::Perceive(bit). lstmpr / lstmex: the LSTM hints cmix sets just before (predictor.cpp:462-465).
 * out431 (may be NULL) receives FXCM::Predict()'s vector for the next bit. Returns the model's own final probability. */
int orc_fx_model_update(FxModel* m, int y, int lstmpr, int lstmex, float* out431) {
  static const int e_l[8] = {1830, 1997, 1973, 1851, 1897, 1690, 1998, 1842};   /* :3222 */
  static const uint32_t tri[4] = {0, 4, 3, 7}, trj[4] = {0, 6, 6, 12};             /* :3211 */
  m->y = y; m->lstmpr = lstmpr; m->lstmex = lstmex;
  m->c0 += m->c0 + y;
  if (m->c0 >= 256) {
    m->c4 = (m->c4 << 8) + (uint32_t)(m->c0 & 0xff);
    m->c0 = 1;
    ++m->blpos;
    if ((m->fails & 255) == 0) for (int i = 0; i < 10; i++) m->mxA[i]->elim = imax(256, m->mxA[i]->elim + 1);
    else for (int i = 0; i < 10; i++) m->mxA[i]->elim = imax(0, imin(16, m->mxA[i]->elim - 1));
    m->sscmrate = (m->blpos > 14 * 256 * 1024);
    m->rate = 6 + (m->blpos > 14 * 256 * 1024) + (m->blpos > 28 * 512 * 1024);
  }
  m->bpos = (m->bpos + 1) & 7;
  for (int i = 0; i < 12; i++) fx_mixer_update(m->mxA[i], y);
  m->in1.ncount = 0; m->n2 = 0;
  if (m->fails & 0x00000080) --m->failcount;
  m->fails = m->fails * 2;
  m->failz = m->failz * 2;
  if (y) m->pr = 4095 - m->pr;
  if (m->pr >= e_l[m->bpos]) { ++m->fails; ++m->failcount; }
  if (m->pr >= 848) ++m->failz;

  int pr = model_prediction(m);
  export_value(m, pr);
  const int c0 = m->c0, rate = m->rate;
  int pt, pu = (fx_apm_p(m->apm[0], pr, c0, 3, y) + 7 * pr + 4) >> 3, pv, pz = (int)m->failcount + 1;
  pz += (int)tri[(m->fails >> 5) & 3];
  pz += (int)trj[(m->fails >> 3) & 3];
  pz += (int)trj[(m->fails >> 1) & 3];
  if (m->fails & 1) pz += 8;
  pz = pz / 2;
  pu = fx_apm_p(m->apm[3], pu, (int)((((uint32_t)c0 * 2) ^ m->AH1) & 0x3ffff), rate, y); export_value(m, pu);
  pv = fx_apm_p(m->apm[1], pr, (int)((((uint32_t)c0 * 8) ^ hash3(29, m->failz & 2047, 0xffffffffu)) & 0xffff), rate + 1, y); export_value(m, pv);
  if (m->fails & 255) pv = fx_apm_p(m->apm[4], pv, (int)(hash3((uint32_t)c0, m->stream2b & 0xfffc, m->stream3bR & 0x1ff) & 0x3ffff), rate, y);
  else pv = fx_apm_p(m->apm[4], pv, (int)(hash3((uint32_t)c0, (m->stream2bR & 0xfffc) + 0x10000, m->stream3bR & 0x1ff) & 0x3ffff), rate, y);
  export_value(m, pv);
  pt = fx_apm_p(m->apm[2], pr, (int)((((uint32_t)c0 * 32) ^ m->AH2) & 0xffff), rate, y); export_value(m, pt);
  pz = fx_apm_p(m->apm[5], pu, (int)((((uint32_t)c0 * 4) ^ hash3((uint32_t)imin(9, pz), m->x5 & 0x80ff, 0xffffffffu)) & 0x3ffff), rate, y); export_value(m, pz);
  if (m->fails & 255) pr = (pt * 6 + pu + pv * 11 + pz * 14 + 31) >> 5;
  else pr = (pt * 4 + pu * 5 + pv * 12 + pz * 11 + 31) >> 5;
  export_value(m, pr);
  m->pr = pr;
  const int nexp = m->in1.pidx;
  m->in1.pidx = 0;   /* ResetPredictions */
  if (out431) memcpy(out431, m->in1.exported, FX_OUTPUTS * sizeof(float));
  return nexp <= FX_OUTPUTS ? pr : -1;
}

static void load_dictionary(FxModel* m, const char* path) {  /* dosym + loaddict + wfgets :352-381, :413-433: one word per line, index = line number */
  FILE* f = fopen(path, "rb");
  if (!f) return;
  m->dictW = (char**)calloc(44516, sizeof(char*));
  char* line = (char*)malloc(8192 * 8);
  int n = 0, c;
  for (;;) {
    int i = 0;
    while (i < 8192 * 8 - 1 && (c = getc(f)) != EOF) { line[i++] = (char)c; if (c == '\n') { line[i - 1] = 0; break; } }
    line[i] = 0;
    if (i == 0 || n >= 44516) break;
    m->dictW[n] = (char*)calloc((size_t)i + 1, 1);
    memcpy(m->dictW[n], line, (size_t)i);
    n++;
  }
  free(line);
  fclose(f);
  m->sizeDict = n;
}
FxModel* orc_fx_model_new_dict(const char* dictionary_path);
FxModel* orc_fx_model_new(void) { return orc_fx_model_new_dict(NULL); }
FxModel* orc_fx_model_new_dict(const char* dictionary_path) {   /* Predictor::Predictor + PredictorInit :4845-4876, :3313-3405 */
  static const uint32_t c_r[27] = {3, 4, 6, 4, 6, 6, 2, 3, 3, 3, 6, 4, 3, 4, 5, 6, 2, 6, 4, 4, 4, 4, 4, 4, 4, 4, 4};
  static const uint32_t c_s[27] = {28, 26, 28, 31, 34, 31, 33, 33, 35, 35, 29, 32, 33, 34, 30, 36, 31, 32, 32, 32, 32, 32, 33, 32, 32, 32, 32};
  static const uint32_t c_s3[27] = {43, 33, 34, 28, 34, 29, 32, 33, 37, 35, 33, 28, 31, 35, 28, 30, 33, 34, 32, 32, 32, 32, 32, 32, 32, 32, 32};
  static const uint32_t c_s4[27] = {9, 8, 9, 5, 8, 12, 15, 8, 8, 12, 10, 7, 7, 8, 8, 13, 13, 14, 8, 8, 12, 12, 12, 12, 12, 12, 12};
  /* one row per context map: kind (0 ContextMap, 1 ContextMap1, 2 ContextMap2), index, size, contexts, parameter row, state table
   * (0..5 = STA1,2,4,5,6,7), keep flag, st2 input on/off, st2 table (0 zeros, 1 st2_p1, 2 st2_p2) */
  enum { S1 = 0, S2 = 1, S4 = 2, S5 = 3, S6 = 4, S7 = 5 };
  const uint32_t G = 4096u * 4096u;
  static const struct { int kind, idx; uint32_t mul, div; int c, prm, sta, keep, u, st2; } MAPS[] = {
      {2, 0, 8, 1, 3, 0, S6, 0xf0, 1, 1},  {2, 1, 16, 1, 1, 1, S6, 0xf0, 1, 1},  {2, 2, 8, 1, 1, 2, S6, 0xf0, 1, 1},   {2, 3, 8, 1, 1, 3, S6, 0xf0, 1, 1},
      {2, 4, 8, 1, 2, 4, S6, 0xf0, 1, 1},  {2, 5, 8, 1, 6, 5, S6, 0xf0, 1, 1},   {2, 6, 1, 64, 1, 6, S1, 0, 1, 1},     {2, 7, 2, 1, 1, 7, S5, 0xf0, 1, 1},
      {2, 8, 8, 2, 4, 8, S4, 0, 1, 1},     {2, 9, 8, 1, 4, 17, S6, 0xf0, 1, 1},  {2, 10, 8, 1, 6, 18, S5, 0xf0, 1, 1}, {2, 11, 8, 1, 5, 19, S5, 0xf0, 1, 1},
      {2, 12, 8, 1, 2, 20, S6, 0xf0, 1, 1}, {2, 13, 16, 1, 2, 21, S6, 0xf0, 1, 1}, {2, 14, 4, 2, 1, 23, S6, 0xf0, 1, 1}, {2, 16, 1, 2, 1, 17, S6, 0xf0, 1, 1},
      {2, 17, 2, 1, 2, 17, S6, 0xf0, 1, 1}};
  FxModel* m = (FxModel*)calloc(1, sizeof *m);
  m->c0 = 1; m->pr = 2048; m->rate = 6; m->AH2 = 0x765BA55C;
  m->n3bState = m->n2bState = 0xffffffff;
  m->buffer = (uint8_t*)calloc(BMASK + 1, 1);
  m->ind3 = (uint16_t*)calloc(0x2000000, 2);
  m->mhashmask = 0x200000 - 1;
  m->mhash = (MatchSlot*)calloc(0x200000 + 32, sizeof(MatchSlot));
  fx_statemap1_init(&m->smA[0], 1 << 9, 1023); fx_statemap1_init(&m->smA[1], 1 << 19, 1023); fx_statemap1_init(&m->smA[2], 1 << 16, 1023);
  static const int scm_bits[7] = {8, 8, 8, 9, 8, 8, 7};
  for (int i = 0; i < 7; i++) fx_sscm_init(&m->scmA[i], scm_bits[i], 8);
  static const int mx[12][4] = {{2048, 237, 8, 69}, {6 * 256, 204, 8, 19}, {6 * 256 * 4, 70, 1, 34}, {8 * 256, 54, 1, 23}, {6 * 256, 55, 1, 24}, {7 * 256 * 4, 55, 1, 24},
                                {0x4000, 70, 1, 34}, {0x4000, 55, 1, 24}, {0x20000, 55, 1, 24}, {0x20000, 55, 1, 24}, {8 * 7 * 2 * 2, 6, 0, 4}, {1, 6, 0, 4}};
  for (int i = 0; i < 12; i++) {
    m->mxA[i] = fx_mixer_new(i < 10 ? 512 : 16, mx[i][0], mx[i][1], mx[i][2], mx[i][3]);
    free(m->mxA[i]->tx);
    m->mxA[i]->tx = i < 10 ? m->in1.n : m->in2;   /* all first-layer mixers read the one input vector */
  }
  static const int apm_n[6] = {256, 0x10000, 0x10000, 0x40000, 0x40000, 0x40000};
  for (int i = 0; i < 6; i++) m->apm[i] = fx_apm_new(apm_n[i]);
  fx_rcm_init(&m->rcmA, (int)G, 6);
  for (size_t k = 0; k < sizeof MAPS / sizeof MAPS[0]; k++) {
    const int r = MAPS[k].prm;
    m->cmC2[MAPS[k].idx] = orc_fx_cm_new(2, MAPS[k].mul * G / MAPS[k].div, MAPS[k].c | (int)(c_r[r] << 8) | (int)(c_s[r] << 16), (int)c_s3[r], MAPS[k].sta,
                                         (int)c_s4[r], MAPS[k].keep, MAPS[k].u, MAPS[k].st2);
  }
#define NEWCM(kind, m_bytes, c, r, sta, keep, u, st2) orc_fx_cm_new(kind, m_bytes, (c) | (int)(c_r[r] << 8) | (int)(c_s[r] << 16), (int)c_s3[r], sta, (int)c_s4[r], keep, u, st2)
  m->cmC2[15] = NEWCM(2, 8 * 64 * 4096, 1, 24, S1, 0, 0, 0);
  m->cmC1[0] = NEWCM(1, 32 * 4096, 2, 9, S6, 0, 0, 0);
  m->cmC1[1] = NEWCM(1, 2 * 32 * 4096, 3, 10, S7, 0, 1, 1);
  m->cmC1[2] = NEWCM(1, 32 * 4096, 4, 11, S2, 0, 1, 1);
  m->cmC1[3] = NEWCM(1, 128 * 4096, 2, 16, S1, 0, 0, 0);
  m->cmC1[4] = NEWCM(1, 16 * 4096, 5, 12, S7, 0, 1, 1);
  m->cmC1[6] = NEWCM(1, 1 * 16 * 4096, 1, 5, S6, 0, 0, 1);
  m->cmC1[7] = NEWCM(1, 16 * 4096, 4, 12, S2, 0, 1, 1);
  m->cmC[0] = NEWCM(0, 16 * 4096, 7, 13, S2, 0, 1, 1);
  m->cmC[1] = NEWCM(0, 64 * 2 * 4096, 3, 14, S5, 0xf0, 0, 0);
  m->cmC[2] = NEWCM(0, 2 * 4096, 2, 15, S2, 0xf0, 0, 0);
  m->cmC[3] = NEWCM(0, 32 * 4096, 2, 22, S2, 0x00, 1, 2);
  m->cmC[4] = NEWCM(0, 512 * 4096, 1, 25, S1, 0xf0, 1, 1);
  m->cmC[5] = NEWCM(0, 512 * 4096, 1, 26, S1, 0xf0, 1, 1);
#undef NEWCM
  br_init(&m->brcxt, kBrackets, 8, 0, 8, 256);
  br_init(&m->qocxt, kQuotes, 4, 1, 8, 256);
  br_init(&m->fccxt, kFchar, 20, 0, 8, 256);
  br_init(&m->htcxt, kHtml, 2, 0, 16, 0xfff);
  m->colcxt.nlChar = LF; m->colcxt.limit = 31;
  m->smatch = orc_fx_sparsematch_new();
  m->cWord = 0; m->pWord = 3;
  m->so = m->colonstr = kEmpty;
  if (dictionary_path) load_dictionary(m, dictionary_path);
  for (int i = 0; i < FX_OUTPUTS; i++) m->in1.exported[i] = 0.5f;   /* model_predictions(0.5f, num_models) :94 */
  return m;
}
/* test hook: continue as if blpos bytes of the block had been coded (the rates follow the position, update1 :4772-4774) */
void orc_fx_model_set_blpos(FxModel* m, int blpos) {
  m->blpos = blpos;
  m->sscmrate = (blpos > 14 * 256 * 1024);
  m->rate = 6 + (blpos > 14 * 256 * 1024) + (blpos > 28 * 512 * 1024);
}
/* diagnostics for the tests: the byte contexts the maps hold, the ten mixer selectors, a few parser registers */
int orc_fx_model_debug(const FxModel* m, uint32_t* out) {
  int n = 0;
  for (int i = 0; i < 12; i++) out[n++] = (uint32_t)m->mxA[i]->cxt;
  out[n++] = m->stream2b; out[n++] = m->stream3b; out[n++] = m->stream2bR; out[n++] = m->stream3bR; out[n++] = m->word0; out[n++] = (uint32_t)m->fc;
  out[n++] = m->BrFcIdx; out[n++] = m->FcIdx; out[n++] = (uint32_t)m->isParagraph; out[n++] = m->fccxt.context; out[n++] = m->brcxt.context;
  out[n++] = m->qocxt.context; out[n++] = m->worcxt.fword; out[n++] = wc_word(&m->worcxt, 1); out[n++] = wc_type(&m->worcxt, 1); out[n++] = (uint32_t)m->ordX;
  out[n++] = (uint32_t)m->ordW; out[n++] = m->isMatch; out[n++] = m->fails; out[n++] = (uint32_t)m->col;
  out[n++] = (uint32_t)col_b(&m->colcxt, 1, 0); out[n++] = (uint32_t)col_b(&m->colcxt, 1, 1); out[n++] = m->colcxt.nlChar; out[n++] = (uint32_t)m->colcxt.rows;
  out[n++] = (uint32_t)col_len(&m->colcxt, 1, 0); out[n++] = (uint32_t)m->nl1; out[n++] = (uint32_t)m->colcxt.abovecellpos; out[n++] = m->numlen0;
  out[n++] = (uint32_t)(m->isText | m->isMath << 1 | m->isPre << 2 | m->isNowiki << 3); out[n++] = (uint32_t)m->deccode; out[n++] = (uint32_t)m->lastCW;
  return n;
}
uint32_t fx_cm_context(const FxCm* x, int i);
void orc_fx_model_contexts(const FxModel* m, uint32_t* out) {
  int n = 0;
  for (int k = 0; k < 6; k++) for (int i = 0; i < 8; i++) out[n++] = fx_cm_context(m->cmC[k], i);
  for (int k = 0; k < 8; k++) for (int i = 0; i < 8; i++) out[n++] = m->cmC1[k] ? fx_cm_context(m->cmC1[k], i) : 0;
  for (int k = 0; k < 18; k++) for (int i = 0; i < 8; i++) out[n++] = fx_cm_context(m->cmC2[k], i);
}
