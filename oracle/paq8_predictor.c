/* oracle/paq8_predictor.c -- TEST INFRASTRUCTURE ONLY (never linked into the product).
 *
 * The vendored paq8 model as cmix drives it, assembled from the restated blocks: Predictor::update (reference
 * src/models/paq8.cpp:8248-8362: the byte-level globals, the final APM / APM1 stages per block type) around
 * contextModel2 (:8101-8207: block-header parsing, the order-N ContextMap2, run maps, match models, the 15 sub-models in
 * their call order -- the order matters, they share one pseudo-random stream -- and the 1552-input / 28-selector
 * mixer). What PAQ8::Predict() hands to cmix is every value passed to AddPrediction (:504-507) in call order: the
 * 1552 mixer inputs as probabilities, the 28 first-layer outputs, and 10 or 11 stage outputs -- 1591 columns
 * (layer-0 columns 434..2024 of the cmix predictor).
 *
 * SCOPE: general data and text blocks. The image (1/4/8/24/32-bit, BMP / TGA payloads), audio (WAV) and JPEG
 * sub-models are not restated; their DETECTORS are (so that ordinary data takes exactly the reference's path), and a
 * stream that would switch one of them on makes orc_p8_predictor_update() return a negative code instead of a
 * prediction -- never a silently different number.
 * Pinned against the reference's own paq8::Predictor in tests/test_oracle_paq8core.py. */
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "paq8_tables.h"

typedef struct CM2 CM2;
typedef struct RCM RCM;
typedef struct Match Match;
typedef struct SMatch SMatch;
typedef struct Forest Forest;
typedef struct P8Sparse P8Sparse;
typedef struct P8CtxModel P8CtxModel;
typedef struct P8Small P8Small;
typedef struct Record Record;
typedef struct WordM WordM;
typedef struct Xml Xml;
typedef struct TextM TextM;
typedef struct Exe Exe;
typedef struct Lpm Lpm;
typedef struct OrcP8Mixer OrcP8Mixer;
typedef struct OrcP8Apm1 OrcP8Apm1;
typedef struct OrcP8StateMap32 OrcP8StateMap32;

CM2* orc_p8_cm2_new(uint64_t size_bytes, uint32_t count);
int orc_p8_cm2_step(CM2* c, int y_prev, int bpos, const uint64_t* ctx, int nset, int16_t* out, int* nout);
RCM* orc_p8_rcm_new(int m);
void orc_p8_rcm_set(RCM* r, uint64_t cx, int c1);
int orc_p8_rcm_mix(RCM* r, int bpos, int c0, int16_t* out);
Match* orc_p8_match_new(uint32_t size);
int orc_p8_match_step(Match* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* nout, int* expected_out);
SMatch* orc_p8_sparsematch_new(uint64_t size);
int orc_p8_sparsematch_step(SMatch* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* nout, int* sets);
Forest* orc_p8_dmc_new(int level);
int orc_p8_dmc_mix(Forest* f, int y, int bpos, int16_t* out);
P8Sparse* orc_p8_sparse_new(int which, int level);
int orc_p8_sparse_step(P8Sparse* m, int y, int bpos, int c0, const uint32_t* g, int seenbefore, int howmany, const uint8_t* last, int16_t* out);
P8CtxModel* orc_p8_ctxmodel_new(int which, int level);
int orc_p8_ctxmodel_step(P8CtxModel* m, int y, int bpos, int c0, uint32_t c4, uint32_t f4, int pos, const uint8_t* last, int16_t* out);
P8Small* orc_p8_small_new(int which);
int orc_p8_small_step(P8Small* m, int y, int bpos, int c0, uint32_t c4, uint32_t f4, uint32_t w5, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out);
Record* orc_p8_record_new(int level);
int orc_p8_record_step(Record* r, int y, int bpos, int c0, uint32_t c4, uint32_t* io, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* sets);
WordM* orc_p8_word_new(int level);
int orc_p8_word_step(WordM* m, int y, int bpos, int c0, uint32_t c4, uint32_t f4, uint32_t b3, int blpos, const uint8_t* hist, uint32_t bmask, int pos,
                     int16_t* out, uint32_t* g_out);
Xml* orc_p8_xml_new(int level);
int orc_p8_xml_step(Xml* x, int y, int bpos, int c0, uint32_t c4, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, uint32_t* xml_out);
TextM* orc_p8_text_new(uint32_t size_bytes);
int orc_p8_text_step(TextM* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* sel, uint32_t* stats);
Exe* orc_p8_exe_new(int level);
int orc_p8_exe_step(Exe* e, int y, int bpos, int c0, uint32_t c4, int blpos, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* sets,
                    uint32_t* x86_out);
Lpm* orc_p8_lpm_new(void);
int orc_p8_lpm_step(Lpm* m, int y, int bpos, int c0, const uint8_t* last, int16_t* out);
OrcP8Mixer* orc_p8_mixer_new(int n, int m, int s, int w);
int orc_p8_mixer_step(OrcP8Mixer* x, int y_prev, const int16_t* in, int nx, const int* cx, const int* range, int ncx, float* exported, int* nexp);
OrcP8Apm1* orc_p8_apm1_new(int n);
int orc_p8_apm1_p(OrcP8Apm1* a, int y, int pr, int cxt, int rate);
OrcP8StateMap32* orc_p8_statemap32_new(int n);
int orc_p8_statemap32_p(OrcP8StateMap32* s, int y, int cx, int limit);
OrcP8StateMap32* orc_p8_apm_new(int n);
int orc_p8_apm_p(OrcP8StateMap32* s, int y, int pr, int cx, int limit);
int orc_p8_stretch(int p);
int orc_p8_ilog(int x);
uint64_t orc_p8_combine64(uint64_t seed, uint64_t x);
uint32_t orc_p8_finalize64(uint64_t h, int bits);
uint64_t orc_p8_hash2(uint64_t a, uint64_t b);
uint64_t orc_p8_hash3(uint64_t a, uint64_t b, uint64_t c);
uint64_t orc_p8_hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d);

enum { FT_DEFAULT, FT_HDR, FT_JPEG, FT_EXE, FT_TEXT, FT_IMAGE1, FT_IMAGE4, FT_IMAGE8, FT_IMAGE8GRAY, FT_IMAGE24, FT_IMAGE32, FT_AUDIO };  /* preprocessor.h:11-12 */
enum { P8_NUM_INPUTS = 1552, P8_NUM_SETS = 28, P8_NUM_OUTPUTS = P8_NUM_INPUTS + P8_NUM_SETS + 11 };
enum { ORC_P8_ERR_IMAGE_BLOCK = -1, ORC_P8_ERR_JPEG = -2, ORC_P8_ERR_BMP = -3, ORC_P8_ERR_TGA = -4, ORC_P8_ERR_WAV = -5, ORC_P8_ERR_INTERNAL = -9 };

typedef struct {
  int level;
  uint8_t* buf; uint32_t bmask;     /* Buf of MEM()*8 bytes (:8368) */
  /* globals :167-200, :3866-3870, :4538 */
  unsigned long long nbytes;   /* bytes seen (the sanity check below: the first byte has fewer inputs; not `pos`, which a test may place) */
  int pos, c0, bpos, blpos;
  uint32_t c4, b2, b3, w4, w5, f4, tt, x4, x5, last_prediction;
  /* contextModel2 statics */
  CM2* cm; TextM* text; Match* match; SMatch* smatch; Forest* dmc; RCM *rcm7, *rcm9, *rcm10;
  OrcP8StateMap32* sm[2];
  OrcP8Mixer* mixer;
  uint32_t cxt[16];
  int ft2, filetype, size, info;
  P8Sparse *sparse0, *sparse1; P8CtxModel *nest, *dist, *indirect; P8Small *pic, *rec1; Record* rec; WordM* word; Xml* xml; Exe* exe; Lpm* lpm;
  uint32_t gword[9];                /* wordModel's globals: spaces, spacecount, words, wordcount, wordlen, wordlen1, frstchar, spafdo, col */
  /* detectors (imgModel :5386-5504, audioModel :5810-5865) */
  struct { uint32_t Header, Offset, Bpp, Size, Palette, HdrLess, Width, Height, BitMask; } bmp;
  struct { uint32_t Header, IdLength, Bpp, ImgType, MapSize, Width, Height; } tga;
  struct { uint32_t Header, Size, Channels, BitsPerSample, Chunk, Data; } wav;
  uint32_t wav_length;
  /* Predictor */
  int pr;
  OrcP8StateMap32* text_apm[4]; OrcP8Apm1* text_apm1[3]; OrcP8Apm1* gen_apm1[7];
  uint64_t misses;
  uint32_t match_length, match_expected, stat_record, text_first_letter, text_mask;
  int type, nexp;
  float out[P8_NUM_OUTPUTS];
} P8Predictor;

P8Predictor* orc_p8_predictor_new(int level) {
  P8Predictor* p = (P8Predictor*)calloc(1, sizeof *p);
  const uint64_t mem = 0x10000ull << level;  /* MEM() :190-192 */
  p->level = level;
  p->buf = (uint8_t*)calloc(mem * 8, 1); p->bmask = (uint32_t)(mem * 8 - 1);
  p->c0 = 1; p->last_prediction = 2048; p->pr = 2048;
  p->cm = orc_p8_cm2_new(mem * 16, 10);
  p->text = orc_p8_text_new((uint32_t)(mem * 16));
  p->match = orc_p8_match_new((uint32_t)(mem * 2));
  p->smatch = orc_p8_sparsematch_new(mem / 2);
  p->dmc = orc_p8_dmc_new(level);
  p->rcm7 = orc_p8_rcm_new((int)mem); p->rcm9 = orc_p8_rcm_new((int)mem); p->rcm10 = orc_p8_rcm_new((int)mem);
  p->sm[0] = orc_p8_statemap32_new(256); p->sm[1] = orc_p8_statemap32_new(256 * 256);
  p->mixer = orc_p8_mixer_new(P8_NUM_INPUTS, 77472, P8_NUM_SETS, 32);
  p->sparse0 = orc_p8_sparse_new(0, level); p->sparse1 = orc_p8_sparse_new(1, level);
  p->nest = orc_p8_ctxmodel_new(0, level); p->dist = orc_p8_ctxmodel_new(1, level); p->indirect = orc_p8_ctxmodel_new(2, level);
  p->pic = orc_p8_small_new(0); p->rec1 = orc_p8_small_new(1);
  p->rec = orc_p8_record_new(level); p->word = orc_p8_word_new(level); p->xml = orc_p8_xml_new(level); p->exe = orc_p8_exe_new(level);
  p->lpm = orc_p8_lpm_new();
  for (int i = 0; i < 4; i++) p->text_apm[i] = orc_p8_apm_new(0x10000);
  for (int i = 0; i < 3; i++) p->text_apm1[i] = orc_p8_apm1_new(0x10000);
  p->gen_apm1[0] = orc_p8_apm1_new(0x2000);
  for (int i = 1; i < 7; i++) p->gen_apm1[i] = orc_p8_apm1_new(0x10000);
  for (int i = 0; i < P8_NUM_OUTPUTS; i++) p->out[i] = 0.5f;  /* model_predictions(0.5, ...) :500 */
  return p;
}

#define RB(i) ((uint32_t)p->buf[((uint32_t)p->pos - (uint32_t)(i)) & p->bmask])
static uint32_t i4(const P8Predictor* p, int i) { return RB(i) + 256 * RB(i - 1) + 65536 * RB(i - 2) + 16777216 * RB(i - 3); }
static int i2(const P8Predictor* p, int i) { return (int)(RB(i) + 256 * RB(i - 1)); }
static uint32_t m4(const P8Predictor* p, int i) { return RB(i - 3) + 256 * RB(i - 2) + 65536 * RB(i - 1) + 16777216 * RB(i); }
static int m2(const P8Predictor* p, int i) { return (int)(RB(i) * 256 + RB(i - 1)); }
static unsigned ilog2u(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }
static uint64_t hash1(uint64_t a) { return (a + 1) * 0x9E3779B97F4A7C15ull; }

/* The detectors of the sub-models that are not restated, at a byte boundary. 0: ordinary data, go on. */
static int jpeg_detect(const P8Predictor* p) {  /* jpegModel :6098 -- SOI followed by a valid marker */
  const uint32_t b1 = RB(1);
  return (RB(4) == 0xFF && RB(3) == 0xD8 && RB(2) == 0xFF && ((b1 & 0xFE) == 0xC0 || b1 == 0xC4 || (b1 >= 0xDB && b1 <= 0xFE))) ? ORC_P8_ERR_JPEG : 0;
}
static int img_detect(P8Predictor* p) {  /* imgModel :5393-5483 with w == 0, eoi == 0 */
  const int pos = p->pos;
  if (pos >= 40 && !p->bmp.Header &&
      ((RB(54) == 'B' && RB(53) == 'M' && ((p->bmp.Offset = i4(p, 44)) & 0xFFFFFBF7) == 0x36 && i4(p, 40) == 0x28) ||
       (p->bmp.HdrLess = (i4(p, 40) == 0x28)))) {
    p->bmp.Width = i4(p, 36);
    p->bmp.Height = (uint32_t)abs((int)i4(p, 32));
    p->bmp.Bpp = (uint32_t)i2(p, 26);
    p->bmp.Size = i4(p, 20);
    p->bmp.Palette = i4(p, 4);
    const uint32_t bpp = p->bmp.Bpp;
    p->bmp.Header = (i4(p, 24) == 0) && (i2(p, 28) == 1) && (bpp == 1 || bpp == 4 || bpp == 8 || bpp == 24 || bpp == 32) && p->bmp.Width < 30000 &&
                    p->bmp.Height < 10000 && (!p->bmp.Palette || (1u << (bpp & 31)) >= p->bmp.Palette);
    if (p->bmp.Header) return ORC_P8_ERR_BMP;
  } else p->bmp.Offset -= (p->bmp.Offset > 0);
  if (pos >= 8 && !p->tga.Header) {
    if ((m4(p, 8) & 0xFFFFFF) == 0x010100 && (m4(p, 4) & 0xFFFFFFC7) == 0x00000100 && (RB(1) == 16 || RB(1) == 24 || RB(1) == 32)) {
      p->tga.Header = (uint32_t)pos; p->tga.IdLength = RB(8); p->tga.MapSize = RB(1) / 8; p->tga.Bpp = 8; p->tga.ImgType = 1;
    } else if ((m4(p, 8) & 0xFFFEFF) == 0x000200 && !m4(p, 4)) {
      p->tga.Header = (uint32_t)pos; p->tga.IdLength = RB(8); p->tga.ImgType = RB(6); p->tga.Bpp = (p->tga.ImgType == 2) ? 24 : 8;
    }
  } else if (p->tga.Header) {
    const uint32_t q = (uint32_t)pos - p->tga.Header;
    if (q == 8) {
      p->tga.Width = (uint32_t)i2(p, 4); p->tga.Height = (uint32_t)i2(p, 2);
      p->tga.Header *= (!i4(p, 8) && p->tga.Width && p->tga.Width < 0x3FFF && p->tga.Height && p->tga.Height < 0x3FFF);
    } else if (q == 10) {
      const uint16_t i = (uint16_t)m2(p, 2);
      if ((i & 0xFFF7) == (32 << 8)) p->tga.Bpp = 32;
      if ((uint32_t)(i & 0xFFD7) != (p->tga.Bpp << 8)) memset(&p->tga, 0, sizeof p->tga);
    }
    if (p->tga.Header && q == 10 + p->tga.IdLength + p->tga.MapSize * 256) {
      const int w = (int)((p->tga.Width * p->tga.Bpp) >> 3);
      if (w * (int)p->tga.Height > 64) return ORC_P8_ERR_TGA;
      p->tga.Header = 0;
    }
  }
  return 0;
}
static int wav_detect(P8Predictor* p) {  /* audioModel :5814-5851 with eoi == 0 */
  const int pos = p->pos;
  if (pos >= 4 && !p->wav.Header && m4(p, 4) == 0x52494646) { p->wav.Header = (uint32_t)pos; p->wav.Chunk = 0; p->wav_length = 0; }
  else if (p->wav.Header) {
    const int q = pos - (int)p->wav.Header;
    const uint32_t length = p->wav_length;
    if (q == 4) { p->wav.Size = i4(p, 4); p->wav.Header *= (p->wav.Size <= 0x3FFFFFFF); }
    else if (q == 8) p->wav.Header *= (m4(p, 4) == 0x57415645);
    else if (q == (int)(16 + length) && (m4(p, 8) != 0x666d7420 || ((p->wav.Chunk = i4(p, 4) - 16) & 0xFFFFFFFD) != 0)) {
      p->wav_length = ((i4(p, 4) + 1) & (uint32_t)(-2)) + 8;
      p->wav.Header *= !(m4(p, 8) == 0x666d7420 && (i4(p, 4) & 0xFFFFFFFD) != 16);
    } else if (q == (int)(20 + length)) {
      p->wav.Channels = RB(2);
      p->wav.Header *= ((p->wav.Channels == 1 || p->wav.Channels == 2) && (m4(p, 4) & 0xFFFFFCFF) == 0x01000000);
    } else if (q == (int)(32 + length)) {
      p->wav.BitsPerSample = RB(2);
      p->wav.Header *= ((p->wav.BitsPerSample == 8 || p->wav.BitsPerSample == 16) && (m2(p, 2) & 0xE7FF) == 0);
    } else if (q == (int)(40 + length + p->wav.Chunk) && m4(p, 8) != 0x64617461) {
      p->wav.Chunk += ((i4(p, 4) + 1) & (uint32_t)(-2)) + 8;
      p->wav.Header *= (p->wav.Chunk <= 0xFFFFF);
    } else if (q == (int)(40 + length + p->wav.Chunk)) {
      p->wav.Data = (i4(p, 4) + 1) & (uint32_t)(-2);
      if (p->wav.Data && (p->wav.Data % (p->wav.Channels * (p->wav.BitsPerSample / 8))) == 0) return ORC_P8_ERR_WAV;
    }
  }
  return 0;
}

/* contextModel2 :8101-8207. Returns the mixer's probability, or a negative ORC_P8_ERR_*. */
static int context_model2(P8Predictor* p, int y) {
  const int bpos = p->bpos, c0 = p->c0;
  int16_t in[P8_NUM_INPUTS + 64];
  int sel[P8_NUM_SETS], zero[P8_NUM_SETS] = {0};
  int nx = 0, ns = 0, k = 0;
  if (bpos == 0) {  /* block header: type byte, 4-byte size, 4-byte info for the types that carry one */
    --p->size; ++p->blpos;
    if (p->size == -1) { p->info = 0; p->ft2 = (int)RB(1); }
    const int has_info = (p->ft2 == FT_TEXT || (p->ft2 >= FT_IMAGE1 && p->ft2 <= FT_IMAGE32));
    if (p->size == -5 && !has_info) { p->size = (int)(RB(4) << 24 | RB(3) << 16 | RB(2) << 8 | RB(1)); p->blpos = 0; }
    if (p->size == -9) {
      p->size = (int)(RB(8) << 24 | RB(7) << 16 | RB(6) << 8 | RB(5));
      p->info = (int)(RB(4) << 24 | RB(3) << 16 | RB(2) << 8 | RB(1));
      p->blpos = 0;
      if (p->ft2 == FT_TEXT && p->info) p->size = p->info - 8;
    }
    if (!p->blpos) p->filetype = p->ft2;
    if (p->size == 0) p->filetype = FT_DEFAULT;
    p->type = p->filetype;
  }
  in[nx++] = 64;
  const uint8_t last1 = (uint8_t)RB(1);
  uint64_t set[10];
  int n = 0;
  if (bpos == 0) {  /* orders 1-6, 8 and 14 plus the letters-only order to the ContextMap2; orders 7, 10, 12 to the run maps */
    const uint8_t B = (uint8_t)(p->c4 & 0xFF);
    p->cxt[15] = isalpha(B) ? (uint32_t)orc_p8_combine64(p->cxt[15], (uint64_t)tolower(B)) : 0;
    set[n++] = p->cxt[15];
    for (int i = 14; i > 0; --i) p->cxt[i] = (uint32_t)orc_p8_combine64(p->cxt[i - 1], B);
    for (int i = 0; i < 7; ++i) set[n++] = p->cxt[i];
    orc_p8_rcm_set(p->rcm7, p->cxt[7], last1);
    set[n++] = p->cxt[8];
    orc_p8_rcm_set(p->rcm9, p->cxt[10], last1);
    orc_p8_rcm_set(p->rcm10, p->cxt[12], last1);
    set[n++] = p->cxt[14];
  }
  in[nx++] = (int16_t)((orc_p8_stretch(orc_p8_statemap32_p(p->sm[0], y, c0, 1023)) + 1) >> 1);
  in[nx++] = (int16_t)((orc_p8_stretch(orc_p8_statemap32_p(p->sm[1], y, c0 | (int)((uint32_t)last1 << 8), 1023)) + 1) >> 1);
  int order = orc_p8_cm2_step(p->cm, y, bpos, set, n, in + nx, &k); nx += k;
  orc_p8_rcm_mix(p->rcm7, bpos, c0, in + nx++);
  orc_p8_rcm_mix(p->rcm9, bpos, c0, in + nx++);
  orc_p8_rcm_mix(p->rcm10, bpos, c0, in + nx++);

  int expected = 0;
  k = 0;
  p->match_length = (uint32_t)orc_p8_match_step(p->match, y, bpos, c0, p->buf, p->bmask, p->pos, in + nx, &k, &expected); nx += k;
  if (bpos == 0) p->match_expected = (uint32_t)expected;  /* Stats->Match.expectedByte changes at byte boundaries only (:3593) */
  const int ismatch = orc_p8_ilog((int)(p->match_length & 0xffff));
  if (p->filetype >= FT_IMAGE1 && p->filetype <= FT_IMAGE32) return ORC_P8_ERR_IMAGE_BLOCK;
  if (bpos == 0) {
    int e;
    if (p->filetype != FT_EXE && (e = jpeg_detect(p)) != 0) return e;
    if (p->size > 0 && (e = img_detect(p)) != 0) return e;
    if ((e = wav_detect(p)) != 0) return e;
  }

  uint8_t last[64];
  for (int i = 0; i < 64; i++) last[i] = (uint8_t)RB(i + 1);
  int sm_sets[2], rec_sets[3], text_sets[8], exe_sets[6];
  uint32_t scratch = 0, tstats[6];
  k = 0;
  orc_p8_sparsematch_step(p->smatch, y, bpos, c0, p->buf, p->bmask, p->pos, in + nx, &k, sm_sets); nx += k;
  const uint32_t g[9] = {p->c4, p->f4, p->x4, p->w4, p->tt, p->gword[2], p->gword[0], p->gword[6], p->gword[7]};
  nx += orc_p8_sparse_step(p->sparse0, y, bpos, c0, g, ismatch, order, last, in + nx);
  nx += orc_p8_sparse_step(p->sparse1, y, bpos, c0, g, ismatch, order, last, in + nx);
  nx += orc_p8_ctxmodel_step(p->dist, y, bpos, c0, p->c4, p->f4, p->pos, last, in + nx);
  nx += orc_p8_small_step(p->pic, y, bpos, c0, p->c4, p->f4, p->w5, p->buf, p->bmask, p->pos, in + nx);
  uint32_t io[6] = {(uint32_t)p->blpos, 0, (uint32_t)p->filetype, p->stat_record, p->match_length, p->match_expected};
  io[1] = (bpos > 0) ? P8_ASCII_GROUP_C0[(1 << bpos) - 2 + (c0 & ((1 << bpos) - 1))] : 0;
  nx += orc_p8_record_step(p->rec, y, bpos, c0, p->c4, io, p->buf, p->bmask, p->pos, in + nx, rec_sets);
  p->stat_record = io[3];
  nx += orc_p8_small_step(p->rec1, y, bpos, c0, p->c4, p->f4, p->w5, p->buf, p->bmask, p->pos, in + nx);
  nx += orc_p8_word_step(p->word, y, bpos, c0, p->c4, p->f4, p->b3, p->blpos, p->buf, p->bmask, p->pos, in + nx, p->gword);
  nx += orc_p8_ctxmodel_step(p->nest, y, bpos, c0, p->c4, p->f4, p->pos, last, in + nx);
  nx += orc_p8_ctxmodel_step(p->indirect, y, bpos, c0, p->c4, p->f4, p->pos, last, in + nx);
  nx += orc_p8_dmc_mix(p->dmc, y, bpos, in + nx);
  nx += orc_p8_xml_step(p->xml, y, bpos, c0, p->c4, p->buf, p->bmask, p->pos, in + nx, &scratch);
  nx += orc_p8_text_step(p->text, y, bpos, c0, p->buf, p->bmask, p->pos, in + nx, text_sets, tstats);
  if (bpos == 0) { p->text_first_letter = tstats[4]; p->text_mask = tstats[5]; }
  nx += orc_p8_exe_step(p->exe, y, bpos, c0, p->c4, p->blpos, p->buf, p->bmask, p->pos, in + nx, exe_sets, &scratch);
  nx += orc_p8_lpm_step(p->lpm, y, bpos, c0, last, in + nx);
  /* 1552 once a byte boundary has been passed; fewer during the very first byte, when the context maps have no contexts yet */
  if (nx > P8_NUM_INPUTS || (p->nbytes > 0 && nx != P8_NUM_INPUTS)) { fprintf(stderr, "paq8 oracle: %d mixer inputs, expected %d\n", nx, P8_NUM_INPUTS); return ORC_P8_ERR_INTERNAL; }

  /* the 28 weight-set selectors, absolute positions in the 77472-row table, in the order the models call set() */
  int base = 0;
  sel[ns++] = base + sm_sets[0]; sel[ns++] = base + sm_sets[1]; base += 4 * 64 + 4 * 2048;
  for (int i = 0; i < 3; i++) sel[ns++] = base + rec_sets[i];
  base += 1024 + 512 + 11 * 32;
  static const int text_range[8] = {2048, 2048, 4096, 4096, 2048, 2048, 4096, 8192};
  for (int i = 0; i < 8; i++) { sel[ns++] = base + text_sets[i]; base += text_range[i]; }
  for (int i = 0; i < 6; i++) sel[ns++] = base + exe_sets[i];
  base += 3 * 1024 + 3 * 8192;
  const uint32_t words = p->gword[2];
  sel[ns++] = base + ((((order - 3) > 0 ? order - 3 : 0) << 3) | bpos); base += 64;
  order = (order - 5) > 0 ? order - 5 : 0;
  const uint32_t d = (uint32_t)c0 << (8 - bpos);
  uint32_t c = (d + (bpos == 1 ? p->b3 / 2 : 0)) & 192;
  if (!bpos) c = (words * 16) & 192;
  const uint32_t c1 = RB(1);
  sel[ns++] = base + (int)((uint32_t)order * 256 + (p->w4 & 240) + (p->b2 >> 4)); base += 1536;
  sel[ns++] = base + (int)((uint32_t)order * 256 + (p->w4 & 3) * 64 + ((words >> 1) & 63)); base += 1536;
  sel[ns++] = base + (int)((uint32_t)bpos * 256 + c1); base += 2048;
  sel[ns++] = base + (int)((uint32_t)(bpos < 5 ? bpos : 5) * 256 + (p->tt & 63) + c); base += 1536;
  sel[ns++] = base + (int)((uint32_t)order * 256 + ((d | (c1 >> bpos)) & 248) + (uint32_t)bpos); base += 1536;
  sel[ns++] = base + (int)((uint32_t)bpos * 256 + ((((words << bpos) & 255) >> bpos) | (d & 255))); base += 2048;
  sel[ns++] = base + (int)(p->last_prediction / 16); base += 256;
  sel[ns++] = base + c0; base += 256;
  if (ns != P8_NUM_SETS || base != 77472) return ORC_P8_ERR_INTERNAL;
  int nexp = 0;
  const int pr = orc_p8_mixer_step(p->mixer, y, in, nx, sel, zero, ns, p->out, &nexp);
  p->nexp = nexp;  /* AddPrediction's running index: the stage outputs follow */
  return nexp == nx + P8_NUM_SETS ? pr : ORC_P8_ERR_INTERNAL;
}

/* state injection (the twin of oracle/ref_paq8core.cpp refp8_set_pos): the byte position, before the first update */
void orc_p8_predictor_set_pos(P8Predictor* p, int pos) { p->pos = pos; }

/* Predictor::update :8248-8362 = PAQ8::Perceive(bit). Returns the new prediction (12 bits) or a negative ORC_P8_ERR_*;
 * out1591 (may be NULL) receives PAQ8::Predict()'s vector for the next bit. */
int orc_p8_predictor_update(P8Predictor* p, int y, float* out1591) {
  static const uint32_t WRT_mpw[16] = {4, 4, 3, 2, 2, 2, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0}, WRT_mtt[16] = {0, 0, 1, 2, 3, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7};  /* :3868-3869 */
  p->c0 += p->c0 + y;
  p->misses += p->misses + (uint64_t)((p->pr >> 11) != y);
  if (p->c0 >= 256) {
    p->buf[(uint32_t)p->pos++ & p->bmask] = (uint8_t)p->c0;
    p->nbytes++;
    p->c0 -= 256;
    const uint32_t b = (uint32_t)p->c0;
    p->c4 = (p->c4 << 8) + b;
    uint32_t i = WRT_mpw[b >> 4];
    p->w4 = p->w4 * 4 + i;
    if (p->b2 == 3) i = 2;
    p->w5 = p->w5 * 4 + i;
    p->b3 = p->b2;
    p->b2 = b;
    p->x4 = p->x4 * 256 + b; p->x5 = (p->x5 << 8) + b;
    if (b == '.' || b == '!' || b == '?' || b == '/' || b == ')') {
      p->w5 = (p->w5 << 8) | 0x3ff; p->f4 = (p->f4 & 0xfffffff0) + 2; p->x5 = (p->x5 << 8) + b; p->x4 = p->x4 * 256 + b;
      if (b != '!') { p->w4 |= 12; p->tt = (p->tt & 0xfffffff8) + 1; p->b3 = '.'; }
    }
    uint32_t cc = b;
    if (cc == 32) --cc;
    p->tt = p->tt * 8 + WRT_mtt[cc >> 4];
    p->f4 = p->f4 * 16 + (cc >> 4);
    p->c0 = 1;
  }
  p->bpos = (p->bpos + 1) & 7;
  int pr0 = context_model2(p, y);
  if (pr0 < 0) return pr0;
  const int c0 = p->c0, bpos = p->bpos;
  const uint32_t c4 = p->c4, mlen = ilog2u(p->match_length + 1) < 3 ? ilog2u(p->match_length + 1) : 3, eb = p->match_expected;
  float* o = p->out + p->nexp;
  const float cf = (float)(1.0 / 4095);
#define EXPORT(v) (*o++ = (float)(v) * cf)
  EXPORT(pr0);
  int pr, pr1, pr2, pr3;
  if (p->type == FT_TEXT) {  /* :8281-8297 */
    const int limit = 0x3FF >> ((p->blpos < 0xFFF) * 2);
    pr = orc_p8_apm_p(p->text_apm[0], y, pr0, (c0 << 8) | (int)(p->text_mask & 0xF) | (int)((p->misses & 0xF) << 4), limit); EXPORT(pr);
    pr1 = orc_p8_apm_p(p->text_apm[1], y, pr0, (int)orc_p8_finalize64(orc_p8_hash4((uint64_t)bpos, p->misses & 3, c4 & 0xffff, p->text_mask >> 4), 16), limit); EXPORT(pr1);
    pr2 = orc_p8_apm_p(p->text_apm[2], y, pr0, (int)orc_p8_finalize64(orc_p8_hash3((uint64_t)c0, eb, mlen), 16), limit); EXPORT(pr2);
    pr3 = orc_p8_apm_p(p->text_apm[3], y, pr0, (int)orc_p8_finalize64(orc_p8_hash3((uint64_t)c0, c4 & 0xffff, p->text_first_letter), 16), limit); EXPORT(pr3);
    pr0 = (pr0 + pr1 + pr2 + pr3 + 2) >> 2; EXPORT(pr0);
    pr1 = orc_p8_apm1_p(p->text_apm1[0], y, pr0, (int)orc_p8_finalize64(orc_p8_hash3(eb, mlen, c4 & 0xff), 16), 7); EXPORT(pr1);
    pr2 = orc_p8_apm1_p(p->text_apm1[1], y, pr, (int)orc_p8_finalize64(orc_p8_hash2((uint64_t)c0, c4 & 0x00ffffff), 16), 6); EXPORT(pr2);
    pr3 = orc_p8_apm1_p(p->text_apm1[2], y, pr, (int)orc_p8_finalize64(orc_p8_hash2((uint64_t)c0, c4 & 0xffffff00), 16), 6); EXPORT(pr3);
    pr = (pr + pr1 + pr2 + pr3 + 2) >> 2; EXPORT(pr);
    pr = (pr + pr0 + 1) >> 1; EXPORT(pr);
  } else {                   /* :8339-8358; the image types never get here (refused above) */
    pr = orc_p8_apm1_p(p->gen_apm1[0], y, pr0, (int)((mlen << 11) | ((uint32_t)c0 << 3) | (uint32_t)(p->misses & 0x7)), 7); EXPORT(pr);
    const uint16_t ctx1 = (uint16_t)((uint32_t)c0 | RB(1) << 8);
    const uint16_t ctx2 = (uint16_t)((uint32_t)c0 ^ orc_p8_finalize64(hash1(c4 & 0xffff), 16));
    const uint16_t ctx3 = (uint16_t)((uint32_t)c0 ^ orc_p8_finalize64(hash1(c4 & 0xffffff), 16));
    pr1 = orc_p8_apm1_p(p->gen_apm1[1], y, pr0, ctx1, 7); EXPORT(pr1);
    pr2 = orc_p8_apm1_p(p->gen_apm1[2], y, pr0, ctx2, 7); EXPORT(pr2);
    pr3 = orc_p8_apm1_p(p->gen_apm1[3], y, pr0, ctx3, 7); EXPORT(pr3);
    pr0 = (pr0 + pr1 + pr2 + pr3 + 2) >> 2;
    pr1 = orc_p8_apm1_p(p->gen_apm1[4], y, pr, (int)((eb << 8) | RB(1)), 7); EXPORT(pr1);
    pr2 = orc_p8_apm1_p(p->gen_apm1[5], y, pr, ctx2, 7); EXPORT(pr2);
    pr3 = orc_p8_apm1_p(p->gen_apm1[6], y, pr, ctx3, 7); EXPORT(pr3);
    pr = (pr + pr1 + pr2 + pr3 + 2) >> 2; EXPORT(pr);
    pr = (pr + pr0 + 1) >> 1; EXPORT(pr);
  }
#undef EXPORT
  p->pr = pr;
  p->last_prediction = (uint32_t)pr;
  if (out1591) memcpy(out1591, p->out, sizeof p->out);
  return pr;
}
