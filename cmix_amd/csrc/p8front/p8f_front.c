/* p8front/p8f_front.c -- HOST FRONT END of the paq8 stage: the driver. Predictor::update (reference
 * src/models/paq8.cpp:8248-8362: the byte-level globals) around contextModel2 (:8101-8207: block-header parsing, the
 * contexts of the order-N ContextMap2, run maps, the 15 sub-models in their call order) with every learning table
 * replaced by a recorder (p8f_emit.h): per step it leaves the hashed contexts, one op word per small map, the host part
 * of the 28 mixer selectors and the host part of the APM-chain contexts in a P8Chunk (../p8_rec.h). The device kernels
 * (../p8stage.hip) do the rest: tables, mixer, APM chains, the 1591 exported values.
 *
 * SCOPE: general data and text blocks. The image (1/4/8/24/32-bit, BMP / TGA payloads), audio (WAV) and JPEG
 * sub-models are not built; their DETECTORS are (so that ordinary data takes exactly the reference's path), and a
 * stream that would switch one of them on makes p8f_front_run() return a negative code -- never a silently different
 * number. Parity: tests/test_p8stage_host.py, tests/test_zgpu_p8stage.py (columns 434..2024 of reference traces). */
#include <ctype.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "p8f_emit.h"
#include "p8f_front.h"
#include "p8f_tables.h"

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
typedef struct P8fStateMap32 P8fStateMap32;

CM2* p8f_cm2_new(uint64_t size_bytes, uint32_t count);
int p8f_cm2_step(CM2* c, int y_prev, int bpos, const uint64_t* ctx, int nset, int16_t* out, int* nout);
RCM* p8f_rcm_new(int m);
void p8f_rcm_set(RCM* r, uint64_t cx, int c1);
int p8f_rcm_mix(RCM* r, int bpos, int c0, int16_t* out);
Match* p8f_match_new(uint32_t size);
int p8f_match_step(Match* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* nout, int* expected_out);
SMatch* p8f_sparsematch_new(uint64_t size);
int p8f_sparsematch_step(SMatch* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* nout, int* sets);
Forest* p8f_dmc_new(int level);
int p8f_dmc_mix(Forest* f, int y, int bpos, int16_t* out);
P8Sparse* p8f_sparse_new(int which, int level);
int p8f_sparse_step(P8Sparse* m, int y, int bpos, int c0, const uint32_t* g, int seenbefore, int howmany, const uint8_t* last, int16_t* out);
P8CtxModel* p8f_ctxmodel_new(int which, int level);
int p8f_ctxmodel_step(P8CtxModel* m, int y, int bpos, int c0, uint32_t c4, uint32_t f4, int pos, const uint8_t* last, int16_t* out);
P8Small* p8f_small_new(int which);
int p8f_small_step(P8Small* m, int y, int bpos, int c0, uint32_t c4, uint32_t f4, uint32_t w5, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out);
Record* p8f_record_new(int level);
int p8f_record_step(Record* r, int y, int bpos, int c0, uint32_t c4, uint32_t* io, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* sets);
WordM* p8f_word_new(int level);
int p8f_word_step(WordM* m, int y, int bpos, int c0, uint32_t c4, uint32_t f4, uint32_t b3, int blpos, const uint8_t* hist, uint32_t bmask, int pos,
                  int16_t* out, uint32_t* g_out);
Xml* p8f_xml_new(int level);
int p8f_xml_step(Xml* x, int y, int bpos, int c0, uint32_t c4, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, uint32_t* xml_out);
TextM* p8f_text_new(uint32_t size_bytes);
int p8f_text_step(TextM* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* sel, uint32_t* stats);
Exe* p8f_exe_new(int level);
int p8f_exe_step(Exe* e, int y, int bpos, int c0, uint32_t c4, int blpos, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* sets,
                 uint32_t* x86_out);
Lpm* p8f_lpm_new(void);
int p8f_lpm_step(Lpm* m, int y, int bpos, int c0, const uint8_t* last, int16_t* out);
typedef struct Audio8 Audio8;
typedef struct Wav16 Wav16;
Audio8* p8f_audio8_new(void);
int p8f_audio8_step(Audio8* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int info, int blpos, uint32_t* record, int16_t* out, int* sets, int* ranges);
Wav16* p8f_wav16_new(int level);
int p8f_wav16_step(Wav16* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int info, uint32_t* record, int16_t* out, int* sets, int* ranges, int* cm_active);
struct Jpeg* p8f_jpeg_new(int level);
int p8f_jpeg_step(struct Jpeg* j, int y, int bpos, const uint8_t* hist, uint32_t bmask, int pos, int16_t* out, int* sets, int* ranges, int* kind);
typedef struct Im4 Im4;
Im4* p8f_im4_new(int level);
int p8f_im4_step(Im4* m, int y, int bpos, int c0, uint32_t c4, const uint8_t* hist, uint32_t bmask, int pos, int w, int16_t* out, int* sets, int* ranges);
typedef struct Im1 Im1;
Im1* p8f_im1_new(void);
int p8f_im1_step(Im1* m, int y, int bpos, const uint8_t* hist, uint32_t bmask, int pos, int w, int16_t* out, int* sets, int* ranges);
typedef struct Im24 Im24;
typedef struct Im8 Im8;
Im8* p8f_im8_new(int level);
int p8f_im8_step(Im8* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int w, int gray, int16_t* out, int* sets, int* ranges, uint32_t* stats);
Im24* p8f_im24_new(int level);
int p8f_im24_step(Im24* m, int y, int bpos, int c0, const uint8_t* hist, uint32_t bmask, int pos, int w, int alpha, int16_t* out, int* sets, int* ranges, uint32_t* stats);
P8fStateMap32* p8f_statemap32_new(int n);
void p8f_statemap32_emit(P8fStateMap32* s, int cx, int zero, int16_t* out);
int p8f_ilog(int x);
uint64_t p8f_combine64(uint64_t seed, uint64_t x);
uint32_t p8f_finalize64(uint64_t h, int bits);
uint64_t p8f_hash2(uint64_t a, uint64_t b);
uint64_t p8f_hash3(uint64_t a, uint64_t b, uint64_t c);
uint64_t p8f_hash4(uint64_t a, uint64_t b, uint64_t c, uint64_t d);

enum { FT_DEFAULT, FT_HDR, FT_JPEG, FT_EXE, FT_TEXT, FT_IMAGE1, FT_IMAGE4, FT_IMAGE8, FT_IMAGE8GRAY, FT_IMAGE24, FT_IMAGE32, FT_AUDIO };  /* preprocessor.h:11-12 */

/* ---- allocation tracking (p8f_alloc.h) ---- */
typedef struct Blk { struct Blk* next; } Blk;
/* A compressor coding a step of an image / audio / JPEG model: the archive is the reference's, but THIS library's decoder cannot restore it (the late-bit
   form of those steps is not built: P8F_ERR_IMAGE_LATE); say so once per process, on the compress side, before the user finds out while decoding.
   CMX_P8_MEDIA_WARNING=0 silences it. */
static void p8f_media_step_notice(const P8Emit* em) {
  static int said = 0;
  if (said || !em->chunk || !em->chunk->xops) return;
  said = 1;
  const char* e = getenv("CMX_P8_MEDIA_WARNING");
  if (e && e[0] == '0') return;
  fprintf(stderr, "cmix_amd: this stream codes steps of paq8's image / audio / JPEG models; the file decodes with the reference binary, NOT with this library's "
                  "decoder (cmix_dropin -d stops at the first such step: the decoder's form of those models is not built)\n");
}

static __thread Blk** g_blocks;
#undef calloc
#undef malloc
void* p8f_tracked_calloc(size_t n, size_t size) {
  Blk* b = (Blk*)calloc(1, n * size + 32);
  if (!b) { fprintf(stderr, "paq8 front end: out of memory (%zu bytes)\n", n * size); abort(); }
  if (g_blocks) { b->next = *g_blocks; *g_blocks = b; }
  return (char*)b + 32;   /* 32: keeps the alignment malloc gives */
}

typedef struct {
  int level;
  uint8_t* buf; uint32_t bmask;     /* Buf of MEM()*8 bytes (:8368) */
  /* globals :167-200, :3866-3870, :4538 */
  int pos, c0, bpos, blpos;
  uint32_t c4, b2, b3, w4, w5, f4, tt, x4, x5;
  /* contextModel2 statics */
  CM2* cm; TextM* text; Match* match; SMatch* smatch; Forest* dmc; RCM *rcm7, *rcm9, *rcm10;
  P8fStateMap32* sm[2];
  uint32_t cxt[16];
  int ft2, filetype, size, info;
  P8Sparse *sparse0, *sparse1; P8CtxModel *nest, *dist, *indirect; P8Small *pic, *rec1; Record* rec; WordM* word; Xml* xml; Exe* exe; Lpm* lpm;
  uint32_t gword[9];                /* wordModel's globals: spaces, spacecount, words, wordcount, wordlen, wordlen1, frstchar, spafdo, col */
  /* detectors (imgModel :5386-5504, audioModel :5810-5865) */
  struct { uint32_t Header, Offset, Bpp, Size, Palette, HdrLess, Width, Height, BitMask; } bmp;
  int img_gray, img_pltorder;                      /* imgModel's statics `gray`, `pltorder` while a palette is being skipped */
  int img_w, img_bpp, img_eoi, img_alpha;          /* imgModel's statics w, bpp, eoi, alpha: an image payload is being modelled while w != 0 */
  Im24* im24;
  Im8* im8;
  Audio8* audio8;
  Wav16* wav16;
  Im1* im1;
  Im4* im4;
  uint32_t wav_eoi, wav_info;                      /* audioModel's statics eoi, info: PCM samples are being modelled while info != 0 */
  int own_silent;                                  /* the step's model has a ContextMap of its own and it has no contexts this byte (2: a coded JPEG step, exported through the model's export map) */
  int jpeg_const;                                  /* a stuffed / restart step of the JPEG model: its one constant input */
  uint32_t img_stats[8];                           /* ModelStats.Image of the byte: W, N, NN, WW, Wp1, Np1, plane, ctx */
  int model;                                       /* P8_MODEL_* of the current step */
  int nsel;                                        /* weight sets of the current step */
  struct Jpeg* jpeg;                                /* jpegModel :5911-6597 (p8f_jpeg.c) */
  struct { uint32_t Header, IdLength, Bpp, ImgType, MapSize, Width, Height; } tga;
  struct { uint32_t Header, Size, Channels, BitsPerSample, Chunk, Data; } wav;
  uint32_t wav_length;
  uint32_t match_length, match_expected, stat_record, text_first_letter, text_mask;
  int type, nx;
  uint64_t nbytes;                                 /* bytes seen (the sanity check of front_step: the first byte has fewer inputs; NOT `pos`, which a test may place) */
  int16_t in[P8_NX + 64];
} P8Predictor;

struct P8Front {
  P8Emit emit;
  P8Predictor* p;
  Blk* blocks;
  uint64_t steps;     /* steps emitted so far = bits of the stream handed in */
  int last_bit, level, err;
};

static P8Predictor* predictor_new(int level) {
  P8Predictor* p = (P8Predictor*)p8f_tracked_calloc(1, sizeof *p);
  const uint64_t mem = 0x10000ull << level;  /* MEM() :190-192 */
  p->level = level;
  p->buf = (uint8_t*)p8f_tracked_calloc(mem * 8, 1); p->bmask = (uint32_t)(mem * 8 - 1);
  p->c0 = 1;
  p->cm = p8f_cm2_new(mem * 16, 10);
  p->text = p8f_text_new((uint32_t)(mem * 16));
  p->match = p8f_match_new((uint32_t)(mem * 2));
  p->smatch = p8f_sparsematch_new(mem / 2);
  p->dmc = p8f_dmc_new(level);
  p->rcm7 = p8f_rcm_new((int)mem); p->rcm9 = p8f_rcm_new((int)mem); p->rcm10 = p8f_rcm_new((int)mem);
  p->sm[0] = p8f_statemap32_new(256); p->sm[1] = p8f_statemap32_new(256 * 256);
  p->sparse0 = p8f_sparse_new(0, level); p->sparse1 = p8f_sparse_new(1, level);
  p->nest = p8f_ctxmodel_new(0, level); p->dist = p8f_ctxmodel_new(1, level); p->indirect = p8f_ctxmodel_new(2, level);
  p->pic = p8f_small_new(0); p->rec1 = p8f_small_new(1);
  p->rec = p8f_record_new(level); p->word = p8f_word_new(level); p->xml = p8f_xml_new(level); p->exe = p8f_exe_new(level);
  p->lpm = p8f_lpm_new();
  p8f_emit_model(p8f_cur, P8_MODEL_IM24);   /* its maps and ContextMap go to the model's own tables (p8_rec.h P8XLayout) */
  p->im24 = p8f_im24_new(level);
  p8f_emit_model(p8f_cur, P8_MODEL_IM8);
  p->im8 = p8f_im8_new(level);
  p8f_emit_model(p8f_cur, P8_MODEL_AUDIO8);
  p->audio8 = p8f_audio8_new();
  p8f_emit_model(p8f_cur, P8_MODEL_WAV16);
  p->wav16 = p8f_wav16_new(level);
  p8f_emit_model(p8f_cur, P8_MODEL_IM1);
  p->im1 = p8f_im1_new();
  p8f_emit_model(p8f_cur, P8_MODEL_IM4);
  p->im4 = p8f_im4_new(level);
  p8f_emit_model(p8f_cur, P8_MODEL_JPEG);
  p->jpeg = p8f_jpeg_new(level);
  p8f_emit_model(p8f_cur, 0);
  return p;
}

#define RB(i) ((uint32_t)p->buf[((uint32_t)p->pos - (uint32_t)(i)) & p->bmask])
static uint32_t i4(const P8Predictor* p, int i) { return RB(i) + 256 * RB(i - 1) + 65536 * RB(i - 2) + 16777216 * RB(i - 3); }
static int i2(const P8Predictor* p, int i) { return (int)(RB(i) + 256 * RB(i - 1)); }
static uint32_t m4(const P8Predictor* p, int i) { return RB(i - 3) + 256 * RB(i - 2) + 65536 * RB(i - 1) + 16777216 * RB(i); }
static int m2(const P8Predictor* p, int i) { return (int)(RB(i) * 256 + RB(i - 1)); }
static unsigned ilog2u(unsigned x) { unsigned n = 0; while (x > 1) { x >>= 1; ++n; } return n; }
static uint64_t hash1(uint64_t a) { return (a + 1) * 0x9E3779B97F4A7C15ull; }

/* The detectors of the sub-models that are not restated, at a byte boundary. 0: ordinary data, go on. Each one follows the
 * reference up to the byte at which the sub-model would switch on (its return value turns non-zero and contextModel2 :8161-8167
 * takes the short path): only there is the stream refused. A header-like pattern that the reference itself drops (a DIB header
 * with no plausible pixel area, an SOI that is not followed by a valid scan header) is ordinary data here too. */

/* imgModel :5386-5483 with w == 0, eoi == 0. *record = Stats.Record, which the palette walk of an 8-bit header rewrites (:5363) and
 * recordModel reads (:4225). */
static void img_check_gray(P8Predictor* p, uint32_t x, int a, uint32_t* record) {   /* CheckIfGrayscale(x, a) :5355-5377 with w == 0 */
  if (!p->img_gray || (x % (uint32_t)(3 + a)) != 0) return;
  for (int i = 0; i < 3 + a && p->img_gray; i++) {
    const uint8_t B = (uint8_t)RB(4 - i);
    if (p->img_gray >> 9) {
      p->img_gray = 0x100 | B;
      p->img_pltorder = 1 - 2 * (B > 0);
      *record = (*record & 0xFFFF) | ((uint32_t)(3 + a) << 16);
      continue;
    }
    if (!i) {
      p->img_gray = p->img_gray & ((((int)B - (p->img_gray & 0xFF)) == p->img_pltorder) << 8);
      p->img_gray |= p->img_gray ? B : 0;
    } else if (i == 3) p->img_gray &= ((!B || B == 0xFF) * 0x1FF);
    else p->img_gray &= ((B == (p->img_gray & 0xFF)) * 0x1FF);
  }
}
/* imgModel's byte-boundary part :5393-5483: header detection; when a payload of more than 64 bytes follows, w / bpp / eoi / alpha are
 * set and the image model runs until pos reaches eoi (img_model below). 0, or P8F_ERR_BMP / P8F_ERR_TGA for the pixel formats whose
 * models are not built (1, 4, 8 bits). */
static int img_detect(P8Predictor* p, uint32_t* record) {
  const int pos = p->pos, eoi = p->img_eoi;
  if (pos >= eoi + 40 && !p->bmp.Header &&
      ((RB(54) == 'B' && RB(53) == 'M' && ((p->bmp.Offset = i4(p, 44)) & 0xFFFFFBF7) == 0x36 && i4(p, 40) == 0x28) ||
       (p->bmp.HdrLess = (i4(p, 40) == 0x28)))) {
    p->bmp.Width = i4(p, 36);
    p->bmp.Height = (uint32_t)abs((int)i4(p, 32));
    p->bmp.Bpp = (uint32_t)i2(p, 26);
    p->bmp.Size = i4(p, 20);
    p->bmp.Palette = i4(p, 4);
    const uint32_t bpp = p->bmp.Bpp, W = p->bmp.Width;
    p->bmp.Header = (i4(p, 24) == 0) && (i2(p, 28) == 1) && (bpp == 1 || bpp == 4 || bpp == 8 || bpp == 24 || bpp == 32) && p->bmp.Width < 30000 &&
                    p->bmp.Height < 10000 && (!p->bmp.Palette || (1u << (bpp & 31)) >= p->bmp.Palette);
    if (p->bmp.Header) {   /* a plausible header: skip the palette (Offset bytes), then look at the pixel area (:5405-5425) */
      p->bmp.Offset = p->bmp.HdrLess ? ((bpp < 24) ? (p->bmp.Palette ? p->bmp.Palette * 4 : (uint32_t)(4 << bpp)) : 0) : p->bmp.Offset - 54;
      p->img_gray = (bpp == 8) ? 0x300 : 0;
      if (p->bmp.HdrLess && (W * 2 == p->bmp.Height) && bpp > 1 &&
          ((p->bmp.Size > 0 && p->bmp.Size == ((W * p->bmp.Height * (bpp + 1)) >> 4)) ||
           ((!p->bmp.Size || p->bmp.Size < ((W * p->bmp.Height * bpp) >> 3)) &&
            (W == 8 || W == 10 || W == 14 || W == 16 || W == 20 || W == 22 || W == 24 || W == 32 || W == 40 || W == 48 || W == 60 || W == 64 || W == 72 ||
             W == 80 || W == 96 || W == 128 || W == 256))))
        p->bmp.Height = p->bmp.BitMask = W;   /* icon / cursor: colour image and 1-bit AND mask of equal size */
    }
  } else {
    p->bmp.Offset -= (p->bmp.Offset > 0);
    if (!p->img_w) img_check_gray(p, p->bmp.Offset, 1, record);   /* CheckIfGrayscale tests !w itself (:5356) */
  }
  if (!p->bmp.Offset && (p->bmp.Header > 0 || p->bmp.BitMask > 0) && pos >= eoi) {   /* :5427-5438 */
    if (!p->bmp.Header && p->bmp.BitMask) { p->bmp.Header = p->bmp.Bpp = 1; p->bmp.Width = p->bmp.BitMask; p->bmp.BitMask = 0; }
    const int bpp = (int)p->bmp.Bpp;
    p->img_bpp = bpp;
    p->img_w = (bpp > 4) ? (int)((p->bmp.Width * (uint32_t)(bpp >> 3) + 3) & (uint32_t)(-4)) : (bpp == 1) ? (int)((((p->bmp.Width - 1) >> 5) + 1) * 4)
                                                                                                       : (int)(((p->bmp.Width * 4 + 31) >> 5) * 4);
    p->img_alpha = (bpp == 32);
    const int n = (int)((uint32_t)p->img_w * p->bmp.Height);
    if (n > 64) {
      p->img_eoi = n + pos;
      /* (every pixel size the header test admits has its model) */
    } else { p->img_eoi = 0; p->bmp.Header = 0; p->img_w = 0; }   /* too small to be an image: dropped, as the reference drops it */
  }
  if (pos >= p->img_eoi + 8 && !p->tga.Header) {
    if ((m4(p, 8) & 0xFFFFFF) == 0x010100 && (m4(p, 4) & 0xFFFFFFC7) == 0x00000100 && (RB(1) == 16 || RB(1) == 24 || RB(1) == 32)) {
      p->tga.Header = (uint32_t)pos; p->tga.IdLength = RB(8); p->tga.MapSize = RB(1) / 8; p->tga.Bpp = 8; p->tga.ImgType = 1;
    } else if ((m4(p, 8) & 0xFFFEFF) == 0x000200 && !m4(p, 4)) {
      p->tga.Header = (uint32_t)pos; p->tga.IdLength = RB(8); p->tga.ImgType = RB(6); p->tga.Bpp = (p->tga.ImgType == 2) ? 24 : 8;
    }
  } else if (!p->img_w && p->tga.Header) {
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
      p->img_w = (int)((p->tga.Width * p->tga.Bpp) >> 3);
      p->img_gray = (p->tga.ImgType == 3);
      p->img_bpp = (int)p->tga.Bpp;
      p->img_alpha = (p->tga.Bpp == 32);
      const int n = p->img_w * (int)p->tga.Height;
      if (n > 64) {
        p->img_eoi = n + pos;
        if (p->img_bpp < 8) return P8F_ERR_TGA;
      } else { p->img_eoi = 0; p->tga.Header = 0; p->img_w = 0; }
    }
  }
  return 0;
}
static int wav_detect(P8Predictor* p) {  /* audioModel's byte-boundary part :5814-5851 */
  const int pos = p->pos;
  if (pos >= (int)(p->wav_eoi + 4) && !p->wav.Header && m4(p, 4) == 0x52494646) { p->wav.Header = (uint32_t)pos; p->wav.Chunk = 0; p->wav_length = 0; }
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
      if (p->wav.Data && (p->wav.Data % (p->wav.Channels * (p->wav.BitsPerSample / 8))) == 0) {   /* PCM samples follow: the audio model runs to eoi */
        p->wav_info = (p->wav.Channels + p->wav.BitsPerSample / 4 - 3) + 1;
        p->wav_eoi = (uint32_t)pos + p->wav.Data;
      }
    }
  }
  return 0;
}


/* contextModel2 :8101-8207 up to the mixer: 0 or a negative P8F_ERR_*. sel[] receives the host part of the 28 selectors. */
static int context_model2(P8Predictor* p, int y, int32_t* sel) {
  const int bpos = p->bpos, c0 = p->c0;
  int16_t* in = p->in;
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
    p->cxt[15] = isalpha(B) ? (uint32_t)p8f_combine64(p->cxt[15], (uint64_t)tolower(B)) : 0;
    set[n++] = p->cxt[15];
    for (int i = 14; i > 0; --i) p->cxt[i] = (uint32_t)p8f_combine64(p->cxt[i - 1], B);
    for (int i = 0; i < 7; ++i) set[n++] = p->cxt[i];
    p8f_rcm_set(p->rcm7, p->cxt[7], last1);
    set[n++] = p->cxt[8];
    p8f_rcm_set(p->rcm9, p->cxt[10], last1);
    p8f_rcm_set(p->rcm10, p->cxt[12], last1);
    set[n++] = p->cxt[14];
  }
  p8f_statemap32_emit(p->sm[0], c0, 0, in + nx); nx++;                                   /* (stretch(sm.p(c0)) + 1) >> 1 :8155 */
  p8f_statemap32_emit(p->sm[1], c0 | (int)((uint32_t)last1 << 8), 0, in + nx); nx++;   /* :8156 */
  p8f_cm2_step(p->cm, y, bpos, set, n, in + nx, &k); nx += k;   /* its return value, the "order", is device state */
  p8f_rcm_mix(p->rcm7, bpos, c0, in + nx++);
  p8f_rcm_mix(p->rcm9, bpos, c0, in + nx++);
  p8f_rcm_mix(p->rcm10, bpos, c0, in + nx++);

  int expected = 0;
  k = 0;
  p->match_length = (uint32_t)p8f_match_step(p->match, y, bpos, c0, p->buf, p->bmask, p->pos, in + nx, &k, &expected); nx += k;
  if (bpos == 0) p->match_expected = (uint32_t)expected;  /* Stats->Match.expectedByte changes at byte boundaries only (:3593) */
  const int ismatch = p8f_ilog((int)(p->match_length & 0xffff));
  /* :8161-8167: an image block, or an image / audio / JPEG payload found inside another block, is modelled by its own sub-model
   * INSTEAD of everything below */
  p->model = P8_MODEL_GENERIC; p->nsel = P8_NSEL;
  int img_w = 0, img_alpha = 0, img_bpp = 24, img_gray = 0, by_block = 0;
  if (p->filetype == FT_IMAGE24 || p->filetype == FT_IMAGE32) { img_w = p->info; img_alpha = p->filetype == FT_IMAGE32; by_block = 1; }
  else if (p->filetype == FT_IMAGE8 || p->filetype == FT_IMAGE8GRAY) { img_w = p->info; img_bpp = 8; img_gray = p->filetype == FT_IMAGE8GRAY; by_block = 1; }
  else if (p->filetype == FT_IMAGE1) { img_w = p->info; img_bpp = 1; by_block = 1; }
  else if (p->filetype == FT_IMAGE4) { img_w = p->info; img_bpp = 4; by_block = 1; }
  else {
    int e;
    if (p->filetype != FT_EXE) {   /* jpegModel :5911-6597: it follows every byte of the stream; a non-zero return makes the step its own */
      P8Emit* const em = p8f_cur;
      int sets[3], ranges[3], kind = 0;
      const int prefix = nx;
      P8XLayout* X = &em->L.xl[P8_MODEL_JPEG - 1];
      p8f_emit_step_model(em, P8_MODEL_JPEG);
      p8f_emit_model(em, P8_MODEL_JPEG);
      const int jp = p8f_jpeg_step(p->jpeg, y, bpos, p->buf, p->bmask, p->pos, in + nx, sets, ranges, &kind);
      p8f_emit_model(em, 0);
      if (jp) {
        if (em->chunk && !em->chunk->xops) return P8F_ERR_IMAGE_LATE;
        p8f_media_step_notice(em);
        nx += kind == 3 ? 70 : kind ? 1 : 0;
        if (em->xdiscovering) {
          for (int i = 0; i < prefix; i++) X->map[i] = (int16_t)i;
          X->prefix_nx = prefix; if (nx > X->nx) X->nx = nx;
          if (kind == 1 || kind == 2) X->map[prefix] = (int16_t)prefix;
          if (kind == 3) {   /* the coded step's exports in call order (:6572-6591): m1.add(128); per context m.add, m1.add, m.add; m1.p()'s three; six more */
            int k = 0;
            for (int i = 0; i < prefix; i++) X->exp[k++] = (int16_t)i;
            X->exp[k++] = (int16_t)(prefix + 70);
            for (int i = 0; i < 32; i++) { X->exp[k++] = (int16_t)(prefix + 2 * i); X->exp[k++] = (int16_t)(prefix + 2 * i + 1); X->exp[k++] = (int16_t)(prefix + 2 * i + 1); }
            for (int i = 0; i < 3; i++) X->exp[k++] = (int16_t)(prefix + 71 + i);
            for (int i = 0; i < 6; i++) X->exp[k++] = (int16_t)(prefix + 64 + i);
            X->exp_n = k;
          }
        } else if (X->prefix_nx != prefix) return P8F_ERR_INTERNAL;
        p->nx = nx; p->model = P8_MODEL_JPEG; p->nsel = kind ? 3 : 0;
        p->own_silent = kind == 3 ? 2 : 0;
        p->jpeg_const = kind == 1 ? 128 : kind == 2 ? 4095 : 0;
        int base = 0;
        for (int i = 0; i < p->nsel; i++) { sel[ns++] = base + sets[i]; base += ranges[i]; }
        for (; ns < P8_NSEL; ns++) sel[ns] = -1;
        return 0;
      }
      p8f_emit_step_model(em, 0);
    }
    if (p->size > 0) {   /* imgModel :5386-5504 */
      if (bpos == 0 && (e = img_detect(p, &p->stat_record)) != 0) return e;
      if (p->pos > p->img_eoi) p->img_w = 0;
      img_w = p->img_w; img_alpha = p->img_alpha; img_bpp = p->img_bpp; img_gray = p->img_gray;
    }
    if (!img_w && bpos == 0 && (e = wav_detect(p)) != 0) return e;
  }
  /* audioModel :5810-5865 (reached only when no image model took the step) */
  int aud_info = 0;
  if (!img_w) {
    if (p->pos > (int)p->wav_eoi) p->wav_info = 0;
    aud_info = (int)p->wav_info;
  }
  if (img_w || aud_info) {   /* im24bitModel :5001-5353 / im8bitModel :4743-4999 / audio8bModel :5552-5657 / wavModel :5659-5804 through the model's own tables and weight sets */
    P8Emit* const em = p8f_cur;
    if (em->chunk && !em->chunk->xops) return P8F_ERR_IMAGE_LATE;
    p8f_media_step_notice(em);
    int sets[16], ranges[16];
    const int prefix = nx;
    const int model = img_w ? (img_bpp == 1 ? P8_MODEL_IM1 : img_bpp == 4 ? P8_MODEL_IM4 : img_bpp == 8 ? P8_MODEL_IM8 : P8_MODEL_IM24) : (((aud_info - 1) & 2) == 0 ? P8_MODEL_AUDIO8 : P8_MODEL_WAV16);
    int nsel = model == P8_MODEL_IM24 ? 13 : model == P8_MODEL_IM8 ? 8 : model == P8_MODEL_IM1 ? 4 : model == P8_MODEL_IM4 ? 6 : 5;
    P8XLayout* X = &em->L.xl[model - 1];
    p8f_emit_step_model(em, model);
    if (em->xdiscovering) for (int i = 0; i < prefix; i++) X->map[i] = (int16_t)i;
    p8f_emit_model(em, model);
    int n, own_active = 1;
    if (model == P8_MODEL_IM4) n = p8f_im4_step(p->im4, y, bpos, c0, p->c4, p->buf, p->bmask, p->pos, img_w, in + nx, sets, ranges);
    else if (model == P8_MODEL_IM1) n = p8f_im1_step(p->im1, y, bpos, p->buf, p->bmask, p->pos, img_w, in + nx, sets, ranges);
    else if (model == P8_MODEL_IM8) n = p8f_im8_step(p->im8, y, bpos, c0, p->buf, p->bmask, p->pos, img_w, img_gray, in + nx, sets, ranges, bpos == 0 ? p->img_stats : NULL);
    else if (model == P8_MODEL_IM24) n = p8f_im24_step(p->im24, y, bpos, c0, p->buf, p->bmask, p->pos, img_w, img_alpha, in + nx, sets, ranges, bpos == 0 ? p->img_stats : NULL);
    else if (model == P8_MODEL_AUDIO8) { n = p8f_audio8_step(p->audio8, y, bpos, c0, p->buf, p->bmask, p->pos, aud_info - 1, p->blpos, &p->stat_record, in + nx, sets, ranges); own_active = 0; }
    else n = p8f_wav16_step(p->wav16, y, bpos, c0, p->buf, p->bmask, p->pos, aud_info - 1, &p->stat_record, in + nx, sets, ranges, &own_active);
    p8f_emit_model(em, 0);
    if (n < 0) return P8F_ERR_IMAGE_PADDING;
    nx += n;
    int base = 0;
    for (int i = 0; i < nsel; i++) { sel[ns++] = base + sets[i]; base += ranges[i]; }
    if (aud_info) {   /* recordModel(m, AUDIO, Stats) :5861: the generic maps, at their generic places, read through the model's input map */
      int rec_sets[3];
      uint32_t io[6] = {(uint32_t)p->blpos, 0, FT_AUDIO, p->stat_record, p->match_length, p->match_expected};
      io[1] = (bpos > 0) ? P8_ASCII_GROUP_C0[(1 << bpos) - 2 + (c0 & ((1 << bpos) - 1))] : 0;
      nx += p8f_record_step(p->rec, y, bpos, c0, p->c4, io, p->buf, p->bmask, p->pos, in + nx, rec_sets);
      p->stat_record = io[3];
      for (int i = 0; i < 3; i++) sel[ns++] = base + rec_sets[i];   /* (its three sets come with their cumulative offsets: 1024, 512, 11 * 32 rows) */
      base += 1024 + 512 + 11 * 32;
      nsel += 3;
    }
    p->own_silent = X->fam_count > 0 && !own_active;
    if (bpos == 0 && em->chunk && em->chunk->xfam_ctx) {   /* the family slots that are called with a context this byte */
      uint32_t* row = em->chunk->xfam_ctx + em->byte_row * (size_t)P8_XL_MAXS;
      if (X->ngen) { row[P8_XL_MAXS - 2] = own_active ? 0 : (uint32_t)X->fam_count; row[P8_XL_MAXS - 1] = (uint32_t)X->nslots; }
    }
    if (em->xdiscovering) { X->prefix_nx = prefix; if (nx > X->nx) X->nx = nx; }
    else if (X->prefix_nx != prefix || nx > X->nx) { fprintf(stderr, "paq8 front end: model step with %d + %d inputs\n", prefix, nx - prefix); return P8F_ERR_INTERNAL; }
    p->nx = nx; p->model = model; p->nsel = nsel;
    if (img_w && !by_block && img_bpp >= 8) p->type = img_bpp == 8 ? (img_gray ? FT_IMAGE8GRAY : FT_IMAGE8) : (img_alpha ? FT_IMAGE32 : FT_IMAGE24);   /* Stats->Type :5493-5495 */
    for (; ns < P8_NSEL; ns++) sel[ns] = -1;
    if (img_w && !by_block && bpos == 7 && p->pos + 1 == p->img_eoi) { memset(&p->tga, 0, sizeof p->tga); p->bmp.Header = 0; p->img_gray = p->img_alpha = 0; }   /* :5498-5501 */
    if (aud_info && bpos == 7 && p->pos + 1 == (int)p->wav_eoi) memset(&p->wav, 0, sizeof p->wav);   /* :5864 */
    return 0;
  }
  if (!img_w && bpos == 7 && p->pos + 1 == (int)p->wav_eoi) memset(&p->wav, 0, sizeof p->wav);
  uint8_t last[64];
  for (int i = 0; i < 64; i++) last[i] = (uint8_t)RB(i + 1);
  int sm_sets[2], rec_sets[3], text_sets[8], exe_sets[6];
  uint32_t scratch = 0, tstats[6];
  k = 0;
  p8f_sparsematch_step(p->smatch, y, bpos, c0, p->buf, p->bmask, p->pos, in + nx, &k, sm_sets); nx += k;
  const uint32_t g[9] = {p->c4, p->f4, p->x4, p->w4, p->tt, p->gword[2], p->gword[0], p->gword[6], p->gword[7]};
  nx += p8f_sparse_step(p->sparse0, y, bpos, c0, g, ismatch, 0, last, in + nx);
  nx += p8f_sparse_step(p->sparse1, y, bpos, c0, g, ismatch, 0, last, in + nx);
  nx += p8f_ctxmodel_step(p->dist, y, bpos, c0, p->c4, p->f4, p->pos, last, in + nx);
  nx += p8f_small_step(p->pic, y, bpos, c0, p->c4, p->f4, p->w5, p->buf, p->bmask, p->pos, in + nx);
  uint32_t io[6] = {(uint32_t)p->blpos, 0, (uint32_t)p->filetype, p->stat_record, p->match_length, p->match_expected};
  io[1] = (bpos > 0) ? P8_ASCII_GROUP_C0[(1 << bpos) - 2 + (c0 & ((1 << bpos) - 1))] : 0;
  nx += p8f_record_step(p->rec, y, bpos, c0, p->c4, io, p->buf, p->bmask, p->pos, in + nx, rec_sets);
  p->stat_record = io[3];
  nx += p8f_small_step(p->rec1, y, bpos, c0, p->c4, p->f4, p->w5, p->buf, p->bmask, p->pos, in + nx);
  nx += p8f_word_step(p->word, y, bpos, c0, p->c4, p->f4, p->b3, p->blpos, p->buf, p->bmask, p->pos, in + nx, p->gword);
  nx += p8f_ctxmodel_step(p->nest, y, bpos, c0, p->c4, p->f4, p->pos, last, in + nx);
  nx += p8f_ctxmodel_step(p->indirect, y, bpos, c0, p->c4, p->f4, p->pos, last, in + nx);
  nx += p8f_dmc_mix(p->dmc, y, bpos, in + nx);
  nx += p8f_xml_step(p->xml, y, bpos, c0, p->c4, p->buf, p->bmask, p->pos, in + nx, &scratch);
  nx += p8f_text_step(p->text, y, bpos, c0, p->buf, p->bmask, p->pos, in + nx, text_sets, tstats);
  if (bpos == 0) { p->text_first_letter = tstats[4]; p->text_mask = tstats[5]; }
  nx += p8f_exe_step(p->exe, y, bpos, c0, p->c4, p->blpos, p->buf, p->bmask, p->pos, in + nx, exe_sets, &scratch);
  nx += p8f_lpm_step(p->lpm, y, bpos, c0, last, in + nx);
  /* 1552 once a byte boundary has been passed; fewer during the very first byte, when the context maps have no contexts yet */
  if (nx > P8_NX || (p->nbytes > 0 && nx != P8_NX)) { fprintf(stderr, "paq8 front end: %d mixer inputs, expected %d\n", nx, P8_NX); return P8F_ERR_INTERNAL; }
  p->nx = nx;

  /* the 28 weight-set selectors, absolute positions in the 77472-row table, in the order the models call set(); the
   * order-N map's return value ("order") and the last prediction are device state: their terms are added there (p8_rec.h) */
  int base = 0;
  sel[ns++] = base + sm_sets[0]; sel[ns++] = base + sm_sets[1]; base += 4 * 64 + 4 * 2048;
  for (int i = 0; i < 3; i++) sel[ns++] = base + rec_sets[i];
  base += 1024 + 512 + 11 * 32;
  static const int text_range[8] = {2048, 2048, 4096, 4096, 2048, 2048, 4096, 8192};
  for (int i = 0; i < 8; i++) { sel[ns++] = base + text_sets[i]; base += text_range[i]; }
  for (int i = 0; i < 6; i++) sel[ns++] = base + exe_sets[i];
  base += 3 * 1024 + 3 * 8192;
  const uint32_t words = p->gword[2];
  sel[ns++] = base + bpos; base += 64;                                                     /* + (max(order - 3, 0) << 3) */
  const uint32_t d = (uint32_t)c0 << (8 - bpos);
  uint32_t c = (d + (bpos == 1 ? p->b3 / 2 : 0)) & 192;
  if (!bpos) c = (words * 16) & 192;
  const uint32_t c1 = RB(1);
  sel[ns++] = base + (int)((p->w4 & 240) + (p->b2 >> 4)); base += 1536;                    /* + max(order - 5, 0) * 256 */
  sel[ns++] = base + (int)((p->w4 & 3) * 64 + ((words >> 1) & 63)); base += 1536;          /* + max(order - 5, 0) * 256 */
  sel[ns++] = base + (int)((uint32_t)bpos * 256 + c1); base += 2048;
  sel[ns++] = base + (int)((uint32_t)(bpos < 5 ? bpos : 5) * 256 + (p->tt & 63) + c); base += 1536;
  sel[ns++] = base + (int)(((d | (c1 >> bpos)) & 248) + (uint32_t)bpos); base += 1536;     /* + max(order - 5, 0) * 256 */
  sel[ns++] = base + (int)((uint32_t)bpos * 256 + ((((words << bpos) & 255) >> bpos) | (d & 255))); base += 2048;
  sel[ns++] = base; base += 256;                                                           /* + last prediction / 16 */
  sel[ns++] = base + c0; base += 256;
  if (ns != P8_NSEL || base != P8_NROWS) return P8F_ERR_INTERNAL;
  return 0;
}

/* Predictor::update :8248-8362 for step t (bit y = bit t-1 of the stream) */
static int front_step(P8Front* f, int y, int32_t* sel, P8ApmRec* apm) {
  static const uint32_t WRT_mpw[16] = {4, 4, 3, 2, 2, 2, 1, 1, 1, 1, 1, 1, 0, 0, 0, 0}, WRT_mtt[16] = {0, 0, 1, 2, 3, 4, 5, 5, 6, 6, 6, 6, 7, 7, 7, 7};  /* :3868-3869 */
  P8Predictor* p = f->p;
  p->c0 += p->c0 + y;
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
  const int rc = context_model2(p, y, sel);
  if (rc < 0) return rc;
  /* (A step of a model with tables of its own -- audio, a 1- / 4-bit image, JPEG: the ones that leave Stats.Type alone -- INSIDE A TEXT BLOCK ends in the
   * text chain, :8281-8296: the record carries the step's own fields in m[], the chain's ten contexts in c[] as for any text step.) */
  /* the final APM stages (:8281-8358): contexts as far as the host knows them (misses and the probabilities are device state) */
  const int c0 = p->c0, bpos = p->bpos;
  const uint32_t lg = ilog2u(p->match_length + 1);
  const uint32_t c4 = p->c4, mlen = lg < 3 ? lg : 3, eb = p->match_expected;
  memset(apm, 0, sizeof *apm);
  apm->model = (uint8_t)p->model;
  if (p->model) { apm->m[0] = (uint16_t)p->own_silent; apm->m[1] = (uint16_t)p->nx; apm->m[2] = (uint16_t)p->nsel; apm->m[3] = (uint16_t)(p->model == P8_MODEL_JPEG ? p->jpeg_const : 0); }
  if (p->type == FT_TEXT) {
    apm->text = P8_APM_TEXT;
    apm->limit = (uint16_t)(0x3FF >> ((p->blpos < 0xFFF) * 2));
    apm->c[0] = (uint16_t)((c0 << 8) | (int)(p->text_mask & 0xF));
    for (uint64_t m = 0; m < 4; ++m) apm->c[1 + m] = (uint16_t)p8f_finalize64(p8f_hash4((uint64_t)bpos, m, c4 & 0xffff, p->text_mask >> 4), 16);
    apm->c[5] = (uint16_t)p8f_finalize64(p8f_hash3((uint64_t)c0, eb, mlen), 16);
    apm->c[6] = (uint16_t)p8f_finalize64(p8f_hash3((uint64_t)c0, c4 & 0xffff, p->text_first_letter), 16);
    apm->c[7] = (uint16_t)p8f_finalize64(p8f_hash3(eb, mlen, c4 & 0xff), 16);
    apm->c[8] = (uint16_t)p8f_finalize64(p8f_hash2((uint64_t)c0, c4 & 0x00ffffff), 16);
    apm->c[9] = (uint16_t)p8f_finalize64(p8f_hash2((uint64_t)c0, c4 & 0xffffff00), 16);
  } else if (p->type == FT_IMAGE24 || p->type == FT_IMAGE32) {   /* Image.Color :8299-8314 */
    const uint32_t* st = p->img_stats;   /* W, N, NN, WW, Wp1, Np1, plane, ctx */
    apm->text = P8_APM_COLOR;
    apm->limit = (uint16_t)(0x3FF >> ((p->blpos < 0xFFF) * 4));
    apm->c[0] = (uint16_t)(c0 << 4);
    apm->c[1] = (uint16_t)p8f_finalize64(p8f_hash3((uint64_t)c0, st[0], st[3]), 16);
    apm->c[2] = (uint16_t)p8f_finalize64(p8f_hash3((uint64_t)c0, st[1], st[2]), 16);
    apm->c[3] = (uint16_t)((c0 << 8) | (int)(st[7] & 0xff));
    apm->c[4] = (uint16_t)p8f_finalize64(p8f_hash4((uint64_t)c0, st[0], (uint64_t)(uint32_t)((c4 & 0xff) - st[4]), st[6]), 16);   /* (c4 & 0xff) - Wp1 in 32-bit unsigned arithmetic */
    apm->c[5] = (uint16_t)p8f_finalize64(p8f_hash4((uint64_t)c0, st[1], (uint64_t)(uint32_t)((c4 & 0xff) - st[5]), st[6]), 16);
  } else if (p->type == FT_IMAGE8GRAY) {   /* Image.Gray :8315-8324 */
    const uint32_t* st = p->img_stats;
    apm->text = P8_APM_GRAY;
    apm->limit = (uint16_t)(0x3FF >> ((p->blpos < 0xFFF) * 4));
    apm->c[0] = (uint16_t)(c0 << 4);
    apm->c[1] = (uint16_t)((c0 << 8) | (int)(st[7] & 0xff));
    apm->c[2] = (uint16_t)((uint32_t)bpos | (st[7] & 0xF8) | (eb << 8));
  } else if (p->type == FT_IMAGE8) {   /* Image.Palette :8325-8340 */
    const uint32_t* st = p->img_stats;   /* W, N, NN, WW */
    apm->text = P8_APM_PALETTE;
    apm->limit = (uint16_t)(0x3FF >> ((p->blpos < 0xFFF) * 4));
    apm->c[0] = (uint16_t)(c0 << 4);
    apm->c[1] = (uint16_t)p8f_finalize64(p8f_hash3((uint64_t)c0, st[0], st[1]), 16);
    apm->c[2] = (uint16_t)p8f_finalize64(p8f_hash3((uint64_t)c0, st[1], st[2]), 16);
    apm->c[3] = (uint16_t)p8f_finalize64(p8f_hash3((uint64_t)c0, st[0], st[3]), 16);
    apm->c[4] = (uint16_t)p8f_finalize64(p8f_hash3((uint64_t)c0, eb, st[1]), 16);
    apm->c[5] = (uint16_t)p8f_finalize64(p8f_hash3((uint64_t)c0, st[0], st[1]), 16);
  } else {   /* the 1 / 4-bit image types never get here (refused above) */
    apm->c[0] = (uint16_t)((mlen << 11) | ((uint32_t)c0 << 3));
    apm->c[1] = (uint16_t)((uint32_t)c0 | RB(1) << 8);
    apm->c[2] = (uint16_t)((uint32_t)c0 ^ p8f_finalize64(hash1(c4 & 0xffff), 16));
    apm->c[3] = (uint16_t)((uint32_t)c0 ^ p8f_finalize64(hash1(c4 & 0xffffff), 16));
    apm->c[4] = (uint16_t)((eb << 8) | RB(1));
  }
  return 0;
}

static void bind(P8Front* f) { p8f_cur = &f->emit; g_blocks = &f->blocks; }
static void release_models(P8Front* f) {
  for (Blk* b = f->blocks; b;) { Blk* n = b->next; free(b); b = n; }
  f->blocks = NULL; f->p = NULL;
}

P8Front* p8f_front_new(int level) {
  P8Front* f = (P8Front*)calloc(1, sizeof *f);
  if (!f) return NULL;
  f->level = level;
  bind(f);
  /* layout pass: two bytes through a throw-away set of models -- which tables exist, in which order contextModel2 walks
   * them, where their inputs sit in the 1552-vector, during the first byte and after it (data-independent) */
  f->emit.discovering = 1; f->emit.xdiscovering = 1;   /* (the image models' maps register themselves when they are constructed) */
  f->emit.L.order_slot = -1; f->emit.L.dmc_off = -1;
  f->p = predictor_new(level);
  int32_t sel[P8_NSEL];
  P8ApmRec apm;
  int nx_first = 0, rc = 0;
  for (int t = 1; t < 24 && rc == 0; ++t) {
    p8f_emit_begin_step(&f->emit, f->p->in, NULL, 0, 0, t >= 8);
    rc = front_step(f, 0, sel, &apm);
    if (t == 7) nx_first = f->p->nx;
  }
  if (rc == 0) rc = p8f_emit_finish_discovery(&f->emit, nx_first, f->p->nx);
  if (rc == 0 && f->emit.L.order_slot < 0) rc = 1;
  release_models(f);
  if (rc != 0 || f->emit.err) { fprintf(stderr, "paq8 front end: layout pass failed\n"); free(f); return NULL; }
  f->emit.discovering = 0;
  /* layout pass of the image models: a second throw-away set of models through an IMAGE24 block of two rows (header: type, size,
   * width -- preprocessor.cpp:303-304) -- where their inputs start (the common prefix), how many there are, which maps exist */
  {
    static const uint8_t img[] = {FT_IMAGE24, 0, 0, 0, 24, 0, 0, 0, 12,   1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12,   1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12,
                                  FT_IMAGE8, 0, 0, 0, 8, 0, 0, 0, 4,   1, 2, 3, 4, 5, 6, 7, 8,
                                  FT_IMAGE8GRAY, 0, 0, 0, 8, 0, 0, 0, 4,   1, 2, 3, 4, 5, 6, 7, 8,
                                  FT_IMAGE1, 0, 0, 0, 4, 0, 0, 0, 2,   0x55, 0x0F, 0x33, 0xF0,
                                  FT_IMAGE4, 0, 0, 0, 4, 0, 0, 0, 2,   0x12, 0x34, 0x56, 0x78,
                                  /* a DEFAULT block with two RIFF / WAVE files: 8-bit stereo (6 samples), 16-bit stereo (4 samples) */
                                  FT_DEFAULT, 0, 0, 0, 44 + 12 + 44 + 16,
                                  'R', 'I', 'F', 'F', 48, 0, 0, 0, 'W', 'A', 'V', 'E', 'f', 'm', 't', ' ', 16, 0, 0, 0, 1, 0, 2, 0, 0x44, 0xAC, 0, 0, 0x88, 0x58, 1, 0, 2, 0, 8, 0,
                                  'd', 'a', 't', 'a', 12, 0, 0, 0,   128, 127, 129, 126, 130, 125, 131, 124, 132, 123, 133, 122,
                                  'R', 'I', 'F', 'F', 52, 0, 0, 0, 'W', 'A', 'V', 'E', 'f', 'm', 't', ' ', 16, 0, 0, 0, 1, 0, 2, 0, 0x44, 0xAC, 0, 0, 0x10, 0xB1, 2, 0, 4, 0, 16, 0,
                                  'd', 'a', 't', 'a', 16, 0, 0, 0,   1, 0, 2, 0, 3, 0, 4, 0, 5, 0, 6, 0, 7, 0, 8, 0,
                                  /* a DEFAULT block with a baseline JPEG: SOI, DQT, SOF0 (8 x 8, one component), SOS, six bytes of coded data (no DHT: the standard tables), EOI */
                                  FT_DEFAULT, 0, 0, 0, 102,
                                  255, 216, 255, 219, 0, 67, 0, 16, 11, 12, 14, 12, 10, 16, 14, 13, 14, 18, 17, 16, 19, 24, 40, 26, 24, 22, 22, 24, 49, 35, 37, 29, 40, 58, 51, 61, 60, 57, 51, 56, 55, 64, 72, 92, 78, 64, 68, 87, 69, 55, 56, 80, 109, 81, 87, 95, 98, 103, 104, 103, 62, 77, 113, 121, 112, 100, 120, 92, 101, 103, 99, 255, 192, 0, 11, 8, 0, 8, 0, 8, 1, 1, 17, 0, 255, 218, 0, 8, 1, 1, 0, 0, 63, 0, 165, 97, 160, 244, 249, 43, 255, 217, 0};
    f->emit.xdiscovering = 1; f->emit.lane_objs = 0; memset(f->emit.xlane_objs, 0, sizeof f->emit.xlane_objs);
    f->p = predictor_new(level);
    int bit = 0;
    for (size_t t = 1; t < 8 * sizeof img && rc == 0; ++t) {
      p8f_emit_begin_step(&f->emit, f->p->in, NULL, 0, 0, t >= 8);
      rc = front_step(f, bit, sel, &apm);
      bit = (img[t >> 3] >> (7 - (t & 7))) & 1;
    }
    for (int m = 0; m < P8_NMODEL - 1 && rc == 0; m++) if (f->emit.L.xl[m].nx == 0) rc = 2 + m;   /* every model must have been met */
    /* the common prefix's maps run in every model's steps */
    for (int l = 0; l < f->emit.L.nlanes; l++) if (f->emit.L.lane[l].off < f->emit.L.xl[0].prefix_nx) f->emit.L.lane[l].modes = ~0u;
    for (int m = 0; m < P8_NMODEL - 1; m++) if (f->emit.L.xl[m].nslots < f->emit.L.xl[m].fam_count) f->emit.L.xl[m].nslots = f->emit.L.xl[m].fam_count;
    release_models(f);
    f->emit.xdiscovering = 0;
    if (rc != 0 || f->emit.err) { fprintf(stderr, "paq8 front end: layout pass of the image models failed (%d)\n", rc); free(f); return NULL; }
  }
  f->emit.lane_objs = 0; memset(f->emit.xlane_objs, 0, sizeof f->emit.xlane_objs);
  f->p = predictor_new(level);
  return f;
}
void p8f_front_free(P8Front* f) {
  if (!f) return;
  release_models(f);
  free(f);
}
const P8Layout* p8f_front_layout(const P8Front* f) { return &f->emit.L; }

/* One step: the records of the stream's next step (Predictor::update after the bit handed in last) into row `step_row` of `out`
 * (its byte-level arrays: row step_row >> 3). A compressor's chunk is a loop of these over known bits (p8f_front_run); a decoder
 * calls p8f_front_emit_step() for step t + 1 as soon as it has decoded bit t and told the front end (p8f_front_set_bit). */
int p8f_front_emit_step(P8Front* f, P8Chunk* out, size_t step_row) {
  if (f->err) return f->err;
  bind(f);
  p8f_emit_begin_step(&f->emit, f->p->in, out, step_row >> 3, step_row, f->steps >= 8);
  if (f->steps == 0) {   /* no step 0: the first prediction is the constructor's */
    memset(out->sel + step_row * P8_NSEL, 0, P8_NSEL * sizeof(int32_t));
    memset(&out->apm[step_row], 0, sizeof(P8ApmRec));
    if (out->model) out->model[0] = 0;
  } else {
    const int rc = front_step(f, f->last_bit, out->sel + step_row * P8_NSEL, &out->apm[step_row]);
    if (rc < 0 || f->emit.err) { f->err = rc < 0 ? rc : P8F_ERR_INTERNAL; return f->err; }
    const int model = f->p->model;
    p8f_emit_directs(&f->emit, model ? f->emit.L.xl[model - 1].prefix_nx : P8_NX);
    if (out->model && (step_row & 7) == 0) out->model[step_row >> 3] = (uint8_t)model;   /* a byte's eight steps share their model */
  }
  ++f->steps;
  return 0;
}
void p8f_front_set_bit(P8Front* f, int bit) { f->last_bit = bit ? 1 : 0; }
/* test hook (state injection; the twin of the reference harness's refp8_set_pos): the model's byte position -- the index into its 2^30-byte history ring
 * (paq8.cpp:167-186). Before the first step only. */
void p8f_front_set_pos(P8Front* f, int pos) { f->p->pos = pos; }

int p8f_front_run(P8Front* f, const uint8_t* bytes, size_t nbytes, P8Chunk* out) {
  if (f->err) return f->err;
  for (size_t i = 0; i < 8 * nbytes; ++i) {
    const int rc = p8f_front_emit_step(f, out, i);
    if (rc) return rc;
    p8f_front_set_bit(f, (bytes[i >> 3] >> (7 - (i & 7))) & 1);
  }
  return 0;
}

const char* p8f_strerror(int code) {
  switch (code) {
    case 0: return "ok";
    case P8F_ERR_IMAGE_BLOCK: return "paq8 stage: 4-bit image block (im4bitModel is not built; the 1-, 8- and 24 / 32-bit image models are)";
    case P8F_ERR_JPEG: return "paq8 stage: JPEG stream detected (jpegModel is outside the stage's scope)";
    case P8F_ERR_BMP: return "paq8 stage: 4-bit BMP payload detected (im4bitModel is not built; the 1-, 8- and 24 / 32-bit image models are)";
    case P8F_ERR_TGA: return "paq8 stage: TGA payload of an unsupported pixel size";
    case P8F_ERR_WAV: return "paq8 stage: WAV header detected (audioModel is outside the stage's scope)";
    case P8F_ERR_IMAGE_PADDING: return "paq8 stage: image rows whose byte width is not a multiple of the pixel size (the reference indexes past its OLS array there, paq8.cpp:5043,5226: undefined)";
    case P8F_ERR_MODEL_IN_TEXT: return "paq8 stage: (unused since round 5: a model step inside a TEXT block runs the text chain)";
    case P8F_ERR_IMAGE_LATE: return "paq8 stage: an image model in a decoder's chunk (the late-bit form of the image models is not built)";
    default: return "paq8 stage: internal inconsistency in the front end";
  }
}
const uint8_t* p8f_state_table(void) { return P8_STATE; }
const int16_t* p8f_stretch_table(void) { return P8_STRETCH; }
const int16_t* p8f_squash_table(void) { return P8_SQUASH; }
