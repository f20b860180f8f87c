/* p8_rec.h -- records that connect the paq8 stage's HOST front end (cmix_amd/csrc/p8front: everything
 * contextModel2 and its sub-models compute from the byte stream alone, reference src/models/paq8.cpp:8101-8207) with
 * its DEVICE kernels (p8stage.hip: every table that learns, the 1552-input mixer, the APM chains). Plain C, included
 * from both sides.
 *
 * Step t of a stream is PAQ8::Perceive(bit t-1) = Predictor::update (:8248-8362); its 1591 exported values are what
 * PAQ8::Predict() returns before bit t is coded (layer-0 columns 434..2024 of the cmix predictor). Step 0 does not
 * exist (the first prediction is the constructor's 0.5 in every slot); steps 1..7 run with no byte context yet, so the
 * ContextMaps add nothing and the mixer's input vector is shorter (P8Layout.first_map). */
#ifndef CMX_P8_REC_H
#define CMX_P8_REC_H
#include <stdint.h>

enum {
  P8_NX = 1552,            /* mixer inputs (:8109) */
  P8_NSEL = 28,            /* weight sets selected per bit */
  P8_NROWS = 77472,        /* total selector range */
  P8_NOUT = P8_NX + P8_NSEL + 11,   /* 1591 exported values */
  P8_FAM_MAXI = 24,        /* ContextMap instances of the family */
  P8_FAM_MAXS = 256,       /* contexts of the family */
  P8_NCM2 = 3,             /* ContextMap2 instances: contextModel2's order-N map, TextModel's, exeModel's */
  P8_NLANE = 64,           /* small learners + host-computed inputs, one lane each */
  P8_ORDER_MAX = 10,       /* ContextMap2::mix's return value for the 10-context order-N map */
  /* the image models (im24bitModel :5001-5353, im8bitModel :4743-4999): a model id per byte, 0 = everything above */
  P8_NMODEL = 8,           /* 0 generic (text / default / exe ...), 1 im24 (IMAGE24 / IMAGE32 blocks, 24 / 32-bit BMP / TGA payloads), 2 im8, 3 audio8, 4 wav16, 5 im1, 6 im4, 7 jpeg */
  P8_XL_NLANE = 192,       /* small maps of one image model (im24: 100 StationaryMap + 59 SmallStationaryContextMap) */
  P8_XL_MAXS = 64,         /* slots of a model's ContextMap family (im24: 47, im8: 52, wav16: 11 + recordModel's 25); the row's last two cells: active range */
  P8_XL_MAXG = 8           /* generic ContextMap instances a model also calls (the audio models: recordModel's four) */
};
enum { P8_MODEL_GENERIC = 0, P8_MODEL_IM24 = 1, P8_MODEL_IM8 = 2, P8_MODEL_AUDIO8 = 3, P8_MODEL_WAV16 = 4, P8_MODEL_IM1 = 5, P8_MODEL_IM4 = 6, P8_MODEL_JPEG = 7 };

/* small lanes (one wavefront): kinds */
enum {
  P8L_NONE = 0,
  P8L_SSCM,      /* SmallStationaryContextMap :891-933   (u16 cells; a = rate) */
  P8L_STAT,      /* StationaryMap :935-974               (u32 cells; a = limit) */
  P8L_IND,       /* IndirectMap :976-1008                (u8 bit histories + StateMap32(256); a = limit) */
  P8L_SM32,      /* StateMap32 :645-690 read out as (stretch(p) + 1) >> 1 (contextModel2 :8155-8156, MatchModel :3672-3679) */
  P8L_PIC,       /* picModel :3844-3864, im1bitModel :4634-4673: a bit-history byte per context + a u16 StateMap (limit: with P8OP_ZERO on the map whose
                  * range holds cell 0 (a == 0), the extra updates of that cell at the model's first call -- every context still sits on it; 0 = picModel's 2) */
  P8L_DIRECT,    /* an input the host front end computes itself (run maps, match lengths, constants): op = the value */
  P8L_RCM,       /* RunContextMap :857-889 over BH<4> :778-813: the byte that followed a hashed context last time and how often in a row; the op word of a
                  * byte's first step carries P8OP_SET | checksum << 8 | the byte just coded, the next lane's (a P8L_NONE placeholder) the item index; one input */
  P8L_HT16,      /* im4bitModel's 14 contexts on its HashTable<16> (:828-856, :4675-4742): ONE lane walks them in the reference's order -- items are found and
                  * replaced by priority, two contexts may come to share one --; a bit history per context and nibble position, a u16 StateMap each; the 14
                  * hashed contexts of a nibble (checksum << 22 | item index) are the raw op words of the 14 P8L_NONE lanes behind it; 42 inputs */
  P8L_JPG,       /* jpegModel's learning half (:6482-6596): ONE lane -- 32 contexts on a BH<9> table (found / moved to the front in the reference's order), a
                  * StateMap each, the model's own 33-input mixer with three weight sets and its second layer, two APM stages; op word: P8OP_MIX | hbcount |
                  * (huffcode & 1) << 2; behind it 64 raw words (the 32 contexts' checksum / item index, every third bit of a code) and 5 more (the three
                  * weight rows, the two APM contexts); 70 inputs + 4 values that are only exported (the mixer's constant input and its three set outputs) */
  P8L_PIC2       /* two P8L_PIC maps whose context ranges overlap in the reference's shared array (im1bitModel's cxt[6] / cxt[7]): one lane, one array,
                  * the two contexts in the reference's order; the second context's op word is the next lane's (a P8L_NONE placeholder); two inputs */
};
/* per step and lane one op word */
#define P8OP_MIX   0x80000000u   /* the map's mix() / p() is called this step (else: untouched, its inputs are 0) */
#define P8OP_SET   0x40000000u   /* preceded by set(): low 30 bits = cell index of the context (masked, times stride) */
#define P8OP_ORDER 0x20000000u   /* the context is the order-N map's return value of this step (sparseModel1's scm6 :4566) */
#define P8OP_ZERO  0x10000000u   /* StateMap32: update, but the input is 0 (MatchModel with ctx == 0 :3675) */
#define P8OP_CTX   0x0FFFFFFFu

typedef struct {
  uint8_t kind, a, mul, div;   /* kind, rate or limit selector, output scale mul / div (:909-919) */
  uint16_t limit;              /* STAT / IND / SM32: count limit */
  uint16_t bits_per_ctx;       /* dmaps: bits consumed before the context restarts (BitsPerContext) */
  int16_t off;                 /* first input position in the 1552-vector */
  int16_t nout;                /* inputs produced (2 for the maps, 1 for SM32 / PIC / DIRECT) */
  uint32_t cells;              /* table size in cells */
  uint32_t init;               /* cell initial value (SM32 with 256 cells: the state-table prior instead) */
  uint32_t modes;              /* a lane of the generic table: bit m set = it is also called in the steps of model m (the common prefix: every model; recordModel's
                                * maps: the audio models) */
} P8Lane;

/* An image model's part of the layout. Its steps run contextModel2's common prefix (:8133-8160: the constant, two StateMap32, the
 * order-N ContextMap2, three run maps, MatchModel -- the same objects at the same input positions as in the generic layout), then the
 * model's own maps instead of everything else (:8161-8167): inputs [0, prefix_nx) + [prefix_nx, nx), fewer weight sets (sel = -1
 * for the unused ones), their own final APM stages (P8ApmRec.kind). AddPrediction() counts on: the exported values of such a step are
 * its nx inputs, nsel second-layer inputs and the chain's values back to back; the rest of the 1591 keep their last values (:504-510). */
typedef struct {
  int prefix_nx, nx;
  int nlanes;
  P8Lane lane[P8_XL_NLANE];
  uint64_t fam_size;                   /* the model's own ContextMap (fam_count == 0: none) */
  int fam_count;
  int16_t fam_off[P8_XL_MAXS];         /* input position of a family slot's five inputs (own contexts first, then the generic instances') */
  /* generic ContextMap instances the model calls after its own maps (the audio models end with recordModel :5861): they join the model's
   * family at slots gen_first[k] .., sharing tables and per-context state with the generic family (handed over at every switch) */
  int ngen, gen_inst[P8_XL_MAXG], gen_first[P8_XL_MAXG];
  int nslots;                          /* fam_count + the generic instances' contexts */
  /* the step's inputs in add() order -> positions in the 1552-vector: the model's own first (position = order), then the generic maps' at their
   * generic positions. own_opt: inputs [opt_lo, opt_lo + opt_n) are the own ContextMap's, absent in a byte it is silent (P8ApmRec.m[0]) */
  int16_t map[P8_NX];
  int opt_lo, opt_n;
  /* jpegModel's coded steps export more than their inputs, interleaved (the model's own mixer exports too, :504-507): positions of the 1552-vector in
   * export order (P8ApmRec.m[0] == 2 selects it; otherwise a step exports its inputs in order) */
  int exp_n;
  int16_t exp[P8_NX];
} P8XLayout;

typedef struct {
  int fam_ninst, fam_slots;
  uint64_t fam_size[P8_FAM_MAXI];      /* bytes, in the order contextModel2 walks the instances (= rnd() draw order) */
  int fam_count[P8_FAM_MAXI];
  int16_t fam_off[P8_FAM_MAXS];        /* input position of a context's five inputs */
  uint64_t cm2_size[P8_NCM2];
  int cm2_count[P8_NCM2];
  int16_t cm2_off[P8_NCM2];            /* seven inputs per context, contiguous */
  int nlanes;
  P8Lane lane[P8_NLANE];
  int16_t dmc_off;                     /* six inputs */
  int order_slot;                      /* family context whose value is hash(2, order) (sparseModel :4513) */
  uint32_t order_ctx[P8_ORDER_MAX + 1];
  uint16_t order_chk[P8_ORDER_MAX + 1];
  int nx_first;                        /* inputs during the first byte */
  int16_t first_map[P8_NX];            /* compact position -> position in the full vector */
  P8XLayout xl[P8_NMODEL - 1];         /* the image models (xl[model - 1]) */
} P8Layout;

/* selectors whose value depends on device state: the host part is in sel[], the device adds
 *   sel[P8_SEL_ORDER3] += max(order - 3, 0) << 3;  sel[P8_SEL_ORDER5_a/b/c] += max(order - 5, 0) * 256   (:8171-8190)
 *   sel[P8_SEL_LASTPR] += last prediction / 16 (:8193) */
enum { P8_SEL_ORDER3 = 19, P8_SEL_ORDER5_A = 20, P8_SEL_ORDER5_B = 21, P8_SEL_ORDER5_C = 24, P8_SEL_LASTPR = 26 };

/* per step: contexts of the final APM stages (Predictor::update :8281-8358) as far as the host knows them.
 * TEXT block:  c[0] = c0 << 8 | mask & 15 (device ORs (misses & 15) << 4), c[1..4] = second APM's context for
 *              misses & 3 = 0..3, c[5], c[6] = third, fourth; c[7..9] = the three APM1 contexts
 * other:       c[0] = mlen << 11 | c0 << 3 (device ORs misses & 7), c[1..3] = ctx1..3, c[4] = expected byte << 8 | c1 */
/* IMAGE24/32 (kind 2, Image.Color :8299-8314): c[0] = c0 << 4 (device ORs misses & 15), c[1..3] = the other three APMs' contexts,
 *              c[4], c[5] = the two APM1 contexts; c[8] = the step's input count nx, c[9] = its weight-set count, c[7] = 1: the model's own ContextMap is silent this byte
 *              (every image / audio kind)
 * IMAGE8GRAY (kind 3, Image.Gray :8315-8324): c[0] as above, c[1], c[2]; IMAGE8 (kind 4, Image.Palette :8325-8340): c[0..3], c[4], c[5] */
enum { P8_APM_GENERIC = 0, P8_APM_TEXT = 1, P8_APM_COLOR = 2, P8_APM_GRAY = 3, P8_APM_PALETTE = 4 };
/* m[0..3] (round 5): a step of a model with tables of its own -- m[0] own_silent (1: the model's own ContextMap is silent this byte, 2: the export order of
 * P8XLayout), m[1] nx (its inputs), m[2] nsel (its weight sets), m[3] the JPEG model's one constant input. Fields of their own since such a step can end in
 * the TEXT chain, whose ten contexts fill c[0..9] (paq8.cpp:8281-8296: a WAV / JPEG / 1- or 4-bit image the detectors find inside a TEXT block). The generic
 * mixer kernels stage only the first 24 bytes of a record (they never see a model's step). */
typedef struct { uint16_t c[10]; uint16_t limit; uint8_t text /* = kind: P8_APM_* */, model /* P8_MODEL_* of the step */; uint16_t m[4]; } P8ApmRec;

/* one chunk of nbytes input bytes = 8 nbytes steps */
typedef struct {
  uint32_t* fam_ctx; uint16_t* fam_chk;                   /* [nbytes][fam_slots]: set at the step with bpos == 0 */
  uint32_t* cm2_ctx[P8_NCM2]; uint16_t* cm2_chk[P8_NCM2]; /* [nbytes][cm2_count[k]] */
  uint32_t* ops;                                          /* [8 nbytes][P8_NLANE] */
  int32_t* sel;                                           /* [8 nbytes][P8_NSEL] absolute rows (-1: no such set at this step) */
  P8ApmRec* apm;                                          /* [8 nbytes] */
  /* image models (NULL while a stream has met none: the stage allocates them at the first such byte) */
  uint8_t* model;                                         /* [nbytes] P8_MODEL_* */
  uint32_t* xops;                                         /* [8 nbytes][P8_XL_NLANE]: the model's small maps (the table of that byte's model) */
  uint32_t* xfam_ctx; uint16_t* xfam_chk;                 /* [nbytes][P8_XL_MAXS] */
} P8Chunk;

#endif
