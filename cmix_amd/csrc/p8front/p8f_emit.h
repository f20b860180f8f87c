/* p8f_emit.h -- the table interface the paq8 front end's sub-models are written against. On the reference's CPU path
 * these calls ARE the tables (ContextMap::set/mix, ContextMap2::set/mix, StationaryMap::set/mix ..., reference
 * src/models/paq8.cpp:891-1358); here they only RECORD what the tables are asked -- hashed contexts per byte, one op
 * word per small map and step -- into the chunk records of ../p8_rec.h, which the device kernels (../p8stage.hip)
 * consume. The sub-models themselves (p8f_word.c, p8f_text.c, ...) compute everything that depends on the byte stream
 * alone; nothing in this directory holds a learned table (the three RunContextMaps moved to the device in round 4: P8L_RCM). What it does
 * hold is arithmetic on the bytes themselves: the least-squares predictors of the image / audio models (f64, wavModel's in long double),
 * the JPEG parser and Huffman decoder. */
#ifndef CMX_P8F_EMIT_H
#define CMX_P8F_EMIT_H
#include <stddef.h>
#include <stdint.h>

#include "../p8_rec.h"

typedef struct {
  P8Layout L;
  int discovering;        /* layout pass: objects register themselves, offsets are learned */
  int16_t* in_base;       /* the step's mixer-input array (models write host-computed inputs into it) */
  P8Chunk* chunk;         /* where the step's records go (NULL: discard) */
  size_t byte_row, step_row;
  int full;               /* a byte boundary has been passed: every object produces its inputs */
  int fam_calls, cm2_calls, lane_objs;   /* per-step walk counters (calling-order checks) */
  int err;
  /* image models (p8_rec.h P8XLayout): while `model` is non-zero, maps and the ContextMap that are created / called belong to
   * that model's own tables (xops / xfam_* of the chunk) */
  int model;
  int step_model;                        /* the model of the step being emitted (set when the front end enters it, kept to the end of the step): generic
                                          * maps called under it (recordModel in an audio step) run at their generic places, read through the model's map */
  int xdiscovering;                      /* layout pass of the image models (the generic objects are already fixed) */
  int xlane_objs[P8_NMODEL - 1];
  int16_t lane_off0[P8_NLANE];           /* discovery: input positions during the first byte */
  int16_t dmc_off0;
  uint8_t claimed[P8_NX], claimed0[P8_NX];
} P8Emit;

extern __thread P8Emit* p8f_cur;

void p8f_emit_begin_step(P8Emit* e, int16_t* in_base, P8Chunk* chunk, size_t byte_row, size_t step_row, int full);
int p8f_emit_finish_discovery(P8Emit* e, int nx_first, int nx_full);
/* after a step: copy the host-computed inputs into their DIRECT lanes' op words (positions below lim_off only: an image model's
 * step ends the generic layout at its common prefix) */
void p8f_emit_directs(P8Emit* e, int lim_off);
/* the models of image model m (1 ..) are being constructed / stepped from here on (0: back to the generic tables) */
void p8f_emit_model(P8Emit* e, int m);
/* the step belongs to model m from here on (its records: xops row cleared, family row's active range reset) */
void p8f_emit_step_model(P8Emit* e, int m);

#endif
