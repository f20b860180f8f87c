/* p8f_front.h -- C interface of the paq8 stage's host front end (p8f_front.c). One P8Front per stream. */
#ifndef CMX_P8F_FRONT_H
#define CMX_P8F_FRONT_H
#include <stddef.h>
#include <stdint.h>

#include "../p8_rec.h"

#ifdef __cplusplus
extern "C" {
#endif
typedef struct P8Front P8Front;
enum { P8F_ERR_IMAGE_BLOCK = -1, P8F_ERR_JPEG = -2, P8F_ERR_BMP = -3, P8F_ERR_TGA = -4, P8F_ERR_WAV = -5, P8F_ERR_IMAGE_PADDING = -6, P8F_ERR_IMAGE_LATE = -7, P8F_ERR_MODEL_IN_TEXT = -8, P8F_ERR_INTERNAL = -9 };

P8Front* p8f_front_new(int level);                 /* cmix runs paq8 at level 11 (reference src/models/paq8.cpp:8368, paq8.h) */
void p8f_front_free(P8Front* f);
const P8Layout* p8f_front_layout(const P8Front* f);
/* the next nbytes bytes of the stream = the next 8 nbytes steps; fills every array of `out` (sized for nbytes).
 * 0, or a negative P8F_ERR_* (a block type outside the stage's scope, or an internal inconsistency). */
int p8f_front_run(P8Front* f, const uint8_t* bytes, size_t nbytes, P8Chunk* out);
/* the same one step at a time (a decoder): emit the records of the next step into row step_row of `out`, then tell the front end
 * the bit that was coded with them. p8f_front_run() is a loop of the two. */
int p8f_front_emit_step(P8Front* f, P8Chunk* out, size_t step_row);
void p8f_front_set_bit(P8Front* f, int bit);
void p8f_front_set_pos(P8Front* f, int pos);   /* test hook: the model's byte position (before the first step) */
const char* p8f_strerror(int code);
/* data tables the device side is built from (the reference's nex() state table, stretch, squash, ilog) */
const uint8_t* p8f_state_table(void);     /* [1024]  nex(s, k) = t[4 s + k] */
const int16_t* p8f_stretch_table(void);   /* [4096] */
const int16_t* p8f_squash_table(void);    /* [4096]  squash(d), index d + 2048 */
const uint8_t* p8f_ilog_table(void);      /* [65536] */
#ifdef __cplusplus
}
#endif
#endif
