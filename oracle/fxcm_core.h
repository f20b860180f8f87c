/* oracle/fxcm_core.h -- TEST INFRASTRUCTURE ONLY. Shared types of the fxcm restatement (oracle/fxcm_*.c); each block
 * cites the reference lines it follows in oracle/fxcm_core.c. */
#ifndef ORACLE_FXCM_CORE_H
#define ORACLE_FXCM_CORE_H
#include <stdint.h>

typedef struct {            /* Inputs<S> + the exported probabilities (reference src/models/fxcmv1.cpp:191-202, :93-106) */
  int16_t n[640]; int ncount;
  float exported[640]; int pidx;
} FxSink;
void fx_add(FxSink* s, int p);
void fx_unexport(FxSink* s);

int fx_squash(int d);
int fx_stretch(int p);
int fx_ilog(int x);
int fx_clp(int z);
int fx_sc(int p);
int fx_dt(int i);
int fx_pre1(int state);
const uint8_t* fx_sta(int which);   /* 0..5 = STA1, STA2, STA4, STA5, STA6, STA7 */

typedef struct { int N, M, cxt, pr, shift1, elim, uperr, err; int16_t *tx, *wx; } FxMixer;
FxMixer* fx_mixer_new(int n, int m, int shift, int elim, int uperr);
void fx_mixer_update(FxMixer* x, int y);
int fx_mixer_p(FxMixer* x);
int fx_mixer_p1(FxMixer* x);

typedef struct { int N, cxt, pr; uint32_t* t; } FxStateMap;
void fx_statemap_init(FxStateMap* s, int n, const uint8_t* nn);
int fx_statemap_set(FxStateMap* s, int y, int c);

typedef struct { int N, cxt, pr, mask, limit; uint32_t* t; } FxStateMap1;
void fx_statemap1_init(FxStateMap1* s, int n, int limit);
int fx_statemap1_set(FxStateMap1* s, int y, int c);

typedef struct { int index; uint16_t* t; } FxApm;
FxApm* fx_apm_new(int contexts);
int fx_apm_p(FxApm* a, int pr, int cxt, int rate, int y);

typedef struct { uint8_t* t; uint32_t n, cp; int16_t rc[512]; } FxRcm;
void fx_rcm_init(FxRcm* r, int m, int rcm_ml);
void fx_rcm_set(FxRcm* r, uint32_t cx, int c1);
int fx_rcm_p(const FxRcm* r, int bpos, int c0);
int fx_rcm_mix(FxRcm* r, FxSink* s, int bpos, int c0);

typedef struct { uint16_t* Data; int Context, Mask, Stride, bCount, bTotal, B, N, cp; } FxSscm;
void fx_sscm_init(FxSscm* m, int bits_of_context, int input_bits);
void fx_sscm_set(FxSscm* m, uint32_t ctx);
void fx_sscm_mix(FxSscm* m, FxSink* s, int y, int r);

typedef struct { FxStateMap* sm; int* cxt; uint32_t mask; uint8_t* CxtState; int index, count; const uint8_t* nn; } FxDsm;
void fx_dsm_init(FxDsm* d, int m, int c, const uint8_t* nn);
void fx_dsm_set(FxDsm* d, FxSink* s, uint32_t cx, int y);
#endif
