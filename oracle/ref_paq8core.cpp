// oracle/ref_paq8core.cpp -- TEST INFRASTRUCTURE ONLY (never linked into the product).
//
// C-ABI window onto the numeric building blocks of the UNMODIFIED paq8 model (reference src/models/paq8.cpp, compiled
// here as part of this translation unit, from where it lies): squash/stretch tables, the two-layer int16 Mixer
// (:513-598, dot_product/train :403-432), APM1 (:600-621), StateMap (:623-645), StateMap32 (:645-690), APM (:691-712).
// SURVEY.md 8(a') lists them as what the paq8/fxcm device stages must reproduce; oracle/paq8_core.c restates them and
// tests/test_oracle_paq8core.py pins the restatement against these entry points. No reference logic is re-implemented
// here; only member access is widened. Built as its own shared object (oracle/_ref/libcmixrefpaq8.so).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <unordered_map>
#include <valarray>
#include <vector>
#include <ctype.h>
#include <dirent.h>
#include <errno.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <sys/types.h>
#include <time.h>

#define private public
#define protected public
#include "models/paq8.cpp"
#undef private
#undef protected

char* dictionary_path = NULL;  // referenced by other reference objects' externs; unused here

extern "C" {

void refp8_tables(int16_t* squash4096, int16_t* stretch4096, int32_t* dt1024, uint8_t* state_table_256x4) {
  for (int i = 0; i < 4096; ++i) squash4096[i] = (int16_t)paq8::squash(i - 2048);
  for (int i = 0; i < 4096; ++i) stretch4096[i] = (int16_t)paq8::stretch(i);
  for (int i = 0; i < 1024; ++i) dt1024[i] = 16384 / (i + i + 3);  // what paq8::Predictor() fills (:8243-8245)
  for (int i = 0; i < 256; ++i)
    for (int j = 0; j < 4; ++j) state_table_256x4[4 * i + j] = paq8::State_table[i][j];
}
void refp8_init_dt() {
  for (int i = 0; i < 1024; ++i) paq8::dt[i] = 16384 / (i + i + 3);
}

// ---- Mixer ----
void* refp8_mixer_new(int n, int m, int s, int w) { return new paq8::Mixer(n, m, s, w); }
void refp8_mixer_free(void* h) { delete (paq8::Mixer*)h; }
// One bit: update() with the previous bit, the inputs, the weight-set selectors, p(). exported[] receives what the
// add() calls of both layers hand to cmix through AddPrediction (squash(x) / 4095), *nexp their number.
int refp8_mixer_step(void* h, int y_prev, const int16_t* x, int nx, const int* cx, const int* range, int ncx,
                     float* exported, int* nexp) {
  paq8::Mixer& m = *(paq8::Mixer*)h;
  paq8::y = y_prev;
  paq8::ResetPredictions();
  m.update();
  for (int i = 0; i < nx; ++i) m.add(x[i]);
  for (int i = 0; i < ncx; ++i) m.set(cx[i], range[i]);
  const int p = m.p();
  *nexp = (int)paq8::prediction_index;
  for (int i = 0; i < *nexp; ++i) exported[i] = paq8::model_predictions[i];
  return p;
}

// ---- APM1 / StateMap / StateMap32 / APM: p() = update with the previous bit, then predict ----
void* refp8_apm1_new(int n) { return new paq8::APM1(n); }
int refp8_apm1_p(void* h, int y_prev, int pr, int cx, int rate) { paq8::y = y_prev; return ((paq8::APM1*)h)->p(pr, cx, rate); }
void* refp8_statemap_new() { return new paq8::StateMap(); }
int refp8_statemap_p(void* h, int y_prev, int cx) { paq8::y = y_prev; return ((paq8::StateMap*)h)->p(cx); }
void* refp8_statemap32_new(int n) { return new paq8::StateMap32(n); }
int refp8_statemap32_p(void* h, int y_prev, int cx, int limit) { paq8::y = y_prev; return ((paq8::StateMap32*)h)->p(cx, limit); }
void* refp8_apm_new(int n) { return new paq8::APM(n); }
int refp8_apm_p(void* h, int y_prev, int pr, int cx, int limit) { paq8::y = y_prev; return ((paq8::APM*)h)->p(pr, cx, limit); }

}  // extern "C"
