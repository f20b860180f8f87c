/* oracle/mixnet_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the reference's final mixing network:
 *   MixerInput (stretch)            reference src/mixer/mixer-input.cpp:11-27
 *   Sigmoid (logit LUT / logistic)  reference src/mixer/sigmoid.cpp:5-25
 *   Mixer (select row, dot, update) reference src/mixer/mixer.cpp:16-72
 *   network wiring, aux context     reference src/predictor.cpp:184-357,361-437
 *   SSE                             oracle/sse_oracle.c
 * Evaluation order and rounding follow the reference as built with
 * g++ -std=c++14 -O3 (no FMA contraction, no reassociation): every product is
 * rounded to float, then added to the running float sum in ascending index.
 * This file is compiled with -ffp-contract=off.
 */
#include "cmix_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define LOGIT_SIZE 100001
#define ROW_LIMIT 10000u
#define OVERFLOW_KEY 0xDEADBEEFu
#define MAP_SLOTS 32768u /* > ROW_LIMIT+1, power of two */

static float g_logit[LOGIT_SIZE];
static int g_logit_ready = 0;

/* Sigmoid::Sigmoid + SlowLogit, sigmoid.cpp:5-10,23-25 */
static void init_logit(void) {
  if (g_logit_ready) return;
  for (int i = 0; i < LOGIT_SIZE; ++i) {
    float p = (i + 0.5f) / LOGIT_SIZE;
    g_logit[i] = logf(p / (1 - p));
  }
  g_logit_ready = 1;
}
void orc_logit_table(float* out) {
  init_logit();
  memcpy(out, g_logit, sizeof g_logit);
}

/* Sigmoid::Logit, sigmoid.cpp:12-17 */
static float logit(float p) {
  int index = (int)(p * LOGIT_SIZE);
  if (index >= LOGIT_SIZE) index = LOGIT_SIZE - 1;
  else if (index < 0) index = 0;
  return g_logit[index];
}

/* Sigmoid::Logistic, sigmoid.cpp:19-21 (float overload of exp) */
float orc_logistic(float x) { return 1 / (1 + expf(-x)); }

/* MixerInput::SetInput, mixer-input.cpp:11-15 with eps 1e-4 (predictor.cpp:194) */
float orc_stretch(float p) {
  init_logit();
  const float mn = 1.0e-4f, mx = 1 - 1.0e-4f;
  if (p < mn) p = mn;
  else if (p > mx) p = mx;
  return logit(p);
}

typedef struct {
  uint64_t steps;
  float* w;  /* n_in */
  float* ew; /* n_extra */
} row_t;

typedef struct {
  int n_in, n_extra;
  float lr;
  float p;                 /* Mixer::p_ */
  uint64_t steps, max_steps;
  uint32_t n_rows;         /* context_map_.size() */
  uint32_t keys[MAP_SLOTS];
  row_t* rows[MAP_SLOTS];
  float extra[ORC_N_MIX0]; /* extra_inputs_ captured at Mix() time */
} mixer_t;

struct orc_mixnet {
  mixer_t* mx[ORC_N_MIX];
  float in0[ORC_N_IN0], in1[ORC_N_IN1], in2[ORC_N_IN2];
  float smin, smax; /* stretched_min_/max_ = Logit(0), Logit(1) */
  orc_sse* sse;
  uint64_t aux_ctx;
};

/* learning rates in construction order, predictor.cpp:199-356 */
static const float LR[ORC_N_MIX] = {
    /* layer 0 */ 0.005f, 0.0005f, 0.005f, 0.0005f, 0.005f, 0.002f, 0.002f, 0.005f, 0.00005f,
    0.0007f, 0.0005f, 0.002f, 0.0005f, 0.001f, 0.001f, 0.001f, 0.005f, 0.001f, 0.001f, 0.005f,
    0.001f, 0.001f, 0.005f, 0.005f, 0.005f, 0.003f,
    /* layer 1 */ 0.005f, 0.0005f, 0.005f, 0.0005f, 0.00001f, 0.005f, 0.005f, 0.005f, 0.0005f,
    0.002f, 0.001f, 0.001f, 0.001f, 0.001f, 0.001f, 0.001f, 0.001f, 0.001f, 0.001f, 0.001f,
    /* layer 2 */ 0.0003f};

static row_t* new_row(int n_in, int n_extra) {
  row_t* r = (row_t*)calloc(1, sizeof *r);
  r->w = (float*)calloc((size_t)n_in, sizeof(float));
  r->ew = (float*)calloc((size_t)(n_extra ? n_extra : 1), sizeof(float));
  return r;
}

static row_t** find_slot(mixer_t* m, uint32_t key) {
  uint32_t h = (key * 2654435761u) >> 17; /* 15 bits */
  for (;;) {
    if (!m->rows[h] || m->keys[h] == key) {
      m->keys[h] = key;
      return &m->rows[h];
    }
    h = (h + 1) & (MAP_SLOTS - 1);
  }
}
static int has_key(mixer_t* m, uint32_t key) {
  uint32_t h = (key * 2654435761u) >> 17;
  for (;;) {
    if (!m->rows[h]) return 0;
    if (m->keys[h] == key) return 1;
    h = (h + 1) & (MAP_SLOTS - 1);
  }
}

/* Mixer::GetContextData, mixer.cpp:16-36. The map key type is unsigned int, so
 * the 64-bit context is truncated to 32 bits. */
static row_t* get_row(mixer_t* m, uint64_t context) {
  uint32_t key = (uint32_t)context;
  if (m->n_rows >= ROW_LIMIT && !has_key(m, key)) key = OVERFLOW_KEY;
  row_t** slot = find_slot(m, key);
  if (!*slot) {
    *slot = new_row(m->n_in, m->n_extra);
    m->n_rows++;
  }
  return *slot;
}

/* Study hook (scripts/spec_chain_study.py): called with the inputs and the selected weight row of every Mix. NULL in the tests. */
void (*orc_mix_probe)(const void* mixer, const float* in, const float* w, int n_in) = 0;

/* Mixer::Mix, mixer.cpp:38-54 */
static float mix(mixer_t* m, const float* in, const float* extra_vec, uint64_t ctx) {
  row_t* r = get_row(m, ctx);
  if (orc_mix_probe) orc_mix_probe(m, in, r->w, m->n_in);
  float p = 0;
  for (int i = 0; i < m->n_in; ++i) p += in[i] * r->w[i];
  m->p = p;
  for (int i = 0; i < m->n_extra; ++i) m->extra[i] = extra_vec[i];
  float e = 0;
  for (int i = 0; i < m->n_extra; ++i) e += m->extra[i] * r->ew[i];
  m->p += e;
  return m->p;
}

/* Mixer::Perceive, mixer.cpp:56-72 */
static void perceive(mixer_t* m, const float* in, uint64_t ctx, int bit) {
  row_t* r = get_row(m, ctx);
  float decay = (float)(0.9 / pow(0.0000001 * m->steps + 0.8, 0.8));
  decay = (float)(decay * (1.5 - ((1.0 * r->steps) / m->max_steps)));
  float update = decay * m->lr * (orc_logistic(m->p) - bit);
  ++m->steps;
  ++r->steps;
  if (r->steps > m->max_steps) m->max_steps = r->steps;
  for (int i = 0; i < m->n_in; ++i) r->w[i] = r->w[i] - update * in[i];
  for (int i = 0; i < m->n_extra; ++i) r->ew[i] = r->ew[i] - update * m->extra[i];
  if ((r->steps & 1023) == 0) {
    const float c = 1.0f - 3.0e-6f;
    for (int i = 0; i < m->n_in; ++i) r->w[i] *= c;
    for (int i = 0; i < m->n_extra; ++i) r->ew[i] *= c;
  }
}

orc_mixnet* orc_mixnet_create(void) {
  init_logit();
  orc_mixnet* n = (orc_mixnet*)calloc(1, sizeof *n);
  for (int k = 0; k < ORC_N_MIX; ++k) {
    mixer_t* m = (mixer_t*)calloc(1, sizeof *m);
    if (k < ORC_N_MIX0) { m->n_in = ORC_N_IN0; m->n_extra = k; }
    else if (k < ORC_N_MIX0 + ORC_N_MIX1) { m->n_in = ORC_N_IN1; m->n_extra = k - ORC_N_MIX0; }
    else { m->n_in = ORC_N_IN2; m->n_extra = 0; }
    m->lr = LR[k];
    m->p = 0.5f;
    m->max_steps = 1;
    n->mx[k] = m;
  }
  /* MixerInput ctor, mixer-input.cpp:3-5: inputs_ start at 0.5 */
  for (int i = 0; i < ORC_N_IN0; ++i) n->in0[i] = 0.5f;
  for (int i = 0; i < ORC_N_IN1; ++i) n->in1[i] = 0.5f;
  for (int i = 0; i < ORC_N_IN2; ++i) n->in2[i] = 0.5f;
  n->smin = logit(0);
  n->smax = logit(1);
  n->sse = orc_sse_create();
  return n;
}

void orc_mixnet_destroy(orc_mixnet* n) {
  if (!n) return;
  for (int k = 0; k < ORC_N_MIX; ++k) {
    mixer_t* m = n->mx[k];
    for (unsigned s = 0; s < MAP_SLOTS; ++s)
      if (m->rows[s]) { free(m->rows[s]->w); free(m->rows[s]->ew); free(m->rows[s]); }
    free(m);
  }
  orc_sse_destroy(n->sse);
  free(n);
}

uint64_t orc_mixnet_aux_context(const orc_mixnet* n) { return n->aux_ctx; }

static float clamp_s(const orc_mixnet* n, float p) { /* mixer-input.cpp:17-27 */
  if (p > n->smax) p = n->smax;
  else if (p < n->smin) p = n->smin;
  return p;
}

float orc_mixnet_step(orc_mixnet* n, const float* probs, const uint64_t* sel_in, int bit,
                      float* mix_out) {
  static const int AUX[3] = {433, 2024, 2077};
  uint64_t sel[ORC_N_MIX];
  memcpy(sel, sel_in, sizeof sel);

  /* Predict(): predictor.cpp:362-387 */
  for (int i = 0; i < ORC_N_IN0; ++i) n->in0[i] = orc_stretch(probs[i]);
  float override = -1;
  { float p = probs[ORC_N_IN0 - 1]; if (p == 0 || p == 1) override = p; }
  /* predictor.cpp:388-393 */
  float avg = 0;
  for (int i = 0; i < 3; ++i) avg += orc_logistic(n->in0[AUX[i]]);
  avg /= 3;
  n->aux_ctx = (uint64_t)(avg * 15);
  sel[ORC_AUX_MIXER] = n->aux_ctx;

  /* predictor.cpp:395-412 */
  float extra0[ORC_N_MIX0], extra1[ORC_N_MIX1];
  for (int i = 0; i < ORC_N_MIX0; ++i) {
    float p = mix(n->mx[i], n->in0, extra0, sel[i]);
    extra0[i] = clamp_s(n, p);
    n->in1[i] = extra0[i];
    n->in2[i] = extra0[i];
  }
  for (int i = 0; i < 3; ++i) {
    float p = clamp_s(n, n->in0[AUX[i]]);
    n->in1[ORC_N_MIX0 + i] = p;
    n->in2[ORC_N_MIX0 + ORC_N_MIX1 + i] = p;
  }
  for (int i = 0; i < ORC_N_MIX1; ++i) {
    float p = mix(n->mx[ORC_N_MIX0 + i], n->in1, extra1, sel[ORC_N_MIX0 + i]);
    extra1[i] = clamp_s(n, p);
    n->in2[ORC_N_MIX0 + i] = extra1[i];
  }
  /* predictor.cpp:413-418 */
  float p = orc_logistic(mix(n->mx[ORC_N_MIX - 1], n->in2, NULL, sel[ORC_N_MIX - 1]));
  p = orc_sse_predict(n->sse, p);
  if (override >= 0) p = override;
  if (mix_out) for (int k = 0; k < ORC_N_MIX; ++k) mix_out[k] = n->mx[k]->p;

  /* Perceive(): predictor.cpp:432-437 */
  for (int k = 0; k < ORC_N_MIX; ++k) {
    const float* in = k < ORC_N_MIX0 ? n->in0 : k < ORC_N_MIX0 + ORC_N_MIX1 ? n->in1 : n->in2;
    perceive(n->mx[k], in, sel[k], bit);
  }
  orc_sse_perceive(n->sse, bit);
  return p;
}

/* state injection: the twin of ref_debug_set_mixer_steps (oracle/ref_harness.cpp) */
void orc_mixnet_set_steps(orc_mixnet* n, uint64_t steps) {
  for (int k = 0; k < ORC_N_MIX; ++k) n->mx[k]->steps = steps;
}
