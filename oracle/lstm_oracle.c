/* oracle/lstm_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement of the byte-level LSTM "byte mixer":
 *   ByteMixer::SetInput/ByteUpdate      reference src/mixer/byte-mixer.cpp:15-38
 *   Lstm::SetInput/Perceive/Predict     reference src/mixer/lstm.cpp:80-150
 *   LstmLayer::ForwardPass/BackwardPass reference src/mixer/lstm-layer.cpp:62-197
 *   Adam                                reference src/mixer/lstm-layer.cpp:11-32
 *   ByteModel::Predict/Perceive/ByteUpdate  reference src/models/byte-model.cpp:8-45
 * with libstdc++'s valarray evaluation order: element-wise expressions are rounded
 * per operation, valarray::sum() adds forward from 0, (expression).sum() adds
 * BACKWARD from the last element (bits/valarray_after.h _Expr::sum), SURVEY.md
 * Appendix B.  Compiled with -ffp-contract=off.
 *
 * Weight initialisation consumes glibc rand() after srand(0xDEADBEEF) exactly as the
 * reference Predictor does: 31 Indirect constructors draw one value each before the
 * LSTM (predictor.cpp:90-178, indirect.cpp:10), then lstm-layer.cpp:52-59.
 */
#include "cmix_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define C_ 200
#define L_ 2
#define H_ 100
#define NH (C_ * L_ + 1)

typedef struct {
  int rowlen;
  float *w, *m, *v, *u;               /* [C][rowlen] */
  float gamma[C_], gamma_u[C_], gamma_m[C_], gamma_v[C_];
  float beta[C_], beta_u[C_], beta_m[C_], beta_v[C_];
  float error[C_];
  float ivar[H_];
  float norm[H_][C_], state[H_][C_];
} gate_t;

typedef struct {
  int insz;                            /* layer_input size */
  gate_t g[3];                         /* forget, input node, output */
  float state[C_], state_error[C_], stored_error[C_];
  float tanh_state[H_][C_], input_gate_state[H_][C_], last_state[H_][C_];
  unsigned epoch;
  unsigned long long update_steps;
} layer_t;

struct orc_lstm {
  int V;
  layer_t ly[L_];
  unsigned input_history[H_];
  float hidden[NH], hidden_error[C_];
  float* layer_input[H_][L_];
  float* output_layer[H_];             /* [V][NH] */
  float* output[H_];                   /* [V] */
  unsigned epoch;
  float lr;
  /* ByteMixer / ByteModel part */
  unsigned char vocab[256];
  int byte_map[256];
  float probs[256];
  int top, mid, bot, ex;
};

static float logistic(float x) { return 1 / (1 + expf(-x)); }

static float frand(void) { return (float)rand() / (float)RAND_MAX; } /* lstm-layer.h:36-38 */

orc_lstm* orc_lstm_create(const uint8_t* vocab256, int skip_rand) {
  orc_lstm* s = (orc_lstm*)calloc(1, sizeof *s);
  int V = 0;
  for (int i = 0; i < 256; ++i) {
    s->vocab[i] = vocab256[i] != 0;
    s->byte_map[i] = V;
    if (s->vocab[i]) ++V;
  }
  s->V = V;
  s->lr = 0.03f;
  srand(0xDEADBEEF); /* predictor.cpp:26 */
  for (int i = 0; i < skip_rand; ++i) (void)rand();
  /* Lstm::Lstm, lstm.cpp:8-32 */
  s->hidden[NH - 1] = 1;
  for (int e = 0; e < H_; ++e) {
    for (int l = 0; l < L_; ++l) {
      int n = l == 0 ? 1 + C_ + V : V + 1 + 2 * C_;
      s->layer_input[e][l] = (float*)calloc((size_t)n, sizeof(float));
      s->layer_input[e][l][n - 1] = 1;
    }
    s->output_layer[e] = (float*)calloc((size_t)V * NH, sizeof(float));
    s->output[e] = (float*)malloc((size_t)V * sizeof(float));
    for (int i = 0; i < V; ++i) s->output[e][i] = (float)(1.0 / V);
  }
  /* LstmLayer::LstmLayer, lstm-layer.cpp:36-60 */
  for (int l = 0; l < L_; ++l) {
    layer_t* y = &s->ly[l];
    y->insz = l == 0 ? 1 + C_ + V : V + 1 + 2 * C_;
    int rowlen = y->insz + V;
    for (int k = 0; k < 3; ++k) {
      gate_t* g = &y->g[k];
      g->rowlen = rowlen;
      g->w = (float*)calloc((size_t)C_ * rowlen, sizeof(float));
      g->m = (float*)calloc((size_t)C_ * rowlen, sizeof(float));
      g->v = (float*)calloc((size_t)C_ * rowlen, sizeof(float));
      g->u = (float*)calloc((size_t)C_ * rowlen, sizeof(float));
      for (int i = 0; i < C_; ++i) g->gamma[i] = 1.0f;
    }
    float val = sqrtf(6.0f / (float)(V + V));
    float low = -val, range = 2 * val;
    for (int i = 0; i < C_; ++i) {
      for (int j = 0; j < rowlen; ++j) {
        y->g[0].w[i * rowlen + j] = low + frand() * range;
        y->g[1].w[i * rowlen + j] = low + frand() * range;
        y->g[2].w[i * rowlen + j] = low + frand() * range;
      }
      y->g[0].w[i * rowlen + rowlen - 1] = 1;
    }
  }
  /* ByteModel ctor, byte-model.cpp:5-6 */
  for (int i = 0; i < 256; ++i) s->probs[i] = (float)(1.0 / 256);
  s->top = 255;
  s->mid = 0;
  s->bot = 0;
  return s;
}

void orc_lstm_destroy(orc_lstm* s) {
  if (!s) return;
  for (int e = 0; e < H_; ++e) {
    for (int l = 0; l < L_; ++l) free(s->layer_input[e][l]);
    free(s->output_layer[e]);
    free(s->output[e]);
  }
  for (int l = 0; l < L_; ++l)
    for (int k = 0; k < 3; ++k) { free(s->ly[l].g[k].w); free(s->ly[l].g[k].m); free(s->ly[l].g[k].v); free(s->ly[l].g[k].u); }
  free(s);
}

int orc_lstm_vocab_size(const orc_lstm* s) { return s->V; }
int orc_lstm_gate_rowlen(const orc_lstm* s, int layer) { return s->ly[layer].g[0].rowlen; }
const float* orc_lstm_gate_weights(const orc_lstm* s, int layer, int gate) { return s->ly[layer].g[gate].w; }

/* LstmLayer::ForwardPass(NeuronLayer&...), lstm-layer.cpp:85-99 */
static void gate_forward(orc_lstm* s, layer_t* y, gate_t* g, const float* input, int symbol) {
  const int V = s->V, e = (int)y->epoch;
  for (int i = 0; i < C_; ++i) {
    const float* w = g->w + (size_t)i * g->rowlen;
    float f = w[symbol];
    for (int j = 0; j < y->insz; ++j) f += input[j] * w[V + j];
    g->norm[e][i] = f;
  }
  float sq = g->norm[e][C_ - 1] * g->norm[e][C_ - 1];   /* (norm*norm).sum(): backward */
  for (int i = C_ - 2; i >= 0; --i) sq += g->norm[e][i] * g->norm[e][i];
  g->ivar[e] = 1.0f / sqrtf((sq / C_) + 1e-5f);
  for (int i = 0; i < C_; ++i) g->norm[e][i] *= g->ivar[e];
  for (int i = 0; i < C_; ++i) g->state[e][i] = g->norm[e][i] * g->gamma[i] + g->beta[i];
}

/* LstmLayer::ForwardPass, lstm-layer.cpp:62-83 */
static void layer_forward(orc_lstm* s, layer_t* y, const float* input, int symbol, float* hidden_out) {
  const int e = (int)y->epoch;
  memcpy(y->last_state[e], y->state, sizeof y->state);
  for (int k = 0; k < 3; ++k) gate_forward(s, y, &y->g[k], input, symbol);
  for (int i = 0; i < C_; ++i) {
    y->g[0].state[e][i] = logistic(y->g[0].state[e][i]);
    y->g[1].state[e][i] = tanhf(y->g[1].state[e][i]);
    y->g[2].state[e][i] = logistic(y->g[2].state[e][i]);
  }
  for (int i = 0; i < C_; ++i) y->input_gate_state[e][i] = 1.0f - y->g[0].state[e][i];
  for (int i = 0; i < C_; ++i) y->state[i] *= y->g[0].state[e][i];
  for (int i = 0; i < C_; ++i) y->state[i] += y->g[1].state[e][i] * y->input_gate_state[e][i];
  for (int i = 0; i < C_; ++i) y->tanh_state[e][i] = tanhf(y->state[i]);
  for (int i = 0; i < C_; ++i) hidden_out[i] = y->g[2].state[e][i] * y->tanh_state[e][i];
  if (++y->epoch == H_) y->epoch = 0;
}

static void clip(float* a, int n, float c) { /* lstm-layer.cpp:101-106 */
  for (int i = 0; i < n; ++i) {
    if (a[i] < -c) a[i] = -c;
    else if (a[i] > c) a[i] = c;
  }
}

/* Adam, lstm-layer.cpp:11-32 */
static void adam(float* g, float* m, float* v, float* w, int n, float lr, float t,
                 unsigned long long limit) {
  const float beta1 = 0.025f, beta2 = 0.9999f, eps = 1e-6f;
  float alpha, b1, b2;
  if (t < limit) {
    alpha = lr * 0.1f / sqrtf(5e-5f * t + 1.0f);
    b1 = (float)(1.0f - powf(beta1, t));
    b2 = (float)(1.0f - powf(beta2, t));
  } else {
    alpha = lr * 0.1f / sqrtf(5e-5f * limit + 1.0f);
    b1 = (float)(1.0f - pow(beta1, limit));
    b2 = (float)(1.0f - pow(beta2, limit));
  }
  for (int i = 0; i < n; ++i) m[i] *= beta1;
  for (int i = 0; i < n; ++i) m[i] += (1.0f - beta1) * g[i];
  for (int i = 0; i < n; ++i) v[i] *= beta2;
  for (int i = 0; i < n; ++i) v[i] += (1.0f - beta2) * g[i] * g[i];
  for (int i = 0; i < n; ++i) w[i] -= alpha * ((m[i] / b1) / (sqrtf(v[i] / b2 + eps)));
}

/* LstmLayer::BackwardPass(NeuronLayer&...), lstm-layer.cpp:145-197 */
static void gate_backward(orc_lstm* s, layer_t* y, gate_t* g, const float* input, int epoch, int layer,
                          int symbol, float* hidden_error) {
  const int V = s->V, rl = g->rowlen;
  if (epoch == H_ - 1) {
    memset(g->gamma_u, 0, sizeof g->gamma_u);
    memset(g->beta_u, 0, sizeof g->beta_u);
    memset(g->u, 0, (size_t)C_ * rl * sizeof(float));
    /* transpose_ snapshot == the weights themselves until Adam at epoch 0 */
  }
  for (int i = 0; i < C_; ++i) g->beta_u[i] += g->error[i];
  for (int i = 0; i < C_; ++i) g->gamma_u[i] += g->error[i] * g->norm[epoch][i];
  for (int i = 0; i < C_; ++i) g->error[i] *= g->gamma[i] * g->ivar[epoch];
  float sm = g->error[C_ - 1] * g->norm[epoch][C_ - 1];  /* (error*norm).sum(): backward */
  for (int i = C_ - 2; i >= 0; --i) sm += g->error[i] * g->norm[epoch][i];
  sm = sm / C_;
  for (int i = 0; i < C_; ++i) g->error[i] -= sm * g->norm[epoch][i];
  if (layer > 0) {
    for (int i = 0; i < C_; ++i) {
      float f = 0;
      for (int j = 0; j < C_; ++j) f += g->error[j] * g->w[(size_t)j * rl + 2 * V + C_ + i];
      hidden_error[i] += f;
    }
  }
  if (epoch > 0) {
    for (int i = 0; i < C_; ++i) {
      float f = 0;
      for (int j = 0; j < C_; ++j) f += g->error[j] * g->w[(size_t)j * rl + 2 * V + i];
      y->stored_error[i] += f;
    }
  }
  for (int i = 0; i < C_; ++i) {
    float* u = g->u + (size_t)i * rl;
    for (int j = 0; j < y->insz; ++j) u[V + j] += g->error[i] * input[j];
    u[symbol] += g->error[i];
  }
  if (epoch == 0) {
    for (int i = 0; i < C_; ++i)
      adam(g->u + (size_t)i * rl, g->m + (size_t)i * rl, g->v + (size_t)i * rl, g->w + (size_t)i * rl, rl,
           s->lr, (float)y->update_steps, 3000);
    adam(g->gamma_u, g->gamma_m, g->gamma_v, g->gamma, C_, s->lr, (float)y->update_steps, 3000);
    adam(g->beta_u, g->beta_m, g->beta_v, g->beta, C_, s->lr, (float)y->update_steps, 3000);
  }
}

/* LstmLayer::BackwardPass, lstm-layer.cpp:108-143 */
static void layer_backward(orc_lstm* s, layer_t* y, const float* input, int epoch, int layer, int symbol,
                           float* hidden_error) {
  gate_t *fg = &y->g[0], *in = &y->g[1], *og = &y->g[2];
  if (epoch == H_ - 1) {
    memcpy(y->stored_error, hidden_error, sizeof y->stored_error);
    memset(y->state_error, 0, sizeof y->state_error);
  } else {
    for (int i = 0; i < C_; ++i) y->stored_error[i] += hidden_error[i];
  }
  for (int i = 0; i < C_; ++i)
    og->error[i] = y->tanh_state[epoch][i] * y->stored_error[i] * og->state[epoch][i] * (1.0f - og->state[epoch][i]);
  for (int i = 0; i < C_; ++i)
    y->state_error[i] += y->stored_error[i] * og->state[epoch][i] * (1.0f - (y->tanh_state[epoch][i] * y->tanh_state[epoch][i]));
  for (int i = 0; i < C_; ++i)
    in->error[i] = y->state_error[i] * y->input_gate_state[epoch][i] * (1.0f - (in->state[epoch][i] * in->state[epoch][i]));
  for (int i = 0; i < C_; ++i)
    fg->error[i] = (y->last_state[epoch][i] - in->state[epoch][i]) * y->state_error[i] * fg->state[epoch][i] * y->input_gate_state[epoch][i];
  memset(hidden_error, 0, C_ * sizeof(float));
  if (epoch > 0) {
    for (int i = 0; i < C_; ++i) y->state_error[i] *= fg->state[epoch][i];
    memset(y->stored_error, 0, sizeof y->stored_error);
  } else if (y->update_steps < 3000) {
    ++y->update_steps;
  }
  gate_backward(s, y, fg, input, epoch, layer, symbol, hidden_error);
  gate_backward(s, y, in, input, epoch, layer, symbol, hidden_error);
  gate_backward(s, y, og, input, epoch, layer, symbol, hidden_error);
  clip(y->state_error, C_, 10.0f);
  clip(y->stored_error, C_, 10.0f);
  clip(hidden_error, C_, 10.0f);
}

/* Lstm::Predict, lstm.cpp:120-150 */
static const float* lstm_predict(orc_lstm* s, unsigned symbol) {
  const int V = s->V, e = (int)s->epoch;
  for (int l = 0; l < L_; ++l) {
    memcpy(s->layer_input[e][l] + V, s->hidden + l * C_, C_ * sizeof(float));
    layer_forward(s, &s->ly[l], s->layer_input[e][l], (int)symbol, s->hidden + l * C_);
    if (l < L_ - 1) memcpy(s->layer_input[e][l + 1] + C_ + V, s->hidden + l * C_, C_ * sizeof(float));
  }
  float max_out = 0;
  for (int i = 0; i < V; ++i) {
    float sum = 0;
    const float* ol = s->output_layer[e] + (size_t)i * NH;
    for (int j = 0; j < NH; ++j) sum += s->hidden[j] * ol[j];
    s->output[e][i] = sum;
    if (sum > max_out) max_out = sum;
  }
  for (int i = 0; i < V; ++i) s->output[e][i] = expf(s->output[e][i] - max_out);
  float tot = 0;
  for (int i = 0; i < V; ++i) tot += s->output[e][i];
  for (int i = 0; i < V; ++i) s->output[e][i] /= tot;
  if (++s->epoch == H_) s->epoch = 0;
  return s->output[e];
}

/* Lstm::Perceive, lstm.cpp:87-118 */
static const float* lstm_perceive(orc_lstm* s, unsigned symbol) {
  const int V = s->V;
  int last_epoch = (int)s->epoch - 1;
  if (last_epoch == -1) last_epoch = H_ - 1;
  unsigned old_input = s->input_history[last_epoch];
  s->input_history[last_epoch] = symbol;
  if (s->epoch == 0) {
    for (int epoch = H_ - 1; epoch >= 0; --epoch) {
      for (int layer = L_ - 1; layer >= 0; --layer) {
        int offset = layer * C_;
        for (int i = 0; i < V; ++i) {
          float error = ((unsigned)i == s->input_history[epoch]) ? (s->output[epoch][i] - 1) : s->output[epoch][i];
          const float* ol = s->output_layer[epoch] + (size_t)i * NH;
          for (int j = 0; j < C_; ++j) s->hidden_error[j] += ol[j + offset] * error;
        }
        int prev_epoch = epoch - 1;
        if (prev_epoch == -1) prev_epoch = H_ - 1;
        unsigned input_symbol = s->input_history[prev_epoch];
        if (epoch == 0) input_symbol = old_input;
        layer_backward(s, &s->ly[layer], s->layer_input[epoch][layer], epoch, layer, (int)input_symbol, s->hidden_error);
      }
    }
  }
  const int e = (int)s->epoch;
  for (int i = 0; i < V; ++i) {
    float error = ((unsigned)i == symbol) ? (s->output[last_epoch][i] - 1) : s->output[last_epoch][i];
    const float* src = s->output_layer[last_epoch] + (size_t)i * NH;
    float* dst = s->output_layer[e] + (size_t)i * NH;
    float le = s->lr * error;
    for (int j = 0; j < NH; ++j) dst[j] = src[j];
    for (int j = 0; j < NH; ++j) dst[j] -= le * s->hidden[j];
  }
  return lstm_predict(s, symbol);
}

/* ByteMixer::ByteUpdate (byte-mixer.cpp:22-38) for one byte model: `in256` is that model's
 * BytePredict() distribution (predictor.cpp:450-457), `byte` the byte just coded. */
void orc_lstm_byte_update(orc_lstm* s, const float* in256, int byte) {
  const int V = s->V;
  float* inputs = (float*)calloc((size_t)V, sizeof(float));
  int off = 0;
  for (int i = 0; i < 256; ++i)
    if (s->vocab[i]) inputs[off++] += in256[i];
  for (int i = 0; i < V; ++i) inputs[i] *= 2; /* 2 / num_models_ : unsigned division, num_models_ == 1 */
  for (int l = 0; l < L_; ++l) memcpy(s->layer_input[s->epoch][l], inputs, (size_t)V * sizeof(float)); /* Lstm::SetInput */
  free(inputs);
  const float* out = lstm_perceive(s, (unsigned)s->byte_map[byte]);
  off = 0;
  for (int i = 0; i < 256; ++i) s->probs[i] = s->vocab[i] ? out[off++] : 0;
  s->top = 255; /* ByteModel::ByteUpdate, byte-model.cpp:39-45 */
  s->bot = 0;
}

/* ByteModel::Predict, byte-model.cpp:8-24 */
float orc_lstm_bit_predict(orc_lstm* s) {
  int mid = s->bot + ((s->top - s->bot) / 2);
  float num = 0.0f;
  for (int i = mid + 1; i <= s->top; ++i) num += s->probs[i];
  float denom = num;
  for (int i = s->bot; i <= mid; ++i) denom += s->probs[i];
  s->ex = s->bot;
  float mx = s->probs[s->bot];
  for (int i = s->bot + 1; i <= s->top; ++i)
    if (s->probs[i] > mx) { mx = s->probs[i]; s->ex = i; }
  return denom == 0 ? 0.5f : num / denom;
}

/* ByteModel::Perceive, byte-model.cpp:30-37 */
void orc_lstm_bit_perceive(orc_lstm* s, int bit) {
  s->mid = s->bot + ((s->top - s->bot) / 2);
  if (bit) s->bot = s->mid + 1;
  else s->top = s->mid;
}

const float* orc_lstm_probs(const orc_lstm* s) { return s->probs; }
int orc_lstm_ex(const orc_lstm* s) { return s->ex; }
