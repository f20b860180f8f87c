// lstm_state.h -- HBM-resident state of one stream's byte-level LSTM byte mixer
// (reference src/mixer/{lstm,lstm-layer,byte-mixer}.*, src/models/byte-model.*).
#ifndef CMX_LSTM_STATE_H
#define CMX_LSTM_STATE_H
#include <stdint.h>

#define LSTM_C 200     // cells per layer          (predictor.cpp:190)
#define LSTM_L 2       // layers
#define LSTM_H 100     // truncated-BPTT horizon
#define LSTM_NH 401    // hidden_ size = C*L + 1 (bias)
#define LSTM_VP 256    // padded vocabulary stride
#define LSTM_UPDATE_LIMIT 3000
// workgroup geometry of the block kernels (lstm_block.hip)
#define LSTM_FB_GL 12       // workgroups per gate layer in the forward block
#define LSTM_FB_R 50        //   gate rows per workgroup (12 x 50 = 3 gates x 200 cells)
#define LSTM_FB_GO 8        // workgroups of the output layer
#define LSTM_FB_THREADS 256
#define LSTM_BP_G 20        // workgroups of the sequential BPTT part
#define LSTM_BP_J 10        //   cells per workgroup
#define LSTM_BP_THREADS 512

// In-launch hand-off counters of the block kernels; everything after `fail` is zeroed ahead of every launch.
struct LstmSync {
  unsigned fail;                           // sticky: a bounded wait ran out (cmx_lstm_failed)
  unsigned pad[3];
  unsigned raw_cnt[LSTM_L][LSTM_H];        // workgroups of a gate layer that delivered their raw sums of epoch e
  unsigned h_flag[LSTM_L][LSTM_H];         // the layer's hidden vector of epoch e is in h_ring
  unsigned logit_cnt[LSTM_H];              // output-layer workgroups that delivered their logits of epoch e
  unsigned bp_cnt[2 * LSTM_H];             // BPTT workgroups that delivered step s
};

// Arguments of one epoch-aligned block of 100 bytes (one BPTT round + 100 x (SGD, forward)) replayed as a
// HIP graph: the graph's kernel parameters are frozen at capture, so everything that changes per block is
// read from this device-resident record, written by a 1-lane kernel ahead of each replay.
struct LstmBlockArgs {
  const float* in_probs;         // byte model distributions of the chunk
  const unsigned char* bytes;    // the chunk's bytes
  float* out_probs;              // LSTM distributions out (may be NULL)
  unsigned long long n0;         // index in the chunk of the block's first byte
  int hc0;                       // hidden_ double-buffer index at the block's first byte
  int us;                        // LstmLayer::update_steps_ of this block's BPTT round
};

struct LstmState {
  int V;                         // vocabulary size (distinct bytes of the input)
  int insz[LSTM_L];              // layer_input sizes: 1+C+V, V+1+2C   (lstm.cpp:13-24)
  int rowlen[LSTM_L];            // gate weight row length = insz + V  (lstm.cpp:27)
  // Passed to the kernels BY VALUE (kernarg segment -> scalar loads); everything that changes per
  // byte and is known to the host (epoch = bytes_done % 100, the hidden_ double-buffer index,
  // LstmLayer::update_steps_) is a plain kernel argument instead of a device-memory field.
  int* dyn;                      // [4] device scalars: [0] = old_input of the current Perceive
  struct LstmBlockArgs* blk;     // device copy of the arguments of the 100-byte block a captured graph is replaying
  int byte_map[256];             // byte value -> vocabulary index (byte-mixer.cpp:9-12)
  unsigned char vocab[256];
  float lr;                      // 0.03
  int xcd;                       // >= 0: the single-workgroup kernels run as block `xcd` of 8 (XCD placement, speed only)
  int poll_sleep;                // A/B switch (CMX_LSTM_SLEEP=1): s_sleep 1 between two polls of an in-launch counter
  int avoid_xcd;                 // >= 0: the block kernels leave this XCD to the mixing network (CMX_MIXNET_XCD: its 27 workgroups share that XCD's L2) --
                                 //   the grids are padded and the blocks with blockIdx % 8 == avoid_xcd exit (observed: block b runs on XCD b % 8; speed only)

  // gate parameters, g = 0 forget, 1 input node, 2 output gate
  float* W[LSTM_L][3];           // [C][rowlen]   reference layout (coalesced for the BPTT matvecs)
  float* WT[LSTM_L][3];          // transposed copy for the forward chains, see lstm_wt_index()
  float* M[LSTM_L][3];           // Adam first moment  [C][rowlen]
  float* Vv[LSTM_L][3];          // Adam second moment [C][rowlen]
  float* gb[LSTM_L][3];          // [8][C]: gamma, beta, gamma_m, gamma_v, beta_m, beta_v, gamma_u, beta_u
  // per-time-step caches (ring of H)
  float* norm[LSTM_L][3];        // [H][C]
  float* gstate[LSTM_L][3];      // [H][C]  activated gate state
  float* ivar[LSTM_L][3];        // [H]
  float* last_state[LSTM_L];     // [H][C]
  float* tanh_state[LSTM_L];     // [H][C]
  float* in_gate_state[LSTM_L];  // [H][C]
  float* state[LSTM_L];          // [C]   LstmLayer::state_ (= stateb[0])
  float* stateb[2][LSTM_L];      // state_ double-buffered like hid[]: a block launch reads [hc], leaves [hc ^ 1]
  float* layer_input[LSTM_L];    // [H][insz]
  float* OL;                     // [H][V][401]   output_layer_
  float* OLT;                    // [H][401][VP]  transposed copy
  float* output;                 // [H][VP]       output_
  float* hid[2];                 // [401] double-buffered hidden_
  unsigned* input_history;       // [H]
  unsigned* bp_symbol;           // [H]  input_symbol seen by BackwardPass at each epoch
  float* raw[LSTM_L][3];         // [C]  pre-normalisation gate sums of the current step
  float* logits;                 // [VP]
  float* E[LSTM_L][3];           // [H][C] final gate errors of the current BPTT round
  const float* adam_tab;         // [3001][4] alpha, 1-beta1^t, 1-beta2^t (host libm)
  float* byte_probs;             // [256] ByteModel::probs_ of the byte mixer
  // block kernels (lstm_block.hip): values that cross workgroups inside a launch
  struct LstmSync* sync;
  float* raw_ring;               // [L][H][3*C]  pre-normalisation gate sums of epoch e
  float* h_ring;                 // [H][401]     hidden_ after epoch e
  float* logit_ring;             // [H][VP]
  float* bp_pub;                 // [2*H][2][C]  BPTT step s: hidden error after the output-layer chain | stored error
};

// Position of weight (cell i, column c) in the transposed copy WT[l][g]. The V one-hot columns come
// first, [c][i]. The dense columns (d = c - V, insz of them) are grouped four at a time so that a
// lane's four consecutive terms are one 16-byte load: [d/4][i][4]; the insz%4 trailing columns are
// stored [d][i] again. (dword loads left the CU's L2->L1 path at ~1/3 of its rate.)
#if defined(__HIPCC__) || defined(__cplusplus)
// grid size and role of a block when one XCD is left out: roles 0 .. n-1 go to the blocks whose index is not congruent to `avoid` modulo 8
static inline
#if defined(__HIPCC__)
__host__ __device__
#endif
int lstm_role_of_block(int b, int avoid) {   // -1: the block has no role
  if (avoid < 0) return b;
  if ((b & 7) == avoid) return -1;
  return b - (b >= avoid ? (b - avoid) / 8 + 1 : 0);
}
static inline int lstm_grid_for_roles(int n, int avoid) {
  if (avoid < 0) return n;
  int g = n;
  while (lstm_role_of_block(g - 1, avoid) < n - 1) ++g;
  return g;
}
static inline
#if defined(__HIPCC__)
__host__ __device__
#endif
size_t lstm_wt_index(int V, int insz, int c, int i) {
  if (c < V) return (size_t)c * LSTM_C + i;
  const int d = c - V, full = insz & ~3;
  if (d < full) return (size_t)V * LSTM_C + ((size_t)(d >> 2) * LSTM_C + i) * 4 + (d & 3);
  return (size_t)V * LSTM_C + (size_t)full * LSTM_C + (size_t)(d - full) * LSTM_C + i;
}
#endif

#endif
