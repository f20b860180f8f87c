// p8mixer.hip -- building block of the paq8 / fxcm stages (SURVEY.md 8a'): paq8's two-layer int16 mixer
// (reference src/models/paq8.cpp:513-598; dot_product / train :403-432) for a chunk of already-known bits, one
// persistent workgroup per stream. Not yet wired into a stage: its inputs (1552 stretch-domain int16 per bit and 28
// weight-set selectors per bit) are what the paq8 context models will produce. C ABI: cmx_p8mixer_* (include/cmix_amd.h).
//
// Integer work, HBM/L2-bound (per bit 28 rows x 1552 int16 are read, trained and written back: 174 KB): no MFMA -- the
// dot product shifts every PAIR of products right by 8 before adding (pmaddwd / psrad), which a matrix unit cannot do;
// sums are modulo 2^32, so the reduction across lanes is exact in any order.
//   layout   weight rows dense [M][N] int16 (the reference creates rows lazily, filled with init_w: same values)
//   wave w   owns sets 4w .. 4w+3 of the 28; lane l holds 16-byte groups l, l+64, l+128, l+192 of a row in registers
//            from the dot product to the training step of the same bit, then stores them back
//   order    the reference trains a bit's rows at the start of the next bit (Mixer::update); nothing reads them in
//            between, so training right after the prediction gives the same rows and leaves no pending state.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>

#include "../../include/cmix_amd.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip
int cmx_device_count(void);

namespace {
constexpr int P8_N = 1552, P8_S = 28, P8_N2 = 32, P8_THREADS = 448, P8_GROUPS = P8_N / 8;  // 194 groups of 8 int16

struct P8MixerDev {
  int16_t* wx;            // [M][P8_N]
  int16_t* wx2;           // [P8_N2] second layer, one row
  const int16_t* squash;  // [4096] squash(d), d + 2048
  const int16_t* stretch; // [4096]
  int M;
};

__device__ __forceinline__ int sat16(int v) { return v > 32767 ? 32767 : v < -32768 ? -32768 : v; }
__device__ __forceinline__ int sq(const int16_t* t, int d) { return d > 2047 ? 4095 : d < -2047 ? 0 : t[d + 2048]; }
__device__ __forceinline__ int lo16(uint32_t v) { return (int)(int16_t)(v & 0xffffu); }
__device__ __forceinline__ int hi16(uint32_t v) { return (int)(int16_t)(v >> 16); }
// one dword = one pair: ((t0*w0 + t1*w1) >> 8), wrapping
__device__ __forceinline__ uint32_t pair_dot(uint32_t t, uint32_t w) {
  const uint32_t s = (uint32_t)(lo16(t) * lo16(w)) + (uint32_t)(hi16(t) * hi16(w));
  return (uint32_t)((int32_t)s >> 8);
}
__device__ __forceinline__ int train1(int t, int w, int err) {  // one weight: paq8.cpp:415-430
  int v = sat16(2 * t);
  v = (v * err) >> 16;
  v = sat16(v + 1) >> 1;
  return sat16(v + w);
}
__device__ __forceinline__ uint32_t pair_train(uint32_t t, uint32_t w, int err) {
  const int a = train1(lo16(t), lo16(w), err), b = train1(hi16(t), hi16(w), err);
  return ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16);
}
}  // namespace

extern "C" __global__ __launch_bounds__(P8_THREADS) void cmx_p8mixer_kernel(const P8MixerDev D, const int16_t* __restrict__ x,
                                                                           const int* __restrict__ rows,
                                                                           const uint8_t* __restrict__ bits, int T,
                                                                           int* __restrict__ p_out, int* __restrict__ pr_out) {
  __shared__ __attribute__((aligned(16))) uint32_t xs[P8_N / 2];  // the bit's inputs, as pairs
  __shared__ int pr_s[P8_N2];                                      // first-layer outputs
  __shared__ uint32_t st_s[P8_N2 / 2];                             // stretch(pr) as pairs (second-layer inputs)
  __shared__ int p_s;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  for (int t = 0; t < T; ++t) {
    // ---- inputs of the bit -> LDS (3104 bytes, 16-byte groups) ----
    const uint4* xg = reinterpret_cast<const uint4*>(x + (size_t)t * P8_N);
    if (tid < P8_GROUPS) reinterpret_cast<uint4*>(xs)[tid] = xg[tid];
    if (tid < P8_N2) pr_s[tid] = 0;
    __syncthreads();
    // ---- first layer: 4 sets per wave ----
    uint4 w[4][4];
    int row[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      row[q] = rows[(size_t)t * P8_S + 4 * wave + q];
      const uint4* wr = reinterpret_cast<const uint4*>(D.wx + (size_t)row[q] * P8_N);
      uint32_t acc = 0;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int grp = lane + 64 * g;
        w[q][g] = grp < P8_GROUPS ? wr[grp] : make_uint4(0, 0, 0, 0);
        const uint4 xv = grp < P8_GROUPS ? reinterpret_cast<const uint4*>(xs)[grp] : make_uint4(0, 0, 0, 0);
        acc += pair_dot(xv.x, w[q][g].x) + pair_dot(xv.y, w[q][g].y) + pair_dot(xv.z, w[q][g].z) + pair_dot(xv.w, w[q][g].w);
      }
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) {
        const int pr = sq(D.squash, (int32_t)(acc * 9u) >> 9);  // paq8.cpp:566
        pr_s[4 * wave + q] = pr;
        if (pr_out) pr_out[(size_t)t * P8_S + 4 * wave + q] = pr;
      }
    }
    __syncthreads();
    // ---- second layer (one row of 32, 28 live inputs): wave 0 ----
    if (wave == 0) {
      const int a = lane < P8_S ? D.stretch[pr_s[lane]] : 0;  // mp->add(stretch(pr[i])), padding zeros
      const int b = __shfl_down(a, 1);
      if ((lane & 1) == 0 && lane < P8_N2) st_s[lane >> 1] = ((uint32_t)a & 0xffffu) | ((uint32_t)b << 16);
      uint32_t acc = 0;
      __builtin_amdgcn_s_waitcnt(0);  // the pairs are in LDS before other lanes of this wave read them
      __builtin_amdgcn_wave_barrier();
      if (lane < P8_N2 / 2) acc = pair_dot(st_s[lane], reinterpret_cast<const uint32_t*>(D.wx2)[lane]);
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
      if (lane == 0) {
        const int p = sq(D.squash, (int32_t)acc >> 9);  // paq8.cpp:578
        p_s = p;
        p_out[t] = p;
      }
    }
    __syncthreads();
    // ---- the coded bit: train this bit's rows (paq8.cpp:528-541) and store them ----
    const int y = bits[t];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int err = (int)(int16_t)(((y << 12) - pr_s[4 * wave + q]) * 7);
      uint4* wr = reinterpret_cast<uint4*>(D.wx + (size_t)row[q] * P8_N);
      // two of a wave's four sets may select the same row only if the caller hands in equal row numbers, which the
      // cumulative bases of Mixer::set exclude
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int grp = lane + 64 * g;
        if (grp < P8_GROUPS && err) {
          const uint4 xv = reinterpret_cast<const uint4*>(xs)[grp];
          uint4 v = w[q][g];
          v.x = pair_train(xv.x, v.x, err); v.y = pair_train(xv.y, v.y, err);
          v.z = pair_train(xv.z, v.z, err); v.w = pair_train(xv.w, v.w, err);
          wr[grp] = v;
        }
      }
    }
    if (wave == 0 && lane < P8_N2 / 2) {
      const int err2 = (int)(int16_t)(((y << 12) - p_s) * 7);
      uint32_t* w2 = reinterpret_cast<uint32_t*>(D.wx2);
      if (err2) w2[lane] = pair_train(st_s[lane], w2[lane], err2);
    }
    __syncthreads();  // rows stored / xs free before the next bit (a later bit may select a row trained here)
  }
}

struct cmx_p8mixer {
  int device = 0;
  P8MixerDev dev{};
  int16_t* d_tables = nullptr;
};

extern "C" {

void cmx_p8mixer_destroy(cmx_p8mixer_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  if (h->dev.wx) (void)hipFree(h->dev.wx);
  if (h->dev.wx2) (void)hipFree(h->dev.wx2);
  if (h->d_tables) (void)hipFree(h->d_tables);
  delete h;
}

cmx_p8mixer_t* cmx_p8mixer_create(int device, int total_rows, const int16_t squash4096[4096], const int16_t stretch4096[4096]) {
  if (cmx_device_count() <= 0) { cmx_set_err("cmx_p8mixer_create: no HIP device visible (a gfx950 GPU is required)"); return nullptr; }
  if (total_rows <= 0 || !squash4096 || !stretch4096) { cmx_set_err("cmx_p8mixer_create: bad argument"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return nullptr; }
  cmx_p8mixer_t* h = new cmx_p8mixer();
  h->device = device;
  h->dev.M = total_rows;
  const size_t n1 = (size_t)total_rows * P8_N;
  bool ok = hipMalloc((void**)&h->dev.wx, n1 * 2) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->dev.wx2, P8_N2 * 2) == hipSuccess;
  ok = ok && hipMalloc((void**)&h->d_tables, 2 * 4096 * 2) == hipSuccess;
  if (ok) {
    // rows are created lazily with init_w = 32 in the reference (Mixer m(NUM_INPUTS, 77472, NUM_SETS, 32), :8109);
    // the second layer's single row starts at 0x7fff (:589)
    ok = hipMemsetD16(h->dev.wx, 32, n1) == hipSuccess;
    ok = ok && hipMemsetD16(h->dev.wx2, 0x7fff, P8_N2) == hipSuccess;
    ok = ok && hipMemcpy(h->d_tables, squash4096, 4096 * 2, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipMemcpy(h->d_tables + 4096, stretch4096, 4096 * 2, hipMemcpyHostToDevice) == hipSuccess;
    ok = ok && hipDeviceSynchronize() == hipSuccess;
  }
  if (!ok) { cmx_set_err("cmx_p8mixer_create: allocation / init failed"); cmx_p8mixer_destroy(h); return nullptr; }
  h->dev.squash = h->d_tables;
  h->dev.stretch = h->d_tables + 4096;
  return h;
}

int cmx_p8mixer_run(cmx_p8mixer_t* h, const int16_t* d_x, const int* d_rows, const uint8_t* d_bits, size_t nbits, int* d_p,
                    int* d_pr, void* stream) {
  if (!h) { cmx_set_err("cmx_p8mixer_run: null handle"); return 1; }
  if (nbits == 0) return 0;
  if (!d_x || !d_rows || !d_bits || !d_p) { cmx_set_err("cmx_p8mixer_run: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipLaunchKernelGGL(cmx_p8mixer_kernel, dim3(1), dim3(P8_THREADS), 0, (hipStream_t)stream, h->dev, d_x, d_rows, d_bits,
                     (int)nbits, d_p, d_pr);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_p8mixer_run: ") + hipGetErrorString(e)); return 1; }
  return 0;
}

}  // extern "C"
