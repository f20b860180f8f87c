// p8match.hip -- building block of the paq8 stage (SURVEY.md 8a'): MatchModel + SparseMatchModel over a chunk of known
// bytes (p8match_dev.h): two lanes of one workgroup around a shared byte-history ring. Bytes in; per bit 28 mixer inputs,
// 3 statistics other models read, 2 mixer selectors out. Not yet fed into a stage. Parity: tests/test_p8match_host.py
// (kernel body on the host vs the oracle), tests/test_zgpu_p8match.py (the kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/cmix_amd.h"
#include "p8match_build.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip
extern "C" int cmx_device_count(void);

__global__ __launch_bounds__(64) void cmx_p8match_kernel(P8MatchDev* d, const uint8_t* bytes, int n, int t0, int16_t* out, int* stats, int* sets) {
  const int tid = threadIdx.x;
  int y = d->last_y;
  for (int t = 0; t < t0; t++) y = (bytes[t >> 3] >> (7 - (t & 7))) & 1;
  for (int t = t0; t < 8 * n; t++) {
    const int bpos = t & 7, cur = bytes[t >> 3];
    const int c0 = (1 << bpos) | (cur >> (8 - bpos));
    p8d_match_step2(d, tid, y, bpos, c0, out + (size_t)t * 28, stats + (size_t)t * 3, sets + (size_t)t * 2);
    __syncthreads();
    p8d_match_step1(d, tid, bpos == 7, cur);
    __syncthreads();
    y = (cur >> (7 - bpos)) & 1;
  }
  if (tid == 0) d->last_y = y;
}

namespace {
struct P8MatchPolicy {
  std::vector<void*> blocks;
  bool ok = true;
  void* zalloc(size_t bytes) {
    void* p = nullptr;
    if (!ok || hipMalloc(&p, bytes + 64) != hipSuccess || hipMemset(p, 0, bytes + 64) != hipSuccess) { ok = false; return nullptr; }
    blocks.push_back(p);
    return p;
  }
  void upload(void* dst, const void* src, size_t bytes) { if (dst && hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice) != hipSuccess) ok = false; }
};
}  // namespace

struct cmx_p8match { int device = 0; P8MatchPolicy pol; P8MatchDev* d_dev = nullptr; };

extern "C" {
void cmx_p8match_destroy(cmx_p8match_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->pol.blocks) (void)hipFree(p);
  if (h->d_dev) (void)hipFree(h->d_dev);
  delete h;
}
cmx_p8match_t* cmx_p8match_create(int device, uint64_t match_bytes, uint64_t sparse_bytes, int hist_log2, const uint8_t nex1024[1024],
                                  const int16_t stretch4096[4096], const uint8_t ilog65536[65536]) {
  if (cmx_device_count() <= 0) { cmx_set_err("cmx_p8match_create: no HIP device visible (a gfx950 GPU is required)"); return nullptr; }
  if (!nex1024 || !stretch4096 || !ilog65536) { cmx_set_err("cmx_p8match_create: bad argument"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return nullptr; }
  cmx_p8match_t* h = new cmx_p8match();
  h->device = device;
  P8MatchDev host;
  bool ok = p8b::build_match(host, h->pol, match_bytes, sparse_bytes, hist_log2, nex1024, stretch4096, ilog65536) && h->pol.ok;
  ok = ok && hipMalloc((void**)&h->d_dev, sizeof(P8MatchDev)) == hipSuccess;
  ok = ok && hipMemcpy(h->d_dev, &host, sizeof(P8MatchDev), hipMemcpyHostToDevice) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  if (!ok) { cmx_set_err("cmx_p8match_create: bad sizes (powers of two, ring 2^12..2^30) or allocation failed"); cmx_p8match_destroy(h); return nullptr; }
  return h;
}
int cmx_p8match_run_from(cmx_p8match_t* h, const uint8_t* d_bytes, size_t nbytes, int first_bit, int16_t* d_out, int* d_stats, int* d_sets, void* stream);
int cmx_p8match_run(cmx_p8match_t* h, const uint8_t* d_bytes, size_t nbytes, int16_t* d_out, int* d_stats, int* d_sets, void* stream) {
  return cmx_p8match_run_from(h, d_bytes, nbytes, 0, d_out, d_stats, d_sets, stream);
}
int cmx_p8match_run_from(cmx_p8match_t* h, const uint8_t* d_bytes, size_t nbytes, int first_bit, int16_t* d_out, int* d_stats, int* d_sets, void* stream) {
  if (!h) { cmx_set_err("cmx_p8match_run: null handle"); return 1; }
  if (first_bit < 0 || first_bit > 7) { cmx_set_err("cmx_p8match_run_from: first_bit must be 0..7"); return 1; }
  if (nbytes == 0) return 0;
  if (!d_bytes || !d_out || !d_stats || !d_sets || nbytes > (1u << 24)) { cmx_set_err("cmx_p8match_run: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipLaunchKernelGGL(cmx_p8match_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, h->d_dev, d_bytes, (int)nbytes, first_bit, d_out, d_stats, d_sets);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_p8match_run: ") + hipGetErrorString(e)); return 1; }
  return 0;
}
}  // extern "C"
