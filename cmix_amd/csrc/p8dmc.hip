// p8dmc.hip -- building block of the paq8 stage (SURVEY.md 8a'): the DMC forest over a chunk of known bits
// (p8dmc_dev.h). d_bits in, 6 mixer inputs per bit out; ~0.7 GB of nodes at cmix's level 11. Not yet fed into a stage.
// Parity: tests/test_p8dmc_host.py (kernel body on the host vs the oracle), tests/test_zgpu_p8dmc.py (the kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/cmix_amd.h"
#include "p8dmc_build.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip
extern "C" int cmx_device_count(void);

__global__ __launch_bounds__(P8DMC_THREADS) void cmx_p8dmc_kernel(P8DmcDev* d, const uint8_t* bits, int16_t* out, int nbits) {
  __shared__ P8DmcShared sh;
  const int tid = threadIdx.x;
  int y = d->last_y;
  const uint32_t done = d->bits_done;
  for (int t = 0; t < nbits; t++) {
    p8d_dmc_step1(d, &sh, tid, y);
    __syncthreads();
    p8d_dmc_step2(d, &sh, tid, (int)((done + (uint32_t)t) & 7), out + (size_t)t * 6);
    __syncthreads();
    p8d_dmc_step3(d, &sh, tid);
    __syncthreads();
    y = bits[t];
  }
  if (tid == 0) { d->last_y = y; d->bits_done = done + (uint32_t)nbits; }
}

namespace {
struct P8DmcPolicy {
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

struct cmx_p8dmc { int device = 0; P8DmcPolicy pol; P8DmcDev* d_dev = nullptr; };

extern "C" {
void cmx_p8dmc_destroy(cmx_p8dmc_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->pol.blocks) (void)hipFree(p);
  if (h->d_dev) (void)hipFree(h->d_dev);
  delete h;
}
cmx_p8dmc_t* cmx_p8dmc_create(int device, int level, const uint8_t nex1024[1024], const int16_t stretch4096[4096]) {
  if (cmx_device_count() <= 0) { cmx_set_err("cmx_p8dmc_create: no HIP device visible (a gfx950 GPU is required)"); return nullptr; }
  if (!nex1024 || !stretch4096 || level < 0 || level > 12) { cmx_set_err("cmx_p8dmc_create: bad argument"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return nullptr; }
  cmx_p8dmc_t* h = new cmx_p8dmc();
  h->device = device;
  P8DmcDev host;
  p8b::build_dmc(host, h->pol, level, nex1024, stretch4096);
  bool ok = h->pol.ok && hipMalloc((void**)&h->d_dev, sizeof(P8DmcDev)) == hipSuccess;
  ok = ok && hipMemcpy(h->d_dev, &host, sizeof(P8DmcDev), hipMemcpyHostToDevice) == hipSuccess && hipDeviceSynchronize() == hipSuccess;
  if (!ok) { cmx_set_err("cmx_p8dmc_create: allocation failed"); cmx_p8dmc_destroy(h); return nullptr; }
  return h;
}
int cmx_p8dmc_run(cmx_p8dmc_t* h, const uint8_t* d_bits, size_t nbits, int16_t* d_out, void* stream) {
  if (!h) { cmx_set_err("cmx_p8dmc_run: null handle"); return 1; }
  if (nbits == 0) return 0;
  if (!d_bits || !d_out || nbits > (1u << 27)) { cmx_set_err("cmx_p8dmc_run: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipLaunchKernelGGL(cmx_p8dmc_kernel, dim3(1), dim3(P8DMC_THREADS), 0, (hipStream_t)stream, h->d_dev, d_bits, d_out, (int)nbits);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_p8dmc_run: ") + hipGetErrorString(e)); return 1; }
  return 0;
}
}  // extern "C"
