// p8cm.hip -- building block of the paq8 stage (SURVEY.md 8a'): the FAMILY of paq8's older ContextMap instances over a
// chunk of known bits as one persistent workgroup, one lane per context, with the process-global rnd() draws handed
// out in the reference's order (p8cm_dev.h). Contexts arrive hashed (cmx_p8cm2_hash: ContextMap::set uses the same
// hash as ContextMap2::set). Not yet fed by a stage. Parity: tests/test_p8cm_host.py (kernel body on the host vs the
// oracle), tests/test_zgpu_p8cm.py (the kernel).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include <string>
#include <vector>

#include "../../include/cmix_amd.h"
#include "p8cm_build.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip
extern "C" int cmx_device_count(void);

__global__ __launch_bounds__(P8CM_MAXS) void cmx_p8cm_kernel(P8CmDev* d, const uint32_t* ctx, const uint16_t* chk, const uint8_t* bits, int16_t* out, int nbits) {
  __shared__ P8CmShared sh;
  const int s = threadIdx.x, S = d->nslots;
  if (s == 0) { sh.r = d->regs; sh.rnd = d->rnd; }
  int last_y = d->last_y, c1 = d->c1;
  __syncthreads();
  for (int t = 0; t < nbits; t++) {
    const P8CmBit u = p8d_cm_bit(d, ctx, chk, bits, out, nullptr, t, &last_y, &c1);
    if (s < S) p8d_cm_touch(d, &sh, u, s);
    __syncthreads();
    if (s < S) p8d_cm_check(d, &sh, s);
    __syncthreads();
    if (s < S) p8d_cm_draw(d, &sh, s);
    __syncthreads();
    if (s < S) p8d_cm_run(d, &sh, u, s);
    __syncthreads();
  }
  if (s == 0) { d->regs = sh.r; d->rnd = sh.rnd; d->last_y = last_y; d->c1 = c1; }
}

namespace {
struct P8FamPolicy {
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

struct cmx_p8cm {
  int device = 0, nslots = 0;
  P8FamPolicy pol;
  P8CmDev* d_dev = nullptr;
};

extern "C" {

void cmx_p8cm_destroy(cmx_p8cm_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->pol.blocks) (void)hipFree(p);
  if (h->d_dev) (void)hipFree(h->d_dev);
  delete h;
}

cmx_p8cm_t* cmx_p8cm_create(int device, int ninst, const uint64_t* size_bytes, const int* counts, const uint8_t nex1024[1024], const int16_t stretch4096[4096],
                            const uint8_t ilog257[257]) {
  if (cmx_device_count() <= 0) { cmx_set_err("cmx_p8cm_create: no HIP device visible (a gfx950 GPU is required)"); return nullptr; }
  if (!size_bytes || !counts || !nex1024 || !stretch4096 || !ilog257) { cmx_set_err("cmx_p8cm_create: bad argument"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return nullptr; }
  cmx_p8cm_t* h = new cmx_p8cm();
  h->device = device;
  P8CmDev* host = new P8CmDev();
  bool ok = p8b::build_family(*host, h->pol, ninst, size_bytes, counts, nex1024, stretch4096, ilog257) && h->pol.ok;
  if (ok) {
    h->nslots = host->nslots;
    const char* serial = getenv("CMX_P8CM_SERIAL");   // A/B switch: the reference's serial walk on lane 0
    if (serial && serial[0] == '1') host->slot_parallel = 0;
  }
  ok = ok && hipMalloc((void**)&h->d_dev, sizeof(P8CmDev)) == hipSuccess;
  ok = ok && hipMemcpy(h->d_dev, host, sizeof(P8CmDev), hipMemcpyHostToDevice) == hipSuccess;
  ok = ok && hipDeviceSynchronize() == hipSuccess;
  delete host;
  if (!ok) { cmx_set_err("cmx_p8cm_create: bad geometry (<= 16 instances, <= 256 contexts, sizes powers of two >= 64 KB) or allocation failed"); cmx_p8cm_destroy(h); return nullptr; }
  return h;
}

int cmx_p8cm_slots(cmx_p8cm_t* h) { return h ? h->nslots : 0; }

int cmx_p8cm_run(cmx_p8cm_t* h, const uint32_t* d_ctx, const uint16_t* d_chk, const uint8_t* d_bits, size_t nbytes, int16_t* d_out, void* stream) {
  if (!h) { cmx_set_err("cmx_p8cm_run: null handle"); return 1; }
  if (nbytes == 0) return 0;
  if (!d_ctx || !d_chk || !d_bits || !d_out || nbytes > (1u << 24)) { cmx_set_err("cmx_p8cm_run: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipLaunchKernelGGL(cmx_p8cm_kernel, dim3(1), dim3(P8CM_MAXS), 0, (hipStream_t)stream, h->d_dev, d_ctx, d_chk, d_bits, d_out, (int)(8 * nbytes));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_p8cm_run: ") + hipGetErrorString(e)); return 1; }
  return 0;
}

}  // extern "C"
