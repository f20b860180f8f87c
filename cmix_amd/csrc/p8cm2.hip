// p8cm2.hip -- building block of the paq8 stage (SURVEY.md 8a'): paq8's ContextMap2 over a chunk of known bits as ONE
// persistent workgroup per instance, one lane per context (p8cm2_dev.h: bucket lists, overlap check, update + mix).
// The front end that produces the per-byte contexts (contextModel2 :8139-8153, TextModel :3006-3518, exeModel :7275) is
// host work (integer state machines over the bytes); here the contexts arrive already hashed (cmx_p8cm2_hash =
// ContextMap2::set). Not yet fed by a stage. Parity: tests/test_p8cm2_host.py runs the kernel body on the host against
// the oracle (pinned against the reference's own class); tests/test_zgpu_p8cm2.py runs the kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#include "../../include/cmix_amd.h"
#include "p8cm2_build.h"

void cmx_set_err(const std::string& s);  // cmx_api.hip
extern "C" int cmx_device_count(void);

__global__ __launch_bounds__(P8CM2_MAXC) void cmx_p8cm2_kernel(P8Cm2Dev* d, const uint32_t* ctx, const uint16_t* chk, const uint8_t* bits, int16_t* out, int nbits) {
  __shared__ P8Cm2Shared sh;
  const int i = threadIdx.x, C = d->C;
  if (i == 0) sh.r = d->regs;
  uint32_t run_bits = d->bits;
  int last_y = d->last_y;
  __syncthreads();
  for (int t = 0; t < nbits; t++) {
    const P8Cm2Bit u = p8d_bit(d, ctx, chk, bits, out, t, &run_bits, &last_y);
    if (i < C) p8d_touch(d, &sh, u, i);
    __syncthreads();
    if (i < C) p8d_conflict(d, &sh, i);
    __syncthreads();
    if (i < C) p8d_run(d, &sh, u, i);
    __syncthreads();
  }
  if (i == 0) { d->regs = sh.r; d->bits = run_bits; d->last_y = last_y; }
}

namespace {
struct P8DevPolicy {
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

struct cmx_p8cm2 {
  int device = 0, count = 0;
  P8DevPolicy pol;
  P8Cm2Dev* d_dev = nullptr;
};

extern "C" {

void cmx_p8cm2_hash(uint64_t ctx, uint32_t index, uint64_t size_bytes, uint32_t* ctx32, uint16_t* chk16) {
  p8b::hash(ctx, index, p8b::hashbits(size_bytes), ctx32, chk16);
}

void cmx_p8cm2_destroy(cmx_p8cm2_t* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipDeviceSynchronize();
  for (void* p : h->pol.blocks) (void)hipFree(p);
  if (h->d_dev) (void)hipFree(h->d_dev);
  delete h;
}

cmx_p8cm2_t* cmx_p8cm2_create(int device, uint64_t size_bytes, int count, const uint8_t nex1024[1024], const int16_t stretch4096[4096],
                              const uint8_t ilog257[257]) {
  if (cmx_device_count() <= 0) { cmx_set_err("cmx_p8cm2_create: no HIP device visible (a gfx950 GPU is required)"); return nullptr; }
  if (!nex1024 || !stretch4096 || !ilog257) { cmx_set_err("cmx_p8cm2_create: bad argument"); return nullptr; }
  if (hipSetDevice(device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return nullptr; }
  cmx_p8cm2_t* h = new cmx_p8cm2();
  h->device = device; h->count = count;
  P8Cm2Dev host;
  bool ok = p8b::build(host, h->pol, size_bytes, count, nex1024, stretch4096, ilog257) && h->pol.ok;
  if (ok) {
    const char* serial = getenv("CMX_P8CM2_SERIAL");   // A/B switch: the reference's serial walk on lane 0
    if (serial && serial[0] == '1') host.slot_parallel = 0;
  }
  ok = ok && hipMalloc((void**)&h->d_dev, sizeof(P8Cm2Dev)) == hipSuccess;
  ok = ok && hipMemcpy(h->d_dev, &host, sizeof(P8Cm2Dev), hipMemcpyHostToDevice) == hipSuccess;
  ok = ok && hipDeviceSynchronize() == hipSuccess;
  if (!ok) { cmx_set_err("cmx_p8cm2_create: bad size / count (power of two >= 64 KB, 1..64 contexts) or allocation failed"); cmx_p8cm2_destroy(h); return nullptr; }
  return h;
}

int cmx_p8cm2_run(cmx_p8cm2_t* h, const uint32_t* d_ctx, const uint16_t* d_chk, const uint8_t* d_bits, size_t nbytes, int16_t* d_out, void* stream) {
  if (!h) { cmx_set_err("cmx_p8cm2_run: null handle"); return 1; }
  if (nbytes == 0) return 0;
  if (!d_ctx || !d_chk || !d_bits || !d_out || nbytes > (1u << 24)) { cmx_set_err("cmx_p8cm2_run: bad argument"); return 1; }
  if (hipSetDevice(h->device) != hipSuccess) { cmx_set_err("hipSetDevice failed"); return 1; }
  hipLaunchKernelGGL(cmx_p8cm2_kernel, dim3(1), dim3(P8CM2_MAXC), 0, (hipStream_t)stream, h->d_dev, d_ctx, d_chk, d_bits, d_out, (int)(8 * nbytes));
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) { cmx_set_err(std::string("cmx_p8cm2_run: ") + hipGetErrorString(e)); return 1; }
  return 0;
}

}  // extern "C"
