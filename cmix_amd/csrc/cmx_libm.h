// cmx_libm.h -- bit-faithful device re-implementations of the glibc 2.35 libm
// float routines that sit on the reference's per-bit float path:
//   expf   (Sigmoid::Logistic, reference src/mixer/sigmoid.cpp:19-21;
//           softmax, src/mixer/lstm.cpp:143)
//   tanhf  (LSTM candidate/cell, src/mixer/lstm-layer.cpp:71,78)
// The reference stream is only reproducible if these return the very same
// float as the host libm the -O3 reference binary calls, so
// each function follows glibc's published algorithm operation by operation:
//   * expf: sysdeps/ieee754/flt-32/e_expf.c (Szabolcs Nagy's exp2f-table
//     algorithm, N = 32) in the form the x86-64 `__expf_fma` ifunc variant
//     executes on any FMA-capable host: the double-precision polynomial with the
//     four contractions the glibc build contains (verified against the
//     disassembly of libm.so.6 and exhaustively against expf() for all 2^32
//     inputs by tests/test_libm_host.py).
//   * tanhf/expm1f: sysdeps/ieee754/flt-32/s_tanhf.c, s_expm1f.c (Sun msun
//     lineage) -- plain float arithmetic, no contraction.
// Compile with -ffp-contract=off. The header is valid both as host C++ (for the
// exhaustive host-side check) and as HIP device code.
#ifndef CMX_LIBM_H
#define CMX_LIBM_H

#include <stdint.h>

#if defined(__HIPCC__)
#define CMX_HD __host__ __device__ __forceinline__
#else
#define CMX_HD static inline
#endif

CMX_HD uint32_t cmx_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }
CMX_HD float cmx_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
CMX_HD uint64_t cmx_d2u(double d) { return __builtin_bit_cast(uint64_t, d); }
CMX_HD double cmx_u2d(uint64_t u) { return __builtin_bit_cast(double, u); }

#if defined(__HIP_DEVICE_COMPILE__)
__device__ static const uint64_t cmx_exp2f_tab[32] = {
#else
static const uint64_t cmx_exp2f_tab[32] = {
#endif
    0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
    0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
    0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
    0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
    0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
    0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
    0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
    0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};

// glibc __expf (FMA build). T[] = asuint64(2^(i/32)) - (i << 47). `tab`: where the caller keeps the 32 table words (a kernel
// on a latency-critical path keeps a copy in LDS: from device-global memory the lookup is an L2 round trip).
CMX_HD float cmx_expf_t(float x, const uint64_t* tab) {
  const double InvLn2N = 0x1.71547652b82fep+5;  // 32/ln2
  const double Shift = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-20, C1 = 0x1.ebfce50fac4f3p-13, C2 = 0x1.62e42ff0c52d6p-6;
  uint32_t ux = cmx_f2u(x);
  uint32_t abstop = (ux >> 20) & 0x7ff;
  if (abstop >= 0x42b) {  // |x| >= 88 or non-finite
    if (ux == 0xff800000u) return 0.0f;
    if (abstop >= 0x7f8) return x + x;
    if (x > 0x1.62e42ep6f) return cmx_u2f(0x7f800000u);  // overflow
    if (x < -0x1.9fe368p6f) return 0.0f;                  // underflow
  }
  double xd = (double)x;
  double kd = __builtin_fma(InvLn2N, xd, Shift);
  uint64_t ki = cmx_d2u(kd);
  kd = kd - Shift;
  double r = __builtin_fma(InvLn2N, xd, -kd);
  uint64_t t = tab[ki & 31] + (ki << 47);
  double s = cmx_u2d(t);
  double z = __builtin_fma(C0, r, C1);
  double r2 = r * r;
  double y = __builtin_fma(C2, r, 1.0);
  y = __builtin_fma(z, r2, y);
  y = y * s;
  return (float)y;
}

CMX_HD float cmx_expf(float x) { return cmx_expf_t(x, cmx_exp2f_tab); }

// glibc __expm1f (s_expm1f.c)
CMX_HD float cmx_expm1f(float x) {
  const float one = 1.0f, huge = 1.0e+30f, tiny = 1.0e-30f;
  const float o_threshold = 8.8721679688e+01f, ln2_hi = 6.9313812256e-01f,
              ln2_lo = 9.0580006145e-06f, invln2 = 1.4426950216e+00f;
  const float Q1 = -3.3333335072e-02f, Q2 = 1.5873016091e-03f, Q3 = -7.9365076090e-05f,
              Q4 = 4.0082177293e-06f, Q5 = -2.0109921195e-07f;
  float y, hi, lo, c = 0.0f, t, e, hxs, hfx, r1;
  int32_t k;
  uint32_t hx = cmx_f2u(x);
  uint32_t xsb = hx & 0x80000000u;
  hx &= 0x7fffffffu;

  if (hx >= 0x4195b844u) {    // |x| >= 27 ln2
    if (hx >= 0x42b17218u) {  // |x| >= 88.721...
      if (hx > 0x7f800000u) return x + x;
      if (hx == 0x7f800000u) return xsb == 0 ? x : -1.0f;
      if (x > o_threshold) return huge * huge;
    }
    if (xsb != 0) return tiny - one;
  }

  if (hx > 0x3eb17218u) {    // |x| > 0.5 ln2
    if (hx < 0x3F851592u) {  // |x| < 1.5 ln2
      if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
      else { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
    } else {
      k = (int32_t)(invln2 * x + (xsb == 0 ? 0.5f : -0.5f));
      t = (float)k;
      hi = x - t * ln2_hi;
      lo = t * ln2_lo;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  } else if (hx < 0x33000000u) {  // |x| < 2^-25
    t = huge + x;
    return x - (t - (huge + x));
  } else {
    k = 0;
  }

  hfx = 0.5f * x;
  hxs = x * hfx;
  r1 = one + hxs * (Q1 + hxs * (Q2 + hxs * (Q3 + hxs * (Q4 + hxs * Q5))));
  t = 3.0f - r1 * hfx;
  e = hxs * ((r1 - t) / (6.0f - x * t));
  if (k == 0) return x - (x * e - hxs);
  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5f * (x - e) - 0.5f;
  if (k == 1) {
    if (x < -0.25f) return -2.0f * (e - (x + 0.5f));
    return one + 2.0f * (x - e);
  }
  if (k <= -2 || k > 56) {
    y = one - (e - x);
    y = cmx_u2f(cmx_f2u(y) + ((uint32_t)k << 23));
    return y - one;
  }
  if (k < 23) {
    t = cmx_u2f(0x3f800000u - (0x1000000u >> k));
    y = t - (e - x);
    y = cmx_u2f(cmx_f2u(y) + ((uint32_t)k << 23));
  } else {
    t = cmx_u2f((uint32_t)(0x7f - k) << 23);
    y = x - (e + t);
    y += one;
    y = cmx_u2f(cmx_f2u(y) + ((uint32_t)k << 23));
  }
  return y;
}

// glibc __tanhf (s_tanhf.c)
CMX_HD float cmx_tanhf(float x) {
  const float one = 1.0f, two = 2.0f, tiny = 1.0e-30f;
  float t, z;
  uint32_t jx = cmx_f2u(x);
  uint32_t ix = jx & 0x7fffffffu;
  if (ix >= 0x7f800000u) {
    if ((int32_t)jx >= 0) return one / x + one;
    return one / x - one;
  }
  if (ix < 0x41b00000u) {  // |x| < 22
    if (ix == 0) return x;
    if (ix < 0x24000000u) return x * (one + x);  // |x| < 2^-55
    float ax = cmx_u2f(ix);
    if (ix >= 0x3f800000u) {  // |x| >= 1
      t = cmx_expm1f(two * ax);
      z = one - two / (t + two);
    } else {
      t = cmx_expm1f(-two * ax);
      z = -t / (t + two);
    }
  } else {
    z = one - tiny;
  }
  return ((int32_t)jx >= 0) ? z : -z;
}

// Sigmoid::Logistic, reference src/mixer/sigmoid.cpp:19-21
CMX_HD float cmx_logistic(float x) { return 1.0f / (1.0f + cmx_expf(-x)); }
CMX_HD float cmx_logistic_t(float x, const uint64_t* tab) { return 1.0f / (1.0f + cmx_expf_t(-x, tab)); }

#endif  // CMX_LIBM_H
