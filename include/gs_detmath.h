/* gs_detmath.h -- the one piece of arithmetic that the HIP mapper kernels and the CPU
 * oracle must agree on bit-for-bit.
 *
 * Why: the tile mapper turns f32 geometry into INTEGER decisions (which tiles a splat
 * touches, reference taichi_lib/grid_query.py:73-91).  Every step of that computation is an
 * IEEE-754 correctly rounded operation (+ - * / sqrt floor ceil, compiled with
 * -ffp-contract=off on both sides) except one: ln(alpha / alpha_threshold)
 * (grid_query.py:76, perspective/projection.py:61).  libm's logf (glibc) and the device
 * logf (ocml) differ in the last bit for some arguments, which would make "tile indices
 * bit-exact" unprovable.  gs_det_logf is a natural logarithm built only from correctly
 * rounded f32 operations and integer bit manipulation, so x86 and gfx950 produce the same
 * bits.  Algorithm: the classic argument reduction x = 2^k * (1+f), sqrt(2)/2 < 1+f < sqrt(2),
 * then log(1+f) = f - hfsq + s*(hfsq+R(z)), s = f/(2+f), z = s*s, R an even minimax
 * polynomial (the published fdlibm/msun single-precision scheme; error < 1 ulp).
 *
 * Plain C (also valid C++/HIP).  No FMA is used, and callers compile with
 * -ffp-contract=off, so no contraction can change the rounding.
 */
#ifndef GS_DETMATH_H
#define GS_DETMATH_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__) || defined(__HIP__)
#define GS_HD __host__ __device__ __forceinline__
#else
#define GS_HD static inline
#endif

GS_HD uint32_t gs_f32_bits(float x) {
  uint32_t u;
  memcpy(&u, &x, 4);
  return u;
}

GS_HD float gs_bits_f32(uint32_t u) {
  float x;
  memcpy(&x, &u, 4);
  return x;
}

/* Natural log for finite, normal, positive x (the mapper only ever passes
 * alpha/threshold in (1, 1/threshold]).  x <= 0, NaN -> NaN; +inf -> +inf;
 * subnormals are scaled first. */
GS_HD float gs_det_logf(float x) {
  const float ln2_hi = 6.9313812256e-01f; /* 0x3f317180 */
  const float ln2_lo = 9.0580006145e-06f; /* 0x3717f7d1 */
  const float Lg1 = 6.6666668653e-01f;    /* 0x3f2aaaab */
  const float Lg2 = 4.0000000596e-01f;    /* 0x3ecccccd */
  const float Lg3 = 2.8571429849e-01f;    /* 0x3e924925 */
  const float Lg4 = 2.2222198546e-01f;    /* 0x3e638e29 */

  uint32_t ix = gs_f32_bits(x);
  int k = 0;
  if (ix == 0u || ix == 0x80000000u) return gs_bits_f32(0xff800000u); /* log(0) = -inf */
  if (ix >= 0x80000000u) return gs_bits_f32(0x7fc00000u);              /* negative or -NaN */
  if (ix >= 0x7f800000u) return x;                                     /* +inf / NaN */
  if (ix < 0x00800000u) {                                              /* subnormal */
    x = x * 33554432.0f; /* 2^25 */
    ix = gs_f32_bits(x);
    k = -25;
  }
  /* normalise so that 1+f lies in [sqrt(2)/2, sqrt(2)) */
  ix += 0x3f800000u - 0x3f3504f3u;
  k += (int)(ix >> 23) - 127;
  ix = (ix & 0x007fffffu) + 0x3f3504f3u;
  x = gs_bits_f32(ix);

  float f = x - 1.0f;
  float s = f / (2.0f + f);
  float z = s * s;
  float w = z * z;
  float t1 = w * (Lg2 + w * Lg4);
  float t2 = z * (Lg1 + w * Lg3);
  float R = t2 + t1;
  float hfsq = (0.5f * f) * f;
  float dk = (float)k;
  return (s * (hfsq + R) + dk * ln2_lo) - hfsq + f + dk * ln2_hi;
}

#endif /* GS_DETMATH_H */
