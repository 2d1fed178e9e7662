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
 *
 * gs_det_sqrtf: the correctly rounded square root.  On the host that is sqrtf.  On the device it is NOT
 * `__fsqrt_rn`: hipcc's header maps that name to the native v_sqrt_f32 (1 ulp) unless OCML_BASIC_ROUNDED_OPERATIONS is
 * defined -- rounds 1 and 2 used it, and C5 at its full size (6 M Gaussians) found the one splat in 5 million whose
 * bounding box ends 2e-4 px past a tile edge in IEEE arithmetic and 1 ulp short of it with v_sqrt_f32 (two overlaps
 * of 19 169 107 missing).  The device form below takes v_sqrt_f32 and settles the last bit with two exact fma
 * residuals against the neighbouring floats (the scheme LLVM's own correctly rounded f32 sqrt lowering uses), so it does
 * not depend on -fhip-fp32-correctly-rounded-divide-sqrt either.
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

/* Correctly rounded sqrt for x >= 0 (NaN / negative / inf as the native instruction returns them). */
GS_HD float gs_det_sqrtf(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  /* below 2^-96 v_sqrt_f32's result would be subnormal-adjacent: scale by 2^32, unscale the root by 2^-16 */
  const int tiny = x < 1.262177448e-29f; /* 2^-96 */
  const float xs = tiny ? x * 4294967296.0f : x;
  float r = __builtin_amdgcn_sqrtf(xs);
  if (xs > 0.0f && xs < __builtin_huge_valf()) {
    const float r_dn = gs_bits_f32(gs_f32_bits(r) - 1u), r_up = gs_bits_f32(gs_f32_bits(r) + 1u);
    /* x - r_dn * r <= 0: the root lies at or below the midpoint of (r_dn, r) -> r_dn; x - r_up * r > 0: above the
     * midpoint of (r, r_up) -> r_up (exact residuals: one rounding each, of a quantity whose sign is what matters) */
    const float e_dn = __builtin_fmaf(-r_dn, r, xs), e_up = __builtin_fmaf(-r_up, r, xs);
    if (e_dn <= 0.0f) r = r_dn;
    if (e_up > 0.0f) r = r_up;
  }
  return tiny ? r * 1.52587890625e-05f : r; /* 2^-16 */
#else
  return __builtin_sqrtf(x);
#endif
}

#endif /* GS_DETMATH_H */
