// glibc_libm.h — bit-exact re-implementations of glibc 2.35's x86-64 `exp`, `log`,
// `log1p`, `expm1` for MI355X device code (and for the host table builders).
//
// Why this exists: the reference computes every non-unanimous consensus column through
// Rust `f64::exp / ln / ln_1p / exp_m1` (crates/fgumi-consensus/src/phred.rs:156-189,
// base_builder.rs:677), which resolve to the platform libm — glibc 2.35 on this image
// (SURVEY.md §8c "third-party arithmetic").  ROCm's ocml transcendentals are not
// bit-identical to glibc, and the output contract is byte-identical BAM, so the kernels
// carry their own implementation of glibc's published algorithms:
//
//   * exp, log   — Szabolcs Nagy's table-driven routines (glibc sysdeps/ieee754/dbl-64/
//                  e_exp.c, e_log.c).  x86-64 glibc selects the `__exp_fma`/`__log_fma`
//                  ifunc variants on every FMA+AVX2 CPU; those are built with -mfma, so GCC
//                  contracted specific a*b+c pairs.  The contraction pattern below was read
//                  off the compiled `e_exp-fma.o` / `e_log-fma.o` of libm-2.35.a and is
//                  spelled with explicit fma(); everything else is plain IEEE mul/add.
//   * log1p, expm1 — fdlibm-derived (s_log1p.c, s_expm1.c); glibc 2.35 ships no FMA
//                  variant of these, so they are plain IEEE double expressions.
//
// The translation unit that includes this header MUST be compiled with
// -ffp-contract=off (hipcc defaults to `fast`), otherwise the compiler would fuse the
// non-fused expressions and break bit-exactness.  `tests/test_cpu_library.py`
// (test_glibc_port_matches_box_libm) checks these functions against the box's real libm
// (host build), `tests/test_gpu_parity.py` (test_device_libm_bit_exact) does the same for the
// device build, and `fgx_create` repeats a 7 168-point comparison in every process.
//
// Licence note: these routines restate glibc's algorithms and carry its constants (glibc is
// LGPL-2.1-or-later; Szabolcs Nagy's exp / log come from ARM optimized-routines, MIT OR
// Apache-2.0 WITH LLVM-exception; log1p / expm1 are fdlibm-derived, Sun Microsystems permissive
// notice).  Treat this file and glibc_tables.h as derived from those sources.
#pragma once
#include <stdint.h>
#include <string.h>
#include "glibc_tables.h"

#if defined(__HIPCC__)
#define FGX_HD __host__ __device__ __forceinline__
#define FGX_TABLE_QUAL __device__ __constant__
#else
#define FGX_HD inline
#endif

namespace fgx {

// Tables live in a struct so host and device code can index the same symbols.  On the
// device they sit in constant memory (4 KB total, L1/scalar-cache resident).
struct GlibcTables {
  uint64_t exp_tab[256];
  uint64_t log_tab[256];
};

#if defined(__HIPCC__)
__device__ __constant__ static const GlibcTables g_glibc_dev = {FGX_EXP_TAB_INIT, FGX_LOG_TAB_INIT};
#endif
static const GlibcTables g_glibc_host = {FGX_EXP_TAB_INIT, FGX_LOG_TAB_INIT};

FGX_HD const GlibcTables& glibc_tables() {
#if defined(__HIP_DEVICE_COMPILE__)
  return g_glibc_dev;
#else
  return g_glibc_host;
#endif
}

FGX_HD double fgx_asdouble(uint64_t u) {
  double d;
#if defined(__HIP_DEVICE_COMPILE__)
  d = __longlong_as_double((long long)u);
#else
  memcpy(&d, &u, 8);
#endif
  return d;
}
FGX_HD uint64_t fgx_asuint64(double d) {
  uint64_t u;
#if defined(__HIP_DEVICE_COMPILE__)
  u = (uint64_t)__double_as_longlong(d);
#else
  memcpy(&u, &d, 8);
#endif
  return u;
}
FGX_HD double fgx_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---------------------------------------------------------------------------------------
// exp  (glibc e_exp.c, `__exp_fma` variant)
// ---------------------------------------------------------------------------------------
FGX_HD double g_exp_special(double tmp, uint64_t sbits, uint64_t ki) {
  double scale, y;
  if ((ki & 0x80000000ULL) == 0) {
    // k > 0: exponent of scale may have overflowed by <= 460.
    sbits -= 1009ULL << 52;
    scale = fgx_asdouble(sbits);
    y = 0x1p1009 * fgx_fma(scale, tmp, scale);
    return y;
  }
  // k < 0: take care in the subnormal range.
  sbits += 1022ULL << 52;
  scale = fgx_asdouble(sbits);
  double st = scale * tmp;  // NOT fused in the compiled code (value is reused for lo)
  y = scale + st;
  if (y < 1.0) {
    double hi, lo;
    lo = scale - y + st;
    hi = 1.0 + y;
    lo = 1.0 - hi + y + lo;
    y = (hi + lo) - 1.0;
    if (y == 0.0) y = 0.0;  // avoid -0.0
  }
  return 0x1p-1022 * y;
}

FGX_HD double g_exp(double x) {
  const GlibcTables& T = glibc_tables();
  const double InvLn2N = fgx_asdouble(FGX_EXP_INVLN2N_BITS);
  const double Shift = fgx_asdouble(FGX_EXP_SHIFT_BITS);
  const double NegLn2hiN = fgx_asdouble(FGX_EXP_NEGLN2HIN_BITS);
  const double NegLn2loN = fgx_asdouble(FGX_EXP_NEGLN2LON_BITS);
  const double C2 = fgx_asdouble(FGX_EXP_C2_BITS), C3 = fgx_asdouble(FGX_EXP_C3_BITS);
  const double C4 = fgx_asdouble(FGX_EXP_C4_BITS), C5 = fgx_asdouble(FGX_EXP_C5_BITS);

  uint64_t ix = fgx_asuint64(x);
  uint32_t abstop = (uint32_t)(ix >> 52) & 0x7ff;
  if (abstop - 0x3c9u >= 0x3fu) {  // top12(512.0) - top12(0x1p-54) = 0x408 - 0x3c9
    if (abstop - 0x3c9u >= 0x80000000u) return 1.0 + x;  // tiny x
    if (abstop >= 0x409u) {                               // |x| >= 1024 or inf/nan
      if (ix == 0xfff0000000000000ULL) return 0.0;
      if (abstop >= 0x7ffu) return 1.0 + x;
      if (ix >> 63) return 0.0;                 // __math_uflow(0)
      return fgx_asdouble(0x7ff0000000000000ULL);  // __math_oflow(0)
    }
    abstop = 0;  // large x: handled by g_exp_special below
  }
  double kd = fgx_fma(x, InvLn2N, Shift);  // z + Shift, fused
  uint64_t ki = fgx_asuint64(kd);
  kd -= Shift;
  double r = fgx_fma(kd, NegLn2hiN, x);
  r = fgx_fma(kd, NegLn2loN, r);
  uint64_t idx = 2 * (ki % 128);
  uint64_t top = ki << (52 - 7);
  double tail = fgx_asdouble(T.exp_tab[idx]);
  uint64_t sbits = T.exp_tab[idx + 1] + top;
  double r2 = r * r;
  double p23 = fgx_fma(r, C3, C2);
  double tr = r + tail;
  double p45 = fgx_fma(r, C5, C4);
  double a = fgx_fma(p23, r2, tr);
  double r4 = r2 * r2;
  double tmp = fgx_fma(r4, p45, a);
  if (abstop == 0) return g_exp_special(tmp, sbits, ki);
  double scale = fgx_asdouble(sbits);
  return fgx_fma(scale, tmp, scale);
}

// ---------------------------------------------------------------------------------------
// log  (glibc e_log.c, `__log_fma` variant)
// ---------------------------------------------------------------------------------------
FGX_HD double g_log(double x) {
  const GlibcTables& T = glibc_tables();
  const double Ln2hi = fgx_asdouble(FGX_LOG_LN2HI_BITS), Ln2lo = fgx_asdouble(FGX_LOG_LN2LO_BITS);
  uint64_t ix = fgx_asuint64(x);
  uint32_t top = (uint32_t)(ix >> 48);
  const uint64_t LO = 0x3fee000000000000ULL;  // asuint64(1.0 - 0x1p-4)
  const uint64_t HI = 0x3ff1090000000000ULL;  // asuint64(1.0 + 0x1.09p-4)
  if (ix - LO < HI - LO) {
    if (ix == 0x3ff0000000000000ULL) return 0.0;
    const double B0 = fgx_asdouble(FGX_LOG_B0_BITS), B1 = fgx_asdouble(FGX_LOG_B1_BITS),
                 B2 = fgx_asdouble(FGX_LOG_B2_BITS), B3 = fgx_asdouble(FGX_LOG_B3_BITS),
                 B4 = fgx_asdouble(FGX_LOG_B4_BITS), B5 = fgx_asdouble(FGX_LOG_B5_BITS),
                 B6 = fgx_asdouble(FGX_LOG_B6_BITS), B7 = fgx_asdouble(FGX_LOG_B7_BITS),
                 B8 = fgx_asdouble(FGX_LOG_B8_BITS), B9 = fgx_asdouble(FGX_LOG_B9_BITS),
                 B10 = fgx_asdouble(FGX_LOG_B10_BITS);
    double r = x - 1.0;
    double r2 = r * r;
    double r3 = r * r2;
    double a = fgx_fma(r, B2, B1);
    a = fgx_fma(r2, B3, a);  // B1 + r*B2 + r2*B3
    double b = fgx_fma(r, B5, B4);
    b = fgx_fma(r2, B6, b);  // B4 + r*B5 + r2*B6
    double c = fgx_fma(r, B8, B7);
    c = fgx_fma(r2, B9, c);
    c = fgx_fma(r3, B10, c);  // B7 + r*B8 + r2*B9 + r3*B10
    double in = fgx_fma(c, r3, b);
    in = fgx_fma(in, r3, a);
    // rhi = r + w - w with w = r*2^27, both steps fused by GCC
    double t = fgx_fma(r, 0x1p27, r);
    double rhi = fgx_fma(-0x1p27, r, t);
    double rlo = r - rhi;
    double rhi2 = rhi * rhi;
    double hi = fgx_fma(rhi2, B0, r);
    double lo = fgx_fma(rhi2, B0, r - hi);
    lo = fgx_fma(B0 * rlo, rhi + r, lo);
    double y = fgx_fma(in, r3, lo);
    return hi + y;
  }
  if (top - 0x0010u >= 0x7ff0u - 0x0010u) {
    if (ix * 2 == 0) return -fgx_asdouble(0x7ff0000000000000ULL);  // log(0) = -inf
    if (ix == 0x7ff0000000000000ULL) return x;
    if ((top & 0x8000u) || (top & 0x7ff0u) == 0x7ff0u) return (x - x) / (x - x);  // NaN
    ix = fgx_asuint64(x * 0x1p52);
    ix -= 52ULL << 52;
  }
  const double A0 = fgx_asdouble(FGX_LOG_A0_BITS), A1 = fgx_asdouble(FGX_LOG_A1_BITS),
               A2 = fgx_asdouble(FGX_LOG_A2_BITS), A3 = fgx_asdouble(FGX_LOG_A3_BITS),
               A4 = fgx_asdouble(FGX_LOG_A4_BITS);
  const uint64_t OFF = 0x3fe6000000000000ULL;
  uint64_t tmp = ix - OFF;
  int i = (int)((tmp >> (52 - 7)) % 128);
  int k = (int)((int64_t)tmp >> 52);
  uint64_t iz = ix - (tmp & (0xfffULL << 52));
  double invc = fgx_asdouble(T.log_tab[2 * i]);
  double logc = fgx_asdouble(T.log_tab[2 * i + 1]);
  double z = fgx_asdouble(iz);
  double r = fgx_fma(z, invc, -1.0);
  double kd = (double)k;
  double w = fgx_fma(kd, Ln2hi, logc);
  double hi = w + r;
  double lo = fgx_fma(kd, Ln2lo, (w - hi) + r);
  double r2 = r * r;
  double p = fgx_fma(r, A2, A1);
  double q = fgx_fma(r, A4, A3);
  double r3 = r * r2;
  double lo2 = fgx_fma(r2, A0, lo);
  double s = fgx_fma(q, r2, p);
  double y = fgx_fma(r3, s, lo2);
  return y + hi;
}

// ---------------------------------------------------------------------------------------
// log1p  (glibc s_log1p.c, fdlibm-derived; no FMA variant in 2.35)
// ---------------------------------------------------------------------------------------
FGX_HD double g_log1p(double x) {
  const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               two54 = 1.80143985094819840000e+16;
  const double Lp1 = 6.666666666666735130e-01, Lp2 = 3.999999999940941908e-01,
               Lp3 = 2.857142874366239149e-01, Lp4 = 2.222219843214978396e-01,
               Lp5 = 1.818357216161805012e-01, Lp6 = 1.531383769920937332e-01,
               Lp7 = 1.479819860511658591e-01;
  double hfsq, f = 0.0, c = 0.0, s, z, R, u, z2, z4, z6, R1, R2, R3, R4;
  int32_t k, hx, hu = 0, ax;
  hx = (int32_t)(fgx_asuint64(x) >> 32);
  ax = hx & 0x7fffffff;
  k = 1;
  if (hx < 0x3FDA827A) {          // x < 0.41422
    if (ax >= 0x3ff00000) {       // x <= -1.0
      if (x == -1.0) return -two54 / 0.0 * 1.0;
      return (x - x) / (x - x);
    }
    if (ax < 0x3e200000) {        // |x| < 2**-29
      if (ax < 0x3c900000) return x;
      return x - x * x * 0.5;
    }
    if (hx > 0 || hx <= (int32_t)0xbfd2bec3) {
      k = 0; f = x; hu = 1;       // -0.2929 < x < 0.41422
    }
  } else if (hx >= 0x7ff00000) {
    return x + x;
  }
  if (k != 0) {
    if (hx < 0x43400000) {
      u = 1.0 + x;
      hu = (int32_t)(fgx_asuint64(u) >> 32);
      k = (hu >> 20) - 1023;
      c = (k > 0) ? 1.0 - (u - x) : x - (u - 1.0);
      c /= u;
    } else {
      u = x;
      hu = (int32_t)(fgx_asuint64(u) >> 32);
      k = (hu >> 20) - 1023;
      c = 0;
    }
    hu &= 0x000fffff;
    uint64_t ub = fgx_asuint64(u);
    if (hu < 0x6a09e) {
      ub = (ub & 0xffffffffULL) | ((uint64_t)(uint32_t)(hu | 0x3ff00000) << 32);
    } else {
      k += 1;
      ub = (ub & 0xffffffffULL) | ((uint64_t)(uint32_t)(hu | 0x3fe00000) << 32);
      hu = (0x00100000 - hu) >> 2;
    }
    u = fgx_asdouble(ub);
    f = u - 1.0;
  }
  hfsq = 0.5 * f * f;
  if (hu == 0) {                  // |f| < 2**-20
    if (f == 0.0) {
      if (k == 0) return 0.0;
      c += k * ln2_lo;
      return k * ln2_hi + c;
    }
    R = hfsq * (1.0 - 0.66666666666666666 * f);
    if (k == 0) return f - R;
    return k * ln2_hi - ((R - (k * ln2_lo + c)) - f);
  }
  s = f / (2.0 + f);
  z = s * s;
  R1 = z * Lp1; z2 = z * z;
  R2 = Lp2 + z * Lp3; z4 = z2 * z2;
  R3 = Lp4 + z * Lp5; z6 = z4 * z2;
  R4 = Lp6 + z * Lp7;
  R = R1 + z2 * R2 + z4 * R3 + z6 * R4;
  if (k == 0) return f - (hfsq - s * (hfsq + R));
  return k * ln2_hi - ((hfsq - (s * (hfsq + R) + (k * ln2_lo + c))) - f);
}

// ---------------------------------------------------------------------------------------
// expm1  (glibc s_expm1.c, fdlibm-derived; no FMA variant in 2.35)
// ---------------------------------------------------------------------------------------
FGX_HD double g_expm1(double x) {
  const double one = 1.0, huge = 1.0e+300, tiny = 1.0e-300,
               o_threshold = 7.09782712893383973096e+02,
               ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double Q1 = -3.33333333333331316428e-02, Q2 = 1.58730158725481460165e-03,
               Q3 = -7.93650757867487942473e-05, Q4 = 4.00821782732936239552e-06,
               Q5 = -2.01099218183624371326e-07;
  double y, hi, lo, c = 0.0, t, e, hxs, hfx, r1, h2, h4, R1, R2, R3;
  int32_t k, xsb;
  uint32_t hx = (uint32_t)(fgx_asuint64(x) >> 32);
  xsb = (int32_t)(hx & 0x80000000u);
  hx &= 0x7fffffffu;
  if (hx >= 0x4043687Au) {          // |x| >= 56*ln2
    if (hx >= 0x40862E42u) {        // |x| >= 709.78
      if (hx >= 0x7ff00000u) {
        uint32_t low = (uint32_t)fgx_asuint64(x);
        if (((hx & 0xfffffu) | low) != 0) return x + x;
        return (xsb == 0) ? x : -1.0;
      }
      if (x > o_threshold) return huge * huge;
    }
    if (xsb != 0) return tiny - one;
  }
  if (hx > 0x3fd62e42u) {           // |x| > 0.5 ln2
    if (hx < 0x3FF0A2B2u) {         // |x| < 1.5 ln2
      if (xsb == 0) { hi = x - ln2_hi; lo = ln2_lo; k = 1; }
      else          { hi = x + ln2_hi; lo = -ln2_lo; k = -1; }
    } else {
      k = (int32_t)(invln2 * x + ((xsb == 0) ? 0.5 : -0.5));
      t = k;
      hi = x - t * ln2_hi;
      lo = t * ln2_lo;
    }
    x = hi - lo;
    c = (hi - x) - lo;
  } else if (hx < 0x3c900000u) {    // |x| < 2**-54
    t = huge + x;
    return x - (t - (huge + x));
  } else {
    k = 0;
  }
  hfx = 0.5 * x;
  hxs = x * hfx;
  R1 = one + hxs * Q1; h2 = hxs * hxs;
  R2 = Q2 + hxs * Q3; h4 = h2 * h2;
  R3 = Q4 + hxs * Q5;
  r1 = R1 + h2 * R2 + h4 * R3;
  t = 3.0 - r1 * hfx;
  e = hxs * ((r1 - t) / (6.0 - x * t));
  if (k == 0) return x - (x * e - hxs);
  e = (x * (e - c) - c);
  e -= hxs;
  if (k == -1) return 0.5 * (x - e) - 0.5;
  if (k == 1) {
    if (x < -0.25) return -2.0 * (e - (x + 0.5));
    return one + 2.0 * (x - e);
  }
  if (k <= -2 || k > 56) {
    y = one - (e - x);
    uint64_t yb = fgx_asuint64(y);
    yb += (uint64_t)((uint32_t)k << 20) << 32;
    y = fgx_asdouble(yb);
    return y - one;
  }
  if (k < 20) {
    t = fgx_asdouble((uint64_t)(uint32_t)(0x3ff00000 - (0x200000 >> k)) << 32);
    y = t - (e - x);
    uint64_t yb = fgx_asuint64(y);
    yb += (uint64_t)((uint32_t)k << 20) << 32;
    y = fgx_asdouble(yb);
  } else {
    t = fgx_asdouble((uint64_t)((uint32_t)(0x3ff - k) << 20) << 32);
    y = x - (e + t);
    y += one;
    uint64_t yb = fgx_asuint64(y);
    yb += (uint64_t)((uint32_t)k << 20) << 32;
    y = fgx_asdouble(yb);
  }
  return y;
}

}  // namespace fgx
