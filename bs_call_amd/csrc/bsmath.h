/*
 * bsmath.h — fixed-operation-order FP64 log / exp / log-factorial.
 *
 * Why this exists: bs_call's per-site model (reference src/genotype_model.c:165-245,
 * src/stats_utils.c:25-91) calls libm log()/exp()/lgamma().  glibc's versions are table
 * driven, ifunc-selected and not reproducible on a GPU.  Every function in this header is
 * built ONLY from IEEE-754 correctly rounded primitives (+ - * / fma, int<->double
 * conversion, bit moves) in a fixed order, so the gfx950 kernel and a host C build
 * produce bit-identical results.  Accuracy vs. a correctly rounded result is < 1 ulp
 * (tests/test_bsmath.py measures it against libm).
 *
 * Build rules: compile with -ffp-contract=off on both sides; every fused operation is an
 * explicit BSM_FMA().  Constants come from tools/gen_bsmath_consts.py.
 *
 * This header is product code (it is compiled into the HIP kernels).  oracle/ may include it
 * (for its "bsm" flavour); nothing here includes anything from oracle/.
 */
#ifndef BSCALL_AMD_BSMATH_H
#define BSCALL_AMD_BSMATH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define BSM_FN __host__ __device__ static __forceinline__
#else
#define BSM_FN static inline __attribute__((always_inline))
#endif

#define BSM_FMA(a, b, c) __builtin_fma((a), (b), (c))

BSM_FN uint64_t bsm_bits(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return u;
}
BSM_FN double bsm_from_bits(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, 8);
  return x;
}

#define BSM_LN2_HI 0x1.62e4200000000p-1  /* top 32 bits of ln 2: k*LN2_HI is exact */
#define BSM_LN2_LO 0x1.fdf473de6af28p-22 /* ln 2 - LN2_HI */
#define BSM_INV_LN2 0x1.71547652b82fep+0
#define BSM_LN10 2.30258509299404568402 /* same literal as the reference's LOG10 macro (include/bs_call.h:36) */

/*
 * Natural log.  x = 2^k * m, m in [sqrt(2)/2, sqrt(2)); f = m-1; s = f/(2+f); z = s*s
 *   log(m) = f - f^2/2 + s*(f^2/2 + R(z)),  R(z) = sum_{j=1..10} 2/(2j+1) z^j   (atanh series)
 * One IEEE division, 10 FMAs for R.  Truncation error of R < 2e-18.
 */
BSM_FN double bsm_log(double x) {
  uint64_t ix = bsm_bits(x);
  int k = 0;
  if ((ix << 1) == 0) return -__builtin_inf();              /* log(+-0) = -inf */
  if ((int64_t)ix < 0) return __builtin_nan("");            /* log(<0)  = nan  */
  if ((ix >> 52) == 0x7ff) return x;                        /* +inf, nan        */
  if ((ix >> 52) == 0) {                                    /* subnormal: scale by 2^54 (exact) */
    x *= 0x1p54;
    ix = bsm_bits(x);
    k = -54;
  }
  uint32_t hx = (uint32_t)(ix >> 32);
  hx += 0x3ff00000u - 0x3fe6a09eu;
  k += (int)(hx >> 20) - 0x3ff;
  hx = (hx & 0x000fffffu) + 0x3fe6a09eu;
  double m = bsm_from_bits(((uint64_t)hx << 32) | (ix & 0xffffffffull));
  double f = m - 1.0; /* exact */
  double s = f / (2.0 + f);
  double z = s * s;
  double R = 0x1.8618618618618p-4;            /* 2/21 */
  R = BSM_FMA(R, z, 0x1.af286bca1af28p-4);    /* 2/19 */
  R = BSM_FMA(R, z, 0x1.e1e1e1e1e1e1ep-4);    /* 2/17 */
  R = BSM_FMA(R, z, 0x1.1111111111111p-3);    /* 2/15 */
  R = BSM_FMA(R, z, 0x1.3b13b13b13b14p-3);    /* 2/13 */
  R = BSM_FMA(R, z, 0x1.745d1745d1746p-3);    /* 2/11 */
  R = BSM_FMA(R, z, 0x1.c71c71c71c71cp-3);    /* 2/9  */
  R = BSM_FMA(R, z, 0x1.2492492492492p-2);    /* 2/7  */
  R = BSM_FMA(R, z, 0x1.999999999999ap-2);    /* 2/5  */
  R = BSM_FMA(R, z, 0x1.5555555555555p-1);    /* 2/3  */
  R = R * z;
  double hfsq = (0.5 * f) * f;
  double dk = (double)k;
  double t = BSM_FMA(dk, BSM_LN2_LO, s * (hfsq + R));
  return dk * BSM_LN2_HI - ((hfsq - t) - f);
}

/*
 * exp.  k = round(x/ln2); r = x - k*ln2 (two-step, |r| <= 0.3466); exp(r) = 1 + (r + r^2 q(r)),
 * q = Taylor 1/2 .. r^11/13! (truncation < 5e-18); result scaled by 2^k with exact power-of-two
 * multiplies (two steps near the ends of the exponent range so the only rounding there is the
 * final one).
 */
BSM_FN double bsm_exp(double x) {
  if (x != x) return x;
  if (x > 0x1.62e42fefa39efp+9) return __builtin_inf(); /* > 709.78: overflow */
  if (x < -0x1.74910d52d3051p+9) return 0.0;            /* < -745.13: underflow to +0 */
  double t = BSM_FMA(x, BSM_INV_LN2, 0x1.8p52);
  int k = (int)(uint32_t)bsm_bits(t); /* low word of the shifted value = round(x/ln2), two's complement */
  double kd = t - 0x1.8p52;
  double r = BSM_FMA(kd, -BSM_LN2_HI, x);
  r = BSM_FMA(kd, -BSM_LN2_LO, r);
  double q = 0x1.6124613a86d09p-33;           /* 1/13! */
  q = BSM_FMA(q, r, 0x1.1eed8eff8d898p-29);   /* 1/12! */
  q = BSM_FMA(q, r, 0x1.ae64567f544e4p-26);   /* 1/11! */
  q = BSM_FMA(q, r, 0x1.27e4fb7789f5cp-22);   /* 1/10! */
  q = BSM_FMA(q, r, 0x1.71de3a556c734p-19);   /* 1/9!  */
  q = BSM_FMA(q, r, 0x1.a01a01a01a01ap-16);   /* 1/8!  */
  q = BSM_FMA(q, r, 0x1.a01a01a01a01ap-13);   /* 1/7!  */
  q = BSM_FMA(q, r, 0x1.6c16c16c16c17p-10);   /* 1/6!  */
  q = BSM_FMA(q, r, 0x1.1111111111111p-7);    /* 1/5!  */
  q = BSM_FMA(q, r, 0x1.5555555555555p-5);    /* 1/4!  */
  q = BSM_FMA(q, r, 0x1.5555555555555p-3);    /* 1/3!  */
  q = BSM_FMA(q, r, 0x1.0000000000000p-1);    /* 1/2!  */
  double p = 1.0 + BSM_FMA(r * r, q, r);      /* in [0.70, 1.42] */
  if (k < -1021) { /* subnormal or near-subnormal result: 2^(k+1000) is normal, 2^-1000 rounds once */
    p *= bsm_from_bits((uint64_t)(k + 1000 + 1023) << 52);
    return p * 0x1p-1000;
  }
  if (k > 1023) { /* k == 1024 only (x just below the overflow threshold) */
    p *= 0x1p1023;
    return p * 2.0;
  }
  return p * bsm_from_bits((uint64_t)(k + 1023) << 52);
}

/*
 * log(n!) for n >= 256, i.e. the lgamma(n+1) branch of the reference's lfact2 macro
 * (include/bs_call.h:335).  Stirling with z = n+1 >= 257:
 *   (z-1/2) ln z - z + ln(2 pi)/2 + 1/(12 z) - 1/(360 z^3) + 1/(1260 z^5)
 * the next term is < 1e-21.
 */
BSM_FN double bsm_lfact_big(int n) {
  double z = (double)n + 1.0;
  double lz = bsm_log(z);
  double zi = 1.0 / z;
  double zi2 = zi * zi;
  double c = 0x1.a01a01a01a01ap-11;            /* 1/1260 */
  c = BSM_FMA(c, zi2, -0x1.6c16c16c16c17p-9);  /* -1/360 */
  c = BSM_FMA(c, zi2, 0x1.5555555555555p-4);   /* 1/12 */
  c = c * zi;
  double a = BSM_FMA(z - 0.5, lz, -z);
  return (a + 0x1.d67f1c864beb5p-1) + c;
}

#endif /* BSCALL_AMD_BSMATH_H */
