/*
 * callmath.h — device forms of bsmath.h and the small numeric helpers shared by the calling kernel (kernels.hip) and
 * the fused chain kernel (fused.hip): same operations in the same order as bsmath.h, hence the same bits.
 */
#ifndef BSCALL_AMD_CALLMATH_H
#define BSCALL_AMD_CALLMATH_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bsmath.h"

#ifndef BSC_DMA_AUX
#define BSC_DMA_AUX 2 /* cache policy bits of the LDS-DMA loads: nt — every pile-up is read exactly once */
#endif
#define IN_DW 26  /* dwords per pileup  (104 B) */
#define OUT_DW 50 /* dwords per gt_meth (200 B) */
#define MAX_OUT_DW 52 /* gt_vcf stride (208 B) */
#define SLOT_DW (64 * IN_DW) /* per-wave LDS slot: 64 pile-ups = 6 656 B >= 32 results (6 400 / 6 656 B) */

/* LDS accesses that hand data from one lane to another inside a wave: the wave runs in lockstep and its LDS operations
 * execute in order, so only the COMPILER has to be told — a wavefront-scope fence orders the memory operations, the
 * (free) s_wave_barrier keeps the scheduler from moving anything across. */
#define WAVE_LDS_ORDER()                                     \
  do {                                                       \
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   \
    __builtin_amdgcn_wave_barrier();                         \
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");   \
  } while (0)

/* number of set bits of a wave-wide mask below the calling lane (v_mbcnt_lo / _hi: two instructions, against the seven of
 * popcount(m & ((1 << lane) - 1)) in 64-bit arithmetic) */
__device__ static __forceinline__ unsigned lane_rank(unsigned long long m) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

/* ---- device forms of bsmath.h with wave-uniform branches only -------------------------------------------
 * Same operations in the same order as bsm_log_t / bsm_exp_t, hence the same bits, but without per-lane
 * branches: the table path of log() runs for every lane, its near-1 polynomial only when some lane of the wave
 * needs it, and the rare special cases are left to a wave-uniform fallback onto the full functions.  64-bit integer steps are done on the high
 * word where the constants' low words are zero. */

/* log(x) for positive, normal, finite x: the table path (x = 2^k z, z in [0x1.6p-1, 0x1.6p0)) */
/*
 * a * ks + kv / a * kv + ks with both other operands compile-time constants, one kept in an SGPR pair and one in a
 * VGPR pair (a VALU instruction reads at most one SGPR operand).  Written as the three-address v_fma_f64 by hand:
 * for a constant addend the compiler emits "v_mov_b64 tmp, kv; v_fmac_f64 tmp, ks, a", i.e. one more VALU issue per
 * polynomial step of every log and exp (4 % of the kernel's instructions).
 */
__device__ static __forceinline__ double fma_sv(double a, double ks, double kv) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(ks), "v"(kv));
  return d;
}
__device__ static __forceinline__ double fma_vs(double a, double kv, double ks) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(kv), "s"(ks));
  return d;
}

/* max(a, b) as the one v_max_f64 it is (fmax() puts a canonicalising v_max_f64 x, x in front of every operand) */
__device__ static __forceinline__ double max_raw(double a, double b) {
  double d;
  asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

/* a * ks + a */
__device__ static __forceinline__ double fma_s_self(double a, double ks) {
  double d;
  asm("v_fma_f64 %0, %1, %2, %1" : "=v"(d) : "v"(a), "s"(ks));
  return d;
}

__device__ static __forceinline__ double log_main(double x, const double *tab) {
  const uint64_t ix = bsm_bits(x);
  const uint32_t hx = (uint32_t)(ix >> 32);
  const uint32_t tmp = hx - 0x3fe60000u;
  const uint32_t i = (tmp >> 13) & 127u;
  const int k = (int)tmp >> 20;
  const uint32_t hz = hx - (tmp & 0xfff00000u);
  const double z = bsm_from_bits(((uint64_t)hz << 32) | (uint32_t)ix);
  const double2 ic = *reinterpret_cast<const double2 *>(tab + 2 * i);
  const double r = BSM_FMA(z, ic.x, -1.0);
  const double kd = (double)k;
  const double w = BSM_FMA(kd, BSM_LOG_LN2HI, ic.y);
  const double hi = r + w;
  double lo = BSM_FMA(kd, BSM_LOG_LN2LO, (w - hi) + r);
  const double r2 = r * r;
  const double r3 = r * r2;
  const double q1 = fma_sv(r, BSM_LOG_A2, BSM_LOG_A1);
  const double q3 = fma_sv(r, BSM_LOG_A4, BSM_LOG_A3);
  lo = BSM_FMA(r2, BSM_LOG_A0, lo);
  const double q = BSM_FMA(q3, r2, q1);
  return BSM_FMA(q, r3, lo) + hi;
}

/* log(x) for 1 - 2^-4 <= x < 1 + 0x1.09p-4 (x == 1 gives +0 through the same operations) */
__device__ static __forceinline__ double log_near1(double x) {
  const double s = x - 1.0;
  const double s2 = s * s;
  const double s3 = s * s2;
  double t7 = fma_sv(s, BSM_LOG_B8, BSM_LOG_B7); /* the heads of the three chains and the splitter: three-address, see above */
  t7 = BSM_FMA(s2, BSM_LOG_B9, t7);
  t7 = BSM_FMA(s3, BSM_LOG_B10, t7);
  double t4 = fma_sv(s, BSM_LOG_B5, BSM_LOG_B4);
  t4 = BSM_FMA(s2, BSM_LOG_B6, t4);
  double t1 = fma_sv(s, BSM_LOG_B2, BSM_LOG_B1);
  t1 = BSM_FMA(s2, BSM_LOG_B3, t1);
  const double p = BSM_FMA(BSM_FMA(t7, s3, t4), s3, t1);
  const double a = fma_s_self(s, 0x1p27);
  const double shi = BSM_FMA(-0x1p27, s, a);
  const double slo = s - shi;
  const double shi2 = shi * shi;
  const double nhi = BSM_FMA(shi2, BSM_LOG_B0, s);
  double nlo = BSM_FMA(shi2, BSM_LOG_B0, s - nhi);
  nlo = BSM_FMA(s + shi, slo * BSM_LOG_B0, nlo);
  return nhi + BSM_FMA(p, s3, nlo);
}

/*
 * log(x) on the device: table path for every lane; the near-1 polynomial only when some lane of the wave needs
 * it (wave-uniform branch: of the 12 methylation terms only 4 are near 1 with any frequency, so most
 * evaluations skip it); the full bsm_log_t only when some lane is not positive-normal-finite (never, for valid
 * parameters).
 */
__device__ static __forceinline__ double log_dev(double x, const double *tab) {
  const uint32_t hx = (uint32_t)(bsm_bits(x) >> 32);
  const bool near = hx - 0x3fee0000u < 0x3ff10900u - 0x3fee0000u;
  const bool ok = hx - 0x00100000u < 0x7fe00000u;
  double y;
  if (__all(near)) { /* e.g. the log of the normalising sum of a tile of confident calls: the table path is not wanted at all */
    y = log_near1(x);
  } else {
    y = log_main(x, tab);
    if (__any(near)) {
      const double yn = log_near1(x);
      y = near ? yn : y;
    }
  }
  if (__builtin_expect(__any(!ok), 0)) y = ok ? y : bsm_log_t(x, tab);
  return y;
}

/* exp(x) where x == 0 or 2^-54 <= |x| < 512 (no over/underflow handling, no tiny-x shortcut) */
__device__ static __forceinline__ double exp_mid(double x, const uint64_t *tab) {
  const double kd0 = fma_vs(x, BSM_EXP_INVLN2N, BSM_EXP_SHIFT);
  const uint32_t ki = (uint32_t)bsm_bits(kd0);
  const double kd = kd0 - BSM_EXP_SHIFT;
  const double r = BSM_FMA(kd, BSM_EXP_NEGLN2LON, BSM_FMA(kd, BSM_EXP_NEGLN2HIN, x));
  const ulonglong2 ts = *reinterpret_cast<const ulonglong2 *>(tab + 2u * (ki & 127u));
  const double tail = bsm_from_bits(ts.x);
  const uint64_t sbits = ts.y + ((uint64_t)(ki << 13) << 32); /* + (ki << 45): only the high word changes */
  const double r2 = r * r;
  const double p23 = fma_sv(r, BSM_EXP_C3, BSM_EXP_C2);
  const double p45 = fma_sv(r, BSM_EXP_C5, BSM_EXP_C4);
  const double t = BSM_FMA(p23, r2, tail + r);
  const double tmp = BSM_FMA(r2 * r2, p45, t);
  const double scale = bsm_from_bits(sbits);
  return BSM_FMA(scale, tmp, scale);
}

__device__ static __forceinline__ double exp_dev(double x, const uint64_t *tab) {
  const uint32_t abstop = (uint32_t)(bsm_bits(x) >> 52) & 0x7ffu;
  const bool ok = (abstop - 0x3c9u < 0x408u - 0x3c9u) || x == 0.0;
  double y = exp_mid(x, tab);
  if (__builtin_expect(__any(!ok), 0)) y = ok ? y : bsm_exp_t(x, tab);
  return y;
}

/*
 * One term exp(x), x <= 0, of the normalising sum of calc_gt_prob (src/genotype_model.c:240-243: sum += exp(ll[i] - max),
 * index order, one of the terms being exp(0) = 1 exactly).  Deep coverage puts most arguments below -512, where glibc's
 * exp leaves its main path; running the whole function for the wave whenever one lane is there doubled the cost of the
 * loop at 200x (profiles/r03_cfg4_*).  Instead:
 *   -700 <= x <= -512   glibc's `specialcase` for k < 0 with a result that stays normal (>= 2^-1022, i.e. x > -708.39):
 *                       scale' = scale * 2^1022, y = scale' + RN(scale' * tmp), result 2^-1022 * y — two roundings where
 *                       the main path has one fma, so it is evaluated as glibc does, for the wave, only when some lane needs it;
 *   x < -700            the term is replaced by exp(-700) (the argument is clamped: one v_max_f64): below 2^-1000 either way,
 *                       and no term below 2^-1000 can change the sum.  Proof: before the term 1.0 is added a partial sum
 *                       below 2^-53 is absorbed by it (RN(1 + q) = 1), and a partial sum that is not below 2^-53 contains a
 *                       term >= 2^-57, on whose arrival everything that came before it below 2^-1000 was absorbed
 *                       (q < ulp / 2); from the 1.0 on the partial sum is >= 1 and absorbs every term below 2^-53.  So the
 *                       sum has the same bits whatever stands in for the tiny terms (tests/test_gpu_parity.py at 200x /
 *                       300x / 1400x against the oracle, which evaluates every term with libm).
 * Anything else that is not in exp_mid's range (never: |x| >= 2^-54 or x == 0 for differences of sums of logs, and a
 * difference to the maximum is never positive) takes the full function.  tools/check_exp_far.c: the far form equals libm on
 * 2e8 arguments in [-700, -512].
 */
__device__ static __forceinline__ double exp_term_dev(double x, const uint64_t *tab) {
  double xc; /* max(x, -700) as the one instruction it is (fmax() comes with a canonicalising v_max_f64 x, x in front) */
  asm("v_max_f64 %0, %1, %2" : "=v"(xc) : "v"(x), "s"(-700.0));
  const uint32_t hx = (uint32_t)(bsm_bits(xc) >> 32);
  const bool far = hx >= 0xC0800000u; /* xc <= -512 (a negative double's high word grows with its magnitude) */
  const bool ok = ((hx & 0x7fffffffu) - 0x3c900000u < 0x40900000u - 0x3c900000u) || x == 0.0; /* 2^-54 <= |xc| < 1024, or 0 */
  const double kd0 = fma_vs(xc, BSM_EXP_INVLN2N, BSM_EXP_SHIFT);
  const uint32_t ki = (uint32_t)bsm_bits(kd0);
  const double kd = kd0 - BSM_EXP_SHIFT;
  const double r = BSM_FMA(kd, BSM_EXP_NEGLN2LON, BSM_FMA(kd, BSM_EXP_NEGLN2HIN, xc));
  const ulonglong2 ts = *reinterpret_cast<const ulonglong2 *>(tab + 2u * (ki & 127u));
  const double tail = bsm_from_bits(ts.x);
  const uint64_t sbits = ts.y + ((uint64_t)(ki << 13) << 32);
  const double r2 = r * r;
  const double p23 = fma_sv(r, BSM_EXP_C3, BSM_EXP_C2);
  const double p45 = fma_sv(r, BSM_EXP_C5, BSM_EXP_C4);
  const double t = BSM_FMA(p23, r2, tail + r);
  const double tmp = BSM_FMA(r2 * r2, p45, t);
  const double scale = bsm_from_bits(sbits);
  double y = BSM_FMA(scale, tmp, scale);
  if (__any(far)) {
    asm volatile(""); /* keeps this a branch: the block is cheap enough for the compiler to run it for every wave otherwise */
    const double sc2 = bsm_from_bits(sbits + (1022ull << 52));
    const double st = sc2 * tmp;
    const double yf = 0x1p-1022 * (sc2 + st);
    y = far ? yf : y;
  }
  if (__builtin_expect(__any(!ok), 0)) y = ok ? y : bsm_exp_t(x, tab);
  return y;
}

/*
 * x / ln(10), correctly rounded (the reference divides by its LOG10 macro, src/genotype_model.c:244).
 * Markstein's theorem: with rc = RN(1/c), q0 = RN(x*rc), r = x - c*q0 (exact in an fma) the value
 * RN(q0 + r*rc) is the correctly rounded quotient, barring underflow in r.  tools/check_div.c compares this
 * against true division on 2e9 arguments: the only differences are x = -0 and |x| < 2^-1000, which take
 * the true division below (neither occurs for (ll - max) - log(sum)).
 */
#define BSC_RLN10 0x1.bcb7b1526e50dp-2 /* RN(1 / 2.30258509299404568402) */
__device__ static __forceinline__ double div_ln10_dev(double x) {
  const uint32_t ax = (uint32_t)(bsm_bits(x) >> 32) & 0x7fffffffu;
  const bool ok = (ax - 0x01700000u < 0x7ff00000u - 0x01700000u) || bsm_bits(x) == 0; /* 2^-1000 <= |x| < inf, or +0 */
  const double q0 = x * BSC_RLN10;
  const double r = BSM_FMA(-BSM_LN10, q0, x);
  double q = BSM_FMA(r, BSC_RLN10, q0);
  if (__builtin_expect(__any(!ok), 0)) q = ok ? q : x / BSM_LN10;
  return q;
}

/*
 * get_Z (src/genotype_model.c:23-42).  The three quotients share the divisor d = (x1 + x2)(l - t): one true
 * division gives y = RN(1/d), then each quotient is a Markstein step (q0 = RN(n*y), r = n - d*q0 exact in an
 * fma, q = RN(q0 + r*y)), which is the correctly rounded n/d when y is the correctly rounded reciprocal, except for a
 * divisor whose significand is all ones — those lanes take the true divisions (wave-uniform fallback).  No
 * over/underflow can interfere: bsc_create bounds l - t to [2^-20, 1], counts are < 2^33, so |n|, d, q are in
 * [2^-60, 2^70].  tools/check_div_getz.c: 2.7e9 quotients over the model's operand ranges, all equal to IEEE
 * division.  An empty class pair (x1 + x2 == 0) gives inf/nan here; the caller never uses that result.
 */
__device__ static __forceinline__ void get_Z(double x1, double x2, double k1, double k2, double l, double t, double &Z0,
                                             double &Z1, double &Z2) {
  const double lpt = l + t;
  const double lmt = l - t;
  const double d = (x1 + x2) * lmt;
  const double a2 = 2.0 - lpt;
  const double n0 = x1 * (lpt + 2.0 * k2) - x2 * (a2 + 2.0 * k1);
  const double n1 = x1 * (2.0 + lpt + 4.0 * k2) - x2 * (a2 + 4.0 * k1);
  const double n2 = x1 * (lpt + 4.0 * k2) - x2 * (a2 + 4.0 * k1);
  const double y = 1.0 / d;
  double s0 = n0 * y, s1 = n1 * y, s2 = n2 * y;
  s0 = BSM_FMA(BSM_FMA(-d, s0, n0), y, s0);
  s1 = BSM_FMA(BSM_FMA(-d, s1, n1), y, s1);
  s2 = BSM_FMA(BSM_FMA(-d, s2, n2), y, s2);
  const uint64_t db = bsm_bits(d);
  const bool allones = ((uint32_t)db & ((uint32_t)(db >> 32) | 0xfff00000u)) == 0xffffffffu;
  if (__builtin_expect(__any(allones), 0)) {
    s0 = allones ? n0 / d : s0;
    s1 = allones ? n1 / d : s1;
    s2 = allones ? n2 / d : s2;
  }
  s0 = s0 < -1.0 ? -1.0 : (s0 > 1.0 ? 1.0 : s0);
  Z0 = 0.5 * (lmt * s0 + 2.0 - lpt);
  s1 = s1 < -1.0 ? -1.0 : (s1 > 1.0 ? 1.0 : s1);
  Z1 = 0.5 * (lmt * s1 + 2.0 - lpt);
  s2 = s2 < -1.0 ? -1.0 : (s2 > 1.0 ? 1.0 : s2);
  Z2 = 0.5 * (lmt * s2 + 2.0 - lpt);
}

/*
 * Logs that need no evaluation per site.  When one class of a strand's pair is empty — no C among the C2T reads of a T site,
 * every C of a site converted, a G2A strand seen only as A — get_Z's three quotients are beyond +-1 by a wide margin
 * ((3 - l - t) / (l - t) >= 2 and the like, whatever the counts), the clamp makes them exactly -1 or +1, the three Z equal
 *   Zc = 0.5 * (lmt * -1.0 + 2.0 - lpt)   (the pair's first class empty)   or   Zd = 0.5 * (lmt * 1.0 + 2.0 - lpt)   (its second),
 * and the class's three log arguments depend on nothing but its quality index q (through k = q_prob[q].k):
 *   PT_A  log(1.0 - Zc + k)        class 7: CC        class 4: GG
 *   PT_B  log(1.0 - 0.5 * Zc + k)  class 7: CT        class 4: AG
 *   PT_C  log(0.5 * (1.0 - Zc) + k)  class 7: AC, CG  class 4: CG, GT
 *   PT_D  log(Zd + k)              class 5: CC        class 6: GG
 *   PT_E  log(0.5 * Zd + k)        class 5: CT, AC, CG  class 6: AG, CG, GT
 * 5 x 44 doubles, filled once per workgroup by the same expressions and the same log as the per-site path (pure_log_entry),
 * so a tabulated value is the value the site would have computed.  Such classes are half of all (site, class) pairs of WGBS
 * data at 30x; leaving them out of the log rounds halves those rounds (call_body.inc).
 */
#define PT_Q 44
#define PT_A 0
#define PT_B 1
#define PT_C 2
#define PT_D 3
#define PT_E 4
#define PT_WORDS (5 * PT_Q)
__device__ static __forceinline__ double pure_log_entry(unsigned idx, double l, double t, const double *s_k, const double *logtab) {
  const unsigned id = idx / PT_Q, q = idx - id * PT_Q;
  const double lpt = l + t, lmt = l - t; /* as get_Z forms them */
  const double Zc = 0.5 * (lmt * -1.0 + 2.0 - lpt), Zd = 0.5 * (lmt * 1.0 + 2.0 - lpt);
  const double k = s_k[q];
  double x;
  switch (id) {
    case PT_A: x = 1.0 - Zc + k; break;
    case PT_B: x = 1.0 - 0.5 * Zc + k; break;
    case PT_C: x = 0.5 * (1.0 - Zc) + k; break;
    case PT_D: x = Zd + k; break;
    default: x = 0.5 * Zd + k; break;
  }
  return bsm_log_t(x, logtab);
}

/* LDS-DMA: 16 bytes per lane, global (per-lane address) -> LDS (wave-uniform base + lane * 16). */
__device__ static __forceinline__ void dma16(const void *g, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, BSC_DMA_AUX);
}

/* lfact2 (include/bs_call.h:335) */
__device__ static __forceinline__ double lfact_dev(int x, const double *lf, const double *logtab) {
  return x < 256 ? lf[x] : bsm_lfact_big_t(x, logtab);
}

/* fisher() (src/stats_utils.c:25-91) */
__device__ static double fisher_dev(int c0, int c1, int c2, int c3, const double *lf, const double *logtab,
                                    const uint64_t *exptab) {
#define LF(x) lfact_dev((x), lf, logtab)
  const int row0 = c0 + c1, row1 = c2 + c3, col0 = c0 + c2, col1 = c1 + c3;
  const int n = row0 + row1;
  if (n == 0) return 1.0;
  const double delta = (double)c0 - (double)(row0 * col0) / (double)n;
  const double knst = LF(col0) + LF(col1) + LF(row0) + LF(row1) - LF(n);
  double l = bsm_exp_t(knst - LF(c0) - LF(c1) - LF(c2) - LF(c3), exptab);
  double p = l;
  if (delta > 0.0) {
    int mn = c1 < c2 ? c1 : c2;
    for (int i = 0; i < mn; i++) {
      l *= (double)((c1 - i) * (c2 - i)) / (double)((c0 + i + 1) * (c3 + i + 1));
      p += l;
    }
    mn = c0 < c3 ? c0 : c3;
    const int k = (int)ceil(2.0 * delta);
    if (k <= mn) {
      c0 -= k; c3 -= k; c1 += k; c2 += k;
      l = bsm_exp_t(knst - LF(c0) - LF(c1) - LF(c2) - LF(c3), exptab);
      p += l;
      for (int i = 0; i < mn - k; i++) {
        l *= (double)((c0 - i) * (c3 - i)) / (double)((c1 + i + 1) * (c2 + i + 1));
        p += l;
      }
    }
  } else {
    int mn = c0 < c3 ? c0 : c3;
    for (int i = 0; i < mn; i++) {
      l *= (double)((c0 - i) * (c3 - i)) / (double)((c1 + i + 1) * (c2 + i + 1));
      p += l;
    }
    mn = c1 < c2 ? c1 : c2;
    int k = (int)ceil(-2.0 * delta);
    if (!k) k = 1;
    if (k <= mn) {
      c0 += k; c3 += k; c1 -= k; c2 -= k;
      l = bsm_exp_t(knst - LF(c0) - LF(c1) - LF(c2) - LF(c3), exptab);
      p += l;
      for (int i = 0; i < mn - k; i++) {
        l *= (double)((c1 - i) * (c2 - i)) / (double)((c0 + i + 1) * (c3 + i + 1));
        p += l;
      }
    }
  }
  return p;
#undef LF
}


/* the 2x2 strand table of a heterozygous call (src/call_genotypes.c:64-100): f[] = counts[0][*] (forward), r[] =
 * counts[1][*] (reverse); {allele 1 fwd, allele 2 fwd, allele 1 rev, allele 2 rev} */
__device__ static __forceinline__ void strand_table(unsigned mxi, const uint32_t f[8], const uint32_t r[8], int &t0, int &t1,
                                                    int &t2, int &t3) {
  t0 = t1 = t2 = t3 = 0;
  switch (mxi) {
    case 1: /* AC */
      t0 = f[0] + f[4]; t1 = f[1] + f[5] + f[7]; t2 = r[0] + r[4]; t3 = r[1] + r[5] + r[7];
      break;
    case 2: /* AG */
      t0 = f[0]; t1 = f[2] + f[6]; t2 = r[0]; t3 = r[2] + r[6];
      break;
    case 3: /* AT */
      t0 = f[0] + f[4]; t1 = f[3] + f[7]; t2 = r[0] + r[4]; t3 = r[3] + r[7];
      break;
    case 5: /* CG */
      t0 = f[1] + f[5] + f[7]; t1 = f[2] + f[4] + f[6]; t2 = r[1] + r[5] + r[7]; t3 = r[2] + r[4] + r[6];
      break;
    case 6: /* CT */
      t0 = f[1] + f[5]; t1 = f[3]; t2 = r[1] + r[5]; t3 = r[3];
      break;
    case 8: /* GT: the reverse row uses the FORWARD class-6 count, as the reference does (:98) */
      t0 = f[2] + f[4] + f[6]; t1 = f[3] + f[7]; t2 = r[2] + r[4] + f[6]; t3 = r[3] + r[7];
      break;
    default:
      break;
  }
}

#endif /* BSCALL_AMD_CALLMATH_H */
