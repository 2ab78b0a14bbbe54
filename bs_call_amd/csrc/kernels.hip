/*
 * kernels.hip — gfx950 (MI355X / CDNA4) kernels of the bs_call per-site calling path.
 *
 *   bsc_call_kernel    pile-up -> gt_meth for a batch of genome positions: the body of the reference's
 *                      calc-thread loop (src/call_genotypes.c:44-60,109-113) and calc_gt_prob()
 *                      (src/genotype_model.c:44-246, get_Z :23-42).  Heterozygous calls are appended to a
 *                      compact list instead of running the divergent Fisher walk in this kernel.
 *   bsc_fisher_kernel  strand table + fisher() (src/call_genotypes.c:61-108, src/stats_utils.c:25-91) over
 *                      the compacted heterozygous sites.
 *   bsc_synth_kernel   synthetic L-pileup generator (synth.h) — bench/test support.
 *
 * Design (DESIGN.md has the numbers): the kernel is a streaming scan, 104 B + 1 B in and 200 B out per
 * site with no reuse, so HBM bandwidth is its roofline; in practice instruction issue (mostly FP64 VALU) is the
 * nearer bound, so the structure below is about issuing fewer instructions as much as about moving bytes well.
 * The reference's records are arrays of structs; a wave reading its 64 structs directly would touch
 * each 128-B line from 2 lanes in 7 separate instructions.  Instead each wave moves its 64 records with
 * fully coalesced 16-B-per-lane transfers through a private LDS slot (see bsc_call_kernel).
 *
 * Numerics: FP64 throughout, no contraction (-ffp-contract=off), the transcendental functions are
 * bsmath.h (bit-exact replicas of glibc's log/exp/lgamma, shared with the host), tables come verbatim from the host.  The order
 * of the additions into ll[g] is the reference's: prior first, then one term per class in class order
 * 0..7; a class with n == 0 contributes +0.0, which leaves ll[g] unchanged exactly as skipping does
 * (no ll[g] can be -0.0: every term is n*ln(...) with n > 0 and no ln() argument path yields -0).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bsmath.h"
#include "devtables.h"
#include "synth.h"

#ifndef TILE
#define TILE 256 /* threads per workgroup of bsc_call_kernel (a multiple of 64) */
#endif
#ifndef BSC_DMA_AUX
#define BSC_DMA_AUX 2 /* cache policy bits of the LDS-DMA loads: nt — every pile-up is read exactly once */
#endif
#ifndef BSC_TILES_PER_WAVE
#define BSC_TILES_PER_WAVE 8 /* launch heuristic: wave-tiles per wave (8 measured best over 1 M .. 50 M positions) */
#endif
#ifndef BSC_WAVES_PER_SIMD
#define BSC_WAVES_PER_SIMD 4 /* occupancy target of bsc_call_kernel: bounds its VGPR budget (512 / waves) */
#endif
#define IN_DW 26  /* dwords per pileup  (104 B) */
#define OUT_DW 50 /* dwords per gt_meth (200 B) */
#define MAX_OUT_DW 52 /* gt_vcf stride (208 B) */
#define SLOT_DW (64 * IN_DW) /* per-wave LDS slot: 64 pile-ups = 6 656 B >= 32 results (6 400 / 6 656 B) */

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
  double t7 = BSM_FMA(s, BSM_LOG_B8, BSM_LOG_B7);
  t7 = BSM_FMA(s2, BSM_LOG_B9, t7);
  t7 = BSM_FMA(s3, BSM_LOG_B10, t7);
  double t4 = BSM_FMA(s, BSM_LOG_B5, BSM_LOG_B4);
  t4 = BSM_FMA(s2, BSM_LOG_B6, t4);
  double t1 = BSM_FMA(s, BSM_LOG_B2, BSM_LOG_B1);
  t1 = BSM_FMA(s2, BSM_LOG_B3, t1);
  const double p = BSM_FMA(BSM_FMA(t7, s3, t4), s3, t1);
  const double a = BSM_FMA(s, 0x1p27, s);
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
  double y = log_main(x, tab);
  if (__any(near)) {
    const double yn = log_near1(x);
    y = near ? yn : y;
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

/* LDS-DMA: 16 bytes per lane, global (per-lane address) -> LDS (wave-uniform base + lane * 16). */
__device__ static __forceinline__ void dma16(const void *g, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, BSC_DMA_AUX);
}

/*
 * The calling kernel.  No workgroup barrier after the table set-up: every wave owns a 6 656-byte LDS slot
 * and walks its own 64-site wave-tiles.
 *   in : the wave-tile's 64 pile-ups are 6 656 contiguous bytes in HBM; 6.5 LDS-DMA instructions (1 KiB
 *        each, no VGPRs) land them in the slot; lane i then reads record i (13 x ds_read_b64, stride 26
 *        dwords: conflict-free).
 *   out: results leave in two halves of 32 records (6 400 contiguous bytes, or 6 656 for gt_vcf stride):
 *        the owning lanes write their record to the slot (25 x ds_write_b64, stride 50 dwords:
 *        conflict-free), then all 64 lanes copy the slot out with 16-byte stores (6.25 KiB-wide
 *        instructions).
 * LDS of one wave is touched only by that wave and a wave's LDS operations execute in order, so the
 * hand-over inside a wave needs no barrier — only the vmcnt wait that retires the DMA.
 */
/* FULL = true: the launch's complete 64-position wave-tiles (wt_begin = 0, wt_end = n_sites / 64) — every lane valid,
 * no guarded paths, and the staging pointers are known to be LDS (ds_write instead of flat stores, which would issue to
 * the LDS and the vector-memory pipe both: profiles/r01_h counted 26 such stores per tile).  FULL = false: the same
 * code with the guards, launched for the last, partial wave-tile only. */
template <bool FULL>
__global__ __launch_bounds__(TILE, BSC_WAVES_PER_SIMD) void bsc_call_kernel_t(const uint32_t *__restrict__ cts,
                                                                      const uint8_t *__restrict__ ref,
                                                                      uint64_t n_sites, uint64_t wt_begin,
                                                                      uint64_t wt_end, uint32_t *__restrict__ out,
                                                                      uint32_t out_dw, uint8_t *__restrict__ skip,
                                                                      const bsc_dev_tables *__restrict__ tb,
                                                                      uint32_t *__restrict__ het_list,
                                                                      unsigned long long *__restrict__ counters) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_slot[TILE / 64][SLOT_DW];
  __shared__ double s_k[44], s_lnk[44], s_half[44], s_one[44];
  __shared__ double s_logtab[256];
  __shared__ unsigned long long s_exptab[256];
  __shared__ unsigned int s_cnt[12]; /* covered, hist[10], het */
  __shared__ uint8_t s_pairs[TILE / 64][256]; /* per wave: the (lane, class) pairs whose logs are needed */

  const unsigned tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const unsigned wid = tid >> 6;
  if (tid < 44) {
    s_k[tid] = tb->k[tid];
    s_lnk[tid] = tb->ln_k[tid];
    s_half[tid] = tb->ln_k_half[tid];
    s_one[tid] = tb->ln_k_one[tid];
  }
  for (unsigned i = tid; i < 256; i += TILE) {
    s_logtab[i] = tb->log_tab[i];
    s_exptab[i] = tb->exp_tab[i];
  }
  if (tid < 12) s_cnt[tid] = 0;
  const double l = 1.0 - tb->under_conv;
  const double t = tb->over_conv;
  const double lrb = tb->lrb, lrb1 = tb->lrb1;
  __syncthreads();

  uint32_t *slot = lds_slot[wid];
  const uint64_t wave_stride = (uint64_t)gridDim.x * (TILE / 64);
  for (uint64_t wt = wt_begin + (uint64_t)blockIdx.x * (TILE / 64) + wid; wt < wt_end; wt += wave_stride) {
    const uint64_t site0 = wt * 64;
    const unsigned nvalid = FULL ? 64u : (unsigned)((n_sites - site0) < 64 ? (n_sites - site0) : 64);
    const bool full = FULL; /* compile-time */
    const uint64_t site = site0 + lane;
    const bool valid = FULL || lane < nvalid;
    const unsigned rf = valid ? ref[site] : 0u;

    /* ---- my record ---- */
    uint32_t w[IN_DW];
    if (full) {
      const char *src = reinterpret_cast<const char *>(cts + site0 * IN_DW) + lane * 16;
#pragma unroll
      for (int j = 0; j < 6; j++) dma16(src + j * 1024, slot + j * 256);
      if (lane < 32) dma16(src + 6 * 1024, slot + 6 * 256);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint2 *rec = reinterpret_cast<const uint2 *>(slot + lane * IN_DW);
#pragma unroll
      for (int i = 0; i < IN_DW / 2; i++) {
        const uint2 v = rec[i];
        w[2 * i] = v.x;
        w[2 * i + 1] = v.y;
      }
    } else { /* last, partial wave-tile of the launch: plain guarded loads (no 16-byte over-read) */
#pragma unroll
      for (int i = 0; i < IN_DW; i++) w[i] = valid ? cts[site * IN_DW + i] : 0u;
    }
    const uint32_t n_reads = w[16];
    const float mapq2 = __uint_as_float(w[25]);
    const bool covered = valid && n_reads != 0;

    /* ---- per-site summary (src/call_genotypes.c:45-59) ----
     * Register diet: the eight class counts stay as u32 (converted to f64 where used), the eight rounded
     * mean qualities are packed one per byte until the result record is written.  Precondition (holds for
     * every pile-up the accumulate stage can produce: base qualities are <= 43, src/input_sam.c:76-86): each
     * class's mean quality is in [0,43].  The reference indexes q_prob[] out of bounds otherwise; here the
     * index is clamped (QI) so that garbage cannot read outside the table. */
    uint32_t cnt[8];
    uint32_t qpack0 = 0, qpack1 = 0; /* qual[0..3], qual[4..7] one byte each (0..255) */
    float tot_qual = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      cnt[j] = w[j] + w[8 + j];
      const float qs = __uint_as_float(w[17 + j]);
      const bool has = cnt[j] != 0;
      const float nn = (float)max(cnt[j], 1u);
      /* The reference promotes the f32 quotient to f64 for the +0.5 and floorf's parameter rounds the sum back to
       * f32.  The f64 sum is exact whenever it matters (|v| >= 2^-30; below that both forms give 0.5), so the detour
       * equals ONE correctly rounded f32 addition: (float)(0.5 + (double)v) == v + 0.5f, bit for bit.  Evaluated for
       * every lane (the empty asm keeps the compiler from wrapping each of the eight divisions in its own exec-mask
       * branch: every class is non-empty somewhere in a 64-site tile, so the branches never skip anything). */
      int q = (int)floorf(qs / nn + 0.5f);
      asm volatile("" : "+v"(q));
      tot_qual += has ? qs : 0.0f;
      const uint32_t qb = has ? ((uint32_t)q & 0xffu) : 0u;
      if (j < 4) qpack0 |= qb << (8 * j);
      else qpack1 |= qb << (8 * (j - 4));
    }
    const float nf = covered ? (float)n_reads : 1.0f;
    const int aq = (int)floorf(tot_qual / nf + 0.5f);
    const int mq = (int)(0.5 + sqrt((double)(mapq2 / nf)));
/* table index of class j: its packed quality, clamped to the table (only garbage input exceeds 43) */
#define QI(j) min((((j) < 4 ? qpack0 : qpack1) >> (8 * ((j)&3))) & 0xffu, 43u)
#define ND(j) ((double)cnt[j])

    /* ---- calc_gt_prob ----
     * The 12 methylation-dependent log() arguments of classes 4..7 are parked in this lane's own 104 bytes of
     * the slot (13 doubles; its pile-up record is in registers by now) and evaluated by ONE rolled loop, the
     * 10 exp() of the normalisation likewise: one code instance each instead of 23, a fraction of the
     * registers, and the independent chains of the other waves on the SIMD hide the latency. */
    double *la = reinterpret_cast<double *>(slot + lane * IN_DW);
    bool has4, has5, has6, has7;
    { /* recomputed from laundered copies: otherwise the compiler keeps the summary's eight compare masks alive in
       * SGPR pairs that it then spills to VGPR lanes (4 extra VALU per class) */
      uint32_t c4 = cnt[4], c5 = cnt[5], c6 = cnt[6], c7 = cnt[7];
      asm volatile("" : "+v"(c4), "+v"(c5), "+v"(c6), "+v"(c7));
      has4 = c4 != 0; has5 = c5 != 0; has6 = c6 != 0; has7 = c7 != 0;
    }
    { /* methylation estimates (src/genotype_model.c:165-171), one strand at a time to keep few values live;
       * the Z of an empty class pair is never used.  C2T reads: classes 5 (C) and 7 (T) */
      const double k5 = s_k[QI(5)], k7 = s_k[QI(7)];
      double Z0, Z1, Z2;
      get_Z(ND(5), ND(7), k5, k7, l, t, Z0, Z1, Z2);
      la[3] = has5 ? Z0 + k5 : 2.0;                 /* class 5 (:188-201): CC */
      la[4] = has5 ? 0.5 * Z1 + k5 : 2.0;           /*                     CT */
      la[5] = has5 ? 0.5 * Z2 + k5 : 2.0;           /*                     AC, CG */
      la[9] = has7 ? 1.0 - Z0 + k7 : 2.0;           /* class 7 (:216-230): CC */
      la[10] = has7 ? 1.0 - 0.5 * Z1 + k7 : 2.0;    /*                     CT */
      la[11] = has7 ? 0.5 * (1.0 - Z2) + k7 : 2.0;  /*                     AC, CG */
    }
    { /* G2A reads: classes 6 (G) and 4 (A) */
      const double k6 = s_k[QI(6)], k4 = s_k[QI(4)];
      double Z3, Z4, Z5;
      get_Z(ND(6), ND(4), k6, k4, l, t, Z3, Z4, Z5);
      la[0] = has4 ? 1.0 - 0.5 * Z4 + k4 : 2.0;     /* class 4 (:173-187): AG */
      la[1] = has4 ? 1.0 - Z3 + k4 : 2.0;           /*                     GG */
      la[2] = has4 ? 0.5 * (1.0 - Z5) + k4 : 2.0;   /*                     CG, GT */
      la[6] = has6 ? Z3 + k6 : 2.0;                 /* class 6 (:202-215): GG */
      la[7] = has6 ? 0.5 * Z4 + k6 : 2.0;           /*                     AG */
      la[8] = has6 ? 0.5 * Z5 + k6 : 2.0;           /*                     CG, GT */
    }
    /*
     * Only the non-empty classes need their three logs (an empty class contributes n * anything = 0), and a site has
     * 1-2 of the 4 informative classes (A/T sites one, C/G sites two, plus the odd error read).  The (lane, class)
     * pairs that need evaluating are listed in a 256-byte LDS index (ballot + prefix count, one byte per pair:
     * class << 6 | lane) and the wave then evaluates 64 pairs at a time, each lane fetching "its" pair's arguments
     * from the owner lane's area and putting the logs back there: ~2 dense rounds of 3 logs instead of 12 sparse ones.
     */
    {
      uint8_t *lst = s_pairs[wid];
      unsigned n_pairs = 0; /* wave-uniform */
      {
        const bool hasc[4] = {has4, has5, has6, has7};
#pragma unroll
        for (int c = 0; c < 4; c++) {
          const unsigned long long m = __ballot(hasc[c]);
          if (hasc[c]) lst[n_pairs + __popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)((c << 6) | lane);
          n_pairs += (unsigned)__popcll(m);
        }
      }
#pragma unroll 1
      for (unsigned k = 0; k < n_pairs; k += 64) {
        const bool act = k + lane < n_pairs;
        const unsigned e = act ? lst[k + lane] : 0u;
        double *pa = reinterpret_cast<double *>(slot + (e & 63u) * IN_DW) + 3u * (e >> 6);
#pragma unroll 1
        for (int t3 = 0; t3 < 3; t3++) {
          const double xa = act ? pa[t3] : 2.0;
          const double ya = log_dev(xa, s_logtab);
          if (act) pa[t3] = ya;
        }
      }
    }
    /* prior from the reference base (src/genotype_model.c:87-108); genotype order AA AC AG AT CC CG CT GG GT TT */
    const bool rA = rf == 1, rC = rf == 2, rG = rf == 3, rT = rf == 4;
    double ll0 = rA ? lrb : 0.0, ll4 = rC ? lrb : 0.0, ll7 = rG ? lrb : 0.0, ll9 = rT ? lrb : 0.0;
    double ll1 = (rA || rC) ? lrb1 : 0.0, ll2 = (rA || rG) ? lrb1 : 0.0, ll3 = (rA || rT) ? lrb1 : 0.0;
    double ll5 = (rC || rG) ? lrb1 : 0.0, ll6 = (rC || rT) ? lrb1 : 0.0, ll8 = (rG || rT) ? lrb1 : 0.0;
    /*
     * One term per class and genotype, classes in order 0..7 (the order of the reference's += statements;
     * SURVEY.md appendix A gives the matrix).  An empty class has n = +0, so each of its terms is n*finite
     * = +-0 and the addition leaves ll unchanged, exactly as the reference's skipped `if (n[c])` block does
     * (ll is never -0: every contribution is n*ln(..) with n > 0, and the priors are >= +0).
     */
#define ACC10(a0, a1, a2, a3, a4, a5, a6, a7, a8, a9) \
  ll0 += (a0); ll1 += (a1); ll2 += (a2); ll3 += (a3); ll4 += (a4); ll5 += (a5); ll6 += (a6); ll7 += (a7); ll8 += (a8); ll9 += (a9)
    { /* class 0, A non-informative (:109-122): AA one; AC AG AT half */
      const double n = ND(0);
      const unsigned qi = QI(0);
      const double one = n * s_one[qi], half = n * s_half[qi], lnk = n * s_lnk[qi];
      ACC10(one, half, half, half, lnk, lnk, lnk, lnk, lnk, lnk);
    }
    { /* class 1, C (:123-136): CC one; AC CG CT half */
      const double n = ND(1);
      const unsigned qi = QI(1);
      const double one = n * s_one[qi], half = n * s_half[qi], lnk = n * s_lnk[qi];
      ACC10(lnk, half, lnk, lnk, one, half, half, lnk, lnk, lnk);
    }
    { /* class 2, G (:137-150): GG one; AG CG GT half */
      const double n = ND(2);
      const unsigned qi = QI(2);
      const double one = n * s_one[qi], half = n * s_half[qi], lnk = n * s_lnk[qi];
      ACC10(lnk, lnk, half, lnk, lnk, half, lnk, one, half, lnk);
    }
    { /* class 3, T (:151-164): TT one; AT CT GT half */
      const double n = ND(3);
      const unsigned qi = QI(3);
      const double one = n * s_one[qi], half = n * s_half[qi], lnk = n * s_lnk[qi];
      ACC10(lnk, lnk, lnk, half, lnk, lnk, half, lnk, half, one);
    }
    { /* class 4, A on G2A reads (:173-187): AA one; AC AT half; AG za; GG zb; CG GT zc */
      const double n = ND(4);
      const unsigned qi = QI(4);
      const double one = n * s_one[qi], half = n * s_half[qi], lnk = n * s_lnk[qi];
      const double za = la[0] * n, zb = la[1] * n, zc = la[2] * n;
      ACC10(one, half, za, half, lnk, zc, lnk, zb, zc, lnk);
    }
    { /* class 5, C on C2T reads (:188-201): CC za; CT zb; AC CG zc */
      const double n = ND(5);
      const double lnk = n * s_lnk[QI(5)];
      const double za = la[3] * n, zb = la[4] * n, zc = la[5] * n;
      ACC10(lnk, zc, lnk, lnk, za, zc, zb, lnk, lnk, lnk);
    }
    { /* class 6, G on G2A reads (:202-215): GG za; AG zb; CG GT zc */
      const double n = ND(6);
      const double lnk = n * s_lnk[QI(6)];
      const double za = la[6] * n, zb = la[7] * n, zc = la[8] * n;
      ACC10(lnk, lnk, zb, lnk, lnk, zc, lnk, za, zc, lnk);
    }
    { /* class 7, T on C2T reads (:216-230): TT one; AT GT half; CC za; CT zb; AC CG zc */
      const double n = ND(7);
      const unsigned qi = QI(7);
      const double one = n * s_one[qi], half = n * s_half[qi], lnk = n * s_lnk[qi];
      const double za = la[9] * n, zb = la[10] * n, zc = la[11] * n;
      ACC10(lnk, zc, lnk, half, za, zc, zb, lnk, half, one);
    }
#undef ACC10
    /* first-max argmax (:231-239) */
    double mx = ll0;
    int mxi = 0;
#define AMAX(g) { const bool gt_ = ll##g > mx; mx = gt_ ? ll##g : mx; mxi = gt_ ? g : mxi; }
    AMAX(1) AMAX(2) AMAX(3) AMAX(4) AMAX(5) AMAX(6) AMAX(7) AMAX(8) AMAX(9)
#undef AMAX
    la[0] = ll0 - mx; la[1] = ll1 - mx; la[2] = ll2 - mx; la[3] = ll3 - mx; la[4] = ll4 - mx;
    la[5] = ll5 - mx; la[6] = ll6 - mx; la[7] = ll7 - mx; la[8] = ll8 - mx; la[9] = ll9 - mx;
    /* normalise (:240-245): sum of exp(ll - max) in index order, rolled */
    double sum = 0.0;
#pragma unroll 1
    for (int g = 0; g < 10; g++) sum += exp_dev(la[g], (const uint64_t *)s_exptab);
    const double lsum = log_dev(sum, s_logtab);
#pragma unroll 1
    for (int g = 0; g < 10; g++) la[g] = div_ln10_dev(la[g] - lsum);
    double gp[10];
#pragma unroll
    for (int g = 0; g < 10; g++) gp[g] = la[g];

    /* ---- heterozygous calls go to the Fisher list; block counters ---- */
    const bool het = covered && ((0x16Eu >> mxi) & 1u); /* gt_het: AC AG AT CG CT GT = bits 1,2,3,5,6,8 */
    if (covered) {
      atomicAdd(&s_cnt[0], 1u);
      atomicAdd(&s_cnt[1 + mxi], 1u);
    }
    {
      const unsigned long long m = __ballot(het);
      if (m) {
        unsigned base = 0;
        if (lane == 0) base = atomicAdd((unsigned int *)&counters[BSC_CNT_HET_LIST], (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (het) het_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)site;
        if (lane == 0) atomicAdd(&s_cnt[11], (unsigned)__popcll(m));
      }
    }
    if (valid) skip[site] = covered ? 0 : 1;

    /* ---- results: two halves of 32 records through the slot ----
     * A lane's record lands on bytes that other lanes used as their la[] areas, and the second half on bytes the
     * copy-out of the first half reads: a wave runs in lockstep, so only the COMPILER could reorder these LDS accesses
     * across lanes' program order; s_wave_barrier (convergent, a scheduling barrier, free at run time) pins them. */
#pragma unroll 1
    for (unsigned half = 0; half < 2; half++) {
      __builtin_amdgcn_wave_barrier();
      const bool mine = valid && (lane >> 5) == half;
      auto write_record = [&](auto *rec) {
        if (covered) {
#pragma unroll
          for (int j = 0; j < 8; j++) rec[j] = make_uint2(cnt[j], 0u); /* counts[j] as u64 */
          rec[8] = make_uint2(qpack0 & 0xffu, (qpack0 >> 8) & 0xffu); /* qual[0..7] as i32 */
          rec[9] = make_uint2((qpack0 >> 16) & 0xffu, qpack0 >> 24);
          rec[10] = make_uint2(qpack1 & 0xffu, (qpack1 >> 8) & 0xffu);
          rec[11] = make_uint2((qpack1 >> 16) & 0xffu, qpack1 >> 24);
#pragma unroll
          for (int g = 0; g < 10; g++) {
            const uint64_t b = bsm_bits(gp[g]);
            rec[12 + g] = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
          }
          rec[22] = make_uint2(0u, 0u); /* fisher_strand = 0.0; bsc_fisher_kernel fills heterozygous sites */
          rec[23] = make_uint2((uint32_t)mq, (uint32_t)aq);
          rec[24] = make_uint2((uint32_t)mxi, 0u); /* max_gt + zero padding */
        } else {
#pragma unroll
          for (int j = 0; j < 25; j++) rec[j] = make_uint2(0u, 0u); /* skipped site: the reference's memset, :179 */
        }
        /* out_dw == 52 (gt_vcf): bytes 200.. = {ready = 0, skip, pad} */
        if (out_dw > OUT_DW) rec[25] = make_uint2(covered ? 0u : 0x100u, 0u);
      };
      if (mine) {
        if (full) write_record(reinterpret_cast<uint2 *>(slot + (lane & 31u) * out_dw)); /* LDS */
        else write_record(reinterpret_cast<uint2 *>(out + site * out_dw));                /* global */
      }
      if (full) { /* copy the 32 records out: 16 bytes per lane, contiguous */
        __builtin_amdgcn_wave_barrier(); /* the staging stores above must not sink below the cross-lane reads */
        const unsigned nvec = 32u * out_dw / 4u; /* 400 or 416 */
        uint4 *dst = reinterpret_cast<uint4 *>(out + (site0 + half * 32u) * out_dw);
        const uint4 *srcv = reinterpret_cast<const uint4 *>(slot);
#pragma unroll
        for (unsigned v = 0; v < 7; v++) {
          const unsigned idx = v * 64u + lane;
          if (idx < nvec) { /* written once, never re-read by this kernel: non-temporal (-2 % kernel time) */
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(reinterpret_cast<const u32x4 *>(srcv)[idx], reinterpret_cast<u32x4 *>(dst) + idx);
          }
        }
      }
    }
  }

  /* block counters -> global */
  __syncthreads();
  if (tid < 12 && s_cnt[tid]) atomicAdd(&counters[BSC_CNT_COVERED + tid], (unsigned long long)s_cnt[tid]);
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

/* One thread per heterozygous site of the compact list. */
extern "C" __global__ __launch_bounds__(256) void bsc_fisher_kernel(const uint32_t *__restrict__ cts,
                                                                    uint32_t *__restrict__ out, uint32_t out_dw,
                                                                    const bsc_dev_tables *__restrict__ tb,
                                                                    const uint32_t *__restrict__ het_list,
                                                                    const unsigned long long *__restrict__ counters) {
  __shared__ double s_lf[256];
  __shared__ double s_logtab[256];
  __shared__ unsigned long long s_exptab[256];
  s_lf[threadIdx.x] = tb->lfact[threadIdx.x];
  s_logtab[threadIdx.x] = tb->log_tab[threadIdx.x];
  s_exptab[threadIdx.x] = tb->exp_tab[threadIdx.x];
  __syncthreads();
  const unsigned nhet = (unsigned)counters[BSC_CNT_HET_LIST];
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nhet; i += gridDim.x * blockDim.x) {
    const uint64_t site = het_list[i];
    const uint32_t *p = cts + site * IN_DW;
    uint32_t f[8], r[8]; /* counts[0][*] forward, counts[1][*] reverse */
#pragma unroll
    for (int j = 0; j < 8; j++) {
      f[j] = p[j];
      r[j] = p[8 + j];
    }
    uint32_t *rec = out + site * out_dw;
    const unsigned mxi = rec[48] & 0xffu; /* max_gt at byte 192 */
    int t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    switch (mxi) { /* src/call_genotypes.c:64-100 */
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
    double z = fisher_dev(t0, t1, t2, t3, s_lf, s_logtab, (const uint64_t *)s_exptab);
    if (z < 1.0e-20) z = 1.0e-20;
    const double fs = bsm_log_t(z, s_logtab) / BSM_LN10;
    const uint64_t b = bsm_bits(fs);
    reinterpret_cast<uint2 *>(rec)[22] = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
  }
}

/* Synthetic L-pileup generator: one thread per site (not on the timed path). */
extern "C" __global__ __launch_bounds__(256) void bsc_synth_kernel(uint64_t seed, uint64_t first_site, uint64_t n,
                                                                   uint32_t coverage, uint32_t flags,
                                                                   uint32_t *__restrict__ cts,
                                                                   uint8_t *__restrict__ ref) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t counts[16], nr, rf;
    float q[8], m2;
    syn_site(seed, first_site + i, coverage, flags, counts, &nr, q, &m2, &rf);
    uint32_t *p = cts + i * IN_DW;
#pragma unroll
    for (int j = 0; j < 16; j++) p[j] = counts[j];
    p[16] = nr;
#pragma unroll
    for (int j = 0; j < 8; j++) p[17 + j] = __float_as_uint(q[j]);
    p[25] = __float_as_uint(m2);
    ref[i] = (uint8_t)rf;
  }
}

/* ---- launchers (called from the C host code in bscall_api.c) ---------------------------------------- */

/* pile-up -> gt_meth for n sites (n < 2^32), then the Fisher pass over the heterozygous list.
 * counters[BSC_CNT_HET_LIST] must be zero on entry (the host queues a memset in front). */
extern "C" int bsc_dev_launch_call(const void *cts, const void *ref, uint64_t n, void *out, uint32_t out_dw, void *skip,
                                   const void *tb, void *het_list, void *counters, int num_cus, void *stream,
                                   void *ev_start, void *ev_mid, void *ev_stop) {
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, s);
  /* Grid: the workgroups resident at once (BSC_WAVES_PER_SIMD per CU: 33 KB LDS, <= 128 VGPRs each) times a number of
   * rounds chosen so that a wave walks about BSC_TILES_PER_WAVE wave-tiles.  Measured (gpurun_out/ab*.txt):
   * a fully persistent grid (1 round) is best for small blocks (the table set-up is paid once per wave slot) but 7 %
   * slower at 50 M positions than 8-16 rounds, whose workgroup turnover keeps the waves of a CU out of phase;
   * beyond 32 rounds the set-up cost shows again. */
  uint64_t resident = (uint64_t)num_cus * (BSC_WAVES_PER_SIMD * 4) / (TILE / 64); /* workgroups that fit the chip at once */
  if (resident < 1) resident = 1;
  const uint64_t n_full = n / 64; /* complete wave-tiles: the specialised kernel; the ragged rest: the guarded one */
  hipError_t e = hipSuccess;
  if (n_full) {
    const uint64_t n_tiles = (n_full + (TILE / 64) - 1) / (TILE / 64);
    uint64_t rounds = n_full / (resident * (TILE / 64) * BSC_TILES_PER_WAVE);
    rounds = rounds < 1 ? 1 : (rounds > 16 ? 16 : rounds);
    uint64_t grid = resident * rounds;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(bsc_call_kernel_t<true>, dim3((unsigned)grid), dim3(TILE), 0, s, (const uint32_t *)cts,
                       (const uint8_t *)ref, n, (uint64_t)0, n_full, (uint32_t *)out, out_dw, (uint8_t *)skip,
                       (const bsc_dev_tables *)tb, (uint32_t *)het_list, (unsigned long long *)counters);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (n & 63u) {
    hipLaunchKernelGGL(bsc_call_kernel_t<false>, dim3(1), dim3(TILE), 0, s, (const uint32_t *)cts, (const uint8_t *)ref, n,
                       n_full, n_full + 1, (uint32_t *)out, out_dw, (uint8_t *)skip, (const bsc_dev_tables *)tb,
                       (uint32_t *)het_list, (unsigned long long *)counters);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (ev_mid) (void)hipEventRecord((hipEvent_t)ev_mid, s);
  hipLaunchKernelGGL(bsc_fisher_kernel, dim3((unsigned)(num_cus * 2)), dim3(256), 0, s, (const uint32_t *)cts,
                     (uint32_t *)out, out_dw, (const bsc_dev_tables *)tb, (const uint32_t *)het_list,
                     (const unsigned long long *)counters);
  e = hipGetLastError();
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, s);
  return (int)e;
}

extern "C" int bsc_dev_launch_synth(uint64_t seed, uint64_t first_site, uint64_t n, uint32_t coverage, uint32_t flags,
                                    void *cts, void *ref, int num_cus, void *stream) {
  if (n == 0) return 0;
  uint64_t grid = (n + 255) / 256;
  const uint64_t cap = (uint64_t)num_cus * 32u;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(bsc_synth_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, seed, first_site, n,
                     coverage, flags, (uint32_t *)cts, (uint8_t *)ref);
  return (int)hipGetLastError();
}
