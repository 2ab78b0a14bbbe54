/*
 * sitestats_dev.h — layout of the LDS statistics histogram and the device helpers shared by the statistics kernel
 * (sitestats.hip) and the fused chain kernels (fused.hip): what the reference's printer adds to bs_stats per position
 * (src/print_vcf.c:382-526), see sitestats.hip for the restatement.
 */
#ifndef BSCALL_AMD_SITESTATS_DEV_H
#define BSCALL_AMD_SITESTATS_DEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bscall_amd.h"
#include "bsmath.h"
#include "devtables.h"

#define SS_THREADS 1024 /* one workgroup per CU: its histograms take 110 KB of LDS */
#define SS_PAIR 64      /* (a, b) < SS_PAIR: counted in the unfused kernel's LDS pair table */
#define SS_PAIR_G 512   /* the context's pair table in HBM: u64 [ref / non-ref][all / passed][a < SS_PAIR_G][b < SS_PAIR_G]; a
                           cell's posterior is evaluated once, when the statistics are read, times its count */
#define SS_PAIR_G_BYTES (4ull * SS_PAIR_G * SS_PAIR_G * 8ull)
#define SS_OVF_CAP (1u << 20) /* fused chain: cytosines beyond even that table (>= 512 informative reads of one kind), listed */
#define SS_COV_LDS 1024 /* coverage rows kept in LDS; deeper positions go to global memory directly */

/* LDS histogram layout (u32 words) */
#define SS_MISC 0                    /* snps, indels, multi, dbSNP_sites, dbSNP_var, CpG_ref, CpG_nonref: 14 words */
#define SS_MUT (SS_MISC + 14)        /* 24 */
#define SS_DBMUT (SS_MUT + 24)       /* 24 */
#define SS_QUAL (SS_DBMUT + 24)      /* 1024 */
#define SS_FILT (SS_QUAL + 1024)     /* 64 */
#define SS_FST (SS_FILT + 64)        /* qd, fs, mq: 3 x 512 */
#define SS_COV (SS_FST + 1536)       /* SS_COV_LDS x 6 */
#define SS_WORDS (SS_COV + SS_COV_LDS * 6)

static_assert(offsetof(bsc_site_stats, mut_counts) == 14 * 8, "misc block is 14 words");
static_assert(offsetof(bsc_site_stats, qual) == (14 + 48) * 8, "layout");
static_assert(offsetof(bsc_site_stats, filter_counts) == (14 + 48 + 1024) * 8, "layout");
static_assert(offsetof(bsc_site_stats, qd_stats) == (14 + 48 + 1024 + 64) * 8, "layout");
static_assert(offsetof(bsc_site_stats, cov) == (14 + 48 + 1024 + 64 + 1536) * 8, "layout");
static_assert(SS_COV == 14 + 48 + 1024 + 64 + 1536, "the LDS histogram mirrors the integer part of bsc_site_stats");

/* genotype -> alleles as base codes 1..4 (AA AC AG AT CC CG CT GG GT TT) */
__device__ static __forceinline__ void ss_alleles(int g, int &a, int &b) {
  a = g < 4 ? 1 : (g < 7 ? 2 : (g < 9 ? 3 : 4));
  b = g < 4 ? g + 1 : (g < 7 ? g - 2 : (g < 9 ? g - 4 : 4));
}

/* stats_mut index (include/bs_call.h:46) of ref X -> allele Y, base codes 1..4, X != Y */
__device__ static __forceinline__ int ss_mut_xy(int x, int y) { return 3 * (x - 1) + (y < x ? y - 1 : y - 2); }

/* mut_type[gt][rfix] (src/print_vcf.c:46-57) from its rule: the one non-reference allele of a call that carries the
 * reference base or is homozygous; 12 = mut_no */
__device__ static __forceinline__ int ss_mut_type(int gt, int rfix) {
  if (rfix == 0) return 12;
  int a, b;
  ss_alleles(gt, a, b);
  if (a == b) return a == rfix ? 12 : ss_mut_xy(rfix, a);
  if (a == rfix) return ss_mut_xy(rfix, b);
  if (b == rfix) return ss_mut_xy(rfix, a);
  return 12;
}

/* lfact2 (include/bs_call.h:335) */
__device__ static __forceinline__ double ss_lfact(int x, const double *lf, const double *logtab) {
  return x < 256 ? lf[x] : bsm_lfact_big_t(x, logtab);
}

/* One posterior: bins lane and lane + 64 of meth[i] / sum (src/print_vcf.c:494-505). */
__device__ static __forceinline__ void ss_posterior(uint32_t a, uint32_t b, unsigned lane, const double *s_lf,
                                                    const double *s_logtab, const double *s_logp,
                                                    const unsigned long long *s_exptab, double z[2]) {
  const double konst = ss_lfact((int)(a + b + 1u), s_lf, s_logtab) - ss_lfact((int)a, s_lf, s_logtab) -
                       ss_lfact((int)b, s_lf, s_logtab);
  const double da = (double)a, db = (double)b;
  double v[2];
#pragma unroll
  for (int r = 0; r < 2; r++) {
    const unsigned bin = lane + 64u * r;
    double e = 0.0;
    if (bin == 0) e = a ? 0.0 : bsm_exp_t(konst, (const uint64_t *)s_exptab);
    else if (bin == 100) e = b ? 0.0 : bsm_exp_t(konst, (const uint64_t *)s_exptab);
    else if (bin < 100) e = bsm_exp_t(konst + s_logp[bin - 1] * da + s_logp[99 - bin] * db, (const uint64_t *)s_exptab);
    v[r] = e;
  }
  double sum = v[0] + v[1];
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  z[0] = v[0] / sum;
  z[1] = v[1] / sum;
}

/*
 * h[base + value]++ for every lane with `on`, executed by whole waves (wave-uniform control flow).  Most of these
 * statistics have one or two dominant values (QUAL 255, MQ 60, FILTER 0, FS 0 ...), and 64 LDS atomics on one address
 * are 64 serial passes: the most common values are peeled off with a ballot and added once, by one lane, before the
 * remaining lanes go individually.  base2 >= 0: a second histogram receives the same increments.
 */
template <int PEEL>
__device__ static __forceinline__ void ss_hist_add(uint32_t *h, bool on, uint32_t base, uint32_t value, int base2 = -1) {
  unsigned long long mask = __ballot(on);
#pragma unroll
  for (int it = 0; it < PEEL; it++) {
    if (!mask) break;
    const int src = __builtin_ctzll(mask);
    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane(value, src);
    const unsigned long long same = __ballot(on && value == v);
    if ((int)(threadIdx.x & 63u) == src) {
      atomicAdd(&h[base + v], (uint32_t)__popcll(same));
      if (base2 >= 0) atomicAdd(&h[(uint32_t)base2 + v], (uint32_t)__popcll(same));
    }
    on = on && value != v;
    mask &= ~same;
  }
  if (on) {
    atomicAdd(&h[base + value], 1u);
    if (base2 >= 0) atomicAdd(&h[(uint32_t)base2 + value], 1u);
  }
}

#endif /* BSCALL_AMD_SITESTATS_DEV_H */
