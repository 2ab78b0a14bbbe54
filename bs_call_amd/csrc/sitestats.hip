/*
 * sitestats.hip — the sums the reference's printer adds to bs_stats per position (src/print_vcf.c:382-526), computed
 * from a block's bsc_vcf_core records and the gt_meth records behind them.  Every field is a sum over positions, so
 * the kernel is a histogram reduction: one thread per position, the block's counts in LDS (u32), flushed to the
 * context's bsc_site_stats with one 64-bit atomic per non-zero bin.  The methylation profiles add, per CpG cytosine,
 * a 101-bin posterior that depends only on its two informative counts (a, b): evaluating 100 exp() per cytosine made
 * this kernel slower than the calling kernel, so cytosines are only COUNTED per (a, b) here (a, b < 64; the rare deeper
 * ones are evaluated on the spot) and bsc_meth_eval_kernel turns the counts into profiles when the statistics are
 * read: count x posterior, once per distinct pair.
 *
 * What the reference does there, restated:
 *   - every position with depth > 0 that reaches _print_vcf_entry: cov[depth].all++ (:386-393);
 *   - written records only (!skip): `alt` was walked to its terminating 0 while the ALT field was encoded (:177-181),
 *     so `alt[0] != '.'` holds for every record and `alt[1] == ','` for none: snps++, qual[variant][phred]++,
 *     cov[depth].var++ for all of them, multi never (:401-420); the FILTER statistics (:421-425); dbSNP (:426-441);
 *   - CpG bookkeeping for records whose CG field is "CG" (:442-516): a '+' strand call (genotype AC, CC, CT) becomes
 *     the pending cytosine; a '-' strand call (AG, GG, GT) right after it completes a CpG (reference or not by the
 *     reference context), passed if neither carries a filter; both kinds add their methylation posterior over 0..100 %
 *     when they have informative reads;
 *   - the mutation type of the call against the reference base (:517-525).
 * The pending cytosine (prev_cpg_x / prev_cpg_flt) is static in the reference, i.e. it survives from block to block:
 * here it travels through two alternating device words (carry), so blocks must be passed in order.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sitestats_dev.h"

extern "C" __global__ __launch_bounds__(SS_THREADS) void bsc_site_stats_kernel(
    const uint8_t *__restrict__ core, const uint8_t *__restrict__ gtm, uint32_t gtm_stride,
    const uint8_t *__restrict__ dbsnp, uint32_t n, const bsc_dev_tables *__restrict__ tb,
    const double *__restrict__ logp, const uint32_t *__restrict__ carry_in, uint32_t *__restrict__ carry_out,
    unsigned long long *__restrict__ out_words, double *__restrict__ out_meth, unsigned long long *__restrict__ out_pairs) {
  __shared__ uint32_t h[SS_WORDS];
  __shared__ uint32_t s_pair[4 * SS_PAIR * SS_PAIR]; /* [ref / non-ref][all / passed][a][b] */
  __shared__ double s_meth[4 * 101]; /* CpG_ref_meth[2][101], CpG_nonref_meth[2][101] */
  __shared__ double s_lf[256], s_logtab[256], s_logp[100];
  __shared__ unsigned long long s_exptab[256];
  const unsigned lane = threadIdx.x & 63u;
  for (unsigned i = threadIdx.x; i < SS_WORDS; i += SS_THREADS) h[i] = 0;
  for (unsigned i = threadIdx.x; i < 4 * SS_PAIR * SS_PAIR; i += SS_THREADS) s_pair[i] = 0;
  for (unsigned i = threadIdx.x; i < 404; i += SS_THREADS) s_meth[i] = 0.0;
  for (unsigned i = threadIdx.x; i < 256; i += SS_THREADS) {
    s_lf[i] = tb->lfact[i];
    s_logtab[i] = tb->log_tab[i];
    s_exptab[i] = tb->exp_tab[i];
  }
  for (unsigned i = threadIdx.x; i < 100; i += SS_THREADS) s_logp[i] = logp[i];
  __syncthreads();

  /* this lane's bins of the four methylation profiles: bin lane and bin lane + 64; [ref / non-ref][all / passed] */
  double acc[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++) acc[a][b][0] = acc[a][b][1] = 0.0;

  const uint32_t n_tiles = (n + 63u) / 64u;
  for (uint32_t tile = blockIdx.x * (SS_THREADS / 64) + (threadIdx.x >> 6); tile < n_tiles;
       tile += gridDim.x * (SS_THREADS / 64)) {
    const uint32_t i = tile * 64u + lane;
    bool do_meth = false, m_ref = false, m_pass = false;
    uint32_t m_a = 0, m_b = 0;
    /* per-lane facts first, the histogram updates afterwards with the whole wave (ss_hist_add) */
    bool called = false, emit = false, pass = false, het = false, rs = false, cpg_site = false, ref_cpg = false;
    bool pair = false, pair_pass = false, fs_ok = false;
    uint32_t phred = 0, flt = 0, qd = 0, fsv = 0, mqv = 0, cdp = 0, cinf = 0;
    int mut = 12;
    if (i < n) {
      const uint4 c0 = *reinterpret_cast<const uint4 *>(core + (uint64_t)i * 64u);
      const uint32_t pos = c0.x;
      called = pos != 0; /* the position reached _print_vcf_entry with depth > 0 */
      if (called) {
        const uint32_t gt = (c0.y >> 8) & 0xffu, rfix = (c0.y >> 16) & 0xffu;
        emit = (c0.y & 0xffu) != 0;
        flt = c0.z & 0xffu;
        phred = (c0.z >> 8) & 0xffu;
        const char cg = (char)(c0.z >> 24);
        /* cx_ref[5] sits at bytes 14..18: prf_ctxt */
        const uint4 c1 = *reinterpret_cast<const uint4 *>(core + (uint64_t)i * 64u + 16u);
        const char r1 = (char)(c0.w >> 24), r2 = (char)c1.x, r3 = (char)(c1.x >> 8); /* prf_ctxt[1..3] */
        const int32_t fs = (int32_t)c1.z; /* bytes 24..27 */
        qd = c1.w > 255u ? 255u : c1.w;   /* bytes 28..31 */
        fs_ok = fs >= 0;
        fsv = (uint32_t)(fs > 255 ? 255 : (fs < 0 ? 0 : fs));
        const uint64_t *cnt = reinterpret_cast<const uint64_t *>(gtm + (uint64_t)i * gtm_stride);
        uint64_t counts[8];
#pragma unroll
        for (int k = 0; k < 8; k++) counts[k] = cnt[k];
        const int32_t mq = *reinterpret_cast<const int32_t *>(gtm + (uint64_t)i * gtm_stride + 184u);
        mqv = (uint32_t)(mq < 0 ? 0 : (mq > 255 ? 255 : mq));
        const uint32_t d_inf = (uint32_t)(counts[4] + counts[5] + counts[6] + counts[7]);
        const uint32_t dp = (uint32_t)(counts[0] + counts[1] + counts[2] + counts[3]) + d_inf;
        cdp = dp < BSC_COV_CAP ? dp : BSC_COV_CAP - 1u;
        cinf = d_inf < BSC_COV_CAP ? d_inf : BSC_COV_CAP - 1u;
        const bool plus = gt == 1 || gt == 4 || gt == 6, minus = gt == 2 || gt == 7 || gt == 8; /* cs_str "+", "-" */
        if (emit) {
          pass = flt == 0;
          rs = dbsnp ? dbsnp[i] != 0 : false;
          het = !(gt == 0 || gt == 4 || gt == 7 || gt == 9);
          if (cg == 'C') { /* FORMAT CG == "CG" */
            if (plus) {
              ref_cpg = r2 == 'C' && r3 == 'G';
              m_a = (uint32_t)counts[5];
              m_b = (uint32_t)counts[7];
            } else if (minus) {
              ref_cpg = r1 == 'C' && r2 == 'G';
              /* the pending cytosine: the record just before this one, if it was a written '+' strand CG call */
              uint32_t p_pos, p_flt;
              bool p_ok;
              if (i > 0) {
                const uint4 p = *reinterpret_cast<const uint4 *>(core + (uint64_t)(i - 1u) * 64u);
                const uint32_t pg = (p.y >> 8) & 0xffu;
                p_ok = p.x != 0 && (p.y & 0xffu) && (char)(p.z >> 24) == 'C' && (pg == 1 || pg == 4 || pg == 6);
                p_pos = p.x;
                p_flt = p.z & 0xffu;
              } else {
                p_pos = carry_in[0];
                p_flt = carry_in[1];
                p_ok = p_pos != 0;
              }
              pair = p_ok && pos - p_pos == 1u;
              pair_pass = pair && !(p_flt || flt);
              m_a = (uint32_t)counts[6];
              m_b = (uint32_t)counts[4];
            }
            cpg_site = plus || minus;
            do_meth = cpg_site && m_a + m_b != 0;
            m_ref = ref_cpg;
            m_pass = pass;
          }
          mut = ss_mut_type((int)gt, (int)rfix);
        }
        /* the last position of the block is the next block's "record just before" */
        if (i == n - 1u) {
          const bool pend = emit && cg == 'C' && plus;
          carry_out[0] = pend ? pos : 0u;
          carry_out[1] = pend ? flt : 0u;
        }
      } else if (i == n - 1u) {
        carry_out[0] = 0u;
        carry_out[1] = 0u;
      }
    }
    /* coverage table: rows below SS_COV_LDS in LDS, deeper ones (rare) straight to global memory */
#define SS_COV_ADD(on, row, col)                                                                        \
  do {                                                                                                  \
    ss_hist_add<1>(h, (on) && (row) < SS_COV_LDS, SS_COV, (row) * 6u + (col)); /* col may differ per lane */ \
    if ((on) && (row) >= SS_COV_LDS) atomicAdd(&out_words[SS_COV + (uint64_t)(row) * 6u + (col)], 1ull); \
  } while (0)
    SS_COV_ADD(called, cdp, 0u);                                        /* gcov->all++ (:393) */
    SS_COV_ADD(emit, cdp, 1u);                                          /* gcov->var++: every written record */
    ss_hist_add<1>(h, emit, SS_MISC + 0, 0u);                           /* snps[all] — see the header */
    ss_hist_add<1>(h, emit && pass, SS_MISC + 1, 0u);                   /* snps[passed] */
    ss_hist_add<2>(h, emit, SS_QUAL + 0 * 256, phred, SS_QUAL + 1 * 256); /* qual[all_sites], qual[variant_sites] */
    ss_hist_add<2>(h, emit, SS_FST + 0 * 512, qd * 2u + het);
    ss_hist_add<2>(h, emit && fs_ok, SS_FST + 1 * 512, fsv * 2u + het);
    ss_hist_add<2>(h, emit, SS_FST + 2 * 512, mqv * 2u + het);
    ss_hist_add<2>(h, emit, SS_FILT, (het ? 32u : 0u) + (flt & 31u));
    if (__any(rs)) {
      ss_hist_add<1>(h, rs, SS_MISC + 6, 0u, SS_MISC + 8);              /* dbSNP_sites[all], dbSNP_var[all] */
      ss_hist_add<1>(h, rs && pass, SS_MISC + 7, 0u, SS_MISC + 9);
    }
    if (__any(cpg_site)) {
      ss_hist_add<1>(h, pair, SS_MISC + 10, ref_cpg ? 0u : 2u);          /* CpG_ref / CpG_nonref [all] */
      ss_hist_add<1>(h, pair_pass, SS_MISC + 11, ref_cpg ? 0u : 2u);
      ss_hist_add<1>(h, cpg_site, SS_QUAL + 2 * 256, (ref_cpg ? 0u : 256u) + phred);
      SS_COV_ADD(cpg_site, cdp, ref_cpg ? 2u : 3u);
      SS_COV_ADD(cpg_site, cinf, ref_cpg ? 4u : 5u);
    }
    if (__any(mut != 12)) {
      const bool m = mut != 12;
      ss_hist_add<1>(h, m, SS_MUT, (uint32_t)mut * 2u);
      ss_hist_add<1>(h, m && pass, SS_MUT + 1, (uint32_t)mut * 2u);
      if (__any(m && rs)) {
        ss_hist_add<1>(h, m && rs, SS_DBMUT, (uint32_t)mut * 2u);
        ss_hist_add<1>(h, m && rs && pass, SS_DBMUT + 1, (uint32_t)mut * 2u);
      }
    }
    /* methylation posteriors (:492-515): counted per (a, b) ... */
    if (do_meth && m_a < SS_PAIR && m_b < SS_PAIR) {
      const uint32_t cell = ((m_ref ? 0u : 2u) * SS_PAIR + m_a) * SS_PAIR + m_b;
      atomicAdd(&s_pair[cell], 1u);
      if (m_pass) atomicAdd(&s_pair[cell + SS_PAIR * SS_PAIR], 1u);
      do_meth = false;
    }
    /* ... or, beyond the table, evaluated now: the wave takes these cytosines one at a time, two bins per lane */
    unsigned long long mm = __ballot(do_meth);
    while (mm) {
      const int src = __builtin_ctzll(mm);
      mm &= mm - 1;
      const uint32_t a = (uint32_t)__builtin_amdgcn_readlane(m_a, src), b = (uint32_t)__builtin_amdgcn_readlane(m_b, src);
      const bool is_ref = __builtin_amdgcn_readlane((uint32_t)m_ref, src) != 0;
      const bool is_pass = __builtin_amdgcn_readlane((uint32_t)m_pass, src) != 0;
      double z2[2];
      ss_posterior(a, b, lane, s_lf, s_logtab, s_logp, s_exptab, z2);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        acc[is_ref ? 0 : 1][0][r] += z2[r];
        if (is_pass) acc[is_ref ? 0 : 1][1][r] += z2[r];
      }
    }
  }
  /* registers -> LDS -> global */
#pragma unroll
  for (int ref = 0; ref < 2; ref++)
    for (int ps = 0; ps < 2; ps++)
      for (int r = 0; r < 2; r++) {
        const unsigned bin = lane + 64u * r;
        if (bin < 101 && acc[ref][ps][r] != 0.0) atomicAdd(&s_meth[(ref * 2 + ps) * 101 + bin], acc[ref][ps][r]);
      }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < SS_WORDS; i += SS_THREADS)
    if (h[i]) atomicAdd(&out_words[i], (unsigned long long)h[i]);
  for (unsigned i = threadIdx.x; i < 404; i += SS_THREADS)
    if (s_meth[i] != 0.0) atomicAdd(&out_meth[i], s_meth[i]);
  for (unsigned i = threadIdx.x; i < 4 * SS_PAIR * SS_PAIR; i += SS_THREADS)
    if (s_pair[i]) { /* LDS cell [q][a][b] -> the context's [q][SS_PAIR_G][SS_PAIR_G] table */
      const unsigned q = i / (SS_PAIR * SS_PAIR), ab = i % (SS_PAIR * SS_PAIR);
      atomicAdd(&out_pairs[((uint64_t)q * SS_PAIR_G + ab / SS_PAIR) * SS_PAIR_G + ab % SS_PAIR], (unsigned long long)s_pair[i]);
    }
}

/* Pair counts -> profiles: the lanes scan 64 cells at a time, the wave evaluates the posterior of each cell that has a
 * count (two bins per lane) and adds it times the count; the cells are consumed (zeroed).  Then the fused chain's list of
 * cytosines beyond the table, one wave per entry. */
extern "C" __global__ __launch_bounds__(256) void bsc_meth_eval_kernel(unsigned long long *__restrict__ pairs,
                                                                       const bsc_dev_tables *__restrict__ tb,
                                                                       const double *__restrict__ logp,
                                                                       double *__restrict__ out_meth,
                                                                       const unsigned long long *__restrict__ ovf_list,
                                                                       const unsigned long long *__restrict__ counters) {
  __shared__ double s_meth[4 * 101];
  __shared__ double s_lf[256], s_logtab[256], s_logp[100];
  __shared__ unsigned long long s_exptab[256];
  const unsigned lane = threadIdx.x & 63u;
  for (unsigned i = threadIdx.x; i < 404; i += 256) s_meth[i] = 0.0;
  {
    const unsigned i = threadIdx.x;
    s_lf[i] = tb->lfact[i];
    s_logtab[i] = tb->log_tab[i];
    s_exptab[i] = tb->exp_tab[i];
    if (i < 100) s_logp[i] = logp[i];
  }
  __syncthreads();
  double acc[2][2][2];
#pragma unroll
  for (int a = 0; a < 2; a++)
    for (int b = 0; b < 2; b++) acc[a][b][0] = acc[a][b][1] = 0.0;
  const unsigned n_cells = 2u * SS_PAIR_G * SS_PAIR_G; /* a multiple of 64 */
  const unsigned wave = blockIdx.x * 4u + (threadIdx.x >> 6), n_waves = gridDim.x * 4u;
  for (unsigned c0 = wave * 64u; c0 < n_cells; c0 += n_waves * 64u) {
    const unsigned c = c0 + lane;
    const unsigned ref = c / (SS_PAIR_G * SS_PAIR_G), ab = c % (SS_PAIR_G * SS_PAIR_G);
    unsigned long long *p_all = pairs + ((uint64_t)ref * 2u) * SS_PAIR_G * SS_PAIR_G + ab;
    unsigned long long *p_pass = p_all + SS_PAIR_G * SS_PAIR_G;
    const unsigned long long my_all = *p_all;
    unsigned long long my_pass = 0;
    if (my_all) {
      my_pass = *p_pass;
      *p_all = 0;
      *p_pass = 0;
    }
    unsigned long long mm = __ballot(my_all != 0);
    while (mm) {
      const int src = __builtin_ctzll(mm);
      mm &= mm - 1;
      const unsigned cc = c0 + (unsigned)src;
      const unsigned r2 = cc / (SS_PAIR_G * SS_PAIR_G), ab2 = cc % (SS_PAIR_G * SS_PAIR_G);
      const double n_all = (double)__shfl(my_all, src), n_pass = (double)__shfl(my_pass, src);
      double z[2];
      ss_posterior(ab2 / SS_PAIR_G, ab2 % SS_PAIR_G, lane, s_lf, s_logtab, s_logp, s_exptab, z);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        acc[r2][0][r] += z[r] * n_all;
        acc[r2][1][r] += z[r] * n_pass;
      }
    }
  }
  if (ovf_list) {
    unsigned long long n_ovf = counters[BSC_CNT_OVF]; /* the host zeroes it after this kernel */
    if (n_ovf > SS_OVF_CAP) n_ovf = SS_OVF_CAP;      /* ... and reports the excess as an error */
    for (unsigned long long k = wave; k < n_ovf; k += n_waves) {
      const unsigned long long e = ovf_list[k];
      const uint32_t ca = (uint32_t)(e & 0xffffffu), cb = (uint32_t)((e >> 24) & 0xffffffu);
      const bool is_ref = (e >> 48) & 1u, is_pass = (e >> 49) & 1u;
      double z[2];
      ss_posterior(ca, cb, lane, s_lf, s_logtab, s_logp, s_exptab, z);
#pragma unroll
      for (int r = 0; r < 2; r++) {
        acc[is_ref ? 0 : 1][0][r] += z[r];
        if (is_pass) acc[is_ref ? 0 : 1][1][r] += z[r];
      }
    }
  }
#pragma unroll
  for (int ref = 0; ref < 2; ref++)
    for (int ps = 0; ps < 2; ps++)
      for (int r = 0; r < 2; r++) {
        const unsigned bin = lane + 64u * r;
        if (bin < 101 && acc[ref][ps][r] != 0.0) atomicAdd(&s_meth[(ref * 2 + ps) * 101 + bin], acc[ref][ps][r]);
      }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < 404; i += 256)
    if (s_meth[i] != 0.0) atomicAdd(&out_meth[i], s_meth[i]);
}

extern "C" int bsc_dev_launch_meth_eval(void *pairs, const void *tb, const void *logp, void *stats, const void *ovf_list,
                                        const void *counters, int num_cus, void *stream) {
  double *meth = (double *)((char *)stats + offsetof(bsc_site_stats, CpG_ref_meth));
  hipLaunchKernelGGL(bsc_meth_eval_kernel, dim3((unsigned)num_cus), dim3(256), 0, (hipStream_t)stream,
                     (unsigned long long *)pairs, (const bsc_dev_tables *)tb, (const double *)logp, meth,
                     (const unsigned long long *)ovf_list, (const unsigned long long *)counters);
  return (int)hipGetLastError();
}

/* GC content by coverage for the unfused path (gt_cov_stats.gc_pcent, src/print_vcf.c:394-398): the depth of a position is
 * the sum of its gt_meth counts, its genome position the record's; the fused chain has its own form (bsc_gc_cov_kernel). */
extern "C" __global__ __launch_bounds__(1024) void bsc_gc_cov_gtm_kernel(const uint8_t *__restrict__ core, const uint8_t *__restrict__ gtm,
                                                                         uint32_t gtm_stride, uint32_t n,
                                                                         const uint8_t *__restrict__ gc_bins, uint32_t n_bins,
                                                                         uint32_t start_pos, unsigned long long *__restrict__ table) {
  __shared__ uint32_t t[256 * 101];
  for (unsigned i = threadIdx.x; i < 256 * 101; i += 1024) t[i] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < n; i += gridDim.x * 1024u) {
    const uint32_t pos = *reinterpret_cast<const uint32_t *>(core + (uint64_t)i * 64u);
    if (!pos) continue; /* the position did not reach the printer */
    const uint64_t *c = reinterpret_cast<const uint64_t *>(gtm + (uint64_t)i * gtm_stride);
    uint64_t d = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) d += c[j];
    const uint32_t dd = d < BSC_COV_CAP ? (uint32_t)d : BSC_COV_CAP - 1u;
    const uint32_t bn = (pos - start_pos) / 100u;
    if (bn >= n_bins) continue;
    const uint32_t g = gc_bins[bn];
    if (g > 100u) continue;
    if (dd < 256u) atomicAdd(&t[dd * 101u + g], 1u);
    else atomicAdd(&table[(uint64_t)dd * 101u + g], 1ull);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < 256 * 101; i += 1024)
    if (t[i]) atomicAdd(&table[i], (unsigned long long)t[i]);
}

extern "C" int bsc_dev_launch_gc_cov_gtm(const void *core, const void *gtm, uint32_t gtm_stride, uint32_t n, const void *gc_bins,
                                         uint32_t n_bins, uint32_t start_pos, void *table, int num_cus, void *stream) {
  if (n == 0) return 0;
  unsigned grid = (n + 1023u) / 1024u;
  if (grid > (unsigned)num_cus) grid = (unsigned)num_cus;
  hipLaunchKernelGGL(bsc_gc_cov_gtm_kernel, dim3(grid), dim3(1024), 0, (hipStream_t)stream, (const uint8_t *)core, (const uint8_t *)gtm,
                     gtm_stride, n, (const uint8_t *)gc_bins, n_bins, start_pos, (unsigned long long *)table);
  return (int)hipGetLastError();
}

extern "C" int bsc_dev_launch_site_stats(const void *core, const void *gtm, uint32_t gtm_stride, const void *dbsnp,
                                         uint32_t n, const void *tb, const void *logp, const void *carry_in,
                                         void *carry_out, void *stats, void *pairs, int num_cus, void *stream) {
  if (n == 0) return 0;
  const unsigned n_tiles = (n + 63u) / 64u;
  unsigned grid = (n_tiles + (SS_THREADS / 64) - 1u) / (SS_THREADS / 64);
  if (grid > (unsigned)num_cus) grid = (unsigned)num_cus; /* one 110-KB workgroup per CU */
  unsigned long long *words = (unsigned long long *)stats;
  double *meth = (double *)((char *)stats + offsetof(bsc_site_stats, CpG_ref_meth));
  hipLaunchKernelGGL(bsc_site_stats_kernel, dim3(grid), dim3(SS_THREADS), 0, (hipStream_t)stream, (const uint8_t *)core,
                     (const uint8_t *)gtm, gtm_stride, (const uint8_t *)dbsnp, n, (const bsc_dev_tables *)tb,
                     (const double *)logp, (const uint32_t *)carry_in, (uint32_t *)carry_out, words, meth,
                     (unsigned long long *)pairs);
  return (int)hipGetLastError();
}
