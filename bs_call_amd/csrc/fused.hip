/*
 * fused.hip — the print side as ONE pass over the pile-ups: pile-up -> call -> VCF record -> site statistics, the
 * 200-byte gt_meth records never reaching HBM.  What the unfused chain does in three kernels and 630 bytes of traffic
 * per position (bsc_call_kernel 105 in + 201 out, bsc_vcf_core_kernel 202 in + 64 out, bsc_site_stats_kernel 136 in) is
 * done here in 105 + 64.
 *
 *   bsc_chain_kernel_t     one wave per RUN of consecutive tiles.  The printer's record of position i needs the called
 *                          genotypes of i-2 .. i+2 (src/print_vcf.c:548-594): a tile computes 64 consecutive sites and
 *                          forms the records of all but the last two (their right neighbours are the next tile's);
 *                          the genotypes of the two sites before a tile are the previous tile's (two words carried in
 *                          the wave's LDS), so a tile advances by 62 — 97 % of the lanes productive — except the
 *                          first tile of a run, which has nobody to inherit from and spends its first two lanes on
 *                          the left halo (60 records).  Every neighbour's genotype is in the wave's own LDS: no
 *                          exchange between waves, no second pass; and the reads-in form meets the reads of one tile
 *                          again in the next, a few microseconds later in the same CU's caches.  Per lane:
 *                          the calling kernel's statements (call_body.inc, shared textually with kernels.hip), then
 *                          the record formation of _print_vcf_entry (:32-381; the restatement is vcfcore.hip's), then
 *                          the statistics block (:382-526; sitestats.hip's), each lane's 64-byte bsc_vcf_core staged
 *                          in the wave's slot and written with 16-byte stores.
 *                          Heterozygous calls (one position in ~1 000) need Fisher's exact test on the strand table
 *                          (src/call_genotypes.c:61-108), a divergent walk: each wave lists its own in HBM and, after its
 *                          last tile, tests them 64 at a time, patches FS / FILTER into the records it wrote and adds
 *                          those positions' statistics (nothing about a heterozygous position depends on, or is needed
 *                          by, its neighbours' statistics: "CG" status needs the homozygous CC / GG pair).  CpG
 *                          cytosines are counted per (informative counts a, b) — LDS table, 512 x 512 table in HBM, a
 *                          list beyond that — and turned into methylation profiles when the statistics are read
 *                          (bsc_meth_eval_kernel, sitestats.hip).
 *   bsc_gc_cov_kernel      GC content by coverage (src/print_vcf.c:394-398) from the per-position depths the main kernel
 *                          leaves when the contig's GC bins are set (bsc_set_gc_bins).
 *
 * A call handles one WINDOW of a block (the reference's unit: a maximal run of overlapping templates; the printer's
 * sliding-window state is flushed at its end): the windows of a block give exactly the records of the whole block,
 * which is what lets a contig be walked in fixed 4 Mi-position windows.  The window's buffers
 * carry up to 2 positions of pile-ups and up to 4 / 2 reference codes of context either side.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "accdev.h"
#include "bsmath.h"
#include "callmath.h"
#include "devtables.h"
#include "sitestats_dev.h"

#ifndef FW
#define FW 16 /* waves per workgroup: one 1024-thread workgroup per CU (the statistics histogram lives in its LDS) */
#endif
#ifndef BSC_CHAIN_STAGGER
#define BSC_CHAIN_STAGGER 0 /* A/B variant: wave w of a workgroup starts w * this many 64-cycle sleeps late */
#endif
#define FT 60  /* records of the first tile of a run (64 sites computed, two of halo either side) */
#define FT2 62 /* records of every further tile of a run: the left halo is the previous tile's (carried) */
#ifndef BSC_RUN_CAP
#define BSC_RUN_CAP 32 /* longest run of tiles the launcher gives a wave at a time (BSC_CHAIN_RUN_CAP overrides: experiments) */
#endif
#define SUM_DW 12 /* dwords of a site summary (accumulate.hip, summary form): per class its count | the forward-strand part of it << 16
                     (8), the packed mean qualities (2), aq | mq << 16, n — 48 bytes: a tile's 64 rows are 3 KB, every tile's first
                     row on a 16-byte boundary.  Counts beyond 16 bits: counters[BSC_CNT_DEEP], and the reads-in twin runs instead */
#define F_COV_LDS 256 /* coverage rows of the statistics histogram kept in LDS (deeper positions: global atomics) */
#define F_WORDS (SS_COV + F_COV_LDS * 6)
#define F_PAIR 32 /* (a, b) < F_PAIR: CpG cytosines counted in the workgroup's LDS pair table; up to SS_PAIR_G: global */

struct bsc_chain_args {
  uint32_t x;       /* genome position (1-based) of the block's first position */
  uint32_t n_block; /* positions in the block */
  uint32_t first;   /* block-relative index of the window's first position */
  uint32_t n;       /* positions in the window */
  uint32_t lc, rc;  /* positions of pile-up context in the buffers left / right of the window (0..2) */
  uint32_t lr;      /* reference codes in the buffer left of the window (0..4) */
  /* the launch's records are cut into n_runs runs, run k for wave k mod (number of waves): the first run_extra runs have
   * run_tiles + 1 tiles, the others run_tiles; a run of t tiles forms FT + (t - 1) FT2 consecutive records, run 0 from window
   * index `origin` on */
  uint32_t origin, n_runs, run_tiles, run_extra;
  /* MULTI (several blocks in one launch, bsc_blocks_records): this struct describes ONE SEGMENT — a block's head, main part or
   * tail, always with first = 0, n = n_block, no context — and the launch runs the segments' runs one after another: */
  uint32_t run0;    /* the segment's first run among the launch's */
  uint32_t ref_off; /* where the block's reference codes start in the reference buffer */
  uint32_t pos_off; /* index of the block's first position in the per-position arrays (records, second halves, dbSNP flags,
                       depths): a multiple of 64 */
  uint32_t bin0, bin_end; /* the block's bins in bin_off[]: [bin0, bin_end) */
  uint32_t first_block, last_block; /* the launch's first block takes the context's CpG carry, its last one leaves it */
  int32_t all_positions;
  uint32_t reg_start, reg_stop;
  int32_t with_stats;
  uint32_t ovf_cap;
  uint32_t het_cap; /* entries of a wave's heterozygous list */
  uint32_t depth_off; /* != 0: the window's depths (u16 per position, 0 = no record formed) are written at het_list + this
                         many dwords, for the GC-by-coverage kernel */
  uint32_t edge_gap;  /* guarded single-block launches: records between run 0 and run 1 that belong to another launch — a
                         block's head tile and the tiles behind its main part as ONE launch (chain_launch_t) */
  uint32_t run_if;    /* 0: run; 1: only if counters[BSC_CNT_DEEP] != 0; 2: only if it is 0 (devtables.h: bsc_chain_launch.run_if) */
};

/* the reads-in form (READS = true): the block's ordered reads instead of pile-ups (accdev.h), and where a heterozygous
 * call's strand counts wait for Fisher's test (the pile-up they come from exists only in the wave's LDS) */
#define F_HET_DW 20 /* dwords per listed heterozygous call: index | max_gt << 28, counts[0][0..7], counts[0] + counts[1], mq, 2 spare */
struct bsc_reads_args {
  const bsc_read_desc *rd;   /* the block's live reads, grouped by the 64-position bin of their first base (accumulate.hip) */
  const uint32_t *bin_off;   /* n_bins + 1: index of each bin's first read; [n_bins] = number of live reads */
  const uint8_t *seq;
  uint32_t *f_scratch;       /* per wave 64 x 8 dwords: forward counts of a tile in which some count exceeds a byte */
  uint32_t n_bins, min_qual;
};

/*
 * The kernel's one parameter.  The first members are read where the kernel starts and stay in scalar registers; the ones
 * after `a` are wanted once per tile or less (a pointer for a rare statistics row, the region, the reads-in arguments during
 * the pile-up phase only), and kept live across the tile loop they were what the register allocator parked in VGPR lanes —
 * a v_readlane per use, on the VALU, which is this kernel's bottleneck (profiles/r02_e_chain_sq_counters.txt: 108 spilled
 * SGPRs).  K_COLD reads such a member from the kernel-argument segment where it is used, through a pointer the compiler
 * cannot see through (so it cannot hoist the load out of the loop either): one s_load on the otherwise idle scalar
 * memory path.
 */
struct bsc_chain_kargs {
  const uint32_t *cts;
  const uint8_t *ref;
  uint8_t *core_p; /* the records (the kernel's `core_out`) */
  const bsc_dev_tables *tb;
  bsc_chain_args a;
  double l, t, lrb, lrb1; /* 1 - under_conv, over_conv, log(ref_bias), log(0.5 (1 + ref_bias)): calc_gt_prob's scalars (host-computed) */
  const uint8_t *dbsnp;
  uint32_t *het_list;
  unsigned long long *counters;
  const uint32_t *carry_in;
  uint32_t *carry_out;
  unsigned long long *stat_words, *pair_cells, *ovf_list;
  uint8_t *aux_out;
  uint8_t *emit_out;
  bsc_reads_args ra;
  /* MULTI: the launch's segments and, segment by segment, their first runs (n_segs + 1 entries: the last = all runs) */
  const bsc_chain_args *segs;
  const uint32_t *seg_run0;
  uint32_t n_segs;
};
typedef const __attribute__((address_space(4))) bsc_chain_kargs *bsc_kargs_p;
#define K_COLD(f)                                                                    \
  ({                                                                                 \
    bsc_kargs_p p_ = (bsc_kargs_p)__builtin_amdgcn_kernarg_segment_ptr();            \
    asm volatile("" : "+s"(p_));                                                     \
    p_->f;                                                                           \
  })

/* the reads-in arguments, member by member (a struct cannot be copied out of the constant address space as a whole) */
#define K_LOAD_RA(ra)                                                                \
  bsc_reads_args ra;                                                                 \
  {                                                                                  \
    bsc_kargs_p p_ = (bsc_kargs_p)__builtin_amdgcn_kernarg_segment_ptr();            \
    asm volatile("" : "+s"(p_));                                                     \
    ra.rd = p_->ra.rd;                                                               \
    ra.bin_off = p_->ra.bin_off;                                                     \
    ra.seq = p_->ra.seq;                                                             \
    ra.f_scratch = p_->ra.f_scratch;                                                 \
    ra.n_bins = p_->ra.n_bins;                                                       \
    ra.min_qual = p_->ra.min_qual;                                                   \
  }                                                                                  \
  acc_reads R_;                                                                      \
  R_.rd = ra.rd;                                                                     \
  R_.bin_off = ra.bin_off + (MULTI ? a.bin0 : 0u);                                   \
  R_.seq = ra.seq;                                                                   \
  R_.n_bins = ra.n_bins;                                                             \
  R_.x = a.x;

struct bsc_vcf_core_f {
  uint32_t pos;
  uint8_t emit, gt, ref_code, gt_enc, flt, phred, n_gl;
  char cg;
  char alt[2];
  char cx_ref[5];
  char cx_gt[5];
  int32_t fs;
  uint32_t qd;
  uint32_t dp;
  float gl[6];
  uint32_t _pad;
};
static_assert(sizeof(bsc_vcf_core_f) == 64, "bsc_vcf_core is 64 bytes");

/*
 * The same LDS-DMA as callmath.h's dma16(), issued through inline assembly so that the compiler's wait-count insertion
 * does not see it.  Why: for an LDS-DMA it cannot tell which LDS array is written, so it puts `s_waitcnt vmcnt(0)` in
 * front of EVERY later LDS operation — the histogram updates that are meant to run while the next tile's pile-ups are in
 * flight would each wait for them.  Hidden operations can only make compiler-computed waits longer, never too short
 * (the counter retires in order); the wait that matters is the explicit vmcnt(0) at the top of the next tile.
 */
__device__ static __forceinline__ void dma16_hidden(const void *g, void *lds_wave_base) {
  const uint32_t l = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void *)lds_wave_base;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off nt" : : "v"(g), "s"(l) : "memory", "m0");
}

#define WAVE_LDS_SYNC() WAVE_LDS_ORDER() /* callmath.h */

/* "NACGT"[code] for code 0..4 (v_perm_b32: selector 0..3 picks a byte of the second operand, 4..7 of the first) */
#define F_BASE_CHAR(code) (__builtin_amdgcn_perm(0x00000054u, 0x4743414Eu, (code)) & 0xffu)

/* genotype -> its two alleles as base codes 1..4 (AA AC AG AT CC CG CT GG GT TT) */
__device__ static __forceinline__ void f_alleles(int g, int &a, int &b) {
  a = g < 4 ? 1 : (g < 7 ? 2 : (g < 9 ? 3 : 4));
  b = g < 4 ? g + 1 : (g < 7 ? g - 2 : (g < 9 ? g - 4 : 4));
}

/* mac1 (src/print_vcf.c:191-214): either allele of a heterozygous call supported by at most one base */
__device__ static __forceinline__ bool f_mac1(int gt, const uint32_t c[8]) {
  const uint64_t sA = (uint64_t)c[0] + c[4], sC = (uint64_t)c[1] + c[5] + c[7], sG = (uint64_t)c[2] + c[6] + c[4],
                 sT = (uint64_t)c[3] + c[7];
  switch (gt) {
    case 1: return sC <= 1 || sA <= 1;                                       /* AC */
    case 2: return (uint64_t)c[2] + c[6] <= 1 || c[0] <= 1;                  /* AG */
    case 3: return sT <= 1 || sA <= 1;                                       /* AT */
    case 5: return sG <= 1 || sC <= 1;                                       /* CG */
    case 6: return c[3] <= 1 || (uint64_t)c[1] + c[5] <= 1;                  /* CT */
    case 8: return sT <= 1 || sG <= 1;                                       /* GT */
    default: return false;
  }
}

/* h[idx]++ for the lanes with `on`: the first such lane's bin is bumped once for every lane that shares it (one value
 * usually dominates), the others go individually.  Whole waves call this (wave-uniform control flow). */
__device__ static __forceinline__ void f_hist_peel1(uint32_t *h, bool on, uint32_t idx, unsigned lane) {
  const unsigned long long m = __ballot(on);
  if (m) {
    const int src = __builtin_ctzll(m);
    const uint32_t v = (uint32_t)__builtin_amdgcn_readlane(idx, src);
    const unsigned long long same = __ballot(on && idx == v);
    if ((int)lane == src) atomicAdd(&h[v], (uint32_t)__popcll(same));
    if (on && idx != v) atomicAdd(&h[idx], 1u);
  }
}

/* per-lane facts of one position for the statistics (what bsc_site_stats_kernel derives from the records) */
struct f_facts {
  bool called, emit, pass, het, rs, cpg_site, ref_cpg, pair, pair_pass, fs_ok, do_meth;
  uint32_t phred, flt, qd, fsv, mqv, cdp, cinf, m_a, m_b;
  int mut;
};

/* the histogram updates of src/print_vcf.c:386-525 for 64 positions at once (wave-uniform control flow), as the
 * heterozygous positions need them (never a CpG cytosine: the methylation part is in the main kernel only) */
__device__ static __forceinline__ void f_stats_update(uint32_t *h, const f_facts &F, unsigned long long *stat_words) {
#define F_COV_ADD(on, row, col)                                                                        \
  do {                                                                                                 \
    ss_hist_add<1>(h, (on) && (row) < F_COV_LDS, SS_COV, (row) * 6u + (col));                           \
    if ((on) && (row) >= F_COV_LDS) atomicAdd(&stat_words[SS_COV + (uint64_t)(row) * 6u + (col)], 1ull); \
  } while (0)
  F_COV_ADD(F.called, F.cdp, 0u);
  F_COV_ADD(F.emit, F.cdp, 1u);
  ss_hist_add<1>(h, F.emit, SS_MISC + 0, 0u);
  ss_hist_add<1>(h, F.emit && F.pass, SS_MISC + 1, 0u);
  ss_hist_add<2>(h, F.emit, SS_QUAL + 1 * 256, F.phred); /* qual[variant_sites]; the workgroup copies it to [all_sites] at the end */
  ss_hist_add<2>(h, F.emit, SS_FST + 0 * 512, F.qd * 2u + F.het);
  ss_hist_add<2>(h, F.emit && F.fs_ok, SS_FST + 1 * 512, F.fsv * 2u + F.het);
  ss_hist_add<2>(h, F.emit, SS_FST + 2 * 512, F.mqv * 2u + F.het);
  ss_hist_add<2>(h, F.emit, SS_FILT, (F.het ? 32u : 0u) + (F.flt & 31u));
  if (__any(F.rs)) {
    ss_hist_add<1>(h, F.rs, SS_MISC + 6, 0u, SS_MISC + 8);
    ss_hist_add<1>(h, F.rs && F.pass, SS_MISC + 7, 0u, SS_MISC + 9);
  }
  if (__any(F.cpg_site)) {
    ss_hist_add<1>(h, F.pair, SS_MISC + 10, F.ref_cpg ? 0u : 2u);
    ss_hist_add<1>(h, F.pair_pass, SS_MISC + 11, F.ref_cpg ? 0u : 2u);
    ss_hist_add<1>(h, F.cpg_site, SS_QUAL + 2 * 256, (F.ref_cpg ? 0u : 256u) + F.phred);
    F_COV_ADD(F.cpg_site, F.cdp, F.ref_cpg ? 2u : 3u);
    F_COV_ADD(F.cpg_site, F.cinf, F.ref_cpg ? 4u : 5u);
  }
  if (__any(F.mut != 12)) {
    const bool m = F.mut != 12;
    ss_hist_add<1>(h, m, SS_MUT, (uint32_t)F.mut * 2u);
    ss_hist_add<1>(h, m && F.pass, SS_MUT + 1, (uint32_t)F.mut * 2u);
    if (__any(m && F.rs)) {
      ss_hist_add<1>(h, m && F.rs, SS_DBMUT, (uint32_t)F.mut * 2u);
      ss_hist_add<1>(h, m && F.rs && F.pass, SS_DBMUT + 1, (uint32_t)F.mut * 2u);
    }
  }
#undef F_COV_ADD
}

/*
 * The heterozygous calls a wave has collected (lane k < n_pend holds one): Fisher's exact test on the strand table
 * (src/call_genotypes.c:61-108, src/stats_utils.c:25-91), FS and the FILTER bits that depend on it patched into the
 * record the wave stored earlier, the position's statistics.  The whole wave calls this (wave-uniform control flow);
 * the records are read back from memory, so the caller has waited for its record stores (vmcnt).
 */
template <bool READS, bool SUMM>
__device__ static __forceinline__ void f_fisher_pending(uint32_t pend_e, unsigned n_pend, unsigned lane, const uint32_t *__restrict__ cts,
                                                        const uint8_t *__restrict__ dbsnp, const bsc_chain_args &a,
                                                        uint8_t *__restrict__ core_out, const double *__restrict__ s_lf, const double *s_logtab,
                                                        const unsigned long long *s_exptab, uint32_t *h,
                                                        unsigned long long *__restrict__ stat_words, uint8_t *__restrict__ emit_p) {
  f_facts F;
  F.called = F.emit = F.pass = F.het = F.rs = F.cpg_site = F.ref_cpg = F.pair = F.pair_pass = F.fs_ok = F.do_meth = false;
  F.phred = F.flt = F.qd = F.fsv = F.mqv = F.cdp = F.cinf = F.m_a = F.m_b = 0;
  F.mut = 12;
  if (lane < n_pend) {
    const uint32_t i = pend_e & 0x0fffffffu;
    const unsigned mxi = pend_e >> 28;
    /* READS: cts is the lane's entry of the wave's list (F_HET_DW dwords), else the block's pile-ups in HBM */
    const uint32_t *p = READS ? cts : cts + (uint64_t)(i + a.lc) * (SUMM ? SUM_DW : IN_DW);
    uint32_t f[8], r[8], c[8];
#pragma unroll
    for (int j = 0; j < 8; j++) { /* the list entry was written by this wave: read past the vector L1 */
      f[j] = READS ? __hip_atomic_load(&p[1 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (SUMM ? p[j] >> 16 : p[j]);
      c[j] = READS ? __hip_atomic_load(&p[9 + j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : (SUMM ? p[j] & 0xffffu : f[j] + p[8 + j]);
      r[j] = c[j] - f[j];
    }
    int t0, t1, t2, t3;
    strand_table(mxi, f, r, t0, t1, t2, t3);
    double z = fisher_dev(t0, t1, t2, t3, s_lf, s_logtab, (const uint64_t *)s_exptab);
    if (z < 1.0e-20) z = 1.0e-20;
    const double fisher = bsm_log_t(z, s_logtab) / BSM_LN10; /* gt_meth.fisher_strand (:105-107) */
    const int fs = (int)(-fisher * 10.0 + 0.5);                /* src/print_vcf.c:151 */
    uint8_t *rec = core_out + (uint64_t)i * 64u;
    const uint4 c0 = *reinterpret_cast<const uint4 *>(rec);
    if (c0.x != 0) { /* the position reached _print_vcf_entry with depth > 0 */
      const bool emit = (c0.y & 0xffu) != 0;
      const int gt = (int)((c0.y >> 8) & 0xffu), rfix = (int)((c0.y >> 16) & 0xffu);
      const uint32_t phred = (c0.z >> 8) & 0xffu;
      const uint32_t qd = *reinterpret_cast<const uint32_t *>(rec + 28);
      /* mq as the calling statements form it (call_body.inc; src/call_genotypes.c:59) */
      int mq;
      if (READS) mq = (int)__hip_atomic_load(&p[17], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      else if (SUMM) mq = (int)(p[10] >> 16);
      else {
        const uint32_t n_reads = p[16];
        const float mapq2 = __uint_as_float(p[25]);
        mq = (int)(0.5 + sqrt((double)(mapq2 / (float)n_reads)));
      }
      int ga, gb;
      f_alleles(gt, ga, gb);
      const bool het = ga != gb; /* the printer's genotype; max_gt differs from it only in a rounding tie */
      uint32_t flt = 0;
      *reinterpret_cast<int32_t *>(rec + 24) = fs;
      if (emit) {
        if (phred < 20) flt |= 1;
        if (qd < 2) flt |= 2;
        if (fs > 60) flt |= 4;
        if (mq < 40) flt |= 8;
        if (!flt && het && f_mac1(gt, c)) flt |= 128;
        rec[8] = (uint8_t)flt;
        if (emit_p) { /* the record's BCF2 length (see the tile's own computation of it): FS and the FILTER bits it decides are known now, so the
                         encoder's size pass need not look at this record either — unless it carries a name */
          uint32_t cm = c[0] > c[1] ? c[0] : c[1], n_amq = 0;
#pragma unroll
          for (int k = 2; k < 8; k++) cm = c[k] > cm ? c[k] : cm;
#pragma unroll
          for (int k = 0; k < 8; k++) n_amq += c[k] > 0u ? 1u : 0u;
#define F_PI(v) ((v) <= 127u ? 2u : ((v) <= 32767u ? 3u : 5u)) /* put_int of a non-negative value */
          const uint32_t fb = flt & 15u, ngl = (c0.z >> 16) & 0xffu, dp1 = *reinterpret_cast<const uint32_t *>(rec + 32);
          const uint32_t ft_len = ((fb & 1u) ? 4u : 0u) + ((fb & 2u) ? 4u : 0u) + ((fb & 4u) ? 5u : 0u) + ((fb & 8u) ? 5u : 0u) + (uint32_t)__popc(fb) - 1u;
          const uint32_t ft = fb ? (ft_len >= 15u ? 3u : 1u) + ft_len : 5u;
          const uint32_t cs = ((0x209u >> gt) & 1u) ? 2u : (((0x72u >> gt) & 1u) + ((0x1A4u >> gt) & 1u));
          const uint32_t len = 32u + 13u + ((c0.w & 0xffu) ? 2u : 0u) + ((c0.w & 0xff00u) ? 2u : 0u) + 5u + 2u + ft + 2u + F_PI(dp1) + 2u + F_PI((uint32_t)mq) + 2u + F_PI(phred) +
                               2u + F_PI(qd) + 3u + 4u * ngl + 3u + 8u * (cm <= 127u ? 1u : (cm <= 32767u ? 2u : 4u)) + (n_amq ? 3u + n_amq : 0u) + 3u + cs + 4u + 8u +
                               (((0x16Eu >> gt) & 1u) ? 2u + F_PI((uint32_t)fs) : 0u);
#undef F_PI
          const bool named = dbsnp ? dbsnp[i] != 0 : false;
          emit_p[i] = (uint8_t)((named || len > 254u || fs < 0 || gt > 9 || ngl > 6u) ? 255u : len);
        }
      }
      const uint32_t d_inf = c[4] + c[5] + c[6] + c[7], dpt = c[0] + c[1] + c[2] + c[3] + d_inf;
      F.called = true;
      F.emit = emit;
      F.flt = flt;
      F.phred = phred;
      F.qd = qd > 255u ? 255u : qd;
      F.fs_ok = fs >= 0;
      F.fsv = (uint32_t)(fs > 255 ? 255 : (fs < 0 ? 0 : fs));
      F.mqv = (uint32_t)(mq < 0 ? 0 : (mq > 255 ? 255 : mq));
      F.cdp = dpt < BSC_COV_CAP ? dpt : BSC_COV_CAP - 1u;
      if (emit) {
        F.pass = flt == 0;
        F.rs = dbsnp ? dbsnp[i] != 0 : false;
        F.het = het;
        F.mut = ss_mut_type(gt, rfix);
      }
    }
  }
  if (a.with_stats) f_stats_update(h, F, stat_words);
}

typedef const __attribute__((address_space(4))) uint32_t *f_cptr;

/* FULL: the launch's tiles are complete (no bounds checks).  What a tile starts from: READS — the block's reads (the wave piles
 * them up itself); SUMM — site summaries, 48 bytes per position (the accumulate kernel's summary form: counts and the per-site
 * summary of src/call_genotypes.c:44-59 already made); neither — pile-ups, 104 bytes per position.  MULTI: several blocks. */
template <bool FULL, bool READS, bool MULTI, bool SUMM>
__global__ __launch_bounds__(64 * FW, FW / 4) void bsc_chain_kernel_t(const bsc_chain_kargs K) {
  constexpr unsigned ROW_DW = SUMM ? SUM_DW : IN_DW; /* dwords per position of the input array */
  const bsc_dev_tables *__restrict__ const tb = K.tb; /* the set-up below only */
  /* MULTI: the segment the wave's current run belongs to, in scalar registers (F_RUN_SETUP); until the first run the launch-wide
   * members of the kernel's own copy (with_stats ...) */
  bsc_chain_args a_m = K.a;
  const bsc_chain_args &a = MULTI ? a_m : K.a;
/* a member that differs from segment to segment (the others are the same in every segment and in K.a) */
#define A_COLD(f) (MULTI ? a.f : K_COLD(a.f))
/* the three arrays every tile touches: read from the argument segment where used, like the cold members */
#define cts K_COLD(cts)
#define ref (K_COLD(ref) + (MULTI ? a.ref_off : 0u))
#define core_base K_COLD(core_p)
#define core_out (K_COLD(core_p) + (MULTI ? (uint64_t)a.pos_off * 64u : 0ull))
  __shared__ __attribute__((aligned(16))) uint32_t lds_slot[FW][SLOT_DW];
  __shared__ double s_k[44], s_lnk[44], s_half[44], s_one[44];
  __shared__ double s_logtab[256];
  __shared__ unsigned long long s_exptab[256];
  __shared__ double s_pthr[64 * 4];   /* QUAL without the log: per binade of om = 1 - z its four steps ... */
  __shared__ uint8_t s_pbase[64];     /* ... and the value below them (devtables.h; ln x! of Fisher's test is read from memory) */
  __shared__ double s_ptab[PT_WORDS]; /* logs of the methylation arguments of a class whose partner class is empty (callmath.h) */
  __shared__ __attribute__((aligned(16))) double s_prior[5 * 10]; /* the priors of the ten genotypes by reference code (src/genotype_model.c:87-108) */
  __shared__ unsigned int s_cnt[12];  /* covered, hist[10], het */
  __shared__ uint8_t s_pairs[FW][256]; /* per wave: the (lane, class) pairs whose logs are needed (call_body.inc) */
  __shared__ uint32_t s_gw[FW][66];    /* per wave, entry 2 + lane: the printer's called genotype + 1 (0 = none) of the lane's
                                          site | its IUPAC letter << 8 | (carries C) << 16 | (carries G) << 17; entries 0, 1:
                                          the two sites before the tile, carried over from the run's previous tile */
  __shared__ uint16_t s_rf[FW][68];    /* per wave: reference code | its letter << 8, from 2 before the first computed site */
  __shared__ uint16_t s_pend[FW][66];  /* per wave, entry 2 + lane (entry 1: carried like s_gw's): bit 0 = a written '+' strand
                                          CG call (the pending cytosine of src/print_vcf.c:447-455), bits 8.. = its FILTER bits */
  __shared__ uint32_t h[F_WORDS];      /* statistics histogram of the workgroup (sitestats_dev.h) */
  __shared__ uint32_t s_pair[4 * F_PAIR * F_PAIR]; /* CpG cytosines per [ref / non-ref][all / passed][a][b] */

  if (K.a.run_if) { /* one of a pair of launches behind the accumulate kernel's summary form: does this one do the work? */
    const bool deep = K_COLD(counters)[BSC_CNT_DEEP] != 0;
    if ((K.a.run_if == 1u) != deep) return;
  }
  const unsigned tid = threadIdx.x;
  const unsigned lane0 = tid & 63u;
  const unsigned wid = __builtin_amdgcn_readfirstlane(tid >> 6); /* wave-uniform: the per-wave LDS bases live in scalar registers */
  if (tid < 44) {
    s_k[tid] = tb->k[tid];
    s_lnk[tid] = tb->ln_k[tid];
    s_half[tid] = tb->ln_k_half[tid];
    s_one[tid] = tb->ln_k_one[tid];
  }
  if (tid < 256) {
    s_logtab[tid] = tb->log_tab[tid];
    s_exptab[tid] = tb->exp_tab[tid];
    s_pthr[tid] = (&tb->phred_thr[0][0])[tid];
  }
  if (tid < 64) s_pbase[tid] = tb->phred_base[tid];
  if (tid < 12) s_cnt[tid] = 0;
  if (a.with_stats) {
    for (unsigned i = tid; i < F_WORDS; i += 64 * FW) h[i] = 0;
    for (unsigned i = tid; i < 4 * F_PAIR * F_PAIR; i += 64 * FW) s_pair[i] = 0;
  }
  __syncthreads();
  if (tid < PT_WORDS) s_ptab[tid] = pure_log_entry(tid, K_COLD(l), K_COLD(t), s_k, s_logtab); /* callmath.h PT_* */
  if (tid >= 256 && tid < 306) { /* log(ref_bias) on the reference's homozygote, log(0.5 (1 + ref_bias)) on the heterozygotes that carry it; N: none */
    const unsigned r = (tid - 256u) / 10u, g = (tid - 256u) % 10u;
    int ga, gb;
    f_alleles((int)g, ga, gb);
    s_prior[tid - 256u] = r == 0 ? 0.0 : (ga == (int)r && gb == (int)r ? K_COLD(lrb) : ((ga == (int)r || gb == (int)r) ? K_COLD(lrb1) : 0.0));
  }
  __syncthreads();
  if (BSC_CHAIN_STAGGER)
    for (unsigned i = 0; i < wid; i++) __builtin_amdgcn_s_sleep(BSC_CHAIN_STAGGER);

  uint32_t *slot = lds_slot[wid];
  uint32_t *sg = s_gw[wid];
  uint16_t *srf = s_rf[wid];
  uint16_t *spd = s_pend[wid];
  /* the reference codes the buffer holds: block indices [first - lr, min(n_block + 2, first + n + 2)) */
  int64_t ref_lo, ref_hi;
#define F_SEG_CONSTS()                                                                                                   \
  do {                                                                                                                   \
    ref_lo = (int64_t)a.first - a.lr;                                                                                    \
    ref_hi = ((int64_t)a.first + a.n + 2 < (int64_t)a.n_block + 2) ? (int64_t)a.first + a.n + 2 : (int64_t)a.n_block + 2; \
  } while (0)
  F_SEG_CONSTS();

/* the tile's 64 pile-ups (6 656 contiguous bytes, the first on a 16-byte boundary) -> the wave's slot by LDS-DMA */
#define F_DMA_TILE(JW0, DMA)                                                                                            \
  do {                                                                                                                  \
    const char *src_ = reinterpret_cast<const char *>(cts + (uint64_t)((int32_t)(JW0) + (int32_t)A_COLD(lc)) * ROW_DW) + lane * 16; \
    constexpr int full_ = (int)(64u * ROW_DW * 4u / 1024u); /* 6 656 bytes = 6.5 KB of pile-ups, 5 632 = 5.5 KB of summaries */ \
    _Pragma("unroll") for (int j_ = 0; j_ < full_; j_++) DMA(src_ + j_ * 1024, slot + j_ * 256);                        \
    constexpr unsigned rem_ = (64u * ROW_DW * 4u % 1024u) / 16u; /* lanes of the last, partial kilobyte */             \
    if (rem_ && lane < rem_) DMA(src_ + full_ * 1024, slot + full_ * 256);                                               \
  } while (0)
  /* ---- the wave's runs: run k = wave index + a multiple of the number of waves; tile tj of a run starts 62 tj sites
   * after the run's first computed site (all wave-uniform, scalar registers) ---- */
  const uint32_t n_waves = gridDim.x * FW;
  uint32_t run_p = FT + (a.run_tiles - 1u) * FT2; /* records of a short run */
  const uint32_t n_runs_all = MULTI ? (uint32_t)((f_cptr)(uintptr_t)K_COLD(seg_run0))[K_COLD(n_segs)] : a.n_runs;
  uint32_t run = blockIdx.x * FW + wid, tj = 0, run_n = 0;
  int32_t run_s = 0; /* window index of the run's first record */
  uint32_t acc_live = 0; /* READS: the reads from here on belong to other blocks (MULTI) / do not exist */
  /* MULTI: the segment of run k — the last one whose first run is not behind k (binary search in the launch's table) — member by
   * member into the scalar registers (a struct cannot be copied out of the constant address space as a whole) */
#define F_SEG_LOAD(k)                                                                                           \
  do {                                                                                                          \
    const f_cptr r0_ = (f_cptr)(uintptr_t)K_COLD(seg_run0);                                                     \
    uint32_t lo_ = 0, hi_ = K_COLD(n_segs);                                                                     \
    while (hi_ - lo_ > 1u) {                                                                                    \
      const uint32_t mid_ = (lo_ + hi_) >> 1;                                                                   \
      if (r0_[mid_] <= (k)) lo_ = mid_;                                                                         \
      else hi_ = mid_;                                                                                          \
    }                                                                                                           \
    const f_cptr g_ = (f_cptr)(uintptr_t)(K_COLD(segs) + lo_);                                                  \
    uint32_t *d_ = reinterpret_cast<uint32_t *>(&a_m);                                                          \
    _Pragma("unroll") for (unsigned i_ = 0; i_ < sizeof(bsc_chain_args) / 4u; i_++) d_[i_] = g_[i_];             \
    F_SEG_CONSTS();                                                                                             \
    run_p = FT + (a.run_tiles - 1u) * FT2;                                                                      \
    if (READS) acc_live = ((f_cptr)(uintptr_t)K_COLD(ra.bin_off))[a.bin_end];                                   \
  } while (0)
#define F_RUN_SETUP(k)                                                                                          \
  do {                                                                                                          \
    if (MULTI) F_SEG_LOAD(k);                                                                                   \
    const uint32_t k_ = (k) - (MULTI ? a.run0 : 0u), ex_ = a.run_extra;                                         \
    run_n = a.run_tiles + (k_ < ex_ ? 1u : 0u);                                                                 \
    run_s = (int32_t)(a.origin + k_ * run_p + FT2 * (k_ < ex_ ? k_ : ex_));                                     \
    if (!FULL && !MULTI && k_ >= 1u) run_s += (int32_t)a.edge_gap;                                              \
  } while (0)
  if (run < n_runs_all) F_RUN_SETUP(run);
  /* heterozygous calls waiting for Fisher's test (window index | max_gt << 28): the wave's own list in HBM, one in ~1 000
   * positions; a.het_cap entries (F_HET_CAP), tested whenever fewer than 64 are free (the epochs below) */
#define F_WL() (K_COLD(het_list) + (uint64_t)(blockIdx.x * FW + wid) * K_COLD(a.het_cap) * (READS ? F_HET_DW : 1u))
  unsigned n_pend = 0; /* wave-uniform */
  unsigned inexact = 0;
  uint32_t acc_span = 0; /* READS: the longest read extent of the launch's blocks */
  if (READS) {
    K_LOAD_RA(ra);
    acc_span = (uint32_t)K_COLD(counters)[BSC_CNT_SPAN];
    if (!MULTI) acc_live = ra.bin_off[ra.n_bins];
  }
#ifndef BSC_CHAIN_NO_PREFETCH
  if (FULL && !READS && run < n_runs_all) { /* the wave's first tile; every later one is requested while its predecessor's statistics run */
    const unsigned lane = lane0;
    F_DMA_TILE(run_s - 2, dma16);
  }
#endif
  /* The tile loop runs in epochs: a wave goes on to its next tile as long as its list of heterozygous calls has room for a
   * whole tile's worth, then — and after its last tile — tests what it has listed.  A list of F_HET_CAP entries is
   * filled once in ~8 000 tiles on WGBS data, so an epoch is normally the whole launch; on input where most calls are
   * heterozygous the test simply runs more often.  (Sized for the worst case instead, the reads-in form's 80-byte entries
   * took 80 bytes of address space per position.) */
  const uint32_t het_room = K_COLD(a.het_cap) - 64u;
  for (;;) {
  n_pend = 0;
  while (run < n_runs_all && n_pend <= het_room) {
    /* the lane number, made opaque once per tile: otherwise every lane-dependent address of the loop body is hoisted
     * out of it, kept alive across the whole kernel and — at 128 VGPRs — spilled to scratch (a vector-memory round trip
     * per use instead of one VALU instruction) */
    unsigned lane = lane0;
    asm volatile("" : "+v"(lane));
    const uint32_t f0 = tj ? 0u : 2u;                     /* first lane that forms a record: the first tile of a run has two of left halo */
    const int32_t jw0 = run_s + (int32_t)(tj * FT2) - 2;  /* window-relative index of the site lane 0 computes */
    /* the wave's next tile (the next of the run, or the first of its next run), if any */
    const bool last_of_run = tj + 1u == run_n;
    const bool have_next = !last_of_run || run + n_waves < n_runs_all;
    int32_t jw0_next = jw0 + (int32_t)FT2;
    if (!MULTI && last_of_run && have_next) { /* (only the pile-up-in form looks a tile ahead, and that one is never MULTI) */
      const uint32_t k_ = run + n_waves, ex_ = a.run_extra;
      jw0_next = (int32_t)(a.origin + k_ * run_p + FT2 * (k_ < ex_ ? k_ : ex_)) - 2;
    }
    const int32_t jw = jw0 + (int32_t)lane;
    /* the site is in the buffers; READS: it is a position of the block (every one can be piled up from the reads) */
    const bool valid = FULL || (READS ? ((int64_t)a.first + jw >= 0 && (int64_t)a.first + jw < (int64_t)a.n_block)
                                      : (jw >= -(int32_t)a.lc && jw < (int32_t)(a.n + a.rc)));
    const bool inner = lane >= f0 && lane < 62u && (FULL || (jw >= 0 && jw < (int32_t)a.n));
    /* Everything per lane is relative to the tile: "lane index" L = block index - b0, b0 = block index of lane 0's
     * site (wave-uniform, 64-bit, in scalar registers); the block occupies lane indices blk_lo .. blk_hi (clamped far
     * outside the tile where the block's ends are not near). */
    const int64_t b0 = (int64_t)a.first + jw0;
    const int32_t blk_lo = (int32_t)(-b0 < -4096 ? -4096 : (-b0 > 4096 ? 4096 : -b0));
    const int64_t to_last = (int64_t)a.n_block - 1 - b0;
    const int32_t blk_hi = (int32_t)(to_last < -4096 ? -4096 : (to_last > 4096 ? 4096 : to_last));
    const uint32_t pos0 = a.x + (uint32_t)b0; /* genome position of lane index 0 (mod 2^32; only inner lanes use it) */

    /* ---- reference codes of lane indices -2 .. 63 + 2 -> srf[0 .. 67] ---- */
    bool tile_n; /* wave-uniform: an N (or the end of the buffer) among them — the printer's context blanking (:570-577) may then
                  * turn a lane's own reference base into N */
    {
      const int64_t r0 = b0 - 2 - ref_lo; /* offset of srf[0]'s code in the buffer */
      const int64_t avail = ref_hi - ref_lo;
      const uint8_t *const refp = ref;
      const int64_t k0 = r0 + lane;
      const uint32_t c0 = (k0 >= 0 && k0 < avail) ? refp[k0] : 0u;
      srf[lane] = (uint16_t)(c0 | (F_BASE_CHAR(c0) << 8));
      uint32_t c1 = 1u;
      if (lane < 4u) {
        const int64_t k1 = r0 + 64 + lane;
        c1 = (k1 >= 0 && k1 < avail) ? refp[k1] : 0u;
        srf[64u + lane] = (uint16_t)(c1 | (F_BASE_CHAR(c1) << 8));
      }
      tile_n = __any(c0 == 0u || c1 == 0u);
    }

    /* ---- my record ---- */
    uint32_t w[IN_DW];
    bool bigf_any = false; /* READS, wave-uniform: some forward count of the tile exceeds a byte */
    if (READS) {
      /* ---- HOT LOOP A for the tile's 64 sites (src/call_genotypes.c:180-226; accdev.h): the pile-up the calling
       * statements read is built in the wave's slot and never leaves it ---- */
      K_LOAD_RA(ra);
      const uint32_t q_span = ra.min_qual < 63u ? 63u - ra.min_qual : 0u; /* q counts iff min_qual <= q < 63 (src/call_genotypes.c:217) */
      /* the tile's first batch of candidate reads: in a run they are mostly the previous tile's (read a moment ago by this
       * very wave: cache hits), so nothing is requested a tile ahead — the descriptors kept live across the calling
       * statements for that were what the register allocator sent to scratch */
      const uint32_t acc_t0 = acc_tile_start(R_, b0, acc_span);
      uint32_t acc_kv;
      bsc_read_desc acc_d;
      acc_fetch(R_, acc_live, acc_t0, lane, acc_kv, acc_d);
      const uint32_t loff = b0 < 0 ? (uint32_t)(-b0) : 0u; /* lanes in front of the block's first position (0 .. 2) */
      const int64_t bl = b0 + 63 < (int64_t)a.n_block - 1 ? b0 + 63 : (int64_t)a.n_block - 1; /* last block index of the tile */
      const uint32_t pa = a.x + (uint32_t)(b0 + (int64_t)loff), p_last = a.x + (uint32_t)bl;
      uint32_t *row = slot + lane * IN_DW;
      const bool inx = acc_tile(R_, acc_live, lane, lane - loff, row, pa, p_last, (uint32_t)bl, ra.min_qual, q_span, acc_t0, acc_kv,
                                acc_d, w);
      uint32_t fmax = 0;
#pragma unroll
      for (int j = 0; j < 8; j++) fmax |= w[j];
      inexact |= (inx && lane >= f0 && lane < 62u) ? 1u : 0u;
      /* the forward-strand counts, which only Fisher's test of a heterozygous call needs again, wait in the two dwords of
       * the lane's slot area that the calling statements leave alone (la[12]), a byte each; a tile with a larger count
       * parks them in the wave's scratch lines in HBM instead */
      reinterpret_cast<uint2 *>(row)[12] = make_uint2(w[0] | (w[1] << 8) | (w[2] << 16) | (w[3] << 24),
                                                      w[4] | (w[5] << 8) | (w[6] << 16) | (w[7] << 24));
      bigf_any = __any(fmax > 255u);
      if (__builtin_expect(bigf_any, 0)) {
        uint4 *fs = reinterpret_cast<uint4 *>(ra.f_scratch + ((uint64_t)(blockIdx.x * FW + wid) * 64u + lane) * 8u);
        fs[0] = make_uint4(w[0], w[1], w[2], w[3]);
        fs[1] = make_uint4(w[4], w[5], w[6], w[7]);
      }
    } else if (FULL) {
#ifdef BSC_CHAIN_NO_PREFETCH
      F_DMA_TILE(jw0, dma16);
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the tile was requested before the loop / by the previous tile */
      const uint2 *rec = reinterpret_cast<const uint2 *>(slot + lane * ROW_DW);
#pragma unroll
      for (int i = 0; i < (int)ROW_DW / 2; i++) {
        const uint2 v = rec[i];
        w[2 * i] = v.x;
        w[2 * i + 1] = v.y;
      }
    } else {
      const uint32_t *const ctsp = cts;
#pragma unroll
      for (int i = 0; i < (int)ROW_DW; i++) w[i] = valid ? ctsp[(uint64_t)(jw + (int32_t)a.lc) * ROW_DW + i] : 0u;
    }
    WAVE_LDS_SYNC();
    /* ---- the per-site summary (src/call_genotypes.c:44-59): made here from the pile-up, or taken as the accumulate kernel
     * made it (the same statements there: call_summary.inc) ---- */
    uint32_t n_reads, cnt[8], qpack0, qpack1;
    bool covered;
    int aq, mq;
    if (SUMM) {
      n_reads = w[11];
      covered = valid && n_reads != 0;
#pragma unroll
      for (int j = 0; j < 8; j++) cnt[j] = w[j] & 0xffffu;
      qpack0 = w[8];
      qpack1 = w[9];
      aq = (int)(w[10] & 0xffffu);
      mq = (int)(w[10] >> 16);
    } else {
      uint32_t n_o, c_o[8], q0_o, q1_o;
      bool cov_o;
      int aq_o, mq_o;
      {
#include "call_summary.inc"
        n_o = n_reads;
        cov_o = covered;
#pragma unroll
        for (int j = 0; j < 8; j++) c_o[j] = cnt[j];
        q0_o = qpack0;
        q1_o = qpack1;
        aq_o = aq;
        mq_o = mq;
      }
      n_reads = n_o;
      covered = cov_o;
#pragma unroll
      for (int j = 0; j < 8; j++) cnt[j] = c_o[j];
      qpack0 = q0_o;
      qpack1 = q1_o;
      aq = aq_o;
      mq = mq_o;
    }
    const unsigned rf = valid ? (unsigned)srf[lane + 2u] & 0xffu : 0u; /* my site's reference code (lane index = lane) */
    const double l = K_COLD(l), t = K_COLD(t), lrb = K_COLD(lrb), lrb1 = K_COLD(lrb1);
    /* a run's first tile: lane 1 — the site just left of its first record — forms just enough of its record for its right
     * neighbour (below); in the other tiles that site's facts are the carried ones */
    const bool former = valid && lane < 62u && (inner || (f0 == 2u && lane == 1u));
    const uint8_t *const dbsnp_p = K_COLD(dbsnp);
    const uint32_t rs_found = (dbsnp_p && inner) ? (uint32_t)dbsnp_p[jw + (int32_t)(MULTI ? a.pos_off : 0u)] : 0u;
    /* May the lane's record be written?  The hom-ref skip rule (src/print_vcf.c:85-96,139) on max_gt and the lane's own reference
     * base: exactly the printer's decision unless gt_prob[] ties (call_body.inc looks after those) or the context blanking replaces
     * the base (tile_n: then every lane counts as written).  The region clip (:154-158) is not looked at: such lanes are
     * normalised for nothing. */
    const bool want_all = tile_n || K_COLD(a.all_positions) != 0;
    /* wave-uniform: the block's ends and every N are out of reach of the tile's records (lane indices -1 .. 63 in the block, no
     * blanked base): their neighbours and reference context need no case distinction */
    const bool plain = !tile_n && blk_lo <= -1 && blk_hi >= 63;
#define CALL_WANT_GP(g_) (former && (want_all || (rs_found & 2u) || !(((g_) == 0 && rf == 1u) || ((g_) == 9 && rf == 4u))))
#define CALL_PRINTER_GT
#define CALL_PRIOR_TABLE s_prior
#define CALL_COMPACT (!READS) /* READS: the forward counts wait in the lanes' slot areas (la[12]) until the heterozygous calls are listed */
#define CALL_COMPACT_CAP 39u
#define CALL_SUMMARY_GIVEN
#include "call_body.inc"
#undef CALL_SUMMARY_GIVEN
#undef CALL_WANT_GP

    /* ---- block counters (window positions only, not the halo) ---- */
    const bool defer = covered && inner && ((0x16Eu >> mxi) & 1u); /* gt_het[max_gt]: Fisher's test (:61), after the tile */
    if (covered && inner) {
      atomicAdd(&s_cnt[0], 1u);
      atomicAdd(&s_cnt[1 + mxi], 1u);
    }
    if (READS) { /* heterozygous calls: listed now, with what Fisher's test and the FILTER bits after it need — the counts
                  * are in registers and the forward-strand stash is still intact */
      const unsigned long long m = __ballot(defer);
      if (m) {
        if (defer) {
          uint32_t f[8];
          if (__builtin_expect(bigf_any, 0)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the lane's own stores to the scratch lines */
            const uint32_t *fs = K_COLD(ra.f_scratch) + ((uint64_t)(blockIdx.x * FW + wid) * 64u + lane) * 8u;
#pragma unroll
            for (int j = 0; j < 8; j++) f[j] = __hip_atomic_load(&fs[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            const uint2 st = reinterpret_cast<const uint2 *>(slot + lane * IN_DW)[12];
#pragma unroll
            for (int j = 0; j < 4; j++) {
              f[j] = (st.x >> (8 * j)) & 0xffu;
              f[4 + j] = (st.y >> (8 * j)) & 0xffu;
            }
          }
          uint4 *e = reinterpret_cast<uint4 *>(F_WL() + (uint64_t)(n_pend + (unsigned)__popcll(m & ((1ull << lane) - 1ull))) * F_HET_DW);
          /* MULTI: the index among the launch's positions (the pass after the loop knows nothing of blocks) */
          e[0] = make_uint4((uint32_t)(jw + (int32_t)(MULTI ? a.pos_off : 0u)) | ((uint32_t)mxi << 28), f[0], f[1], f[2]);
          e[1] = make_uint4(f[3], f[4], f[5], f[6]);
          e[2] = make_uint4(f[7], cnt[0], cnt[1], cnt[2]);
          e[3] = make_uint4(cnt[3], cnt[4], cnt[5], cnt[6]);
          e[4] = make_uint4(cnt[7], (uint32_t)mq, 0u, 0u);
        }
        n_pend += (unsigned)__popcll(m);
      }
    }

    /* ---- the printer's genotype: first-max argmax of gt_prob[], recomputed (src/print_vcf.c:584-591) ----
     * Published for the neighbours with everything they need from it: its IUPAC letter and whether it carries C / G. */
    {
      const uint32_t gz = covered ? (uint32_t)pgi + 1u : 0u; /* call_body.inc: max_gt, or the argmax itself where gt_prob[] may tie */
      const uint32_t g0 = gz ? gz - 1u : 0u;
      /* "NAMRWCSYGKT"[gz] */
      const uint32_t lo = __builtin_amdgcn_perm(0x59534357u, 0x524D414Eu, gz & 7u);
      const uint32_t hi = __builtin_amdgcn_perm(0u, 0x00544B47u, gz & 7u);
      const uint32_t iu = (gz >= 8u ? hi : lo) & 0xffu;
      const uint32_t fl = gz ? (((0x72u >> g0) & 1u) | (((0x1A4u >> g0) & 1u) << 1)) : 0u; /* AC CC CG CT ; AG CG GG GT */
      sg[2u + lane] = gz | (iu << 8) | (fl << 16);
    }
    WAVE_LDS_SYNC();

    /* ---- record formation (_print_vcf_entry, src/print_vcf.c:32-381; restated as in vcfcore.hip) ----
     * Lanes 2..61 form their record; lane 1 — the site just left of the tile's first position — runs the same code
     * for the one thing its right neighbour needs from it: whether it is a written '+' strand CG call, and its FILTER
     * bits (the pending cytosine of the CpG bookkeeping).  The record is built as its sixteen dwords. */
    uint32_t od0 = 0u; /* the lane's record position, 0 = it has none: all that is read of the record after the branch */
    bool pend = false, minus_cg = false, st_called = false, st_emit = false, st_het = false, st_rs = false,
         st_cpg = false, st_refcpg = false;
    /* operands of the statistics, each read only under one of the flags above: "some register" (an empty asm defines the
     * value without an instruction) instead of a default that would cost a v_mov per value wherever a branch skips them */
    uint32_t st_phred, st_qd, st_cdp, st_cinf, st_ma, st_mb, st_pos;
#define F_ANY(v) asm volatile("" : "=v"(v)) /* volatile: seven separate registers, not one shared and copied */
    F_ANY(st_phred); F_ANY(st_qd); F_ANY(st_cdp); F_ANY(st_cinf); F_ANY(st_ma); F_ANY(st_mb); F_ANY(st_pos);
#undef F_ANY
    int st_mut = 12;
    const uint32_t me = former ? sg[2u + lane] : 0u;
    const uint32_t dp1 = cnt[0] + cnt[1] + cnt[2] + cnt[3], d_inf = cnt[4] + cnt[5] + cnt[6] + cnt[7];
    uint32_t flt = 0;
    uint32_t ebyte = 0; /* the lane's byte of emit flags: 0 = no record is written for its position */
#ifdef F_EXPERIMENT_SKIP_RECORD /* timing experiment only (tools/build_variant_fused.sh): the tile's cost without the record formation */
    if (false) {
#else
    if (((me & 0xffu) != 0) & (dp1 + d_inf != 0)) { /* one branch, not two nested ones */
#endif
      /* The record is built AND staged inside the branch: sixteen dwords that leave it would each need a default for the
       * lanes that skip it — a v_mov per dword at every level of the branch, ~40 vector instructions a tile. */
      uint32_t od[16];
      od[6] = od[7] = od[9] = od[10] = od[11] = od[12] = od[13] = od[14] = od[15] = 0u; /* 9-13: the GLs of a WRITTEN record, below */
      const int gt = (int)(me & 0xffu) - 1;
      const int L = (int)lane;
      /* called genotypes of lane indices L-2 .. L+2: entries L .. L+4 of sg[], whose first two are the carried sites (a run's
       * first tile has none: its lane 1 clamps at lane 0, its own record is not kept); reference context through the reference's
       * strncpy of a 7-base window (:570-577); srf[] starts at lane index -2 */
      uint32_t ge[5], rr[5];
      if (plain) { /* no block end and no N anywhere near the tile: the five neighbours and the five bases as they stand */
#pragma unroll
        for (int k = 0; k < 5; k++) {
          const uint32_t j2 = lane + (uint32_t)k;
          ge[k] = sg[j2 < f0 ? f0 : j2];
          rr[k] = srf[j2];
        }
      } else {
#pragma unroll
        for (int k = 0; k < 5; k++) {
          const int j = L - 2 + k;
          uint32_t v = (j >= blk_lo && j <= blk_hi) ? sg[j + 2 < (int)f0 ? (int)f0 : j + 2] : 0x4E00u;
          if (j > blk_hi && L + 2 > blk_hi) v = sg[blk_hi + 2]; /* flush_vcf_entries repeats the last genotype, :540 */
          ge[k] = v;
        }
        const int la_ = L + 2 < blk_hi ? L + 2 : blk_hi;    /* look-ahead position that filled the window */
        const int w0 = la_ >= blk_lo + 4 ? la_ - 4 : blk_lo; /* first base of that copy */
        bool blank = false;
        for (int j = w0; j < L - 2; j++) blank |= (srf[j + 2 < 0 ? 0 : j + 2] & 0xffu) == 0; /* an N before the 5 bases */
#pragma unroll
        for (int k = 0; k < 5; k++) {
          const int j = L - 2 + k;
          uint32_t v = 0x4E00u; /* code 0, 'N' */
          if (j >= blk_lo) {
            v = srf[j + 2 < 0 ? 0 : j + 2];
            blank |= (v & 0xffu) == 0;
            if (blank) v = 0x4E00u;
          }
          rr[k] = v;
        }
      }
      const int rfix = (int)(rr[2] & 0xffu);
      int ga, gb;
      f_alleles(gt, ga, gb);
      const bool het = ga != gb;
      bool skp = !K_COLD(a.all_positions) && !(rs_found & 2u) && ((gt == 0 && rfix == 1) || (gt == 9 && rfix == 4));
      const uint32_t pos = pos0 + lane;
      if (!skp) skp = pos < K_COLD(a.reg_start) || pos > K_COLD(a.reg_stop);
      /* CpG status (:227-266) */
      uint32_t cg = '.';
      {
        const uint32_t c = ge[2] & 0xffu, nx = ge[3] & 0xffu, pv = ge[1] & 0xffu;
        const bool nxG = (ge[3] >> 17) & 1u, pvC = (ge[1] >> 16) & 1u, cC = (ge[2] >> 16) & 1u, cG = (ge[2] >> 17) & 1u;
        if ((c == 5u && nx == 8u) || (c == 8u && pv == 5u)) cg = 'C'; /* "CG" */
        else if (cC) cg = nx ? (nxG ? 'H' : 'N') : '?';
        else if (cG) cg = pv ? (pvC ? 'H' : 'N') : (c == 8u ? '?' : '.');
      }
      od[0] = od0 = pos;
      od[2] = cg << 24;
      od[3] = ((rr[0] >> 8) << 16) | ((rr[1] >> 8) << 24);
      od[4] = (rr[2] >> 8) | ((rr[3] >> 8) << 8) | ((rr[4] >> 8) << 16) | (((ge[0] >> 8) & 0xffu) << 24);
      od[5] = ((ge[1] >> 8) & 0xffu) | (((ge[2] >> 8) & 0xffu) << 8) | (((ge[3] >> 8) & 0xffu) << 16) | (((ge[4] >> 8) & 0xffu) << 24);
      od[8] = dp1;
      uint32_t d1 = ((uint32_t)gt << 8) | ((uint32_t)rfix << 16);
      /* QUAL and QD of a position whose record is not written stay 0: the printer computes them (:140-152) and never looks at them
       * (the record's flags and the statistics, :185-217,382-398, are behind `skip`), and such a lane's gt_prob[] is not made */
      int phred = 0;
      uint32_t qd = 0;
      if (!skp) {
        /* phred (:140-148) */
        const double z1 = exp_dev(lg[gt] * BSM_LN10, (const uint64_t *)s_exptab);
        { /* (int)(-10 log(1 - z1) / LOG10) capped at 255, as a staircase in om = 1 - z1 (bscall_api.c: bsc_build_phred_table):
           * the value below the four steps a binade of om can hold, plus the steps om does not exceed */
          const double om = 1.0 - z1;
          uint32_t e = 1023u - ((uint32_t)(bsm_bits(om) >> 32) >> 20); /* 0 < om <= 1: binades 0 .. 53 */
          e = e > 63u ? 63u : e;
          const double2 ta = *reinterpret_cast<const double2 *>(s_pthr + 4u * e), tb2 = *reinterpret_cast<const double2 *>(s_pthr + 4u * e + 2u);
          phred = (int)s_pbase[e] + (om <= ta.x ? 1 : 0) + (om <= ta.y ? 1 : 0) + (om <= tb2.x ? 1 : 0) + (om <= tb2.y ? 1 : 0);
          if (z1 >= 1.0) phred = 255; /* om = 0 (src/print_vcf.c:142-145) */
        }
        /* FS = (int)(-0.0 * 10.0 + 0.5) = 0: fisher_strand is 0 unless gt_het[max_gt]; those are tested after the wave's last tile */
        qd = dp1 > 0 ? (uint32_t)phred / dp1 : (uint32_t)phred;
        if (phred < 20) flt |= 1u;
        if (qd < 2u) flt |= 2u;
        if (mq < 40) flt |= 8u;
        if (!flt && het && !defer && f_mac1(gt, cnt)) flt |= 128u; /* deferred sites: after their FS is known */
        int aix0 = 0, aix1 = 0;
        if (ga != rfix) aix0 = ga;
        if (gb != ga && gb != rfix) { if (aix0) aix1 = gb; else aix0 = gb; }
        /* "\0ACGT"[aix] twice */
        od[3] |= __builtin_amdgcn_perm(0x00000054u, 0x47434100u, (uint32_t)aix0 | ((uint32_t)aix1 << 8)) & 0xffffu;
        const uint32_t gt_enc = het ? ((ga == rfix || gb == rfix) ? 0x24u : 0x48u) : (ga == rfix ? 0x22u : 0x44u);
        d1 |= 1u | (gt_enc << 24);
        /* GL (:319-347): gt_prob read back from the lane's LDS area (la[], call_body.inc); index of alleles a <= b:
         * a (9 - a) / 2 + b - 5 */
#define F_GLIDX(x, y) ((x) * (9 - (x)) / 2 + (y)-5)
        const bool hr = rfix != 0;
        const int r1 = hr ? rfix : 1, a0 = aix0 ? aix0 : 1, a1 = aix1 ? aix1 : 1;
        double zr = hr ? lg[F_GLIDX(r1, r1)] : -99.999;
        double zh0 = lg[r1 < a0 ? F_GLIDX(r1, a0) : F_GLIDX(a0, r1)], zm0 = lg[F_GLIDX(a0, a0)];
        double zh1 = lg[r1 < a1 ? F_GLIDX(r1, a1) : F_GLIDX(a1, r1)], zm1 = lg[F_GLIDX(a1, a1)];
#undef F_GLIDX
        zr = zr < -99.999 ? -99.999 : zr;
        zh0 = zh0 < -99.999 ? -99.999 : zh0;
        zm0 = zm0 < -99.999 ? -99.999 : zm0;
        zh1 = zh1 < -99.999 ? -99.999 : zh1;
        zm1 = zm1 < -99.999 ? -99.999 : zm1;
        const uint32_t nalt = (aix0 ? 1u : 0u) + (aix1 ? 1u : 0u);
        const uint32_t ngl = 1u + (hr ? 2u * nalt : nalt);
        /* with a reference base: ref, het0, hom0, het1, hom1; on an N: ref (-99.999), hom0, hom1 */
        const float f0 = (float)zr, f1 = (float)(hr ? zh0 : zm0), f2 = (float)(hr ? zm0 : zm1), f3 = (float)zh1, f4 = (float)zm1;
        od[9] = __float_as_uint(f0);
        od[10] = ngl > 1u ? __float_as_uint(f1) : 0u;
        od[11] = ngl > 2u ? __float_as_uint(f2) : 0u;
        od[12] = ngl > 3u ? __float_as_uint(f3) : 0u;
        od[13] = ngl > 4u ? __float_as_uint(f4) : 0u;
        od[2] |= flt | ((uint32_t)phred << 8) | (ngl << 16);
        od[7] = qd;
        /* The record's length as BCF2 (csrc/bcfdev.hip bcf_emit_body, with one-byte dictionary indices and no ID), for the byte of emit
         * flags: the encoder's size pass then reads a byte per position instead of the records (round 6).  255 = "ask the record": a
         * heterozygous call (FS and its FILTER bits come after the tile: f_fisher_pending writes the byte again when it has them), a dbSNP-flagged
         * position (its name), anything longer than 254. */
        if (K_COLD(emit_out)) {
          uint32_t cm = cnt[0] > cnt[1] ? cnt[0] : cnt[1], n_amq = 0;
#pragma unroll
          for (int k = 2; k < 8; k++) cm = cnt[k] > cm ? cnt[k] : cm;
#pragma unroll
          for (int k = 0; k < 8; k++) n_amq += cnt[k] > 0u ? 1u : 0u;
#define F_PI(v) ((v) <= 127u ? 2u : ((v) <= 32767u ? 3u : 5u)) /* put_int of a non-negative value */
          const uint32_t fb = flt & 15u;
          const uint32_t ft_len = ((fb & 1u) ? 4u : 0u) + ((fb & 2u) ? 4u : 0u) + ((fb & 4u) ? 5u : 0u) + ((fb & 8u) ? 5u : 0u) + (uint32_t)__popc(fb) - 1u;
          const uint32_t ft = fb ? (ft_len >= 15u ? 3u : 1u) + ft_len : 5u;
          const uint32_t cs = ((0x209u >> gt) & 1u) ? 2u : (((0x72u >> gt) & 1u) + ((0x1A4u >> gt) & 1u));
          const uint32_t len = 32u + 13u + (aix0 ? 2u : 0u) + (aix1 ? 2u : 0u) + 5u + 2u + ft + 2u + F_PI(dp1) + 2u + F_PI((uint32_t)mq) + 2u + F_PI((uint32_t)phred) + 2u +
                               F_PI(qd) + 3u + 4u * ngl + 3u + 8u * (cm <= 127u ? 1u : (cm <= 32767u ? 2u : 4u)) + (n_amq ? 3u + n_amq : 0u) + 3u + cs + 4u + 8u;
#undef F_PI
          ebyte = (het || rs_found || len > 254u) ? 255u : len;
        }
      }
      od[1] = d1;
      /* ---- facts for the statistics and for the right neighbour ---- */
      const bool emit = !skp;
      pend = emit && cg == 'C' && gt == 4; /* a written CC call followed by GG: cs_str "+", FORMAT CG "CG" */
      minus_cg = emit && cg == 'C' && gt == 7;
      if (inner && !defer) {
        st_called = true;
        st_emit = emit;
        st_phred = (uint32_t)phred;
        st_qd = qd > 255u ? 255u : qd;
        st_pos = pos;
        const uint32_t dpt = dp1 + d_inf;
        st_cdp = dpt < BSC_COV_CAP ? dpt : BSC_COV_CAP - 1u;
        st_cinf = d_inf < BSC_COV_CAP ? d_inf : BSC_COV_CAP - 1u;
        if (emit) {
          st_rs = rs_found != 0;
          st_het = het;
          st_mut = ss_mut_type(gt, rfix);
          if (cg == 'C' && (gt == 4 || gt == 7)) {
            st_cpg = true;
            const uint32_t x1 = rr[1] & 0xffu, x2 = rr[2] & 0xffu, x3 = rr[3] & 0xffu;
            st_refcpg = gt == 4 ? (x2 == 2u && x3 == 3u) : (x1 == 2u && x2 == 3u); /* prf_ctxt + 2 / + 1 == "CG" */
            st_ma = gt == 4 ? cnt[5] : cnt[6];
            st_mb = gt == 4 ? cnt[7] : cnt[4];
          }
        }
      }
      /* the record goes to the tile's staging area, which lies over la[]: every read of la[] is above, and a wave's LDS
       * operations execute in order (lane 1 of a run's first tile forms a record for its neighbour's sake only) */
      WAVE_LDS_SYNC();
      if (lane >= f0) {
        uint4 *so = reinterpret_cast<uint4 *>(slot);
#pragma unroll
        for (int k = 0; k < 4; k++) so[(lane - f0) * 4u + k] = make_uint4(od[4 * k], od[4 * k + 1], od[4 * k + 2], od[4 * k + 3]);
      }
    }
    spd[2u + lane] = (uint16_t)((pend ? 1u : 0u) | (flt << 8));
    const uint32_t depth_off = K_COLD(a.depth_off);
    if (depth_off && inner) { /* total depth of every position that reached the printer (the key of gt_cov_stats) */
      const uint32_t dpt = dp1 + d_inf;
      reinterpret_cast<uint16_t *>(K_COLD(het_list) + depth_off)[jw + (int32_t)(MULTI ? a.pos_off : 0u)] = (uint16_t)(od0 ? (dpt < BSC_COV_CAP ? dpt : BSC_COV_CAP - 1u) : 0u);
    }
    WAVE_LDS_SYNC();
    /* ---- results: the tile's 60 records, staged in the slot (every la[] read is done: the sync above), leave
     * contiguously — before the statistics, so that the stores drain while the histograms are updated and the sixteen
     * record dwords are dead by then ---- */
    {
      uint4 *so = reinterpret_cast<uint4 *>(slot);
      if (od0 == 0u && lane >= f0 && lane < 62u) { /* no record (an uncovered position, one in ~1 000): sixty-four zero bytes;
                                                    * after the branch above, i.e. after the last read of la[] */
#pragma unroll
        for (int k = 0; k < 4; k++) so[(lane - f0) * 4u + k] = make_uint4(0u, 0u, 0u, 0u);
      }
      WAVE_LDS_SYNC();
      const uint32_t i0 = (uint32_t)(jw0 + (int32_t)f0); /* window index of the tile's first record */
      /* MULTI: a block's records end where the next block's begin, on a multiple of 64 — the positions between are written as
       * what they are, no record (the packing pass reads every position of the launch) */
      const uint32_t an_ = FULL ? 0u : (MULTI ? (a.n + 63u) & ~63u : K_COLD(a.n));
      const uint32_t nrec = FULL ? 62u - f0 : (i0 < an_ ? (an_ - i0 < 62u - f0 ? an_ - i0 : 62u - f0) : 0u);
      const uint32_t nvec = nrec * 4u;
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      u32x4 *dst = reinterpret_cast<u32x4 *>(core_out + (uint64_t)i0 * 64u);
#pragma unroll
      for (unsigned k = 0; k < 4; k++) {
        const unsigned idx = k * 64u + lane;
        if (idx < nvec) __builtin_nontemporal_store(reinterpret_cast<const u32x4 *>(so)[idx], dst + idx);
      }
      uint8_t *const emit_out = K_COLD(emit_out);
      const uint32_t eb_ = (uint32_t)__shfl((int)ebyte, (int)((lane + f0) & 63u)); /* record r of the tile is lane r + f0's */
      if (emit_out && lane < nrec) /* the records' emit flags once more, a byte per position — 0, or the record's BCF2 length (255: look at
                                    * the record): what the packing pass and the BCF encoder's size pass read instead of the records */
        emit_out[(uint64_t)i0 + (MULTI ? a.pos_off : 0u) + lane] = (uint8_t)eb_;
      uint8_t *const aux_out = K_COLD(aux_out);
      if (aux_out) { /* what the encoder of a written record reads besides the core record (src/print_vcf.c:306-359): MC8 counts,
                      * AMQ qualities, MQ, mean quality, max_gt, the dbSNP flag — the second half of a bsc_vcf_rec */
        WAVE_LDS_SYNC();
        if (lane >= f0 && lane < 62u) {
          const uint32_t rsf = rs_found;
          const bool hasrec = od0 != 0u;
          so[(lane - f0) * 4u + 0] = hasrec ? make_uint4(cnt[0], cnt[1], cnt[2], cnt[3]) : make_uint4(0u, 0u, 0u, 0u);
          so[(lane - f0) * 4u + 1] = hasrec ? make_uint4(cnt[4], cnt[5], cnt[6], cnt[7]) : make_uint4(0u, 0u, 0u, 0u);
          so[(lane - f0) * 4u + 2] = hasrec ? make_uint4(qpack0, qpack1, (uint32_t)mq, (uint32_t)aq) : make_uint4(0u, 0u, 0u, 0u);
          so[(lane - f0) * 4u + 3] = hasrec ? make_uint4((uint32_t)mxi | (rsf << 8), 0u, 0u, 0u) : make_uint4(0u, 0u, 0u, 0u);
        }
        WAVE_LDS_SYNC();
        u32x4 *dsta = reinterpret_cast<u32x4 *>(aux_out + ((uint64_t)i0 + (MULTI ? a.pos_off : 0u)) * 64u);
#pragma unroll
        for (unsigned k = 0; k < 4; k++) {
          const unsigned idx = k * 64u + lane;
          if (idx < nvec) __builtin_nontemporal_store(reinterpret_cast<const u32x4 *>(so)[idx], dsta + idx);
        }
      }
    }
    /* ---- heterozygous calls: listed, Fisher's test after the wave's last tile ---- */
    if (!READS) {
      const unsigned long long m = __ballot(defer);
      if (m) {
        if (defer) F_WL()[n_pend + (unsigned)__popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)jw | ((uint32_t)mxi << 28);
        n_pend += (unsigned)__popcll(m);
      }
    }
#ifndef BSC_CHAIN_NO_PREFETCH
    /* The slot is free from here to the end of the tile (the statistics below touch other LDS arrays only): request the
     * wave's NEXT tile now, so that its pile-ups land — and the record stores above drain — while the histograms are
     * updated, instead of the wave sitting out a full HBM round trip at the top of the next tile. */
    if (FULL && !READS) {
      WAVE_LDS_SYNC();
      if (have_next) F_DMA_TILE(jw0_next, dma16_hidden);
    }
#endif
    if (K_COLD(a.with_stats)) {
      /* ---- the statistics block (src/print_vcf.c:386-525; sitestats.hip has the restatement) for the tile ----
       * Wide histograms take one LDS atomic per lane; where one value dominates (QUAL 255, MQ, FS 0, FILTER 0) the
       * first lane's value is added once for all lanes that share it. */
      const bool pass = flt == 0;
      const uint32_t mqv = (uint32_t)(mq < 0 ? 0 : (mq > 255 ? 255 : mq));
      if (st_called) { /* cov[depth].all / .var (:393, :419) */
        if (st_cdp < F_COV_LDS) {
          atomicAdd(&h[SS_COV + st_cdp * 6u], 1u);
          if (st_emit) atomicAdd(&h[SS_COV + st_cdp * 6u + 1u], 1u);
        } else {
          unsigned long long *const stat_words = K_COLD(stat_words);
          atomicAdd(&stat_words[SS_COV + (uint64_t)st_cdp * 6u], 1ull);
          if (st_emit) atomicAdd(&stat_words[SS_COV + (uint64_t)st_cdp * 6u + 1u], 1ull);
        }
      }
      const unsigned long long m_emit = __ballot(st_emit);
      if (m_emit) {
        const unsigned long long m_pass = __ballot(st_emit && pass);
        if (lane == 0) {
          atomicAdd(&h[SS_MISC + 0], (uint32_t)__popcll(m_emit)); /* snps[all]: every written record, see sitestats.hip */
          atomicAdd(&h[SS_MISC + 1], (uint32_t)__popcll(m_pass));
        }
        const uint32_t hh = st_het ? 1u : 0u;
        f_hist_peel1(h, st_emit, SS_QUAL + 256u + st_phred, lane); /* qual[variant_sites]; [all_sites] is its copy */
        if (st_emit) atomicAdd(&h[SS_FST + 0u * 512u + st_qd * 2u + hh], 1u);
        f_hist_peel1(h, st_emit, SS_FST + 1u * 512u + hh, lane);   /* FS is 0 for every position this kernel finishes */
        f_hist_peel1(h, st_emit, SS_FST + 2u * 512u + mqv * 2u + hh, lane);
        f_hist_peel1(h, st_emit, SS_FILT + (hh ? 32u : 0u) + (flt & 31u), lane);
        if (__any(st_rs)) {
          if (st_rs) {
            atomicAdd(&h[SS_MISC + 6], 1u);
            atomicAdd(&h[SS_MISC + 8], 1u);
            if (pass) {
              atomicAdd(&h[SS_MISC + 7], 1u);
              atomicAdd(&h[SS_MISC + 9], 1u);
            }
          }
        }
        if (__any(st_cpg)) {
          bool pair = false, pair_pass = false;
          if (minus_cg && st_cpg) { /* does the record just before complete a CpG? (:198-205) */
            bool p_ok;
            uint32_t p_flt;
            if (jw == 0 && A_COLD(lc) == 0) { /* first position of a block: the previous block's pending cytosine, if adjacent */
              const uint32_t *const carry_in = K_COLD(carry_in);
              const uint32_t p_pos = carry_in[0];
              /* MULTI: only the launch's first block follows something — the host starts a new launch at a block that begins
               * right behind its predecessor's last position */
              p_ok = p_pos != 0 && st_pos - p_pos == 1u && (!MULTI || a.first_block);
              p_flt = carry_in[1];
            } else {
              const uint32_t pw = spd[1u + lane]; /* the site to the left: the previous lane's, or (lane 0) the carried one */
              p_ok = (pw & 1u) != 0;
              p_flt = pw >> 8;
            }
            pair = p_ok;
            pair_pass = p_ok && !(p_flt || flt);
          }
          if (st_cpg) {
            const uint32_t rsel = st_refcpg ? 0u : 2u;
            if (pair) atomicAdd(&h[SS_MISC + 10 + rsel], 1u);
            if (pair_pass) atomicAdd(&h[SS_MISC + 11 + rsel], 1u);
            atomicAdd(&h[SS_QUAL + 2u * 256u + (st_refcpg ? 0u : 256u) + st_phred], 1u);
            if (st_cdp < F_COV_LDS) atomicAdd(&h[SS_COV + st_cdp * 6u + (st_refcpg ? 2u : 3u)], 1u);
            else atomicAdd(&K_COLD(stat_words)[SS_COV + (uint64_t)st_cdp * 6u + (st_refcpg ? 2u : 3u)], 1ull);
            if (st_cinf < F_COV_LDS) atomicAdd(&h[SS_COV + st_cinf * 6u + (st_refcpg ? 4u : 5u)], 1u);
            else atomicAdd(&K_COLD(stat_words)[SS_COV + (uint64_t)st_cinf * 6u + (st_refcpg ? 4u : 5u)], 1ull);
            if (st_ma + st_mb != 0) { /* methylation posterior (:492-515): counted per (a, b), evaluated when read */
              if (st_ma < F_PAIR && st_mb < F_PAIR) {
                const uint32_t cell = (rsel * F_PAIR + st_ma) * F_PAIR + st_mb;
                atomicAdd(&s_pair[cell], 1u);
                if (pass) atomicAdd(&s_pair[cell + F_PAIR * F_PAIR], 1u);
              } else if (st_ma < SS_PAIR_G && st_mb < SS_PAIR_G) {
                const uint32_t cell = (rsel * SS_PAIR_G + st_ma) * SS_PAIR_G + st_mb;
                unsigned long long *const pair_cells = K_COLD(pair_cells);
                atomicAdd(&pair_cells[cell], 1ull);
                if (pass) atomicAdd(&pair_cells[cell + SS_PAIR_G * SS_PAIR_G], 1ull);
              } else {
                const unsigned long long k = atomicAdd(&K_COLD(counters)[BSC_CNT_OVF], 1ull);
                if (k < K_COLD(a.ovf_cap))
                  K_COLD(ovf_list)[k] = (unsigned long long)st_ma | ((unsigned long long)st_mb << 24) |
                                ((unsigned long long)(st_refcpg ? 1u : 0u) << 48) | ((unsigned long long)(pass ? 1u : 0u) << 49);
              }
            }
          }
        }
        if (__any(st_mut != 12)) {
          if (st_mut != 12) {
            atomicAdd(&h[SS_MUT + (uint32_t)st_mut * 2u], 1u);
            if (pass) atomicAdd(&h[SS_MUT + (uint32_t)st_mut * 2u + 1u], 1u);
            if (st_rs) {
              atomicAdd(&h[SS_DBMUT + (uint32_t)st_mut * 2u], 1u);
              if (pass) atomicAdd(&h[SS_DBMUT + (uint32_t)st_mut * 2u + 1u], 1u);
            }
          }
        }
      }
      /* the window's last position is the pending cytosine, or not, for whatever follows */
      if (jw == (int32_t)A_COLD(n) - 1 && lane >= f0 && lane < 62u && (!MULTI || a.last_block)) {
        uint32_t *const carry_out = K_COLD(carry_out);
        carry_out[0] = pend ? pos0 + lane : 0u;
        carry_out[1] = pend ? flt : 0u;
      }
    }

    /* ---- what the run's next tile inherits: genotype words and pending-cytosine facts of this tile's last two record
     * sites (lanes 60, 61) become entries 0, 1 ---- */
    if (lane < 2u) {
      sg[lane] = sg[62u + lane];
      spd[lane] = spd[62u + lane];
    }
    WAVE_LDS_SYNC();
    if (++tj == run_n) {
      tj = 0;
      run += n_waves;
      if (run < n_runs_all) F_RUN_SETUP(run);
    }
  }

  /* ---- the wave's heterozygous calls: Fisher's exact test, 64 at a time ----
   * A het is one position in ~1 000: tested inside its tile it would idle 63 lanes for the length of Fisher's loops in
   * every such tile, tested by a kernel of its own it costs a launch and a drained device per window. */
  if (n_pend) {
    unsigned long long *const stat_words = K_COLD(stat_words);
    uint32_t *const wl = F_WL();
    const uint8_t *const dbsnp = K_COLD(dbsnp);
    if (lane0 == 0) atomicAdd(&s_cnt[11], n_pend);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); /* the wave's records and its list are in memory */
    for (unsigned k0 = 0; k0 < n_pend; k0 += 64u) {
      const unsigned nb = n_pend - k0 < 64u ? n_pend - k0 : 64u;
      const uint32_t *ent = wl + (uint64_t)(k0 + lane0) * (READS ? F_HET_DW : 1u);
      const uint32_t e = lane0 < nb ? __hip_atomic_load(ent, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
      f_fisher_pending<READS, SUMM>(e, nb, lane0, READS ? ent : cts, dbsnp, a, core_base, K_COLD(tb)->lfact, s_logtab, s_exptab, h, stat_words, K_COLD(emit_out));
    }
  }
  if (run >= n_runs_all) break; /* wave-uniform */
  } /* epochs */
#undef cts
#undef ref
#undef core_out
#undef core_base
  unsigned long long *const counters = K_COLD(counters);
  unsigned long long *const stat_words = K_COLD(stat_words);
  if (READS && __any(inexact)) { /* positions whose quality / MAPQ^2 sums left the exact-float range (accumulate.hip) */
    const unsigned long long m = __ballot(inexact != 0);
    if (lane0 == 0) atomicAdd(&counters[BSC_CNT_INEXACT], (unsigned long long)__popcll(m));
  }

  __syncthreads();
  if (tid < 12 && s_cnt[tid]) atomicAdd(&counters[BSC_CNT_COVERED + tid], (unsigned long long)s_cnt[tid]);
  if (a.with_stats) {
    for (unsigned i = tid; i < F_WORDS; i += 64 * FW)
      if (h[i]) {
        atomicAdd(&stat_words[i], (unsigned long long)h[i]);
        /* qual[all_sites] receives what qual[variant_sites] receives (every written record counts as a variant) */
        if (i >= SS_QUAL + 256u && i < SS_QUAL + 512u) atomicAdd(&stat_words[i - 256u], (unsigned long long)h[i]);
      }
    for (unsigned i = tid; i < 4 * F_PAIR * F_PAIR; i += 64 * FW)
      if (s_pair[i]) { /* LDS cell [q][a][b] -> the context's [q][SS_PAIR_G][SS_PAIR_G] table */
        const unsigned q = i / (F_PAIR * F_PAIR), ab = i % (F_PAIR * F_PAIR);
        atomicAdd(&K_COLD(pair_cells)[(q * SS_PAIR_G + ab / F_PAIR) * SS_PAIR_G + ab % F_PAIR], (unsigned long long)s_pair[i]);
      }
  }
}

/*
 * GC content by coverage (gt_cov_stats.gc_pcent, src/print_vcf.c:394-398): every position that reached the printer adds
 * one to [its total depth][G+C count of its 100-base bin], if the bin holds no N.  Reads the depths the chain kernel left
 * (2 bytes per position) and the contig's bins (1 byte per 100 positions); the table of a workgroup lives in its LDS
 * (256 depths x 101), deeper positions go to the table in HBM directly.
 */
#define GC_ROWS 256
extern "C" __global__ __launch_bounds__(1024) void bsc_gc_cov_kernel(const uint16_t *__restrict__ depth, uint32_t n, uint32_t pos0,
                                                                     const uint8_t *__restrict__ gc_bins, uint32_t n_bins,
                                                                     uint32_t start_pos, unsigned long long *__restrict__ table,
                                                                     const unsigned long long *__restrict__ counters, uint32_t run_if) {
  __shared__ uint32_t t[GC_ROWS * 101];
  if (run_if && (run_if == 1u) != (counters[BSC_CNT_DEEP] != 0)) return; /* the chain launch this one follows stood back (bsc_chain_args.run_if) */
  for (unsigned i = threadIdx.x; i < GC_ROWS * 101; i += 1024) t[i] = 0;
  __syncthreads();
  for (uint32_t i = blockIdx.x * 1024u + threadIdx.x; i < n; i += gridDim.x * 1024u) {
    const uint32_t d = depth[i];
    if (!d) continue;
    const uint32_t bn = (pos0 + i - start_pos) / 100u; /* unsigned like the reference's: a position before start_pos is out of range */
    if (bn >= n_bins) continue;
    const uint32_t g = gc_bins[bn];
    if (g > 100u) continue;
    if (d < GC_ROWS) atomicAdd(&t[d * 101u + g], 1u);
    else atomicAdd(&table[(uint64_t)d * 101u + g], 1ull);
  }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < GC_ROWS * 101; i += 1024)
    if (t[i]) atomicAdd(&table[i], (unsigned long long)t[i]);
}

/* ---- launcher ------------------------------------------------------------------------------------------------ */
/* positions one round of the resident waves covers: a window that is a whole number of these gives every wave the same
 * number of tiles (no partly filled last round) */
extern "C" unsigned bsc_dev_chain_quantum(int num_cus) { return (unsigned)num_cus * FW * FT; }

/* the largest window <= limit in which every resident wave gets ONE run of the same number of tiles: waves x (60 + 62 k) */
extern "C" unsigned bsc_dev_chain_window(int num_cus, unsigned limit) {
  const unsigned waves = (unsigned)num_cus * FW;
  if (limit < waves * FT) return limit / FT * FT;
  const unsigned k = (limit / waves - FT) / FT2;
  return waves * (FT + FT2 * k);
}

/* entries of one wave's heterozygous list: the kernel tests the listed calls whenever fewer than 64 entries are free */
#define F_HET_CAP 512u
static uint32_t chain_het_cap(uint32_t tiles, unsigned grid) {
  (void)tiles;
  (void)grid;
  return F_HET_CAP;
}

/* bytes of the per-wave heterozygous lists (one per resident wave, F_HET_CAP entries); with_depth: plus the window's depths;
 * reads: the reads-in form, whose entries carry the call's strand counts (F_HET_DW dwords) */
extern "C" size_t bsc_dev_chain_het_bytes(uint32_t n, int num_cus, int with_depth, int reads) {
  const size_t lists = (size_t)num_cus * FW * F_HET_CAP * sizeof(uint32_t) * (reads ? F_HET_DW : 1u);
  return lists + (with_depth ? (((size_t)n * 2u + 3u) & ~(size_t)3u) : 0u);
}

/* bytes of the reads-in form's forward-count scratch lines (64 x 8 dwords per resident wave) */
extern "C" size_t bsc_dev_chain_scratch_bytes(int num_cus) { return (size_t)num_cus * FW * 64u * 8u * sizeof(uint32_t); }

/*
 * A block's (or window's) records [0, n) in three parts.  HEAD: without two sites of context in front (lc < 2: the block starts
 * here) the first tile's left halo lies outside the buffers / the block — one guarded tile, records [0, 60).  MAIN: runs of
 * complete tiles (all 64 computed sites exist, so the kernel instance without bounds checks runs them); its last computed site
 * is the one after its last record's right neighbour: main_end + 1 <= n + rc - 1.  TAIL: what is left, under 64 + 62 records,
 * guarded tiles.  The pile-up-in form also needs each tile's first pile-up on a 16-byte boundary for the LDS-DMA ((window
 * index + lc) * 104 bytes: lc even, every run and tile start even); failing that (`aligned` = false) everything is guarded.
 * `waves`: the waves this block may count on (all resident ones, or its share of them in a launch of several blocks).
 */
struct chain_plan {
  uint32_t head, m_runs, m_tiles, m_extra, m_len;
};
static chain_plan chain_plan_block(uint32_t n, uint32_t lc, uint32_t rc, bool aligned, uint64_t waves) {
  chain_plan p = {0u, 0u, 1u, 0u, 0u};
  p.head = lc == 2u ? 0u : (n < (uint32_t)FT ? n : (uint32_t)FT);
  const int64_t avail = (int64_t)n + rc - 2 - p.head;
  if (!((p.head == 0u || p.head == (uint32_t)FT) && avail >= FT && aligned)) {
    p.head = 0;
    return p;
  }
  static int run_cap = 0;
  if (!run_cap) {
    const char *e = getenv("BSC_CHAIN_RUN_CAP");
    run_cap = e && atoi(e) > 0 ? atoi(e) : BSC_RUN_CAP;
  }
  if (waves < 1) waves = 1;
  const uint64_t tiles_all = ((uint64_t)avail + FT2 - 1) / FT2;                   /* about what the block needs */
  const uint64_t per_wave = (tiles_all + waves - 1) / waves;                      /* tiles a wave gets */
  const uint64_t rounds = (per_wave + (uint64_t)run_cap - 1) / (uint64_t)run_cap; /* runs a wave gets */
  uint64_t nr = rounds * waves;
  if (nr > (uint64_t)avail / FT) nr = (uint64_t)avail / FT;                       /* every run has a first tile of 60 records */
  /* 60 nr + 62 more = avail exactly for some nr among any 31 consecutive values when avail is even (60 nr = -2 nr mod 62): a
   * few runs fewer, and nothing is left behind the main part for a guarded launch of its own — which a 4 Mi-position window of
   * a resident contig, with its context either side, would otherwise pay (20 us + a launch gap on 350 us) */
  for (uint64_t k = 0; k <= 30 && k < nr; k++)
    if (((uint64_t)avail - (nr - k) * FT) % FT2 == 0) {
      nr -= k;
      break;
    }
  const uint64_t more = ((uint64_t)avail - nr * FT) / FT2;                        /* further tiles, 62 records each */
  p.m_runs = (uint32_t)nr;
  p.m_tiles = 1u + (uint32_t)(more / nr);
  p.m_extra = (uint32_t)(more % nr);
  p.m_len = (uint32_t)(nr * FT + more * FT2);
  return p;
}

/* the members of the kernel's argument that do not depend on the block */
template <bool READS>
static void chain_common(const bsc_chain_launch *L, uint32_t n_all, bsc_chain_kargs &K, bool &gc) {
  bsc_chain_args &a = K.a;
  memset(&K, 0, sizeof K);
  a.all_positions = L->all_positions;
  a.reg_start = L->reg_start;
  a.reg_stop = L->reg_stop;
  a.with_stats = L->with_stats;
  a.ovf_cap = L->ovf_cap;
  a.het_cap = F_HET_CAP;
  a.run_if = L->run_if;
  gc = L->with_stats && L->gc_bins && L->gc_table;
  a.depth_off = gc ? (uint32_t)(bsc_dev_chain_het_bytes(n_all, L->num_cus, 0, READS) / sizeof(uint32_t)) : 0u;
  if (READS) {
    K.ra.rd = (const bsc_read_desc *)L->rd;
    K.ra.bin_off = (const uint32_t *)L->bin_off;
    K.ra.seq = (const uint8_t *)L->seq;
    K.ra.f_scratch = (uint32_t *)L->f_scratch;
    K.ra.n_bins = L->n_bins;
    K.ra.min_qual = L->min_qual;
  }
  K.cts = (const uint32_t *)L->cts;
  K.ref = (const uint8_t *)L->ref;
  K.core_p = (uint8_t *)L->core_out;
  K.tb = (const bsc_dev_tables *)L->tb;
  K.l = L->par_l;
  K.t = L->par_t;
  K.lrb = L->par_lrb;
  K.lrb1 = L->par_lrb1;
  K.dbsnp = (const uint8_t *)L->dbsnp;
  K.het_list = (uint32_t *)L->het_list;
  K.counters = (unsigned long long *)L->counters;
  K.carry_in = (const uint32_t *)L->carry_in;
  K.carry_out = (uint32_t *)L->carry_out;
  K.stat_words = (unsigned long long *)L->stats;
  K.pair_cells = (unsigned long long *)L->pairs;
  K.ovf_list = (unsigned long long *)L->ovf_list;
  K.aux_out = (uint8_t *)L->aux_out;
  K.emit_out = (uint8_t *)L->emit_out;
}

template <bool READS, bool SUMM>
static int chain_launch_t(const bsc_chain_launch *L) {
  hipStream_t s = (hipStream_t)L->stream;
  bsc_chain_kargs K;
  bool gc;
  chain_common<READS>(L, L->n, K, gc);
  bsc_chain_args &a = K.a;
  a.x = L->x;
  a.n_block = L->n_block;
  a.first = L->first;
  a.n = L->n;
  a.lc = L->lc;
  a.rc = L->rc;
  a.lr = L->lr;
  const chain_plan p = chain_plan_block(L->n, L->lc, L->rc, READS || (!(L->lc & 1u) && !((uintptr_t)L->cts & 15u)),
                                        (uint64_t)L->num_cus * FW);
  if (L->ev_start) (void)hipEventRecord((hipEvent_t)L->ev_start, s);
  if (p.m_runs) {
    a.origin = p.head;
    a.n_runs = p.m_runs;
    a.run_tiles = p.m_tiles;
    a.run_extra = p.m_extra;
    unsigned grid = (p.m_runs + FW - 1) / FW;
    if (grid > (unsigned)L->num_cus) grid = (unsigned)L->num_cus; /* one 1024-thread workgroup per CU, persistent */
    hipLaunchKernelGGL((bsc_chain_kernel_t<true, READS, false, SUMM>), dim3(grid), dim3(64 * FW), 0, s, K);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  /* guarded tiles, one per run: the head, then everything behind the main part — as ONE launch where both exist (each is a
   * wave or three in one workgroup, 18 us of a single wave's latency with the device otherwise idle): run 0 is the head tile,
   * the runs from 1 on lie edge_gap records further on, behind the main part */
  uint32_t edge[2][2] = {{0u, p.head}, {p.head + p.m_len, L->n}};
  const bool merged = p.m_runs && p.head == (uint32_t)FT && edge[1][1] > edge[1][0];
  for (int k = 0; k < 2; k++) {
    if (edge[k][1] <= edge[k][0]) continue;
    a.origin = edge[k][0];
    a.n_runs = (edge[k][1] - edge[k][0] + FT - 1) / FT;
    a.run_tiles = 1;
    a.run_extra = 0;
    a.edge_gap = 0;
    if (merged) {
      if (k == 1) break;
      a.n_runs = 1u + (edge[1][1] - edge[1][0] + FT - 1) / FT;
      a.edge_gap = edge[1][0] - (uint32_t)FT;
    }
    unsigned grid = (a.n_runs + FW - 1) / FW;
    if (grid > (unsigned)L->num_cus) grid = (unsigned)L->num_cus;
    hipLaunchKernelGGL((bsc_chain_kernel_t<false, READS, false, SUMM>), dim3(grid), dim3(64 * FW), 0, s, K);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (gc) { /* the window's positions into the GC-by-coverage table */
    unsigned grid = (L->n + 1023u) / 1024u;
    if (grid > (unsigned)L->num_cus) grid = (unsigned)L->num_cus;
    hipLaunchKernelGGL(bsc_gc_cov_kernel, dim3(grid), dim3(1024), 0, s,
                       reinterpret_cast<const uint16_t *>((const uint32_t *)L->het_list + a.depth_off), L->n, L->x + L->first,
                       (const uint8_t *)L->gc_bins, L->gc_n_bins, L->gc_start_pos, (unsigned long long *)L->gc_table,
                       (const unsigned long long *)L->counters, L->run_if);
  }
  hipError_t e = hipGetLastError();
  if (L->ev_stop) (void)hipEventRecord((hipEvent_t)L->ev_stop, s);
  return (int)e;
}

/*
 * Several whole blocks in one launch sequence (bsc_blocks_records): blk[b_first .. b_last] — reads-in form only.  Every block
 * is cut as above, its share of the waves in proportion to its size; the main parts of all blocks become the segments of ONE
 * launch of the unguarded kernel, their heads and tails the segments of ONE launch of the guarded one.  The segment tables are
 * put together in tab_h (pinned, from byte *cursor on; 8-byte aligned), copied to the same offset of tab_d and read by the
 * kernels through the scalar cache.  L->n = the positions of ALL the call's blocks (what the depth array is sized for).
 */
extern "C" size_t bsc_dev_chain_multi_table_bytes(uint32_t n_blocks) {
  /* per block at most 1 main + 2 edge segments and as many run0 entries, + 2 closing run0 entries per launch group (<= n_blocks) */
  return (size_t)n_blocks * (3u * sizeof(bsc_chain_args) + 5u * sizeof(uint32_t) + 16u) + 64u;
}

extern "C" int bsc_dev_launch_chain_multi(const bsc_chain_launch *L, const bsc_chain_mblock *blk, uint32_t b_first, uint32_t b_last,
                                          void *tab_h, void *tab_d, size_t *cursor) {
  hipStream_t s = (hipStream_t)L->stream;
  bsc_chain_kargs K;
  bool gc;
  chain_common<true>(L, L->n, K, gc);
  const uint32_t nb = b_last - b_first + 1u;
  const uint64_t waves = (uint64_t)L->num_cus * FW;
  uint64_t tiles_all = 0;
  for (uint32_t b = b_first; b <= b_last; b++) tiles_all += (blk[b].n + FT2 - 1) / FT2;
  /* table space: [main segments][edge segments][main run0 (+1)][edge run0 (+1)] */
  char *const base = (char *)tab_h + *cursor;
  bsc_chain_args *const seg_m = (bsc_chain_args *)base, *const seg_e = seg_m + nb;
  uint32_t *const run_m = (uint32_t *)(seg_e + 2u * nb), *const run_e = run_m + (nb + 1u);
  const size_t bytes = (((size_t)((char *)(run_e + 2u * nb + 1u) - base)) + 7u) & ~(size_t)7u;
  uint32_t n_m = 0, n_e = 0, runs_m = 0, runs_e = 0;
  for (uint32_t b = b_first; b <= b_last; b++) {
    const bsc_chain_mblock &B = blk[b];
    bsc_chain_args a = K.a; /* the launch-wide members */
    a.x = B.x;
    a.n_block = a.n = B.n;
    a.first = a.lc = a.rc = a.lr = 0;
    a.ref_off = B.ref_off;
    a.pos_off = B.pos_off;
    a.bin0 = B.bin0;
    a.bin_end = B.bin_end;
    a.first_block = b == b_first;
    a.last_block = b == b_last;
    /* the block's share of the waves, in proportion to its tiles */
    const uint64_t share = tiles_all ? (waves * ((B.n + FT2 - 1) / FT2) + tiles_all - 1) / tiles_all : waves;
    const chain_plan p = chain_plan_block(B.n, 0u, 0u, true, share);
    if (p.m_runs) {
      a.origin = p.head;
      a.n_runs = p.m_runs;
      a.run_tiles = p.m_tiles;
      a.run_extra = p.m_extra;
      a.run0 = runs_m;
      run_m[n_m] = runs_m;
      seg_m[n_m++] = a;
      runs_m += p.m_runs;
    }
    /* guarded: the head, and everything behind the main part up to the next multiple of 64 positions (where the next block's
     * records begin: the positions between get records that say "none") */
    const uint32_t n_pad = (B.n + 63u) & ~63u;
    const uint32_t edge[2][2] = {{0u, p.head}, {p.head + p.m_len, n_pad}};
    for (int k = 0; k < 2; k++) {
      if (edge[k][1] <= edge[k][0]) continue;
      a.origin = edge[k][0];
      a.n_runs = (edge[k][1] - edge[k][0] + FT - 1) / FT;
      a.run_tiles = 1;
      a.run_extra = 0;
      a.run0 = runs_e;
      run_e[n_e] = runs_e;
      seg_e[n_e++] = a;
      runs_e += a.n_runs;
    }
  }
  run_m[n_m] = runs_m;
  run_e[n_e] = runs_e;
  hipError_t e = hipMemcpyAsync((char *)tab_d + *cursor, base, bytes, hipMemcpyHostToDevice, s);
  if (e != hipSuccess) return (int)e;
  const char *const dbase = (const char *)tab_d + *cursor;
  *cursor += bytes;
  if (n_m) {
    K.segs = (const bsc_chain_args *)dbase;
    K.seg_run0 = (const uint32_t *)(dbase + ((const char *)run_m - base));
    K.n_segs = n_m;
    unsigned grid = (runs_m + FW - 1) / FW;
    if (grid > (unsigned)L->num_cus) grid = (unsigned)L->num_cus;
    hipLaunchKernelGGL((bsc_chain_kernel_t<true, true, true, false>), dim3(grid), dim3(64 * FW), 0, s, K);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  }
  if (n_e) {
    K.segs = (const bsc_chain_args *)(dbase + ((const char *)seg_e - base));
    K.seg_run0 = (const uint32_t *)(dbase + ((const char *)run_e - base));
    K.n_segs = n_e;
    unsigned grid = (runs_e + FW - 1) / FW;
    if (grid > (unsigned)L->num_cus) grid = (unsigned)L->num_cus;
    hipLaunchKernelGGL((bsc_chain_kernel_t<false, true, true, false>), dim3(grid), dim3(64 * FW), 0, s, K);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  }
  if (gc) /* block by block: the table is keyed by genome position */
    for (uint32_t b = b_first; b <= b_last; b++) {
      unsigned grid = (blk[b].n + 1023u) / 1024u;
      if (grid > (unsigned)L->num_cus) grid = (unsigned)L->num_cus;
      hipLaunchKernelGGL(bsc_gc_cov_kernel, dim3(grid), dim3(1024), 0, s,
                         reinterpret_cast<const uint16_t *>((const uint32_t *)L->het_list + K.a.depth_off) + blk[b].pos_off, blk[b].n,
                         blk[b].x, (const uint8_t *)L->gc_bins, L->gc_n_bins, L->gc_start_pos, (unsigned long long *)L->gc_table,
                         (const unsigned long long *)L->counters, 0u);
    }
  return (int)hipGetLastError();
}

extern "C" int bsc_dev_launch_chain(const bsc_chain_launch *L) {
  if (L->n == 0) return 0;
  if (L->rd) return chain_launch_t<true, false>(L);
  return L->cts_summary ? chain_launch_t<false, true>(L) : chain_launch_t<false, false>(L);
}
