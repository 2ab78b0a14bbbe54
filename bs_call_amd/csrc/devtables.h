/* devtables.h — device-side constant block and counter indices shared by kernels.hip and bscall_api.c. */
#ifndef BSCALL_AMD_DEVTABLES_H
#define BSCALL_AMD_DEVTABLES_H

#include <stdint.h>

/* q_prob columns (include/bs_call.h:148-150) as separate arrays, lfact_store (src/stats_utils.c:14-21) and
 * the per-run scalars of calc_gt_prob (src/genotype_model.c:47-48,88-89).  Built on the host with libm,
 * exactly as the reference builds them, and uploaded verbatim. */
typedef struct {
  double k[44], ln_k[44], ln_k_half[44], ln_k_one[44];
  double lfact[256];
  double log_tab[256];             /* bsm_log_tab: {invc, logc} x 128 (bsmath_tables.h) */
  unsigned long long exp_tab[256]; /* bsm_exp_tab: {tail, scale} x 128 */
  double under_conv, over_conv;
  double lrb, lrb1; /* log(ref_bias), log(0.5 * (1 + ref_bias)) */
  /* QUAL of a record as a function of om = 1 - exp(LOG10 gt_prob[max_gt]) (src/print_vcf.c:140-148) without the log: for
   * the binade e = 1023 - exponent(om) of om, phred = phred_base[e] + the number of phred_thr[e][0..3] that om does not
   * exceed (bscall_api.c builds and checks the table; fused.hip reads it) */
  double phred_thr[64][4];
  unsigned char phred_base[64];
} bsc_dev_tables;

/* unsigned long long counters[BSC_CNT_WORDS] in device memory */
#define BSC_CNT_HET_LIST 0 /* unused since round 3 (was: length of the calling kernel's heterozygous-site list) */
#define BSC_CNT_COVERED 1  /* then gt_hist[10] at 2..11, het_calls at 12 */
#define BSC_CNT_SPAN 13    /* accumulate: largest template extent of the current block (reset per block) */
#define BSC_CNT_INEXACT 14 /* accumulate: lanes whose quality / MAPQ^2 sums left the exact-float range */
#define BSC_CNT_ERR 15     /* accumulate: min over invalid templates of (index << 8 | BSC_TERR_*); all ones = none */
#define BSC_CNT_RECORDS 16 /* bsc_block_records: written records of the block being packed */
#define BSC_CNT_OVF 17     /* fused chain: CpG cytosines beyond the methylation pair table, listed until the statistics are read */
#define BSC_CNT_DEEP 18    /* accumulate, summary form: != 0 = some position holds more than 65 535 reads of one class — more than the 16-bit
                              counts of a site summary carry: the summary-in chain stands back, the reads-in chain runs (bscall_api.c) */
#define BSC_CNT_WORDS 19

/* what the reference asserts about a block's templates (src/call_genotypes.c:186-188) plus the bounds of the read
 * buffer; checked by bsc_prep_reads_kernel, ordered as the checks are made */
#define BSC_TERR_LEFT 1   /* leftmost position < block start */
#define BSC_TERR_ORI 2    /* orientation > 1 */
#define BSC_TERR_STRAND 3 /* bs_strand > 2 */
#define BSC_TERR_RANGE0 4 /* read 0 outside the read buffer */
#define BSC_TERR_RANGE1 5 /* read 1 outside the read buffer */
#define BSC_TERR_FLAGS 6  /* bits beyond BSC_TPL_WALK_KNOWN | BSC_TPL_WALKED0 in flags (a caller built before the field existed) */

/* arguments of bsc_dev_launch_chain (fused.hip), filled by bsc_chain_device (bscall_api.c) */
typedef struct bsc_chain_launch {
  const void *cts, *ref, *dbsnp;
  uint32_t x, n_block, first, n, lc, rc, lr;
  int32_t all_positions;
  uint32_t reg_start, reg_stop;
  int32_t with_stats;
  const void *tb;
  void *core_out, *het_list, *counters;
  const void *carry_in;
  void *carry_out, *stats, *pairs, *ovf_list;
  uint32_t ovf_cap;
  const void *logp;
  const void *gc_bins; /* the contig's GC bins (device) or NULL */
  uint32_t gc_n_bins, gc_start_pos;
  void *gc_table;      /* u64 [BSC_COV_CAP][101] */
  int num_cus;
  void *stream;
  void *ev_start, *ev_stop;
  double par_l, par_t, par_lrb, par_lrb1; /* 1 - under_conv, over_conv, lrb, lrb1 of the context's tables */
  void *aux_out; /* NULL, or 64 bytes per position: the second half of a bsc_vcf_rec (MC8 counts, AMQ, MQ, aq, max_gt, rs_found) */
  void *emit_out; /* NULL, or one byte per position: bsc_vcf_core.emit once more, for the passes behind the chain (packing, BCF
                     encoding) that want to know which records to fetch without fetching them (zeroed by the caller) */
  /* the reads-in form (rd != NULL): the block's reads grouped by bin instead of cts (accumulate.hip:
   * bsc_dev_launch_bin_reads); first / n may be any window of the block, lc / rc are unused */
  const void *rd, *bin_off, *seq;
  void *f_scratch;
  uint32_t n_bins, min_qual;
  /* != 0: cts holds site summaries, 48 bytes per position (bsc_dev_launch_accumulate_summary), not pile-ups; whole blocks only */
  int32_t cts_summary;
  /* 0: the launch runs; 1: only if counters[BSC_CNT_DEEP] is set; 2: only if it is not — the summary-in chain and its reads-in
   * twin are queued behind the accumulate kernel's summary form, and exactly one of them does the work */
  uint32_t run_if;
} bsc_chain_launch;

/* one block of a launch of several (bsc_dev_launch_chain_multi, bsc_dev_launch_bin_reads_multi) */
typedef struct bsc_chain_mblock {
  uint32_t x, n;     /* first position, positions */
  uint32_t tpl_end;  /* index behind the block's last template among the call's templates */
  uint32_t ref_off;  /* its n + 2 reference codes start here in the reference buffer */
  uint32_t pos_off;  /* its positions start here in the per-position arrays: a multiple of 64 */
  uint32_t bin0, bin_end; /* its bins in bin_off[]: pos_off / 64 .. */
  uint32_t ref_in;   /* bsc_blocks_submit_to_inplace: where its n + 2 codes lie in the packed reference the caller handed over */
} bsc_chain_mblock;

#endif
