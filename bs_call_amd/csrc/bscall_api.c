/*
 * bscall_api.c — host side of libbscall_amd.so, plain C (the reference's host code is C and stays C).
 * Owns the context (tables, HIP stream, grow-only device workspaces, counters) and moves host blocks
 * through the gfx950 kernels in kernels.hip.  See include/bscall_amd.h for what each entry point replaces.
 *
 * No CPU fallback lives here: if HIP reports no usable device the context cannot be created.
 */
#ifndef __HIP_PLATFORM_AMD__
#define __HIP_PLATFORM_AMD__ 1
#endif
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <pthread.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bscall_amd.h"
#include "devtables.h"
#include "bsmath.h"
#include "bsmath_tables.h"
#include "synth.h"

/* launchers implemented in kernels.hip */
int bsc_dev_launch_call(const void *cts, const void *ref, uint64_t n, void *out, uint32_t out_dw, void *skip,
                        const void *tb, void *het_list, void *counters, int num_cus, void *stream, void *ev_start,
                        void *ev_mid, void *ev_stop);
int bsc_dev_launch_accumulate(const void *rd, const void *bin_off, const void *seq, uint32_t x, uint32_t y, uint32_t min_qual, void *cts,
                              void *counters, int num_cus, void *stream);
int bsc_dev_launch_bin_reads(const void *tpl, uint32_t nr, const void *seq, uint64_t seq_bytes, uint32_t x, uint32_t y, void *tflag,
                             void *bin_cnt, void *bin_off, void *bin_cur, void *scan_tmp, size_t scan_tmp_bytes, void *rd,
                             void *counters, void *stream); /* accumulate.hip */
uint32_t bsc_dev_n_bins(uint32_t n_sites);
int bsc_dev_launch_site_stats(const void *core, const void *gtm, uint32_t gtm_stride, const void *dbsnp, uint32_t n,
                              const void *tb, const void *logp, const void *carry_in, void *carry_out, void *stats,
                              void *pairs, int num_cus, void *stream); /* sitestats.hip */
int bsc_dev_launch_gc_cov_gtm(const void *core, const void *gtm, uint32_t gtm_stride, uint32_t n, const void *gc_bins, uint32_t n_bins,
                              uint32_t start_pos, void *table, int num_cus, void *stream); /* sitestats.hip */
int bsc_dev_launch_meth_eval(void *pairs, const void *tb, const void *logp, void *stats, const void *ovf_list,
                             const void *counters, int num_cus, void *stream);
int bsc_dev_scan_tmp_bytes(uint32_t n, size_t *bytes); /* sort.hip */
int bsc_dev_launch_stream_probe(const void *cts, const void *ref, uint64_t n, void *out, void *skip, int num_cus,
                                void *stream); /* probe.hip */
int bsc_dev_launch_compact(const void *core, const void *gtm, uint32_t gtm_stride, const void *dbsnp, uint32_t n,
                           void *tile_cnt, void *tile_off, void *scan_tmp, size_t scan_tmp_bytes, void *out,
                           uint64_t out_cap, void *total, const void *emit, void *emit_ws, int num_cus, void *stream); /* compact.hip */
#define BSC_PAIR_BYTES (4ull * 512u * 512u * 8u) /* sitestats_dev.h SS_PAIR_G: [ref / non-ref][all / passed][a < 512][b < 512] u64 */
#define BSC_OVF_CAP (1u << 20)                   /* sitestats_dev.h SS_OVF_CAP */
int bsc_dev_launch_vcf(const void *gtm, uint32_t stride, const void *skip, const void *ref, const void *dbsnp, uint32_t n,
                       uint32_t x, int all_positions, uint32_t reg_start, uint32_t reg_stop, const void *tb, void *g,
                       void *out, int num_cus, void *stream);
int bsc_dev_launch_chain(const bsc_chain_launch *L); /* fused.hip */
unsigned bsc_dev_chain_quantum(int num_cus);
unsigned bsc_dev_chain_window(int num_cus, unsigned limit);
int bsc_dev_launch_synth(uint64_t seed, uint64_t first_site, uint64_t n, uint32_t coverage, uint32_t flags, void *cts,
                         void *ref, int num_cus, void *stream);
size_t bsc_dev_chain_het_bytes(uint32_t n, int num_cus, int with_depth, int reads);
size_t bsc_dev_chain_scratch_bytes(int num_cus);
size_t bsc_dev_chain_multi_table_bytes(uint32_t n_blocks);
int bsc_dev_launch_chain_multi(const bsc_chain_launch *L, const bsc_chain_mblock *blk, uint32_t b_first, uint32_t b_last, void *tab_h,
                               void *tab_d, size_t *cursor);
int bsc_dev_launch_accumulate_summary(const void *rd, const void *bin_off, const void *seq, uint32_t x, uint32_t y, uint32_t min_qual, void *cts,
                                      void *counters, int num_cus, void *stream);
size_t bsc_dev_summary_bytes(void);
int bsc_dev_launch_accumulate_multi(const void *rd, const void *bin_off, const void *seq, const void *d_blk, uint32_t n_blk, uint32_t n_bins,
                                    uint32_t min_qual, void *cts, void *counters, int num_cus, void *stream);
int bsc_dev_scan_tmp_bytes_u64(uint32_t n, size_t *bytes); /* sort.hip */
size_t bsc_dev_prep_plan_bytes(void);                     /* prepdev.hip */
int bsc_dev_launch_prep(const void *raw, uint32_t nr, const void *seq, uint64_t seq_bytes, const void *misms, uint64_t n_misms,
                        const bsc_prep_params *par, void *ms_work, void *plan, void *out_len, void *out_off, void *scan_tmp,
                        size_t scan_tmp_bytes, void *tpl_out, void *seq_out, uint64_t seq_out_cap, void *cnt, int num_cus, void *stream,
                        const void *prof_ref, uint32_t prof_x, uint32_t prof_n_ref, uint32_t prof_cap, uint32_t prof_used0, void *prof_table,
                        void *max_pos1, void *used_scan, void *prof_mask);
int bsc_dev_launch_bcf(const void *recs, const void *core, const void *aux, const void *n_recs, uint64_t max_recs, int32_t rid,
                       const bsc_bcf_ids *ids, const void *name_pos, const void *name_off, const void *name_bytes, uint32_t n_names,
                       void *tile_bytes, void *tile_off, void *scan_tmp, size_t scan_tmp_bytes, void *out, uint64_t out_cap, void *totals,
                       int num_cus, void *stream, const void *emit_len); /* bcfdev.hip */
int bsc_dev_launch_ref_pad(const void *packed, const void *d_blk, uint32_t n_blk, void *padded, uint32_t n_pos, int num_cus, void *stream);
int bsc_dev_launch_bin_reads_multi(const void *tpl, uint32_t nr, const void *seq, uint64_t seq_bytes, const void *d_blk, uint32_t n_blk,
                                   uint32_t n_bins, void *tflag, void *bin_cnt, void *bin_off, void *bin_cur, void *scan_tmp,
                                   size_t scan_tmp_bytes, void *rd, void *counters, void *stream);
#define BSC_PREP_CNT_SLOTS 64u                               /* csrc/prepdev.hip: PREP_CNT_SLOTS */
#define BSC_PREP_CNT_ALL (8u + 8u * BSC_PREP_CNT_SLOTS)     /* the eight shared words and the slots behind them */
#define BSC_LN10 (2.30258509299404568402) /* the reference's LOG10 literal (include/bs_call.h:36) */
#define BSC_HOST_CHUNK (4u << 20)         /* sites per host->device round trip (4 Mi sites = 1.2 GiB of records) */
#define BSC_MAX_LAUNCH (1ull << 31)       /* sites per launch: site indices in the het list are 32-bit */

struct bsc_context {
  bsc_params params;
  int device;
  int num_cus;
  hipStream_t stream;
  double q_prob[44][5];
  bsc_dev_tables host_tables;
  void *d_tables;
  unsigned long long *d_counters; /* BSC_CNT_WORDS */
  uint64_t sites;                 /* positions processed (host-side count) */
  /* grow-only workspaces (the reference's pileupv / gt_resv are grow-only too, src/call_genotypes.c:172-175) */
  void *d_cts, *d_ref, *d_out, *d_skip;
  size_t cap_cts, cap_ref, cap_out, cap_skip;
  void *d_het;
  size_t cap_het;
  uint64_t max_launch; /* positions per launch of the calling kernel (BSC_MAX_LAUNCH; BSC_MAX_LAUNCH_SITES in the
                          environment lowers it so that the sub-launch loop of longer calls can be tested) */
  void *d_ovf; /* fused chain: CpG cytosines beyond the methylation pair table (BSC_OVF_CAP entries of 8 bytes) */
  void *d_gc_table;      /* GC content by coverage: u64 [BSC_COV_CAP][101] (bsc_set_gc_bins) */
  void *d_gc_own;        /* bsc_set_gc_bins_host: the context's own copy of the bins */
  size_t cap_gc_own;
  const void *d_gc_bins; /* the current contig's bins (caller's device memory) */
  uint32_t gc_n_bins, gc_start_pos;
  hipEvent_t ev_chain[2]; /* bsc_set_profiling: the fused chain's launches */
  int ev_chain_valid;
  /* accumulate stage */
  void *d_tpl, *d_seq, *d_rd;
  size_t cap_tpl, cap_seq, cap_rd;
  /* grouping of the block's reads by bin (accumulate.hip): walked flags per template, bin counts / offsets / cursors, scan scratch */
  void *d_tflag, *d_bcnt, *d_boff, *d_bcur, *d_bscan;
  size_t cap_tflag, cap_bcnt, cap_boff, cap_bcur, cap_bscan;
  void *d_vg, *d_vout, *d_vdb; /* VCF record formation: called genotypes, records, dbSNP flags */
  size_t cap_vg, cap_vout, cap_vdb;
  /* packing of written records (compact.hip): records per tile, their prefix sum, scan scratch, the packed block */
  void *d_tcnt, *d_toff, *d_scantmp, *d_recs;
  size_t cap_tcnt, cap_toff, cap_scantmp, cap_recs;
  /* site statistics (sitestats.hip): the bsc_site_stats block, the printer's CpG carry (two alternating pairs of
   * words: a launch reads one and writes the other), logp[100] (src/init_param.c:56) */
  void *d_sstats, *d_pairs; /* d_pairs: CpG cytosines per (a, b), turned into profiles when the statistics are read */
  uint32_t *d_carry;
  double *d_logp;
  unsigned carry_slot;
  void *h_stage;     /* pinned staging of a submitted block's inputs (templates, reads, ref codes) */
  size_t cap_stage;
  uint64_t pending_sz;      /* positions of the submitted, not yet fetched block (0 = none) */
  uint32_t pending_stride;
  int pending_copied;       /* bsc_block_submit_to: the copy-out is already queued behind the kernels */
  /* the block whose accumulate kernels were queued last (bsc_block_check reads their verdict): host copies of its
   * inputs — the caller's buffers, or the staging area for a submitted block */
  const bsc_template *blk_tpl; /* NULL: the templates are only on the device (bsc_accumulate_device), at blk_d_tpl */
  const void *blk_d_tpl;
  uint32_t blk_x;
  /* bsc_block_records / _submit / _fetch: the pinned {INEXACT, ERR, RECORDS} block, the destination of the block in flight,
   * how many records were copied ahead of the count, the share of positions with a record the next copy is sized from */
  unsigned long long *h_cnt;
  bsc_vcf_rec *rec_out;
  uint64_t rec_cap, rec_copied;
  uint32_t rec_sz;
  int rec_pending;
  double rec_share;
  /* the BCF encoder on the device (bcfdev.hip): bytes per tile of 64 records, their prefix sum, scan scratch, the names' table, the
   * block's stream, {length, refused records}; bsc_block_bcf: the destination of the block in flight (NULL: a records block), how
   * many bytes went ahead of the length, the bytes per position the next copy is sized from */
  void *d_btb, *d_bto, *d_bscn, *d_bnm, *d_bcf, *d_btot;
  size_t cap_btb, cap_bto, cap_bscn, cap_bnm, cap_bcf, cap_btot;
  struct { /* streams handed to the caller (bsc_bcf_stream_detach) come back here (bsc_detached_free) and are taken again by the next block:
            * no hipMalloc / hipFree — a device-wide wait — in the steady state of a run that writes behind the calling */
    void *p[4];
    size_t cap[4];
  } bcf_pool;
  pthread_mutex_t pool_mu;
  int pool_mu_made;
  void *det_ptr[4]; /* the buffers that are out, by pool slot */
  hipStream_t s_det; /* the detached streams' copies out: another thread's, beside the context's own stream */
  void *h_names; /* page-locked: a block entry's names table on its way up (bsc_names_upload) */
  size_t cap_hnm;
  const bsc_bcf_names *names_up; /* set around the encoder's call of a block entry: this table is in d_bnm already */
  uint32_t names_up_n;
  uint64_t names_up_bytes;
  struct { /* the last BCF block's stream was longer than the caller's room: what bsc_block_bcf_again encodes once more, from the per-position
            * arrays still in d_vout / d_out (nothing of the block is computed or counted a second time) */
    int valid, have_names, inexact;
    uint32_t sz, n_names;
    uint64_t name_bytes;
    int32_t rid;
    bsc_bcf_ids ids;
    const void *emit;
    bsc_bcf_names names;
  } again;
  int no_h2d_turns; /* BSC_NO_H2D_TURNS in the environment when the context was made (the A/B of tools/bench_two_contexts.py) */
  void *d_emit; /* the chain's emit flags, a byte per position, for the block entries' packing / encoding passes (bsc_records_queue) */
  size_t cap_emit;
  hipEvent_t ev_h2d; /* recorded behind a block's uploads (bsc_records_queue): the next block of ANY context on this device starts its own
                        uploads behind it (bsc_h2d_turn) */
  const void *emit_hint; /* set around the bsc_vcf_compact_device call of bsc_records_queue: the flags of exactly these arrays */
  uint8_t *bcf_out;
  int stage_timing; /* BSC_STAGE_TIMING in the environment at bsc_create */
  int no_emit_bytes; /* BSC_NO_EMIT_BYTES in the environment at bsc_create (A/B builds) */
  int bcf_blk, bcf_keep; /* the block in flight is a BCF block; its stream stays on the device (bsc_block_bcf_rawdev_keep) */
  uint64_t bcf_cap, bcf_copied, bcf_bytes; /* bcf_bytes: the length of the last block's stream (also when it did not fit) */
  double bcf_share;
  /* bsc_blocks_records_submit / _fetch: the blocks of the launch in flight (their table lives in the staging area), the device
   * copies of that table and of the chain's segment tables, where the per-tile record offsets come back to */
  const bsc_chain_mblock *mb_tab;
  uint32_t mb_n;
  void *d_mblk, *d_mtab;
  size_t cap_mblk, cap_mtab;
  /* bsc_prepare_templates_device: plans, output lengths / offsets, the edited mismatch lists, scan scratch, counters; and, for the
   * host-buffer entry bsc_block_records_raw, the raw templates / reads / lists on the device */
  void *d_pplan, *d_plen, *d_poff, *d_pms, *d_pscan, *d_pcnt, *d_raw, *d_rseq, *d_rms;
  size_t cap_pplan, cap_plen, cap_poff, cap_pms, cap_pscan, cap_pcnt, cap_raw, cap_rseq, cap_rms;
  unsigned long long *h_prep; /* page-locked: what the pre-processing hands the host (bsc_prep_queue) */
  size_t cap_hprep;
  void *d_pmask; /* the read profile's byte per reference code of the block */
  size_t cap_pmask;
  void *d_pprof, *d_pmax, *d_pused; /* the read profile's counts of one call, the templates' last read positions, their running maximum */
  size_t cap_pprof, cap_pmax, cap_pused;
  void *d_refp; /* bsc_blocks_submit_to_inplace: the caller's packed reference codes, before bsc_ref_pad_kernel lays them out */
  size_t cap_refp;
  const uint32_t *mb_toff;
  int dbg_fail_summary; /* bsc_debug_fail_summary_alloc */
  int reads_fused; /* bsc_set_reads_fused: 1 = the one-kernel form always; 0 = site summaries through HBM when they can be allocated */
  void *d_fscr; /* reads-in chain: forward-count scratch lines of the resident waves */
  size_t cap_fscr;
  hipEvent_t ev_rchain[2]; /* bsc_set_profiling: the reads-in chain's launches (read descriptors, ordering, tile search, chain) */
  int ev_rchain_valid;
  hipEvent_t ev_raw[2]; /* bsc_set_profiling: a raw block's launches, pre-processing to encoder (bsc_block_*_rawdev*) */
  int ev_raw_valid;
  int ref_resident; /* the block's reference codes are in d_ref already (the read profile's upload): the chain's own upload is skipped */
  hipEvent_t ev_acc[2]; /* bsc_set_profiling: the accumulate stage's launches (prep, ordering, tile search, accumulate) */
  int ev_acc_valid;
  /* host-buffer pipeline of bsc_call_sites: two chunk buffers, copy streams, events */
  hipStream_t s_in, s_out;
  hipEvent_t ev_in[2], ev_k[2], ev_out[2];
  void *p_cts[2], *p_ref[2], *p_out[2], *p_skip[2];
  size_t p_cap_cts[2], p_cap_ref[2], p_cap_out[2], p_cap_skip[2];
  int pipe_ready;
  /* optional per-launch timing of the calling kernel (bsc_set_profiling) */
  int profiling;
#define BSC_EV_RING 32
  hipEvent_t ev[BSC_EV_RING][3]; /* a ring of event triples: the timings of the last launches can be read after a loop */
  uint64_t ev_count;             /* profiled launches so far (the next one records into ev[ev_count % BSC_EV_RING]) */
};

static __thread char bsc_errbuf[512];

static int bsc_fail(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(bsc_errbuf, sizeof bsc_errbuf, fmt, ap);
  va_end(ap);
  return code;
}

/*
 * Every entry point makes the context's device current for its HIP calls and puts the caller's device back on EVERY exit
 * path (a cleanup handler): a process that shares the HIP runtime with other code — PyTorch, in bench.py and the tests —
 * must not find its current device changed by a call into this library.
 */
typedef struct {
  int prev; /* device to restore, -1 = nothing to do */
} bsc_devguard;
static void bsc_devguard_exit(bsc_devguard *g) {
  if (g->prev >= 0) (void)hipSetDevice(g->prev);
}
#define BSC_ENTER(ctx)                                                               \
  bsc_devguard guard_ __attribute__((cleanup(bsc_devguard_exit), unused)) = {-1};    \
  do {                                                                               \
    int cur_ = -1;                                                                   \
    if (hipGetDevice(&cur_) != hipSuccess || cur_ != (ctx)->device) {                \
      HIP_TRY(hipSetDevice((ctx)->device));                                          \
      guard_.prev = cur_;                                                            \
    }                                                                                \
  } while (0)

/* the same channel for the other C files of the library (dbsnp.c, prep.c) */
int bsc_set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(bsc_errbuf, sizeof bsc_errbuf, fmt, ap);
  va_end(ap);
  return code;
}

#define HIP_TRY(call)                                                                                        \
  do {                                                                                                       \
    hipError_t e_ = (call);                                                                                  \
    if (e_ != hipSuccess) return bsc_fail(BSC_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                               \
  } while (0)

/* for the library's other translation units (csrc/bamdev.hip) */
int bsc_ctx_device(const bsc_context *ctx) { return ctx->device; }
void *bsc_ctx_stream(const bsc_context *ctx) { return (void *)ctx->stream; }

int bsc_abi_version(void) { return BSC_ABI_VERSION; }
const char *bsc_last_error(void) { return bsc_errbuf; }

void bsc_params_default(bsc_params *p) {
  p->under_conv = 0.01; /* DEFAULT_UNDER_CONVERSION, include/bs_call.h:16 */
  p->over_conv = 0.05;  /* DEFAULT_OVER_CONVERSION,  include/bs_call.h:17 */
  p->ref_bias = 2.0;    /* DEFAULT_REF_BIAS,         include/bs_call.h:18 */
  p->min_qual = 20;     /* MIN_QUAL,                 include/bs_call.h:28 */
  p->device = -1;
}

/* fill_base_prob_table (src/genotype_model.c:10-21) + lfact_store_init (src/stats_utils.c:14-21) + the two
 * prior logs (src/genotype_model.c:88-89), with libm as in the reference. */
/*
 * QUAL (src/print_vcf.c:140-148): phred = (int)(-10 * log(1 - z) / LOG10), capped at 255, z = exp(LOG10 * gt_prob[max_gt]).
 * Only the INTEGER leaves the printer, and as a function of om = 1 - z it is a staircase: phred(om) >= k exactly for
 * om <= T[k].  The chain kernel therefore does not evaluate the log (a sixth of its record formation) but looks om up
 * between the steps: T[k] is found here by bisection over the doubles with the very operations the kernel's log path ran
 * (bsm_log_t = glibc's log, the product, the division by LOG10, the truncation), so the lookup returns what those
 * operations return wherever phred() is monotone — and log's error (< 1 ulp) keeps it monotone except within a few hundred
 * ulps of om around a step, where the test suite checks every double (tests/test_phred_table.py).  A binade of om spans
 * 10 log10(2) = 3.01 units of phred, i.e. at most four steps: per binade the value at its upper end and the four steps
 * above it.  om is a multiple of 2^-53 (z < 1 is a double), so binades 0 .. 53 occur; the table has 64.
 */
static int bsc_phred_of_om(double om) {
  const double lg = bsm_log_t(om, bsm_log_tab);
  const double v = (-10.0 * lg) / BSC_LN10;
  const int p = (int)v;
  return p > 255 ? 255 : p;
}
static int bsc_build_phred_table(bsc_dev_tables *t) {
  double T[260]; /* T[k], k = 1 .. 255: the largest om with phred(om) >= k; beyond 255: never */
  for (int k = 1; k <= 255; k++) {
    uint64_t lo = bsm_bits(0x1p-1000), hi = bsm_bits(1.0); /* phred(lo) = 255 >= k > phred(hi) = 0 */
    if (bsc_phred_of_om(bsm_from_bits(lo)) < k || bsc_phred_of_om(bsm_from_bits(hi)) >= k) return -1;
    while (hi - lo > 1) {
      const uint64_t mid = lo + (hi - lo) / 2;
      if (bsc_phred_of_om(bsm_from_bits(mid)) >= k) lo = mid;
      else hi = mid;
    }
    T[k] = bsm_from_bits(lo);
    if (k > 1 && !(T[k] < T[k - 1])) return -1;
  }
  for (int k = 256; k < 260; k++) T[k] = -1.0;
  for (int e = 0; e < 64; e++) {
    const double top = e == 0 ? 1.0 : bsm_from_bits(bsm_bits(ldexp(1.0, 1 - e)) - 1); /* the largest om of the binade */
    const int b = bsc_phred_of_om(top);
    if (bsc_phred_of_om(ldexp(1.0, -e)) - b > 4) return -1;
    t->phred_base[e] = (unsigned char)b;
    for (int j = 0; j < 4; j++) t->phred_thr[e][j] = T[b + 1 + j];
  }
  return 0;
}

int bsc_phred_table(double *thr_64x4, unsigned char *base_64) {
  if (!thr_64x4 || !base_64) return bsc_fail(BSC_ERR_ARG, "bsc_phred_table: NULL argument");
  bsc_dev_tables *t = malloc(sizeof *t);
  if (!t) return bsc_fail(BSC_ERR_NOMEM, "bsc_phred_table: out of memory");
  const int rc = bsc_build_phred_table(t);
  if (!rc) {
    memcpy(thr_64x4, t->phred_thr, sizeof t->phred_thr);
    memcpy(base_64, t->phred_base, sizeof t->phred_base);
  }
  free(t);
  return rc ? bsc_fail(BSC_ERR_ARG, "bsc_phred_table: the QUAL staircase of this build's log is not monotone") : BSC_OK;
}

static void bsc_build_tables(bsc_context *ctx) {
  bsc_dev_tables *t = &ctx->host_tables;
  for (int q = 0; q <= 43; q++) {
    double e = exp(-.1 * (double)q * BSC_LN10);
    if (e > .5) e = .5;
    double k = e / (3.0 - 4.0 * e);
    ctx->q_prob[q][0] = e;
    ctx->q_prob[q][1] = t->k[q] = k;
    ctx->q_prob[q][2] = t->ln_k[q] = log(k);
    ctx->q_prob[q][3] = t->ln_k_half[q] = log(0.5 + k);
    ctx->q_prob[q][4] = t->ln_k_one[q] = log(1.0 + k);
  }
  t->lfact[0] = t->lfact[1] = 0.0;
  double l = 0.0;
  for (int i = 2; i < 256; i++) {
    l += log((double)i);
    t->lfact[i] = l;
  }
  memcpy(t->log_tab, bsm_log_tab, sizeof t->log_tab);
  memcpy(t->exp_tab, bsm_exp_tab, sizeof t->exp_tab);
  t->under_conv = ctx->params.under_conv;
  t->over_conv = ctx->params.over_conv;
  t->lrb = log(ctx->params.ref_bias);
  t->lrb1 = log(0.5 * (1.0 + ctx->params.ref_bias));
}

int bsc_create(const bsc_params *params, bsc_context **out) {
  if (!out) return bsc_fail(BSC_ERR_ARG, "bsc_create: out is NULL");
  *out = NULL;
  bsc_params p;
  if (params) p = *params;
  else bsc_params_default(&p);
  if (!(p.under_conv >= 0.0 && p.under_conv < 1.0) || !(p.over_conv >= 0.0 && p.over_conv < 1.0) || !(p.ref_bias > 0.0))
    return bsc_fail(BSC_ERR_ARG, "bsc_create: conversion rates must be in [0,1) and ref_bias > 0");
  /* get_Z divides by (1 - under_conv - over_conv) (src/genotype_model.c:25-26): it must be positive and finite */
  if (!((1.0 - p.under_conv) - p.over_conv >= 0x1p-20))
    return bsc_fail(BSC_ERR_ARG, "bsc_create: under_conv + over_conv must be < 1 (by at least 2^-20)");
  if (p.min_qual < 1) p.min_qual = 1; /* src/parse_args.c:170-171 */
  if (p.min_qual > 43) p.min_qual = 43;

  int ndev = 0;
  hipError_t e = hipGetDeviceCount(&ndev);
  if (e != hipSuccess || ndev <= 0)
    return bsc_fail(BSC_ERR_NO_DEVICE, "bsc_create: no HIP device (%s); this library has no CPU path",
                    e == hipSuccess ? "count = 0" : hipGetErrorString(e));
  int dev = p.device, cur = -1;
  HIP_TRY(hipGetDevice(&cur));
  if (dev < 0) dev = cur;
  if (dev >= ndev) return bsc_fail(BSC_ERR_ARG, "bsc_create: device %d out of range (%d devices)", dev, ndev);
  bsc_devguard guard_ __attribute__((cleanup(bsc_devguard_exit), unused)) = {-1}; /* the caller's device comes back */
  if (dev != cur) {
    HIP_TRY(hipSetDevice(dev));
    guard_.prev = cur;
  }
  hipDeviceProp_t prop;
  HIP_TRY(hipGetDeviceProperties(&prop, dev));
  if (strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return bsc_fail(BSC_ERR_NO_DEVICE, "bsc_create: device %d is %s; the kernels are built for gfx950 only", dev,
                    prop.gcnArchName);

  bsc_context *ctx = calloc(1, sizeof *ctx);
  if (!ctx) return bsc_fail(BSC_ERR_NOMEM, "bsc_create: out of host memory");
  ctx->params = p;
  ctx->device = dev;
  ctx->num_cus = prop.multiProcessorCount;
  ctx->max_launch = BSC_MAX_LAUNCH;
  ctx->rec_share = 0.55; /* WGBS: a record for every C and G and little else */
  ctx->stage_timing = getenv("BSC_STAGE_TIMING") != NULL;
  ctx->no_emit_bytes = getenv("BSC_NO_EMIT_BYTES") != NULL;
  ctx->no_h2d_turns = getenv("BSC_NO_H2D_TURNS") != NULL;
  {
    const char *ml = getenv("BSC_MAX_LAUNCH_SITES");
    if (ml && *ml) {
      const unsigned long long v = strtoull(ml, NULL, 10) & ~63ull; /* whole wave-tiles: keeps both arrays 16-byte aligned */
      if (v >= 64 && v < BSC_MAX_LAUNCH) ctx->max_launch = v;
    }
  }
  bsc_build_tables(ctx);
  int rc = BSC_OK;
  if (bsc_build_phred_table(&ctx->host_tables)) { /* cannot happen with a monotone log; never a silently wrong QUAL */
    rc = bsc_fail(BSC_ERR_ARG, "bsc_create: the QUAL staircase of this build's log is not monotone");
    bsc_destroy(ctx);
    return rc;
  }
  if (hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) != hipSuccess ||
      hipMalloc(&ctx->d_tables, sizeof(bsc_dev_tables)) != hipSuccess ||
      hipMalloc((void **)&ctx->d_counters, BSC_CNT_WORDS * sizeof(unsigned long long)) != hipSuccess ||
      hipMemcpy(ctx->d_tables, &ctx->host_tables, sizeof(bsc_dev_tables), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemset(ctx->d_counters, 0, BSC_CNT_WORDS * sizeof(unsigned long long)) != hipSuccess ||
      /* hipMemset of device memory returns before the bytes are written, and the context's stream does not wait for the null stream: without
       * this wait the zeroes could land behind the first block's "no error" word (all ones) — once in ~2 400 fresh contexts in
       * tools/fuzz_block.py, "read -4 of template 0 lies outside the read buffer" (the word read back as 0) */
      hipStreamSynchronize(NULL) != hipSuccess) {
    rc = bsc_fail(BSC_ERR_HIP, "bsc_create: device setup failed: %s", hipGetErrorString(hipGetLastError()));
    bsc_destroy(ctx);
    return rc;
  }
  *out = ctx;
  return BSC_OK;
}

static void bsc_h2d_turn_forget(bsc_context *ctx);

int bsc_destroy(bsc_context *ctx) {
  if (!ctx) return BSC_OK;
  bsc_devguard guard_ __attribute__((cleanup(bsc_devguard_exit), unused)) = {-1};
  {
    int cur = -1;
    if (hipGetDevice(&cur) != hipSuccess || cur != ctx->device) {
      (void)hipSetDevice(ctx->device);
      guard_.prev = cur;
    }
  }
  if (ctx->stream) {
    hipStreamSynchronize(ctx->stream);
    hipStreamDestroy(ctx->stream);
  }
  hipFree(ctx->d_tables);
  hipFree(ctx->d_counters);
  hipFree(ctx->d_cts);
  hipFree(ctx->d_ref);
  hipFree(ctx->d_out);
  hipFree(ctx->d_skip);
  hipFree(ctx->d_het);
  hipFree(ctx->d_ovf);
  hipFree(ctx->d_gc_table);
  hipFree(ctx->d_gc_own);
  for (int i = 0; i < 2; i++)
    if (ctx->ev_chain[i]) hipEventDestroy(ctx->ev_chain[i]);
  for (int i = 0; i < 2; i++)
    if (ctx->ev_acc[i]) hipEventDestroy(ctx->ev_acc[i]);
  for (int i = 0; i < 2; i++)
    if (ctx->ev_rchain[i]) hipEventDestroy(ctx->ev_rchain[i]);
  hipFree(ctx->d_fscr);
  hipFree(ctx->d_tpl);
  hipFree(ctx->d_seq);
  hipFree(ctx->d_rd);
  hipFree(ctx->d_tflag);
  hipFree(ctx->d_bcnt);
  hipFree(ctx->d_boff);
  hipFree(ctx->d_bcur);
  hipFree(ctx->d_bscan);
  for (int b = 0; b < 2; b++) {
    hipFree(ctx->p_cts[b]);
    hipFree(ctx->p_ref[b]);
    hipFree(ctx->p_out[b]);
    hipFree(ctx->p_skip[b]);
    if (ctx->ev_in[b]) hipEventDestroy(ctx->ev_in[b]);
    if (ctx->ev_k[b]) hipEventDestroy(ctx->ev_k[b]);
    if (ctx->ev_out[b]) hipEventDestroy(ctx->ev_out[b]);
  }
  if (ctx->s_in) hipStreamDestroy(ctx->s_in);
  if (ctx->s_out) hipStreamDestroy(ctx->s_out);
  hipFree(ctx->d_vg);
  hipFree(ctx->d_sstats);
    hipFree(ctx->d_pairs);
  hipFree(ctx->d_tcnt);
  hipFree(ctx->d_toff);
  hipFree(ctx->d_scantmp);
  hipFree(ctx->d_recs);
  hipFree(ctx->d_btb);
  hipFree(ctx->d_bto);
  hipFree(ctx->d_bscn);
  hipFree(ctx->d_bnm);
  if (ctx->h_names) hipHostFree(ctx->h_names);
  for (int k = 0; k < 4; k++)
    if (ctx->bcf_pool.p[k]) hipFree(ctx->bcf_pool.p[k]);
  if (ctx->s_det) hipStreamDestroy(ctx->s_det);
  if (ctx->pool_mu_made) pthread_mutex_destroy(&ctx->pool_mu);
  hipFree(ctx->d_bcf);
  hipFree(ctx->d_btot);
  hipFree(ctx->d_emit);
  for (int i = 0; i < 2; i++)
    if (ctx->ev_raw[i]) hipEventDestroy(ctx->ev_raw[i]);
  bsc_h2d_turn_forget(ctx);
  hipFree(ctx->d_mblk);
  hipFree(ctx->d_mtab);
  hipFree(ctx->d_refp);
  hipFree(ctx->d_pplan);
  hipFree(ctx->d_plen);
  hipFree(ctx->d_poff);
  hipFree(ctx->d_pms);
  hipFree(ctx->d_pscan);
  hipFree(ctx->d_pcnt);
  hipFree(ctx->d_raw);
  hipFree(ctx->d_rseq);
  hipFree(ctx->d_rms);
  hipFree(ctx->d_pprof);
  hipFree(ctx->d_pmax);
  hipFree(ctx->d_pused);
  hipFree(ctx->d_pmask);
  if (ctx->h_prep) hipHostFree(ctx->h_prep);
  hipFree(ctx->d_carry);
  hipFree(ctx->d_logp);
  hipFree(ctx->d_vout);
  hipFree(ctx->d_vdb);
  if (ctx->h_stage) hipHostFree(ctx->h_stage);
  if (ctx->h_cnt) hipHostFree(ctx->h_cnt);
  for (int r = 0; r < BSC_EV_RING; r++)
    for (int i = 0; i < 3; i++)
      if (ctx->ev[r][i]) hipEventDestroy(ctx->ev[r][i]);
  free(ctx);
  return BSC_OK;
}

int bsc_get_tables(const bsc_context *ctx, double *q_prob_44x5, double *lfact_256) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_get_tables: ctx is NULL");
  if (q_prob_44x5) memcpy(q_prob_44x5, ctx->q_prob, sizeof ctx->q_prob);
  if (lfact_256) memcpy(lfact_256, ctx->host_tables.lfact, sizeof ctx->host_tables.lfact);
  return BSC_OK;
}

static int bsc_reserve(void **p, size_t *cap, size_t need) {
  if (need <= *cap) return BSC_OK;
  if (*p) {
    hipError_t e = hipFree(*p);
    *p = NULL;
    *cap = 0;
    if (e != hipSuccess) return bsc_fail(BSC_ERR_HIP, "hipFree failed: %s", hipGetErrorString(e));
  }
  size_t sz = need + need / 4; /* grow-only with slack */
  hipError_t e = hipMalloc(p, sz);
  if (e != hipSuccess) {
    (void)hipGetLastError(); /* a failed allocation stays behind as the runtime's "last error": the next launch check would report it */
    sz = need;
    e = hipMalloc(p, sz);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    *p = NULL;
    return bsc_fail(BSC_ERR_NOMEM, "hipMalloc(%zu) failed: %s", sz, hipGetErrorString(e));
  }
  *cap = sz;
  return BSC_OK;
}

/* bsc_reserve for a workspace the caller can do without: a failure leaves neither an error message nor the runtime's sticky error */
static int bsc_try_reserve(void **p, size_t *cap, size_t need) {
  if (need <= *cap) return 1;
  char keep[sizeof bsc_errbuf];
  memcpy(keep, bsc_errbuf, sizeof keep);
  const int ok = bsc_reserve(p, cap, need) == BSC_OK;
  if (!ok) {
    (void)hipGetLastError();
    memcpy(bsc_errbuf, keep, sizeof keep);
  }
  return ok;
}

static int bsc_check_stride(uint32_t out_stride) {
  if (out_stride != 200 && out_stride != 208)
    return bsc_fail(BSC_ERR_ARG, "out_stride must be 200 (gt_meth[]) or 208 (gt_vcf[]), got %u", out_stride);
  return BSC_OK;
}

int bsc_call_sites_device(bsc_context *ctx, const void *d_cts, const void *d_ref, uint64_t n, void *d_out,
                          uint32_t out_stride, void *d_skip, void *stream) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_call_sites_device: ctx is NULL");
  int rc = bsc_check_stride(out_stride);
  if (rc) return rc;
  if (n == 0) return BSC_OK;
  if (!d_cts || !d_ref || !d_out || !d_skip) return bsc_fail(BSC_ERR_ARG, "bsc_call_sites_device: NULL buffer");
  if (((uintptr_t)d_cts & 15u) || ((uintptr_t)d_out & 15u))
    return bsc_fail(BSC_ERR_ARG, "bsc_call_sites_device: d_cts and d_out must be 16-byte aligned");
  BSC_ENTER(ctx);
  hipStream_t s = (hipStream_t)stream; /* NULL = HIP's default stream, as everywhere in HIP */
  uint64_t done = 0;
  while (done < n) {
    uint64_t m = n - done;
    if (m > ctx->max_launch) m = ctx->max_launch; /* a multiple of 64 sites: keeps the 16-byte alignment of both arrays */
    rc = bsc_reserve(&ctx->d_het, &ctx->cap_het, (size_t)((m + 63u) / 64u) * 8u); /* one 64-bit mask per wave-tile */
    if (rc) return rc;
    int e = bsc_dev_launch_call((const char *)d_cts + done * 104u, (const char *)d_ref + done, m,
                                (char *)d_out + done * out_stride, out_stride / 4u, (char *)d_skip + done,
                                ctx->d_tables, ctx->d_het, ctx->d_counters, ctx->num_cus, s,
                                ctx->profiling ? ctx->ev[ctx->ev_count % BSC_EV_RING][0] : NULL,
                                ctx->profiling ? ctx->ev[ctx->ev_count % BSC_EV_RING][1] : NULL,
                                ctx->profiling ? ctx->ev[ctx->ev_count % BSC_EV_RING][2] : NULL);
    if (ctx->profiling) ctx->ev_count++;
    if (e) return bsc_fail(BSC_ERR_HIP, "kernel launch failed: %s", hipGetErrorString((hipError_t)e));
    done += m;
  }
  ctx->sites += n;
  return BSC_OK;
}

#define BSC_PIPE_CHUNK (1u << 20) /* sites per pipeline stage of bsc_call_sites (1 Mi sites = 305 MiB of records) */

static int bsc_pipe_init(bsc_context *ctx) {
  if (ctx->pipe_ready) return BSC_OK;
  HIP_TRY(hipStreamCreateWithFlags(&ctx->s_in, hipStreamNonBlocking));
  HIP_TRY(hipStreamCreateWithFlags(&ctx->s_out, hipStreamNonBlocking));
  for (int b = 0; b < 2; b++) {
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_in[b], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_k[b], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&ctx->ev_out[b], hipEventDisableTiming));
  }
  ctx->pipe_ready = 1;
  return BSC_OK;
}

/* Three-stage pipeline over chunks of the block: H2D copy of chunk k+1 (stream s_in), kernels of chunk k (the
 * context's stream), D2H copy of chunk k-1 (stream s_out), two buffer sets.  With pinned host buffers
 * (bsc_alloc_host) the copies are true DMA and overlap the kernels; with pageable memory the runtime stages them. */
int bsc_call_sites(bsc_context *ctx, const bsc_pileup *cts, const uint8_t *ref, uint64_t n, void *out,
                   uint32_t out_stride, uint8_t *skip) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_call_sites: ctx is NULL");
  int rc = bsc_check_stride(out_stride);
  if (rc) return rc;
  if (n == 0) return BSC_OK;
  if (!cts || !ref || !out || !skip) return bsc_fail(BSC_ERR_ARG, "bsc_call_sites: NULL buffer");
  BSC_ENTER(ctx);
  if ((rc = bsc_pipe_init(ctx))) return rc;
  const uint64_t chunk = n < BSC_PIPE_CHUNK ? n : BSC_PIPE_CHUNK;
  const int nbuf = n > chunk ? 2 : 1;
  for (int b = 0; b < nbuf; b++) {
    if ((rc = bsc_reserve(&ctx->p_cts[b], &ctx->p_cap_cts[b], (size_t)chunk * 104u))) return rc;
    if ((rc = bsc_reserve(&ctx->p_ref[b], &ctx->p_cap_ref[b], (size_t)chunk))) return rc;
    if ((rc = bsc_reserve(&ctx->p_out[b], &ctx->p_cap_out[b], (size_t)chunk * out_stride))) return rc;
    if ((rc = bsc_reserve(&ctx->p_skip[b], &ctx->p_cap_skip[b], (size_t)chunk))) return rc;
  }
  uint64_t k = 0;
  for (uint64_t done = 0; done < n; done += chunk, k++) {
    const int b = (int)(k & 1);
    const uint64_t m = (n - done) < chunk ? (n - done) : chunk;
    /* buffer set b is free for new input once the kernels of chunk k-2 have run */
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(ctx->s_in, ctx->ev_k[b], 0));
    HIP_TRY(hipMemcpyAsync(ctx->p_cts[b], cts + done, (size_t)m * 104u, hipMemcpyHostToDevice, ctx->s_in));
    HIP_TRY(hipMemcpyAsync(ctx->p_ref[b], ref + done, (size_t)m, hipMemcpyHostToDevice, ctx->s_in));
    HIP_TRY(hipEventRecord(ctx->ev_in[b], ctx->s_in));
    /* kernels: need the input, and the output buffer of chunk k-2 must have left */
    HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_in[b], 0));
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_out[b], 0));
    rc = bsc_call_sites_device(ctx, ctx->p_cts[b], ctx->p_ref[b], m, ctx->p_out[b], out_stride, ctx->p_skip[b], ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_k[b], ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->s_out, ctx->ev_k[b], 0));
    HIP_TRY(hipMemcpyAsync((char *)out + done * out_stride, ctx->p_out[b], (size_t)m * out_stride, hipMemcpyDeviceToHost,
                           ctx->s_out));
    HIP_TRY(hipMemcpyAsync(skip + done, ctx->p_skip[b], (size_t)m, hipMemcpyDeviceToHost, ctx->s_out));
    HIP_TRY(hipEventRecord(ctx->ev_out[b], ctx->s_out));
  }
  HIP_TRY(hipStreamSynchronize(ctx->s_out));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return BSC_OK;
}

/* Pinned host memory for the buffers handed to the host-buffer entries (hipHostMalloc / hipHostFree). */
void *bsc_alloc_host(uint64_t bytes) {
  void *p = NULL;
  if (hipHostMalloc(&p, (size_t)bytes, hipHostMallocDefault) != hipSuccess) {
    bsc_fail(BSC_ERR_NOMEM, "bsc_alloc_host(%llu) failed", (unsigned long long)bytes);
    return NULL;
  }
  return p;
}

void bsc_free_host(void *p) {
  if (p) hipHostFree(p);
}

/* leftmost position of a template (src/call_genotypes.c:183-185) */
static uint32_t bsc_leftmost(const bsc_template *t) {
  uint32_t x1 = t->pos[0];
  if (x1 == 0) x1 = t->pos[1];
  else if (t->pos[1] > 0 && t->pos[1] < x1) x1 = t->pos[1];
  return x1;
}

/* Uploads a block and queues the accumulate kernels: the pile-up of x..y ends up in ctx->d_cts; bsc_block_check()
 * collects the device's verdict on the templates (the reference's asserts). */
static int bsc_stage_reserve(bsc_context *ctx, size_t need) {
  if (need <= ctx->cap_stage) return BSC_OK;
  if (ctx->h_stage) hipHostFree(ctx->h_stage);
  ctx->h_stage = NULL;
  ctx->cap_stage = 0;
  size_t sz = need + need / 4;
  if (hipHostMalloc(&ctx->h_stage, sz, hipHostMallocDefault) != hipSuccess)
    return bsc_fail(BSC_ERR_NOMEM, "hipHostMalloc(%zu) failed", sz);
  ctx->cap_stage = sz;
  return BSC_OK;
}

/* `stage` != 0: the inputs are first copied into the context's pinned staging area so that the caller's buffers can be
 * recycled as soon as the call returns and the H2D copies are true asynchronous DMA (bsc_block_submit). */
static int bsc_accumulate_queue2(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq,
                                 uint64_t seq_bytes, uint32_t x, uint32_t y, const uint8_t *ref, int stage);

static int bsc_accumulate_queue(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq,
                                uint64_t seq_bytes, uint32_t x, uint32_t y) {
  return bsc_accumulate_queue2(ctx, tpl, nr, seq, seq_bytes, x, y, NULL, 0);
}

/* Workspaces of the accumulate stage for a block of nr templates over positions x .. y (grow-only). */
static int bsc_accumulate_reserve(bsc_context *ctx, uint32_t nr, uint32_t x, uint32_t y, size_t *scan_bytes) {
  const uint64_t sz = (uint64_t)y - x + 1;
  if (sz > 0xffffffffull) return bsc_fail(BSC_ERR_ARG, "accumulate: block longer than 2^32 - 1 positions");
  const size_t nb1 = (size_t)bsc_dev_n_bins((uint32_t)sz) + 1u;
  int rc;
  *scan_bytes = 0;
  if (nr > 0x7fffffffu) return bsc_fail(BSC_ERR_ARG, "accumulate: more than 2^31 - 1 templates in one block");
  if (bsc_dev_scan_tmp_bytes((uint32_t)nb1, scan_bytes)) return bsc_fail(BSC_ERR_HIP, "accumulate: scan size query failed");
  if ((rc = bsc_reserve(&ctx->d_rd, &ctx->cap_rd, (size_t)(nr ? nr : 1) * 48u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_tflag, &ctx->cap_tflag, (size_t)(nr ? nr : 1)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_bcnt, &ctx->cap_bcnt, nb1 * 4u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_boff, &ctx->cap_boff, nb1 * 4u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_bcur, &ctx->cap_bcur, nb1 * 4u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_bscan, &ctx->cap_bscan, *scan_bytes ? *scan_bytes : 1))) return rc;
  return BSC_OK;
}

/* template checks, read descriptors and their grouping by bin on stream s; SPAN, INEXACT = 0 and ERR = all ones first (the
 * kernels take maxima / the minimum) */
static int bsc_reads_prepare(bsc_context *ctx, const void *d_tpl, uint32_t nr, const void *d_seq, uint64_t seq_bytes, uint32_t x,
                             uint32_t y, size_t scan_bytes, hipStream_t s) {
  HIP_TRY(hipMemsetAsync(ctx->d_counters + BSC_CNT_SPAN, 0, 2 * sizeof(unsigned long long), s));
  HIP_TRY(hipMemsetAsync(ctx->d_counters + BSC_CNT_ERR, 0xff, sizeof(unsigned long long), s));
  HIP_TRY(hipMemsetAsync(ctx->d_counters + BSC_CNT_DEEP, 0, sizeof(unsigned long long), s));
  int e = bsc_dev_launch_bin_reads(d_tpl, nr, d_seq, seq_bytes, x, y, ctx->d_tflag, ctx->d_bcnt, ctx->d_boff, ctx->d_bcur,
                                   ctx->d_bscan, scan_bytes, ctx->d_rd, ctx->d_counters, s);
  if (e) return bsc_fail(BSC_ERR_HIP, "read grouping launch failed: %s", hipGetErrorString((hipError_t)e));
  return BSC_OK;
}

/* Queues the accumulate kernels over device-resident templates and read bytes on stream s: pile-up of x .. y -> d_cts
 * ((y - x + 1 rounded up to whole 64-position tiles) x 104 bytes). */
static int bsc_accumulate_launch(bsc_context *ctx, const void *d_tpl, uint32_t nr, const void *d_seq, uint64_t seq_bytes,
                                 uint32_t x, uint32_t y, void *d_cts, size_t scan_bytes, hipStream_t s) {
  if (ctx->profiling) {
    if (!ctx->ev_acc[0])
      for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreate(&ctx->ev_acc[i]));
    HIP_TRY(hipEventRecord(ctx->ev_acc[0], s));
  }
  int rc = bsc_reads_prepare(ctx, d_tpl, nr, d_seq, seq_bytes, x, y, scan_bytes, s);
  if (rc) return rc;
  int e = bsc_dev_launch_accumulate(ctx->d_rd, ctx->d_boff, d_seq, x, y, (uint32_t)ctx->params.min_qual, d_cts, ctx->d_counters,
                                    ctx->num_cus, s);
  if (e) return bsc_fail(BSC_ERR_HIP, "accumulate launch failed: %s", hipGetErrorString((hipError_t)e));
  if (ctx->profiling) {
    HIP_TRY(hipEventRecord(ctx->ev_acc[1], s));
    ctx->ev_acc_valid = 1;
  }
  return BSC_OK;
}

static int bsc_accumulate_queue2(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq,
                                 uint64_t seq_bytes, uint32_t x, uint32_t y, const uint8_t *ref, int stage) {
  if (y < x) return bsc_fail(BSC_ERR_ARG, "accumulate: y (%u) < x (%u) (reference asserts y >= x)", y, x);
  if (nr && (!tpl || !seq)) return bsc_fail(BSC_ERR_ARG, "accumulate: NULL template or read buffer");
  /* one block in flight per context, whichever pair of calls submitted it: the host-buffer paths share the staging area, the
   * device workspaces (templates, reads, reference, results) and the verdict counters */
  if (ctx->pending_sz || ctx->rec_pending)
    return bsc_fail(BSC_ERR_ARG, "a submitted block has not been fetched (bsc_block_fetch / bsc_block_records_fetch first)");
  /* The templates themselves are checked where they are read anyway — by bsc_prep_reads_kernel, on the device
   * (a host loop over a million 40-byte templates costs more than the whole GPU side of the block); the verdict is
   * collected by bsc_block_check(). */
  BSC_ENTER(ctx);
  const uint64_t sz = (uint64_t)y - x + 1;
  const uint64_t n_wt = (sz + 63) / 64;
  int rc;
  size_t scan_bytes = 0;
  if ((rc = bsc_reserve(&ctx->d_cts, &ctx->cap_cts, (size_t)n_wt * 64u * 104u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_tpl, &ctx->cap_tpl, (size_t)(nr ? nr : 1) * sizeof(bsc_template)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_seq, &ctx->cap_seq, (size_t)(seq_bytes ? seq_bytes : 1)))) return rc;
  if ((rc = bsc_accumulate_reserve(ctx, nr, x, y, &scan_bytes))) return rc;
  if (stage) {
    /* the previous block's copies out of the staging area have completed: bsc_block_fetch synchronised the stream */
    const size_t b_tpl = (size_t)nr * sizeof(bsc_template), b_seq = (size_t)seq_bytes, b_ref = ref ? (size_t)sz : 0;
    const size_t o_seq = (b_tpl + 63u) & ~(size_t)63u, o_ref = (o_seq + b_seq + 63u) & ~(size_t)63u;
    if ((rc = bsc_stage_reserve(ctx, o_ref + b_ref + 64u))) return rc;
    char *st = ctx->h_stage;
    if (nr) {
      memcpy(st, tpl, b_tpl);
      memcpy(st + o_seq, seq, b_seq);
      tpl = (const bsc_template *)st;
      seq = (const uint8_t *)(st + o_seq);
    }
    if (ref) {
      memcpy(st + o_ref, ref, b_ref);
      if ((rc = bsc_reserve(&ctx->d_ref, &ctx->cap_ref, (size_t)sz))) return rc;
      HIP_TRY(hipMemcpyAsync(ctx->d_ref, st + o_ref, b_ref, hipMemcpyHostToDevice, ctx->stream));
    }
  }
  ctx->blk_tpl = tpl;
  ctx->blk_d_tpl = ctx->d_tpl;
  ctx->blk_x = x;
  ctx->mb_n = 0;
  if (nr) {
    HIP_TRY(hipMemcpyAsync(ctx->d_tpl, tpl, (size_t)nr * sizeof(bsc_template), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(hipMemcpyAsync(ctx->d_seq, seq, (size_t)seq_bytes, hipMemcpyHostToDevice, ctx->stream));
  }
  return bsc_accumulate_launch(ctx, ctx->d_tpl, nr, ctx->d_seq, seq_bytes, x, y, ctx->d_cts, scan_bytes, ctx->stream);
}

/*
 * Waits for the accumulate kernels queued last and reads their verdict on the block: an invalid template ->
 * BSC_ERR_ARG naming the first one and the reference assert it breaks; *inexact -> positions whose float sums left the
 * exact range (BSC_WARN_INEXACT for the caller).
 */
static int bsc_verdict(bsc_context *ctx, const unsigned long long f[2], int *inexact);

static int bsc_block_check_on(bsc_context *ctx, hipStream_t s, int *inexact) {
  unsigned long long f[2]; /* INEXACT, ERR */
  HIP_TRY(hipMemcpyAsync(f, ctx->d_counters + BSC_CNT_INEXACT, sizeof f, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return bsc_verdict(ctx, f, inexact);
}

static int bsc_block_check(bsc_context *ctx, int *inexact) { return bsc_block_check_on(ctx, ctx->stream, inexact); }

/* f = {INEXACT, ERR} as read from the device counters after the block's prep kernel has run */
static int bsc_verdict(bsc_context *ctx, const unsigned long long f[2], int *inexact) {
  if (f[1] != ~0ull) {
    const uint32_t i = (uint32_t)(f[1] >> 8);
    bsc_template dt;
    const bsc_template *t = &dt;
    if (ctx->blk_tpl) t = ctx->blk_tpl + i;
    else HIP_TRY(hipMemcpy(&dt, (const char *)ctx->blk_d_tpl + (size_t)i * sizeof(bsc_template), sizeof dt, hipMemcpyDeviceToHost));
    uint32_t blk_x = ctx->blk_x;
    for (uint32_t b = 0; b < ctx->mb_n; b++) /* several blocks in flight: the template is named with its own block's start */
      if (i < ctx->mb_tab[b].tpl_end) {
        blk_x = ctx->mb_tab[b].x;
        break;
      }
    switch ((int)(f[1] & 0xffu)) {
      case BSC_TERR_LEFT:
        return bsc_fail(BSC_ERR_ARG, "accumulate: template %u starts at %u, left of the block start %u", i,
                        bsc_leftmost(t), blk_x);
      case BSC_TERR_ORI:
        return bsc_fail(BSC_ERR_ARG, "accumulate: template %u has orientation %u (reference asserts ori < 2)", i,
                        t->orientation);
      case BSC_TERR_STRAND: return bsc_fail(BSC_ERR_ARG, "accumulate: template %u has bs_strand %u", i, t->bs_strand);
      case BSC_TERR_FLAGS:
        return bsc_fail(BSC_ERR_ARG, "accumulate: template %u has flags 0x%x (only BSC_TPL_WALK_KNOWN | BSC_TPL_WALKED0 are defined: ABI %d)", i, t->flags,
                        BSC_ABI_VERSION);
      default:
        return bsc_fail(BSC_ERR_ARG, "accumulate: read %d of template %u lies outside the read buffer",
                        (int)(f[1] & 0xffu) - BSC_TERR_RANGE0, i);
    }
  }
  if (inexact) *inexact = f[0] != 0;
  return BSC_OK;
}

static int bsc_inexact_status(int inexact) {
  if (inexact) {
    bsc_fail(BSC_WARN_INEXACT, "accumulate: position(s) with a quality or MAPQ^2 sum >= 2^24: the reference's float "
             "sums depend on read order there");
    return BSC_WARN_INEXACT;
  }
  return BSC_OK;
}

int bsc_accumulate(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                   uint32_t x, uint32_t y, bsc_pileup *out) {
  if (!ctx || !out) return bsc_fail(BSC_ERR_ARG, "bsc_accumulate: NULL argument");
  int rc = bsc_accumulate_queue(ctx, tpl, nr, seq, seq_bytes, x, y), inexact = 0;
  if (rc) return rc;
  if ((rc = bsc_block_check(ctx, &inexact))) return rc;
  const uint64_t sz = (uint64_t)y - x + 1;
  HIP_TRY(hipMemcpyAsync(out, ctx->d_cts, (size_t)sz * 104u, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return bsc_inexact_status(inexact);
}

/* HOT LOOP A on device-resident reads: asynchronous on `stream`; bsc_block_status() collects the verdict. */
int bsc_accumulate_device(bsc_context *ctx, const void *d_tpl, uint32_t nr, const void *d_seq, uint64_t seq_bytes, uint32_t x,
                          uint32_t y, void *d_cts, void *stream) {
  if (!ctx || !d_cts) return bsc_fail(BSC_ERR_ARG, "bsc_accumulate_device: NULL argument");
  if (y < x) return bsc_fail(BSC_ERR_ARG, "accumulate: y (%u) < x (%u) (reference asserts y >= x)", y, x);
  if (nr && (!d_tpl || !d_seq)) return bsc_fail(BSC_ERR_ARG, "accumulate: NULL template or read buffer");
  if (((uintptr_t)d_cts & 15u) || ((uintptr_t)d_tpl & 7u))
    return bsc_fail(BSC_ERR_ARG, "bsc_accumulate_device: d_cts must be 16-byte and d_tpl 8-byte aligned");
  BSC_ENTER(ctx);
  size_t scan_bytes = 0;
  int rc = bsc_accumulate_reserve(ctx, nr, x, y, &scan_bytes);
  if (rc) return rc;
  ctx->blk_tpl = NULL;
  ctx->blk_d_tpl = d_tpl;
  ctx->blk_x = x;
  ctx->mb_n = 0;
  return bsc_accumulate_launch(ctx, d_tpl, nr, d_seq, seq_bytes, x, y, d_cts, scan_bytes, (hipStream_t)stream);
}

int bsc_block_status(bsc_context *ctx, void *stream) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_block_status: ctx is NULL");
  BSC_ENTER(ctx);
  int inexact = 0;
  int rc = bsc_block_check_on(ctx, (hipStream_t)stream, &inexact);
  if (rc) return rc;
  return bsc_inexact_status(inexact);
}

int bsc_last_accumulate_ms(bsc_context *ctx, float *ms) {
  if (!ctx || !ms) return bsc_fail(BSC_ERR_ARG, "bsc_last_accumulate_ms: NULL argument");
  if (!ctx->profiling || !ctx->ev_acc_valid) return bsc_fail(BSC_ERR_ARG, "bsc_last_accumulate_ms: no profiled launch yet");
  BSC_ENTER(ctx);
  HIP_TRY(hipEventSynchronize(ctx->ev_acc[1]));
  HIP_TRY(hipEventElapsedTime(ms, ctx->ev_acc[0], ctx->ev_acc[1]));
  return BSC_OK;
}

/* Calls n device-resident pile-ups (queued behind whatever is on the context's stream) in chunks and copies each
 * chunk's records to the host while the next chunk is being called (two output buffer sets, copy stream s_out). */
static int bsc_call_resident_to_host(bsc_context *ctx, const void *d_cts, const void *d_ref, uint64_t n, void *out,
                                     uint32_t out_stride, uint8_t *skip) {
  int rc = bsc_pipe_init(ctx);
  if (rc) return rc;
  const uint64_t chunk = n < BSC_PIPE_CHUNK ? n : BSC_PIPE_CHUNK;
  const int nbuf = n > chunk ? 2 : 1;
  for (int b = 0; b < nbuf; b++) {
    if ((rc = bsc_reserve(&ctx->p_out[b], &ctx->p_cap_out[b], (size_t)chunk * out_stride))) return rc;
    if ((rc = bsc_reserve(&ctx->p_skip[b], &ctx->p_cap_skip[b], (size_t)chunk))) return rc;
  }
  uint64_t k = 0;
  for (uint64_t done = 0; done < n; done += chunk, k++) {
    const int b = (int)(k & 1);
    const uint64_t m = (n - done) < chunk ? (n - done) : chunk;
    if (k >= 2) HIP_TRY(hipStreamWaitEvent(ctx->stream, ctx->ev_out[b], 0)); /* buffer set b has left */
    rc = bsc_call_sites_device(ctx, (const char *)d_cts + done * 104u, (const char *)d_ref + done, m, ctx->p_out[b],
                               out_stride, ctx->p_skip[b], ctx->stream);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(ctx->ev_k[b], ctx->stream));
    HIP_TRY(hipStreamWaitEvent(ctx->s_out, ctx->ev_k[b], 0));
    HIP_TRY(hipMemcpyAsync((char *)out + done * out_stride, ctx->p_out[b], (size_t)m * out_stride, hipMemcpyDeviceToHost,
                           ctx->s_out));
    HIP_TRY(hipMemcpyAsync(skip + done, ctx->p_skip[b], (size_t)m, hipMemcpyDeviceToHost, ctx->s_out));
    HIP_TRY(hipEventRecord(ctx->ev_out[b], ctx->s_out));
  }
  HIP_TRY(hipStreamSynchronize(ctx->s_out));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return BSC_OK;
}

int bsc_call_block(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                   uint32_t x, uint32_t y, const uint8_t *ref, void *out, uint32_t out_stride, uint8_t *skip) {
  if (!ctx || !ref || !out || !skip) return bsc_fail(BSC_ERR_ARG, "bsc_call_block: NULL argument");
  int rc = bsc_check_stride(out_stride);
  if (rc) return rc;
  if ((rc = bsc_accumulate_queue(ctx, tpl, nr, seq, seq_bytes, x, y))) return rc;
  const uint64_t sz = (uint64_t)y - x + 1;
  if ((rc = bsc_reserve(&ctx->d_ref, &ctx->cap_ref, (size_t)sz))) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->d_ref, ref, (size_t)sz, hipMemcpyHostToDevice, ctx->stream));
  int inexact = 0;
  if ((rc = bsc_block_check(ctx, &inexact))) return rc; /* nothing is written to out / skip for a bad block */
  /* the pile-up of the whole block is in ctx->d_cts; call it chunk by chunk and stream the records out while the next
   * chunk is computed */
  if ((rc = bsc_call_resident_to_host(ctx, ctx->d_cts, ctx->d_ref, sz, out, out_stride, skip))) return rc;
  return bsc_inexact_status(inexact);
}

int bsc_vcf_records_device(bsc_context *ctx, const void *d_gtm, uint32_t gtm_stride, const void *d_skip,
                           const void *d_ref, const void *d_dbsnp, uint32_t n, uint32_t x,
                           const bsc_vcf_params *params, void *d_out, void *stream) {
  if (!ctx || !params) return bsc_fail(BSC_ERR_ARG, "bsc_vcf_records_device: NULL argument");
  int rc = bsc_check_stride(gtm_stride);
  if (rc) return rc;
  if (n == 0) return BSC_OK;
  if (!d_gtm || !d_skip || !d_ref || !d_out) return bsc_fail(BSC_ERR_ARG, "bsc_vcf_records_device: NULL buffer");
  if (((uintptr_t)d_gtm & 7u) || ((uintptr_t)d_out & 15u))
    return bsc_fail(BSC_ERR_ARG, "bsc_vcf_records_device: d_gtm must be 8-byte and d_out 16-byte aligned");
  BSC_ENTER(ctx);
  if ((rc = bsc_reserve(&ctx->d_vg, &ctx->cap_vg, (size_t)n))) return rc;
  int e = bsc_dev_launch_vcf(d_gtm, gtm_stride, d_skip, d_ref, d_dbsnp, n, x, params->all_positions != 0,
                             params->reg_start, params->reg_stop, ctx->d_tables, ctx->d_vg, d_out, ctx->num_cus, stream);
  if (e) return bsc_fail(BSC_ERR_HIP, "vcf launch failed: %s", hipGetErrorString((hipError_t)e));
  return BSC_OK;
}

int bsc_vcf_records(bsc_context *ctx, const void *gtm, uint32_t gtm_stride, const uint8_t *skip, const uint8_t *ref,
                    const uint8_t *dbsnp, uint32_t n, uint32_t x, const bsc_vcf_params *params, bsc_vcf_core *out) {
  if (!ctx || !params) return bsc_fail(BSC_ERR_ARG, "bsc_vcf_records: NULL argument");
  int rc = bsc_check_stride(gtm_stride);
  if (rc) return rc;
  if (n == 0) return BSC_OK;
  if (!gtm || !skip || !ref || !out) return bsc_fail(BSC_ERR_ARG, "bsc_vcf_records: NULL buffer");
  BSC_ENTER(ctx);
  ctx->again.valid = 0; /* (d_out is about to be rewritten) */
  if ((rc = bsc_reserve(&ctx->d_out, &ctx->cap_out, (size_t)n * gtm_stride))) return rc;
  if ((rc = bsc_reserve(&ctx->d_skip, &ctx->cap_skip, (size_t)n))) return rc;
  if ((rc = bsc_reserve(&ctx->d_ref, &ctx->cap_ref, (size_t)n + 2))) return rc;
  if ((rc = bsc_reserve(&ctx->d_vout, &ctx->cap_vout, (size_t)n * sizeof(bsc_vcf_core)))) return rc;
  if (dbsnp && (rc = bsc_reserve(&ctx->d_vdb, &ctx->cap_vdb, (size_t)n))) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->d_out, gtm, (size_t)n * gtm_stride, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->d_skip, skip, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->d_ref, ref, (size_t)n + 2, hipMemcpyHostToDevice, ctx->stream));
  if (dbsnp) HIP_TRY(hipMemcpyAsync(ctx->d_vdb, dbsnp, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  rc = bsc_vcf_records_device(ctx, ctx->d_out, gtm_stride, ctx->d_skip, ctx->d_ref, dbsnp ? ctx->d_vdb : NULL, n, x, params,
                              ctx->d_vout, ctx->stream);
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(out, ctx->d_vout, (size_t)n * sizeof(bsc_vcf_core), hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return BSC_OK;
}

/* Asynchronous form of bsc_call_block: queue the block and return; the results stay in HBM until bsc_block_fetch. */
int bsc_block_submit(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                     uint32_t x, uint32_t y, const uint8_t *ref, uint32_t out_stride) {
  if (!ctx || !ref) return bsc_fail(BSC_ERR_ARG, "bsc_block_submit: NULL argument");
  if (ctx->pending_sz) return bsc_fail(BSC_ERR_ARG, "bsc_block_submit: the previous block has not been fetched");
  int rc = bsc_check_stride(out_stride);
  if (rc) return rc;
  if (y < x) return bsc_fail(BSC_ERR_ARG, "bsc_block_submit: y (%u) < x (%u)", y, x);
  if ((rc = bsc_accumulate_queue2(ctx, tpl, nr, seq, seq_bytes, x, y, ref, 1))) return rc;
  const uint64_t sz = (uint64_t)y - x + 1;
  ctx->again.valid = 0; /* (d_out is about to be rewritten) */
  if ((rc = bsc_reserve(&ctx->d_out, &ctx->cap_out, (size_t)sz * out_stride))) return rc;
  if ((rc = bsc_reserve(&ctx->d_skip, &ctx->cap_skip, (size_t)sz))) return rc;
  if ((rc = bsc_call_sites_device(ctx, ctx->d_cts, ctx->d_ref, sz, ctx->d_out, out_stride, ctx->d_skip, ctx->stream)))
    return rc;
  ctx->pending_sz = sz;
  ctx->pending_stride = out_stride;
  return BSC_OK;
}

/* bsc_block_submit + the copy-out queued right behind the kernels, into the caller's (pinned) destination: by the time
 * the producer thread has prepared the next block the records are usually already in host memory. */
int bsc_block_submit_to(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                        uint32_t x, uint32_t y, const uint8_t *ref, void *out, uint32_t out_stride, uint8_t *skip) {
  if (!out || !skip) return bsc_fail(BSC_ERR_ARG, "bsc_block_submit_to: NULL destination");
  int rc = bsc_block_submit(ctx, tpl, nr, seq, seq_bytes, x, y, ref, out_stride);
  if (rc) return rc;
  const uint64_t sz = ctx->pending_sz;
  hipError_t e = hipMemcpyAsync(out, ctx->d_out, (size_t)sz * out_stride, hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipMemcpyAsync(skip, ctx->d_skip, (size_t)sz, hipMemcpyDeviceToHost, ctx->stream);
  if (e != hipSuccess) {
    /* work is still in flight on the staging area: the next submit must not overwrite it under the H2D copies */
    (void)hipStreamSynchronize(ctx->stream);
    ctx->pending_sz = 0;
    return bsc_fail(BSC_ERR_HIP, "bsc_block_submit_to: copy-out failed: %s", hipGetErrorString(e));
  }
  ctx->pending_copied = 1;
  return BSC_OK;
}

/*
 * bsc_block_submit_to for several blocks at once (the drop-in glue holds small blocks back, integration/amd_overlap_protocol.h):
 * one upload, one grouping pass over all the blocks' reads, ONE accumulate launch and ONE launch of the calling kernel over the
 * positions of all blocks, one copy-out.  In the pile-up / result arrays every block starts on a multiple of 64 positions
 * (block_off[b], returned): out holds (block_off[n_blocks - 1] + that block's positions rounded up to 64) images of out_stride
 * bytes, block b's from image block_off[b] on — the positions between a block's end and the next multiple of 64 are images of
 * nothing (skip = 1).  bsc_block_fetch(ctx, NULL, NULL) completes it.
 */
static int bsc_blocks_submit_to_(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                                 uint64_t seq_bytes, const uint8_t *ref, void *out, uint32_t out_stride, uint8_t *skip, uint64_t *block_off,
                                 int inplace) {
  if (!ctx || !blocks || !ref || !out || !skip || !block_off) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_submit_to: NULL argument");
  if (ctx->pending_sz || ctx->rec_pending)
    return bsc_fail(BSC_ERR_ARG, "a submitted block has not been fetched (bsc_block_fetch / bsc_block_records_fetch first)");
  int rc = bsc_check_stride(out_stride);
  if (rc) return rc;
  if (n_blocks == 0 || n_blocks > 65536u) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_submit_to: n_blocks must be 1 .. 65536, got %u", n_blocks);
  uint64_t nr64 = 0, pos64 = 0, ref64 = 0;
  for (uint32_t b = 0; b < n_blocks; b++) {
    if (blocks[b].y < blocks[b].x)
      return bsc_fail(BSC_ERR_ARG, "bsc_blocks_submit_to: block %u has y (%u) < x (%u) (reference asserts y >= x)", b, blocks[b].y, blocks[b].x);
    if (blocks[b].y == 0xffffffffu) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_submit_to: positions exceed 32 bits");
    const uint64_t sz = (uint64_t)blocks[b].y - blocks[b].x + 1;
    block_off[b] = pos64;
    nr64 += blocks[b].nr;
    pos64 += (sz + 63u) & ~(uint64_t)63u;
    ref64 += sz + 2;
  }
  if (pos64 > 0x0fffffffull) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_submit_to: more than 2^28 - 1 positions in one call");
  if (nr64 > 0x7fffffffull) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_submit_to: more than 2^31 - 1 templates in one call");
  const uint32_t nr = (uint32_t)nr64, P = (uint32_t)pos64;
  if (nr && (!tpl || !seq)) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_submit_to: NULL template or read buffer");
  BSC_ENTER(ctx);
  size_t scan_bytes = 0;
  if ((rc = bsc_accumulate_reserve(ctx, nr, 1u, P, &scan_bytes))) return rc;
  if ((rc = bsc_reserve(&ctx->d_tpl, &ctx->cap_tpl, (size_t)(nr ? nr : 1) * sizeof(bsc_template)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_seq, &ctx->cap_seq, (size_t)(seq_bytes ? seq_bytes : 1)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_cts, &ctx->cap_cts, (size_t)P * 104u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_ref, &ctx->cap_ref, (size_t)P))) return rc;
  ctx->again.valid = 0; /* (d_out is about to be rewritten) */
  if ((rc = bsc_reserve(&ctx->d_out, &ctx->cap_out, (size_t)P * out_stride))) return rc;
  if ((rc = bsc_reserve(&ctx->d_skip, &ctx->cap_skip, (size_t)P))) return rc;
  if ((rc = bsc_reserve(&ctx->d_mblk, &ctx->cap_mblk, (size_t)n_blocks * sizeof(bsc_chain_mblock)))) return rc;
  /* staging: templates, reads, the reference codes in the device's layout (a block's y - x + 1 codes from its multiple of 64
   * on; its two look-ahead codes are the printer's, not the caller's), the block table.  inplace: the block table alone — templates,
   * reads and the packed reference codes are uploaded from where they lie, and the device lays the codes out (bsc_ref_pad_kernel) */
  const size_t b_tpl = (size_t)nr * sizeof(bsc_template), b_seq = (size_t)seq_bytes, b_blk = (size_t)n_blocks * sizeof(bsc_chain_mblock);
#define AL64(v) (((v) + 63u) & ~(size_t)63u)
  const size_t o_seq = inplace ? 0 : AL64(b_tpl), o_ref = inplace ? 0 : AL64(o_seq + b_seq), o_blk = inplace ? 0 : AL64(o_ref + (size_t)P);
#undef AL64
  if ((rc = bsc_stage_reserve(ctx, o_blk + b_blk + 64u))) return rc;
  if (inplace && (rc = bsc_reserve(&ctx->d_refp, &ctx->cap_refp, (size_t)ref64))) return rc;
  char *st = ctx->h_stage;
  if (nr && !inplace) {
    memcpy(st, tpl, b_tpl);
    memcpy(st + o_seq, seq, b_seq);
  }
  bsc_chain_mblock *mb = (bsc_chain_mblock *)(st + o_blk);
  {
    uint32_t t_end = 0, p_off = 0;
    uint64_t r_in = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
      const uint32_t sz = blocks[b].y - blocks[b].x + 1u;
      t_end += blocks[b].nr;
      mb[b].x = blocks[b].x;
      mb[b].n = sz;
      mb[b].tpl_end = t_end;
      mb[b].ref_off = p_off;
      mb[b].pos_off = p_off;
      mb[b].bin0 = p_off >> 6;
      mb[b].bin_end = (p_off >> 6) + bsc_dev_n_bins(sz);
      mb[b].ref_in = (uint32_t)r_in;
      if (!inplace) {
        memcpy(st + o_ref + p_off, ref + r_in, sz);
        memset(st + o_ref + p_off + sz, 0, ((sz + 63u) & ~63u) - sz);
      }
      r_in += (uint64_t)sz + 2u;
      p_off += (sz + 63u) & ~63u;
    }
  }
  hipStream_t s = ctx->stream;
  ctx->blk_tpl = inplace ? tpl : (const bsc_template *)st;
  ctx->blk_d_tpl = ctx->d_tpl;
  ctx->blk_x = blocks[0].x;
  ctx->mb_tab = mb;
  ctx->mb_n = n_blocks;
  if (nr) {
    HIP_TRY(hipMemcpyAsync(ctx->d_tpl, inplace ? (const void *)tpl : (const void *)st, b_tpl, hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->d_seq, inplace ? (const void *)seq : (const void *)(st + o_seq), b_seq, hipMemcpyHostToDevice, s));
  }
  HIP_TRY(hipMemcpyAsync(ctx->d_mblk, mb, b_blk, hipMemcpyHostToDevice, s));
  if (inplace) {
    HIP_TRY(hipMemcpyAsync(ctx->d_refp, ref, (size_t)ref64, hipMemcpyHostToDevice, s));
    int pe = bsc_dev_launch_ref_pad(ctx->d_refp, ctx->d_mblk, n_blocks, ctx->d_ref, P, ctx->num_cus, s);
    if (pe) {
      (void)hipStreamSynchronize(s);
      return bsc_fail(BSC_ERR_HIP, "reference layout launch failed: %s", hipGetErrorString((hipError_t)pe));
    }
  } else HIP_TRY(hipMemcpyAsync(ctx->d_ref, st + o_ref, (size_t)P, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemsetAsync(ctx->d_counters + BSC_CNT_SPAN, 0, 2 * sizeof(unsigned long long), s));
  HIP_TRY(hipMemsetAsync(ctx->d_counters + BSC_CNT_ERR, 0xff, sizeof(unsigned long long), s));
  int e = bsc_dev_launch_bin_reads_multi(ctx->d_tpl, nr, ctx->d_seq, seq_bytes, ctx->d_mblk, n_blocks, P >> 6, ctx->d_tflag, ctx->d_bcnt,
                                         ctx->d_boff, ctx->d_bcur, ctx->d_bscan, scan_bytes, ctx->d_rd, ctx->d_counters, s);
  if (!e)
    e = bsc_dev_launch_accumulate_multi(ctx->d_rd, ctx->d_boff, ctx->d_seq, ctx->d_mblk, n_blocks, P >> 6, (uint32_t)ctx->params.min_qual,
                                        ctx->d_cts, ctx->d_counters, ctx->num_cus, s);
  if (e) {
    (void)hipStreamSynchronize(s);
    return bsc_fail(BSC_ERR_HIP, "accumulate launch failed: %s", hipGetErrorString((hipError_t)e));
  }
  if ((rc = bsc_call_sites_device(ctx, ctx->d_cts, ctx->d_ref, P, ctx->d_out, out_stride, ctx->d_skip, s))) {
    (void)hipStreamSynchronize(s);
    return rc;
  }
  hipError_t he = hipMemcpyAsync(out, ctx->d_out, (size_t)P * out_stride, hipMemcpyDeviceToHost, s);
  if (he == hipSuccess) he = hipMemcpyAsync(skip, ctx->d_skip, (size_t)P, hipMemcpyDeviceToHost, s);
  if (he != hipSuccess) {
    (void)hipStreamSynchronize(s);
    return bsc_fail(BSC_ERR_HIP, "bsc_blocks_submit_to: copy-out failed: %s", hipGetErrorString(he));
  }
  ctx->sites -= P - (ref64 - 2ull * n_blocks); /* the positions between the blocks were launched, not processed */
  ctx->pending_sz = P;
  ctx->pending_stride = out_stride;
  ctx->pending_copied = 1;
  return BSC_OK;
}

int bsc_blocks_submit_to(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                         uint64_t seq_bytes, const uint8_t *ref, void *out, uint32_t out_stride, uint8_t *skip, uint64_t *block_off) {
  return bsc_blocks_submit_to_(ctx, blocks, n_blocks, tpl, seq, seq_bytes, ref, out, out_stride, skip, block_off, 0);
}

/* no staging copy: templates, reads and reference codes are uploaded from where they lie (page-locked buffers from bsc_alloc_host
 * make that a true DMA behind which the call returns) and must stay unchanged until bsc_block_fetch */
int bsc_blocks_submit_to_inplace(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl,
                                 const uint8_t *seq, uint64_t seq_bytes, const uint8_t *ref, void *out, uint32_t out_stride, uint8_t *skip,
                                 uint64_t *block_off) {
  return bsc_blocks_submit_to_(ctx, blocks, n_blocks, tpl, seq, seq_bytes, ref, out, out_stride, skip, block_off, 1);
}

int bsc_block_fetch(bsc_context *ctx, void *out, uint8_t *skip) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_block_fetch: ctx is NULL");
  if (!ctx->pending_sz) return bsc_fail(BSC_ERR_ARG, "bsc_block_fetch: no block was submitted");
  const int copied = ctx->pending_copied;
  if (!copied && (!out || !skip)) return bsc_fail(BSC_ERR_ARG, "bsc_block_fetch: NULL argument");
  BSC_ENTER(ctx);
  const uint64_t sz = ctx->pending_sz;
  ctx->pending_sz = 0;
  ctx->pending_copied = 0;
  int inexact = 0;
  int rc = bsc_block_check(ctx, &inexact); /* the block's templates were checked on the device; waits for the stream */
  if (rc) return rc;
  if (copied) return bsc_inexact_status(inexact); /* the records are where bsc_block_submit_to was told to put them */
  HIP_TRY(hipMemcpyAsync(out, ctx->d_out, (size_t)sz * ctx->pending_stride, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipMemcpyAsync(skip, ctx->d_skip, (size_t)sz, hipMemcpyDeviceToHost, ctx->stream));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return bsc_inexact_status(inexact);
}

int bsc_set_reads_fused(bsc_context *ctx, int fused) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_set_reads_fused: ctx is NULL");
  ctx->reads_fused = fused != 0;
  return BSC_OK;
}

int bsc_debug_fail_summary_alloc(bsc_context *ctx, int on) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_debug_fail_summary_alloc: ctx is NULL");
  ctx->dbg_fail_summary = on != 0;
  return BSC_OK;
}

int bsc_set_profiling(bsc_context *ctx, int enable) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_set_profiling: ctx is NULL");
  BSC_ENTER(ctx);
  if (enable && !ctx->ev[0][0])
    for (int r = 0; r < BSC_EV_RING; r++)
      for (int i = 0; i < 3; i++) HIP_TRY(hipEventCreate(&ctx->ev[r][i]));
  ctx->profiling = enable != 0;
  ctx->ev_count = 0;
  return BSC_OK;
}

int bsc_kernel_ms_history(bsc_context *ctx, uint32_t age, float *call_ms, float *fisher_ms) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_kernel_ms_history: ctx is NULL");
  if (!ctx->profiling || age >= ctx->ev_count || age >= BSC_EV_RING)
    return bsc_fail(BSC_ERR_ARG, "bsc_kernel_ms_history: no profiled launch %u launches back (the last %d are kept)", age, BSC_EV_RING);
  BSC_ENTER(ctx);
  hipEvent_t *ev = ctx->ev[(ctx->ev_count - 1u - age) % BSC_EV_RING];
  HIP_TRY(hipEventSynchronize(ev[2]));
  float a = 0.f, b = 0.f;
  HIP_TRY(hipEventElapsedTime(&a, ev[0], ev[1]));
  HIP_TRY(hipEventElapsedTime(&b, ev[1], ev[2]));
  if (call_ms) *call_ms = a;
  if (fisher_ms) *fisher_ms = b;
  return BSC_OK;
}

int bsc_last_kernel_ms(bsc_context *ctx, float *call_ms, float *fisher_ms) { return bsc_kernel_ms_history(ctx, 0, call_ms, fisher_ms); }

int bsc_stream_probe_ms(bsc_context *ctx, const void *d_cts, const void *d_ref, uint64_t n, void *d_out, void *d_skip,
                        int reps, void *stream, float *ms) {
  if (!ctx || !d_cts || !d_ref || !d_out || !d_skip || !ms) return bsc_fail(BSC_ERR_ARG, "bsc_stream_probe_ms: NULL argument");
  if (((uintptr_t)d_cts & 15u) || ((uintptr_t)d_out & 15u))
    return bsc_fail(BSC_ERR_ARG, "bsc_stream_probe_ms: d_cts and d_out must be 16-byte aligned");
  BSC_ENTER(ctx);
  hipEvent_t a, b;
  HIP_TRY(hipEventCreate(&a));
  HIP_TRY(hipEventCreate(&b));
  float best = 0.f;
  int rc = BSC_OK;
  const int nrep = reps > 0 ? reps : 1;
  /* measured the way the calling kernel is: launches queued back to back behind a few untimed ones (a single launch from an
   * idle device runs below the steady clock and carries the launch latency), one event pair around the timed ones */
  for (int r = 0; r < 3 + nrep && rc == BSC_OK; r++) {
    if (r == 3) hipEventRecord(a, (hipStream_t)stream);
    if (bsc_dev_launch_stream_probe(d_cts, d_ref, n, d_out, d_skip, ctx->num_cus, stream))
      rc = bsc_fail(BSC_ERR_HIP, "stream probe failed: %s", hipGetErrorString(hipGetLastError()));
  }
  hipEventRecord(b, (hipStream_t)stream);
  if (rc == BSC_OK && hipEventSynchronize(b) != hipSuccess) rc = bsc_fail(BSC_ERR_HIP, "stream probe failed: %s", hipGetErrorString(hipGetLastError()));
  if (rc == BSC_OK) {
    hipEventElapsedTime(&best, a, b);
    best /= (float)nrep;
  } else {
    (void)hipStreamSynchronize((hipStream_t)stream);
  }
  hipEventDestroy(a);
  hipEventDestroy(b);
  *ms = best;
  return rc;
}

int bsc_synchronize(bsc_context *ctx) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_synchronize: ctx is NULL");
  BSC_ENTER(ctx);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return BSC_OK;
}

static int bsc_sstats_init(bsc_context *ctx);

/*
 * The fused chain: one window of a block, pile-ups in, bsc_vcf_core records (+ statistics) out; gt_meth never exists in
 * HBM.  Same records and statistics as bsc_call_sites_device -> bsc_vcf_records_device -> bsc_vcf_stats_device over the
 * whole block.
 */
/* the fields of a chain launch that do not depend on where the pile-ups come from */
static int bsc_chain_fill(bsc_context *ctx, bsc_chain_launch *L, const bsc_window *w, const bsc_vcf_params *params, int with_stats,
                          int reads, const void *d_ref, const void *d_dbsnp, void *d_core, void *d_aux, void *stream) {
  const int gc = with_stats && ctx->d_gc_bins != NULL;
  int rc = bsc_reserve(&ctx->d_het, &ctx->cap_het, bsc_dev_chain_het_bytes(w->n, ctx->num_cus, gc, reads));
  if (rc) return rc;
  memset(L, 0, sizeof *L);
  if (with_stats) {
    if ((rc = bsc_sstats_init(ctx))) return rc;
    L->carry_in = ctx->d_carry + 2 * ctx->carry_slot;
    L->carry_out = ctx->d_carry + 2 * (ctx->carry_slot ^ 1u);
    L->stats = ctx->d_sstats;
    L->pairs = ctx->d_pairs;
    L->ovf_list = ctx->d_ovf;
    L->ovf_cap = BSC_OVF_CAP;
    if (gc) {
      L->gc_bins = ctx->d_gc_bins;
      L->gc_n_bins = ctx->gc_n_bins;
      L->gc_start_pos = ctx->gc_start_pos;
      L->gc_table = ctx->d_gc_table;
    }
    L->logp = ctx->d_logp;
  }
  const uint32_t after = w->n_block - w->first - w->n;
  L->ref = d_ref;
  L->dbsnp = d_dbsnp;
  L->x = w->x;
  L->n_block = w->n_block;
  L->first = w->first;
  L->n = w->n;
  L->lc = w->first < 2u ? w->first : 2u;
  L->rc = after < 2u ? after : 2u;
  L->lr = w->first < 4u ? w->first : 4u;
  L->all_positions = params->all_positions != 0;
  L->reg_start = params->reg_start;
  L->reg_stop = params->reg_stop;
  L->with_stats = with_stats != 0;
  L->tb = ctx->d_tables;
  L->par_l = 1.0 - ctx->host_tables.under_conv;
  L->par_t = ctx->host_tables.over_conv;
  L->par_lrb = ctx->host_tables.lrb;
  L->par_lrb1 = ctx->host_tables.lrb1;
  L->core_out = d_core;
  L->aux_out = d_aux;
  L->het_list = ctx->d_het;
  L->counters = ctx->d_counters;
  L->num_cus = ctx->num_cus;
  L->stream = stream;
  return BSC_OK;
}

static int bsc_window_check(const char *who, const bsc_window *w) {
  if ((uint64_t)w->first + w->n > w->n_block)
    return bsc_fail(BSC_ERR_ARG, "%s: window %u + %u exceeds the block (%u positions)", who, w->first, w->n, w->n_block);
  if (w->n > 0x0fffffffu) return bsc_fail(BSC_ERR_ARG, "%s: window longer than 2^28 - 1 positions", who);
  if ((uint64_t)w->x + w->n_block > 0xffffffffull) return bsc_fail(BSC_ERR_ARG, "%s: positions exceed 32 bits", who);
  return BSC_OK;
}

int bsc_chain_device(bsc_context *ctx, const void *d_cts, const void *d_ref, const void *d_dbsnp, const bsc_window *w,
                     const bsc_vcf_params *params, int with_stats, void *d_core, void *stream) {
  if (!ctx || !w || !params) return bsc_fail(BSC_ERR_ARG, "bsc_chain_device: NULL argument");
  if (w->n == 0) return BSC_OK;
  if (!d_cts || !d_ref || !d_core) return bsc_fail(BSC_ERR_ARG, "bsc_chain_device: NULL buffer");
  int rc = bsc_window_check("bsc_chain_device", w);
  if (rc) return rc;
  if (((uintptr_t)d_core & 15u) || ((uintptr_t)d_cts & 7u))
    return bsc_fail(BSC_ERR_ARG, "bsc_chain_device: d_core must be 16-byte and d_cts 8-byte aligned");
  BSC_ENTER(ctx);
  bsc_chain_launch L;
  if ((rc = bsc_chain_fill(ctx, &L, w, params, with_stats, 0, d_ref, d_dbsnp, d_core, NULL, stream))) return rc;
  L.cts = d_cts;
  if (ctx->profiling) {
    if (!ctx->ev_chain[0])
      for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreate(&ctx->ev_chain[i]));
    L.ev_start = ctx->ev_chain[0];
    L.ev_stop = ctx->ev_chain[1];
    ctx->ev_chain_valid = 1;
  }
  int e = bsc_dev_launch_chain(&L);
  if (e) return bsc_fail(BSC_ERR_HIP, "chain launch failed: %s", hipGetErrorString((hipError_t)e));
  if (with_stats) ctx->carry_slot ^= 1u;
  ctx->sites += w->n;
  return BSC_OK;
}

/*
 * Reads in, records out: the block's reads are checked, described and ordered (accumulate.hip), then ONE kernel piles up,
 * calls, forms the records and adds the statistics of every 60-position tile (fused.hip, READS = true) — neither the
 * pile-up nor gt_meth exists in HBM.  d_aux (optional): the second half of a bsc_vcf_rec per position.
 */
static int bsc_reads_chain_queue(bsc_context *ctx, const void *d_tpl, uint32_t nr, const void *d_seq, uint64_t seq_bytes, uint32_t x,
                                 uint32_t y, const void *d_ref, const void *d_dbsnp, const bsc_vcf_params *params, int with_stats,
                                 void *d_core, void *d_aux, void *d_emit, hipStream_t s) {
  const uint64_t sz64 = (uint64_t)y - x + 1;
  bsc_window w = {x, (uint32_t)sz64, 0u, (uint32_t)sz64};
  int rc = bsc_window_check("reads chain", &w);
  if (rc) return rc;
  size_t scan_bytes = 0;
  if ((rc = bsc_accumulate_reserve(ctx, nr, x, y, &scan_bytes))) return rc;
  if ((rc = bsc_reserve(&ctx->d_fscr, &ctx->cap_fscr, bsc_dev_chain_scratch_bytes(ctx->num_cus)))) return rc;
  bsc_chain_launch L;
  if ((rc = bsc_chain_fill(ctx, &L, &w, params, with_stats, 1, d_ref, d_dbsnp, d_core, d_aux, s))) return rc;
  L.emit_out = d_emit; /* (zeroed by the caller) */
  if (ctx->profiling) {
    if (!ctx->ev_rchain[0])
      for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreate(&ctx->ev_rchain[i]));
    HIP_TRY(hipEventRecord(ctx->ev_rchain[0], s));
  }
  if ((rc = bsc_reads_prepare(ctx, d_tpl, nr, d_seq, seq_bytes, x, y, scan_bytes, s))) return rc;
  int e;
  /* Two forms, same records (tests/test_gpu_reads_chain.py runs both).  TWO KERNELS: the accumulate kernel's summary form
   * leaves 48 bytes per position in HBM — the counts and the per-site summary of src/call_genotypes.c:44-59 — and the chain
   * kernel's summary-in form starts from them: the faster form (round 4, per 50 M positions at 30x: 6.14 ms in one kernel, 5.77
   * through a 104-byte pile-up, less through the summaries): alone, the walk runs 24 waves to a CU and hides its byte loads —
   * inside the chain kernel (128 registers, 16 waves to a CU) it waits for them — and the summary's arithmetic runs where there
   * are issue slots to spare.  ONE KERNEL (READS = true): nothing per position in HBM but the records — taken when the context
   * is told to (bsc_set_reads_fused) or the summaries cannot be allocated. */
  int two_kernels = !ctx->reads_fused;
  if (two_kernels) {
    /* 48 bytes per position of HBM the one-kernel form does not need (a maximal block of 2^28 positions: 13 GB); kept, grow-only,
     * in the context.  No room: the lean form — silently, and with the runtime's error state cleared (ROCm keeps a failed
     * hipMalloc as its last error, which the launch check below would otherwise report as its own).
     * bsc_debug_fail_summary_alloc makes this allocation fail for real (tests/test_gpu_reads_chain.py). */
    const size_t n_pad = ((size_t)w.n + 63u) / 64u * 64u;
    size_t need = n_pad * bsc_dev_summary_bytes();
    if (ctx->dbg_fail_summary) {
      void *never = NULL;
      size_t cap0 = 0;
      if (!bsc_try_reserve(&never, &cap0, (size_t)1 << 60)) need = 0;
      else hipFree(never);
    }
    if (!need || !bsc_try_reserve(&ctx->d_cts, &ctx->cap_cts, need)) two_kernels = 0;
  }
  if (two_kernels) {
    e = bsc_dev_launch_accumulate_summary(ctx->d_rd, ctx->d_boff, d_seq, x, y, (uint32_t)ctx->params.min_qual, ctx->d_cts,
                                          ctx->d_counters, ctx->num_cus, s);
    if (e) return bsc_fail(BSC_ERR_HIP, "accumulate launch failed: %s", hipGetErrorString((hipError_t)e));
    /* The summaries carry 16-bit counts.  A block with a position deeper than 65 535 reads of one class (the accumulate kernel
     * says so in counters[BSC_CNT_DEEP]) is the reads-in twin's: both launches are queued, each looks at the flag where it starts
     * and exactly one does the work — no host round trip, and the statistics are added once. */
    bsc_chain_launch L2 = L;
    L.cts = ctx->d_cts;
    L.cts_summary = 1;
    L.run_if = 2;
    e = bsc_dev_launch_chain(&L);
    if (e) return bsc_fail(BSC_ERR_HIP, "reads chain launch failed: %s", hipGetErrorString((hipError_t)e));
    L = L2;
    L.run_if = 1;
    L.ev_start = NULL; /* the timed pair of events brackets both launches */
  }
  if (!two_kernels || L.run_if) {
    L.rd = ctx->d_rd;
    L.bin_off = ctx->d_boff;
    L.seq = d_seq;
    L.f_scratch = ctx->d_fscr;
    L.n_bins = bsc_dev_n_bins(w.n);
    L.min_qual = (uint32_t)ctx->params.min_qual;
  }
  e = bsc_dev_launch_chain(&L);
  if (e) return bsc_fail(BSC_ERR_HIP, "reads chain launch failed: %s", hipGetErrorString((hipError_t)e));
  if (ctx->profiling) {
    HIP_TRY(hipEventRecord(ctx->ev_rchain[1], s));
    ctx->ev_rchain_valid = 1;
  }
  if (with_stats) ctx->carry_slot ^= 1u;
  ctx->sites += w.n;
  return BSC_OK;
}

int bsc_reads_chain_device(bsc_context *ctx, const void *d_tpl, uint32_t nr, const void *d_seq, uint64_t seq_bytes, uint32_t x,
                           uint32_t y, const void *d_ref, const void *d_dbsnp, const bsc_vcf_params *params, int with_stats,
                           void *d_core, void *d_aux, void *stream) {
  if (!ctx || !params || !d_ref || !d_core) return bsc_fail(BSC_ERR_ARG, "bsc_reads_chain_device: NULL argument");
  if (y < x) return bsc_fail(BSC_ERR_ARG, "bsc_reads_chain_device: y (%u) < x (%u) (reference asserts y >= x)", y, x);
  if (nr && (!d_tpl || !d_seq)) return bsc_fail(BSC_ERR_ARG, "bsc_reads_chain_device: NULL template or read buffer");
  if (((uintptr_t)d_core & 15u) || ((uintptr_t)d_aux & 15u) || ((uintptr_t)d_tpl & 7u))
    return bsc_fail(BSC_ERR_ARG, "bsc_reads_chain_device: d_core / d_aux must be 16-byte and d_tpl 8-byte aligned");
  BSC_ENTER(ctx);
  ctx->blk_tpl = NULL;
  ctx->blk_d_tpl = d_tpl;
  ctx->blk_x = x;
  ctx->mb_n = 0;
  return bsc_reads_chain_queue(ctx, d_tpl, nr, d_seq, seq_bytes, x, y, d_ref, d_dbsnp, params, with_stats, d_core, d_aux, NULL,
                               (hipStream_t)stream);
}

/* bsc_reads_chain_device with the byte per position the block entries keep to themselves: d_len[y - x + 1] (zeroed by the call) receives 0
 * for a position without a written record, else the record's BCF2 length (255: longer, heterozygous or dbSNP-flagged — the record itself
 * says) — what bsc_bcf_sites_len_device sizes the stream from */
int bsc_reads_chain_len_device(bsc_context *ctx, const void *d_tpl, uint32_t nr, const void *d_seq, uint64_t seq_bytes, uint32_t x, uint32_t y,
                               const void *d_ref, const void *d_dbsnp, const bsc_vcf_params *params, int with_stats, void *d_core, void *d_aux,
                               void *d_len, void *stream) {
  if (!ctx || !params || !d_ref || !d_core || !d_len) return bsc_fail(BSC_ERR_ARG, "bsc_reads_chain_len_device: NULL argument");
  if (y < x) return bsc_fail(BSC_ERR_ARG, "bsc_reads_chain_len_device: y (%u) < x (%u)", y, x);
  if (nr && (!d_tpl || !d_seq)) return bsc_fail(BSC_ERR_ARG, "bsc_reads_chain_len_device: NULL template or read buffer");
  if (((uintptr_t)d_core & 15u) || ((uintptr_t)d_aux & 15u) || ((uintptr_t)d_tpl & 7u))
    return bsc_fail(BSC_ERR_ARG, "bsc_reads_chain_len_device: d_core / d_aux must be 16-byte and d_tpl 8-byte aligned");
  BSC_ENTER(ctx);
  ctx->blk_tpl = NULL;
  ctx->blk_d_tpl = d_tpl;
  ctx->blk_x = x;
  ctx->mb_n = 0;
  HIP_TRY(hipMemsetAsync(d_len, 0, (size_t)y - x + 1u, (hipStream_t)stream));
  return bsc_reads_chain_queue(ctx, d_tpl, nr, d_seq, seq_bytes, x, y, d_ref, d_dbsnp, params, with_stats, d_core, d_aux, d_len, (hipStream_t)stream);
}

int bsc_last_reads_chain_ms(bsc_context *ctx, float *ms) {
  if (!ctx || !ms) return bsc_fail(BSC_ERR_ARG, "bsc_last_reads_chain_ms: NULL argument");
  if (!ctx->profiling || !ctx->ev_rchain_valid) return bsc_fail(BSC_ERR_ARG, "bsc_last_reads_chain_ms: no profiled launch yet");
  BSC_ENTER(ctx);
  HIP_TRY(hipEventSynchronize(ctx->ev_rchain[1]));
  HIP_TRY(hipEventElapsedTime(ms, ctx->ev_rchain[0], ctx->ev_rchain[1]));
  return BSC_OK;
}

/* with bsc_set_profiling: device time of the most recent raw block (bsc_block_records_raw[dev], bsc_block_bcf_raw[dev][_keep]) from the first
 * pre-processing launch to the last launch the call queued — pre-processing, grouping, walk, chain, packing or encoder — the host's wait for
 * the prepared size in between included */
int bsc_last_raw_block_ms(bsc_context *ctx, float *ms) {
  if (!ctx || !ms) return bsc_fail(BSC_ERR_ARG, "bsc_last_raw_block_ms: NULL argument");
  if (!ctx->profiling || !ctx->ev_raw_valid) return bsc_fail(BSC_ERR_ARG, "bsc_last_raw_block_ms: no profiled raw block yet");
  BSC_ENTER(ctx);
  HIP_TRY(hipEventSynchronize(ctx->ev_raw[1]));
  HIP_TRY(hipEventElapsedTime(ms, ctx->ev_raw[0], ctx->ev_raw[1]));
  return BSC_OK;
}

uint32_t bsc_chain_window_quantum(const bsc_context *ctx) { return ctx ? bsc_dev_chain_quantum(ctx->num_cus) : 0u; }
uint32_t bsc_chain_window_size(const bsc_context *ctx, uint32_t limit) { return ctx ? bsc_dev_chain_window(ctx->num_cus, limit) : 0u; }

int bsc_last_chain_ms(bsc_context *ctx, float *ms) {
  if (!ctx || !ms) return bsc_fail(BSC_ERR_ARG, "bsc_last_chain_ms: NULL argument");
  if (!ctx->profiling || !ctx->ev_chain_valid) return bsc_fail(BSC_ERR_ARG, "bsc_last_chain_ms: no profiled launch yet");
  BSC_ENTER(ctx);
  HIP_TRY(hipEventSynchronize(ctx->ev_chain[1]));
  HIP_TRY(hipEventElapsedTime(ms, ctx->ev_chain[0], ctx->ev_chain[1]));
  return BSC_OK;
}

/* ---- read pre-processing on the device (prepdev.hip) ------------------------------------------------------------- */
/* The device names a template and the check it fails; the words are csrc/prep.c's: that one template is fetched and run through
 * the host form, whose message (and code) stand. */
static int bsc_prep_device_error(bsc_context *ctx, unsigned long long word, const void *d_raw, const void *d_seq, const void *d_misms,
                                 const bsc_prep_params *par) {
  (void)ctx;
  const uint32_t ti = (uint32_t)(word >> 8), code = (uint32_t)(word & 0xffu);
  if (code == 9u) return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates: seq_out too small (template %u)", ti);
  if (code == 10u) return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates: template %u: read position beyond the profile", ti);
  if (code == 11u) return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates: template %u lies outside the profile's reference", ti);
  bsc_raw_template t;
  HIP_TRY(hipMemcpy(&t, (const char *)d_raw + (size_t)ti * sizeof t, sizeof t, hipMemcpyDeviceToHost));
  if (code == 1u) return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates: template %u has orientation %u", ti, t.orientation);
  if (code == 2u || code == 3u) return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates: read %d of template %u lies outside the read buffer", (int)code - 2, ti);
  if (code == 4u || code == 5u)
    return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates: mismatch list %d of template %u lies outside the list buffer", (int)code - 4, ti);
  /* a soft clip or an indel the reference would abort on: the template's reads and lists (inside the buffers: the device checked)
   * go through csrc/prep.c, numbered 0 there */
  const size_t nb = (size_t)t.len[0] + t.len[1], nmm = (size_t)t.n_misms[0] + t.n_misms[1];
  uint8_t *rb = malloc(nb + 1);
  bsc_misms *mb = malloc((nmm + 1) * sizeof *mb);
  uint8_t *so = NULL;
  size_t so_cap = nb + 16;
  if (!rb || !mb) {
    free(rb);
    free(mb);
    return bsc_fail(BSC_ERR_NOMEM, "bsc_prepare_templates_device: out of memory");
  }
  bsc_raw_template l = t;
  size_t ob = 0, om = 0;
  for (int k = 0; k < 2; k++) {
    if (t.len[k]) (void)hipMemcpy(rb + ob, (const char *)d_seq + t.off[k], t.len[k], hipMemcpyDeviceToHost);
    if (t.n_misms[k])
      (void)hipMemcpy(mb + om, (const char *)d_misms + (size_t)t.misms_off[k] * sizeof *mb, (size_t)t.n_misms[k] * sizeof *mb, hipMemcpyDeviceToHost);
    l.off[k] = ob;
    l.misms_off[k] = om;
    ob += t.len[k];
    om += t.n_misms[k];
  }
  for (size_t z = 0; z < nmm; z++) /* room for what the host form writes before it reaches the failing check */
    if (mb[z].type == BSC_MISMS_INS && mb[z].size <= nb) so_cap += mb[z].size;
  so = malloc(so_cap);
  if (!so) {
    free(rb);
    free(mb);
    return bsc_fail(BSC_ERR_NOMEM, "bsc_prepare_templates_device: out of memory");
  }
  bsc_template o;
  uint64_t used = 0;
  const int rc = bsc_prepare_templates(&l, 1, rb, nb, mb, nmm, par, &o, so, so_cap, &used, NULL); /* only its checks are wanted */
  free(rb);
  free(mb);
  free(so);
  if (rc >= 0 || strstr(bsc_errbuf, "seq_out too small"))
    return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates_device: template %u fails check %u on the device, none on the host", ti, code);
  char msg[sizeof bsc_errbuf];
  snprintf(msg, sizeof msg, "%s", bsc_errbuf);
  char *at = strstr(msg, "template 0 ");
  if (at) { /* "template 0 read k: ..." -> the template's index in the call */
    char tail[sizeof bsc_errbuf];
    snprintf(tail, sizeof tail, "%s", at + 11);
    *at = 0;
    return bsc_fail(rc, "%stemplate %u %s", msg, ti, tail);
  }
  return bsc_fail(rc, "%s (template %u)", msg, ti);
}

/* The pre-processing in two halves: everything queued on `s` — the kernels, and the copies of what the host wants of them (the error word and the
 * base counters, the prepared size, this call's profile counts and the vector's new length) into a page-locked area of the context's —
 * and, behind a wait that is the CALLER's, their reading.  bsc_prepare_templates_device waits in between; the block entries queue the whole
 * reads -> records chain behind the first half and wait once, at the block's end (round 6: a block paid a second wait for its prepared size). */
static int bsc_prep_queue(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes, const void *d_misms, uint64_t n_misms,
                          const bsc_prep_params *par, void *d_tpl_out, void *d_seq_out, uint64_t seq_out_cap, const bsc_read_profile *pf, hipStream_t s) {
  int rc;
  size_t scan_bytes = 0;
  const uint32_t n2 = 2u * nr + 1u;
  if (bsc_dev_scan_tmp_bytes_u64(n2, &scan_bytes)) return bsc_fail(BSC_ERR_HIP, "bsc_prepare_templates_device: scan size query failed");
  if ((rc = bsc_reserve(&ctx->d_pplan, &ctx->cap_pplan, (size_t)(nr ? nr : 1) * 2u * bsc_dev_prep_plan_bytes()))) return rc;
  if ((rc = bsc_reserve(&ctx->d_plen, &ctx->cap_plen, (size_t)n2 * 8u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_poff, &ctx->cap_poff, (size_t)n2 * 8u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_pms, &ctx->cap_pms, (size_t)(n_misms ? n_misms : 1) * sizeof(bsc_misms)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_pscan, &ctx->cap_pscan, scan_bytes ? scan_bytes : 1))) return rc;
  if ((rc = bsc_reserve(&ctx->d_pcnt, &ctx->cap_pcnt, BSC_PREP_CNT_ALL * sizeof(unsigned long long)))) return rc;
  const size_t prof_bytes = pf && nr ? (size_t)pf->cap * 4u * sizeof(unsigned long long) : 0;
  const size_t h_need = (BSC_PREP_CNT_ALL + 2u) * sizeof(unsigned long long) + prof_bytes; /* counters | prepared size | vector length | counts */
  if (h_need > ctx->cap_hprep) {
    if (ctx->h_prep) hipHostFree(ctx->h_prep);
    ctx->h_prep = NULL;
    ctx->cap_hprep = 0;
    if (hipHostMalloc((void **)&ctx->h_prep, h_need, hipHostMallocDefault) != hipSuccess)
      return bsc_fail(BSC_ERR_NOMEM, "bsc_prepare_templates_device: hipHostMalloc(%zu) failed", h_need);
    ctx->cap_hprep = h_need;
  }
  if (prof_bytes) {
    if ((rc = bsc_reserve(&ctx->d_pprof, &ctx->cap_pprof, prof_bytes))) return rc;
    if ((rc = bsc_reserve(&ctx->d_pmax, &ctx->cap_pmax, (size_t)nr * 4u))) return rc;
    if ((rc = bsc_reserve(&ctx->d_pused, &ctx->cap_pused, (size_t)nr * 4u))) return rc;
    if ((rc = bsc_reserve(&ctx->d_pmask, &ctx->cap_pmask, (size_t)pf->n_ref + 28u))) return rc; /* a byte per code of the block (csrc/prepdev.hip) */
    HIP_TRY(hipMemsetAsync(ctx->d_pprof, 0, prof_bytes, s));
  }
  HIP_TRY(hipMemsetAsync(ctx->d_pcnt, 0, BSC_PREP_CNT_ALL * sizeof(unsigned long long), s));
  HIP_TRY(hipMemsetAsync(ctx->d_pcnt, 0xff, sizeof(unsigned long long), s));
  int e = bsc_dev_launch_prep(d_raw, nr, d_seq, seq_bytes, d_misms, n_misms, par, ctx->d_pms, ctx->d_pplan, ctx->d_plen, ctx->d_poff,
                              ctx->d_pscan, scan_bytes, d_tpl_out, d_seq_out, seq_out_cap, ctx->d_pcnt, ctx->num_cus, s,
                              prof_bytes ? pf->ref : NULL, pf ? pf->x : 0u, pf ? pf->n_ref : 0u, pf ? pf->cap : 0u, pf ? pf->used : 0u,
                              ctx->d_pprof, ctx->d_pmax, ctx->d_pused, ctx->d_pmask);
  if (e) {
    (void)hipStreamSynchronize(s);
    return bsc_fail(BSC_ERR_HIP, "read pre-processing launch failed: %s", hipGetErrorString((hipError_t)e));
  }
  unsigned long long *h = ctx->h_prep;
  h[BSC_PREP_CNT_ALL + 1u] = 0; /* the vector's length after the last template (a u32 lands in it) */
  HIP_TRY(hipMemcpyAsync(h, ctx->d_pcnt, BSC_PREP_CNT_ALL * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(h + BSC_PREP_CNT_ALL, (const char *)ctx->d_poff + (size_t)(nr ? 2u * nr : 0u) * 8u, sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  if (prof_bytes) { /* this call's counts and the vector's new length */
    HIP_TRY(hipMemcpyAsync(h + BSC_PREP_CNT_ALL + 2u, ctx->d_pprof, prof_bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(h + BSC_PREP_CNT_ALL + 1u, (const char *)ctx->d_pused + (size_t)(nr - 1u) * 4u, 4, hipMemcpyDeviceToHost, s));
  }
  return BSC_OK;
}

/* ... and what came back, once the stream has passed the copies */
static int bsc_prep_finish(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, const void *d_misms, const bsc_prep_params *par,
                           uint64_t *seq_out_used, bsc_prep_stats *stats, bsc_read_profile *pf) {
  const unsigned long long *hs = ctx->h_prep;
  unsigned long long h[8];
  memcpy(h, hs, sizeof h);
  for (unsigned k = 0; k < BSC_PREP_CNT_SLOTS; k++) { /* the plan kernel's wave sums, slot by slot (csrc/prepdev.hip) */
    h[1] += hs[8u + 8u * k];
    h[2] += hs[8u + 8u * k + 1u];
  }
  if (h[0] != ~0ull) return bsc_prep_device_error(ctx, h[0], d_raw, d_seq, d_misms, par);
  if (pf && nr) {
    const uint32_t used_last = (uint32_t)hs[BSC_PREP_CNT_ALL + 1u];
    const unsigned long long *delta = hs + BSC_PREP_CNT_ALL + 2u;
    if (used_last > pf->used) { /* growing the vector clears everything behind its old end (csrc/prep.c) */
      memset(pf->counts + (size_t)pf->used * 4u, 0, (size_t)(pf->cap - pf->used) * 4u * sizeof(uint64_t));
      pf->used = used_last;
    }
    for (size_t i = 0; i < (size_t)pf->cap * 4u; i++) pf->counts[i] += delta[i];
  }
  if (seq_out_used) *seq_out_used = hs[BSC_PREP_CNT_ALL];
  if (stats) {
    stats->base_clip = h[1];
    stats->base_overlap = h[2];
    stats->base_none = h[3];
    stats->base_trim = h[4];
    stats->base_lowqual = h[5];
    stats->reads = h[6];
    stats->read_bases = h[7];
  }
  return BSC_OK;
}

static int bsc_prep_args_check(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes, const void *d_misms, uint64_t n_misms,
                               const bsc_prep_params *par, void *d_tpl_out, void *d_seq_out, uint64_t seq_out_cap, const bsc_read_profile *pf) {
  if (pf && (!pf->ref || !pf->counts || pf->used > pf->cap || !pf->cap)) return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates_device: bad read profile");
  if (!ctx || !par || (nr && (!d_raw || !d_tpl_out))) return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates_device: NULL argument");
  if ((seq_bytes && !d_seq) || (n_misms && !d_misms) || (seq_out_cap && !d_seq_out))
    return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates_device: NULL buffer");
  if (nr > 0x7fffffffu) return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates_device: more than 2^31 - 1 templates in one call");
  if (((uintptr_t)d_raw & 7u) || ((uintptr_t)d_tpl_out & 7u) || ((uintptr_t)d_misms & 3u))
    return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates_device: d_raw / d_tpl_out must be 8-byte, d_misms 4-byte aligned");
  return BSC_OK;
}

int bsc_prepare_templates_device(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes, const void *d_misms,
                                 uint64_t n_misms, const bsc_prep_params *par, void *d_tpl_out, void *d_seq_out, uint64_t seq_out_cap,
                                 uint64_t *seq_out_used, bsc_prep_stats *stats, bsc_read_profile *pf, void *stream) {
  if (!seq_out_used) return bsc_fail(BSC_ERR_ARG, "bsc_prepare_templates_device: NULL argument");
  int rc = bsc_prep_args_check(ctx, d_raw, nr, d_seq, seq_bytes, d_misms, n_misms, par, d_tpl_out, d_seq_out, seq_out_cap, pf);
  if (rc) return rc;
  *seq_out_used = 0;
  if (stats) memset(stats, 0, sizeof *stats);
  BSC_ENTER(ctx);
  hipStream_t s = (hipStream_t)stream;
  if ((rc = bsc_prep_queue(ctx, d_raw, nr, d_seq, seq_bytes, d_misms, n_misms, par, d_tpl_out, d_seq_out, seq_out_cap, pf, s))) return rc;
  const hipError_t se = hipStreamSynchronize(s);
  if (se != hipSuccess) return bsc_fail(BSC_ERR_HIP, "bsc_prepare_templates_device: %s", hipGetErrorString(se));
  return bsc_prep_finish(ctx, d_raw, nr, d_seq, d_misms, par, seq_out_used, stats, pf);
}

/* ---- written records, packed ------------------------------------------------------------------------------ */
int bsc_vcf_compact_device(bsc_context *ctx, const void *d_core, const void *d_gtm, uint32_t gtm_stride,
                           const void *d_dbsnp, uint32_t n, void *d_out, uint64_t out_cap, void *d_count, void *stream) {
  if (!ctx || !d_count) return bsc_fail(BSC_ERR_ARG, "bsc_vcf_compact_device: NULL argument");
  int rc = gtm_stride ? bsc_check_stride(gtm_stride) : BSC_OK; /* 0: d_gtm is the chain's aux array (bsc_reads_chain_device) */
  if (rc) return rc;
  BSC_ENTER(ctx);
  if (n == 0) {
    HIP_TRY(hipMemsetAsync(d_count, 0, sizeof(unsigned long long), (hipStream_t)stream));
    return BSC_OK;
  }
  if (!d_core || !d_gtm || (!d_out && out_cap)) return bsc_fail(BSC_ERR_ARG, "bsc_vcf_compact_device: NULL buffer");
  if (((uintptr_t)d_core & 15u) || ((uintptr_t)d_gtm & 7u) || ((uintptr_t)d_out & 15u))
    return bsc_fail(BSC_ERR_ARG, "bsc_vcf_compact_device: d_core / d_out must be 16-byte and d_gtm 8-byte aligned");
  const uint32_t n_tiles = (n + 63u) / 64u;
  size_t scan_bytes = 0;
  if (bsc_dev_scan_tmp_bytes(n_tiles, &scan_bytes)) return bsc_fail(BSC_ERR_HIP, "compact: scan size query failed");
  if ((rc = bsc_reserve(&ctx->d_tcnt, &ctx->cap_tcnt, (size_t)n_tiles * 4u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_toff, &ctx->cap_toff, (size_t)n_tiles * 4u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_scantmp, &ctx->cap_scantmp, scan_bytes ? scan_bytes : 1))) return rc;
  const void *emit = gtm_stride == 0 ? ctx->emit_hint : NULL;
  void *emit_ws = NULL;
  if (gtm_stride == 0 && !emit && !ctx->no_emit_bytes) { /* no flags from the chain kernel: the counting pass leaves them */
    if ((rc = bsc_reserve(&ctx->d_emit, &ctx->cap_emit, (size_t)n + 64u))) return rc;
    emit_ws = ctx->d_emit;
  }
  int e = bsc_dev_launch_compact(d_core, d_gtm, gtm_stride, d_dbsnp, n, ctx->d_tcnt, ctx->d_toff, ctx->d_scantmp, scan_bytes,
                                 d_out, out_cap, d_count, emit, emit_ws, ctx->num_cus, stream);
  if (e) return bsc_fail(BSC_ERR_HIP, "compact launch failed: %s", hipGetErrorString((hipError_t)e));
  return BSC_OK;
}

/* ---- the BCF stream of packed records, encoded on the device (bcfdev.hip) ----------------------------------- */
static int bsc_names_check(const char *who, const bsc_bcf_names *names, uint32_t *n_names, uint64_t *name_bytes) {
  *n_names = 0;
  *name_bytes = 0;
  if (!names || !names->n) return BSC_OK;
  if (!names->pos || !names->off || (!names->bytes && names->off[names->n])) return bsc_fail(BSC_ERR_ARG, "%s: a names table with NULL arrays", who);
  for (uint32_t i = 0; i < names->n; i++) {
    if (names->off[i] > names->off[i + 1] || (i && names->pos[i] <= names->pos[i - 1]))
      return bsc_fail(BSC_ERR_ARG, "%s: names table entry %u: positions must ascend, offsets must not descend", who, i);
  }
  *n_names = names->n;
  *name_bytes = names->off[names->n];
  return BSC_OK;
}

/* A block entry's names table (bsc_block_bcf*): positions | offsets | bytes copied into a page-locked area of the context's and queued
 * with the block's other uploads, AHEAD of its kernels — the caller's arrays are free when the call returns, and the submit / fetch
 * split overlaps for dbSNP runs as well (from ordinary memory behind the kernels, the runtime either staged the copy itself, holding the
 * call until the kernels were through, or read the arrays after the call had returned).  bsc_bcf_encode finds ctx->names_up == names. */
static int bsc_names_upload(bsc_context *ctx, const char *who, const bsc_bcf_names *names, hipStream_t s) {
  ctx->names_up = NULL;
  uint32_t n = 0;
  uint64_t nb = 0;
  int rc;
  if ((rc = bsc_names_check(who, names, &n, &nb))) return rc;
  if (!n) return BSC_OK;
  const size_t o_off = (size_t)n * 4u, o_by = o_off + ((size_t)n + 1u) * 4u, total = o_by + (size_t)nb;
  if (total + 1u > ctx->cap_hnm) {
    if (ctx->h_names) hipHostFree(ctx->h_names);
    ctx->h_names = NULL;
    ctx->cap_hnm = 0;
    const size_t sz = total + total / 4 + 4096u;
    if (hipHostMalloc(&ctx->h_names, sz, hipHostMallocDefault) != hipSuccess) return bsc_fail(BSC_ERR_NOMEM, "%s: hipHostMalloc(%zu) failed", who, sz);
    ctx->cap_hnm = sz;
  }
  if ((rc = bsc_reserve(&ctx->d_bnm, &ctx->cap_bnm, total + 1u))) return rc;
  char *h = (char *)ctx->h_names;
  memcpy(h, names->pos, (size_t)n * 4u);
  memcpy(h + o_off, names->off, ((size_t)n + 1u) * 4u);
  if (nb) memcpy(h + o_by, names->bytes, (size_t)nb);
  HIP_TRY(hipMemcpyAsync(ctx->d_bnm, h, total, hipMemcpyHostToDevice, s));
  ctx->names_up = names;
  ctx->names_up_n = n;
  ctx->names_up_bytes = nb;
  return BSC_OK;
}

/* d_recs != NULL: packed records, *d_n_recs of them; else d_core / d_aux: the per-position arrays of the reads-in chain, max_recs positions */
static int bsc_bcf_encode(bsc_context *ctx, const char *who, const void *d_recs, const void *d_core, const void *d_aux, const void *d_n_recs,
                          uint64_t max_recs, int32_t rid, const bsc_bcf_ids *ids, const bsc_bcf_names *names, void *d_out, uint64_t out_cap,
                          void *d_totals, void *stream) {
  if (!ctx || !ids || !d_totals || (out_cap && !d_out)) return bsc_fail(BSC_ERR_ARG, "%s: NULL argument", who);
  if (d_recs ? !d_n_recs : (max_recs && (!d_core || !d_aux))) return bsc_fail(BSC_ERR_ARG, "%s: NULL argument", who);
  if (((uintptr_t)d_recs & 15u) || ((uintptr_t)d_core & 15u) || ((uintptr_t)d_aux & 15u) || ((uintptr_t)d_n_recs & 7u) || ((uintptr_t)d_totals & 7u))
    return bsc_fail(BSC_ERR_ARG, "%s: the records must be 16-byte, the count and the totals 8-byte aligned", who);
  if ((uintptr_t)d_out & 15u) /* the write kernel owns whole 16-byte pieces of the stream, counted from its start */
    return bsc_fail(BSC_ERR_ARG, "%s: d_out must be 16-byte aligned (append blocks at multiples of 16, or encode into a buffer of its own)", who);
  if (max_recs > 0x1fffffffc0ull) return bsc_fail(BSC_ERR_ARG, "%s: more than 2^37 records", who);
  uint32_t n_names = 0;
  uint64_t name_bytes = 0;
  const int names_up = names && ctx->names_up == names; /* a block entry: checked and uploaded with the block's other inputs (bsc_names_upload) */
  int rc;
  if (names_up) {
    n_names = ctx->names_up_n;
    name_bytes = ctx->names_up_bytes;
  } else if ((rc = bsc_names_check(who, names, &n_names, &name_bytes)))
    return rc;
  BSC_ENTER(ctx);
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n_tiles = (uint32_t)((max_recs + 63u) / 64u);
  size_t scan_bytes = 0;
  if (bsc_dev_scan_tmp_bytes_u64(n_tiles + 1u, &scan_bytes)) return bsc_fail(BSC_ERR_HIP, "%s: scan size query failed", who);
  if ((rc = bsc_reserve(&ctx->d_btb, &ctx->cap_btb, ((size_t)n_tiles + 1u) * 8u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_bto, &ctx->cap_bto, ((size_t)n_tiles + 1u) * 8u))) return rc;
  if ((rc = bsc_reserve(&ctx->d_bscn, &ctx->cap_bscn, scan_bytes ? scan_bytes : 1))) return rc;
  const void *d_pos = NULL, *d_off = NULL, *d_nb = NULL;
  if (n_names) { /* positions | offsets | bytes in one workspace */
    const size_t o_off = (size_t)n_names * 4u, o_by = o_off + ((size_t)n_names + 1u) * 4u;
    if (!names_up) { /* the device-level entries: the caller's arrays, on the caller's stream (page-locked arrays make it a true DMA) */
      if ((rc = bsc_reserve(&ctx->d_bnm, &ctx->cap_bnm, o_by + (size_t)name_bytes + 1u))) return rc;
      HIP_TRY(hipMemcpyAsync(ctx->d_bnm, names->pos, (size_t)n_names * 4u, hipMemcpyHostToDevice, s));
      HIP_TRY(hipMemcpyAsync((char *)ctx->d_bnm + o_off, names->off, ((size_t)n_names + 1u) * 4u, hipMemcpyHostToDevice, s));
      if (name_bytes) HIP_TRY(hipMemcpyAsync((char *)ctx->d_bnm + o_by, names->bytes, (size_t)name_bytes, hipMemcpyHostToDevice, s));
    }
    d_pos = ctx->d_bnm;
    d_off = (char *)ctx->d_bnm + o_off;
    d_nb = (char *)ctx->d_bnm + o_by;
  }
  HIP_TRY(hipMemsetAsync(d_totals, 0, 3 * sizeof(unsigned long long), s));
  /* (the per-position form behind the chain: the chain's byte per position holds every record's length, ctx->emit_hint) */
  const int e = bsc_dev_launch_bcf(d_recs, d_core, d_aux, d_n_recs, max_recs, rid, ids, d_pos, d_off, d_nb, n_names, ctx->d_btb, ctx->d_bto, ctx->d_bscn,
                                   scan_bytes, d_out, out_cap, d_totals, ctx->num_cus, stream, d_recs ? NULL : ctx->emit_hint);
  if (e) return bsc_fail(BSC_ERR_HIP, "BCF encoder launch failed: %s", hipGetErrorString((hipError_t)e));
  return BSC_OK;
}

int bsc_bcf_block_device(bsc_context *ctx, const void *d_recs, const void *d_n_recs, uint64_t max_recs, int32_t rid, const bsc_bcf_ids *ids,
                         const bsc_bcf_names *names, void *d_out, uint64_t out_cap, void *d_totals, void *stream) {
  if (!d_n_recs || (max_recs && !d_recs)) return bsc_fail(BSC_ERR_ARG, "bsc_bcf_block_device: NULL argument");
  /* (no records at all: the per-position form over zero positions — nothing is read) */
  return bsc_bcf_encode(ctx, "bsc_bcf_block_device", d_recs, NULL, NULL, d_recs ? d_n_recs : NULL, max_recs, rid, ids, names, d_out, out_cap, d_totals,
                        stream);
}

int bsc_bcf_sites_device(bsc_context *ctx, const void *d_core, const void *d_aux, uint32_t n, int32_t rid, const bsc_bcf_ids *ids,
                         const bsc_bcf_names *names, void *d_out, uint64_t out_cap, void *d_totals, void *stream) {
  return bsc_bcf_encode(ctx, "bsc_bcf_sites_device", NULL, d_core, d_aux, NULL, n, rid, ids, names, d_out, out_cap, d_totals, stream);
}

/* the same with the chain's byte per position (bsc_reads_chain_len_device): the stream is sized from the lengths in it, the records are read
 * once, by the write kernel (a names table, or a dictionary index beyond 127, sends the size pass back to the records) */
int bsc_bcf_sites_len_device(bsc_context *ctx, const void *d_core, const void *d_aux, const void *d_len, uint32_t n, int32_t rid, const bsc_bcf_ids *ids,
                             const bsc_bcf_names *names, void *d_out, uint64_t out_cap, void *d_totals, void *stream) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_bcf_sites_len_device: NULL argument");
  ctx->emit_hint = d_len;
  const int rc = bsc_bcf_encode(ctx, "bsc_bcf_sites_len_device", NULL, d_core, d_aux, NULL, n, rid, ids, names, d_out, out_cap, d_totals, stream);
  ctx->emit_hint = NULL;
  return rc;
}

static void bsc_bcf_pool_take(bsc_context *ctx, size_t need);
/* what bsc_block_bcf asks of bsc_records_queue: the encoder behind the packing, its stream instead of the records on the way back */
typedef struct {
  int32_t rid;
  const bsc_bcf_ids *ids;
  const bsc_bcf_names *names;
  uint8_t *out;
  uint64_t out_cap;
} bsc_bcf_req;

/*
 * Uploads take turns.  Two contexts driven alternately from one host thread (block k + 1 queued while block k is in flight: the pipelined
 * use the split forms exist for) fall into step when their uploads start together: both halves of every pair of blocks then share the
 * link in each direction in turn and finish together, and nothing overlaps (tools/bench_two_contexts.py: 555-570 M positions/s for the BCF
 * entry, 780 with the turns).  So a block's uploads are queued behind the previous block's — whichever
 * context of this process and device queued it: its copy-out and kernels then overlap the next block's uploads by construction.  One event
 * per context, a process-wide pointer to the last one recorded; a context that goes away takes its event out.
 */
static pthread_mutex_t bsc_h2d_lock = PTHREAD_MUTEX_INITIALIZER;
static hipEvent_t bsc_h2d_last = NULL;
static int bsc_h2d_last_dev = -1;
static const bsc_context *bsc_h2d_last_ctx = NULL;

static void bsc_h2d_turn_begin(bsc_context *ctx, hipStream_t s) { /* before a block's first upload */
  if (ctx->no_h2d_turns) return;
  pthread_mutex_lock(&bsc_h2d_lock);
  if (bsc_h2d_last && bsc_h2d_last_ctx != ctx && bsc_h2d_last_dev == ctx->device) (void)hipStreamWaitEvent(s, bsc_h2d_last, 0);
  pthread_mutex_unlock(&bsc_h2d_lock);
}

static void bsc_h2d_turn_end(bsc_context *ctx, hipStream_t s) { /* behind its last upload */
  if (ctx->no_h2d_turns) return;
  if (!ctx->ev_h2d && hipEventCreateWithFlags(&ctx->ev_h2d, hipEventDisableTiming) != hipSuccess) {
    ctx->ev_h2d = NULL;
    (void)hipGetLastError();
    return;
  }
  if (hipEventRecord(ctx->ev_h2d, s) != hipSuccess) return;
  pthread_mutex_lock(&bsc_h2d_lock);
  bsc_h2d_last = ctx->ev_h2d;
  bsc_h2d_last_dev = ctx->device;
  bsc_h2d_last_ctx = ctx;
  pthread_mutex_unlock(&bsc_h2d_lock);
}

static void bsc_h2d_turn_forget(bsc_context *ctx) { /* bsc_destroy */
  pthread_mutex_lock(&bsc_h2d_lock);
  if (bsc_h2d_last_ctx == ctx) {
    bsc_h2d_last = NULL;
    bsc_h2d_last_ctx = NULL;
    bsc_h2d_last_dev = -1;
  }
  pthread_mutex_unlock(&bsc_h2d_lock);
  if (ctx->ev_h2d) (void)hipEventDestroy(ctx->ev_h2d);
  ctx->ev_h2d = NULL;
}

/*
 * One block, reads in -> packed written records out, on the reads-in chain (bsc_reads_chain_queue): H2D of the block,
 * template checks + ordering, ONE kernel from reads to records (+ the second half of every bsc_vcf_rec), packing, and
 * the copy-out — all queued back to back; the host waits ONCE, for the verdict on the templates, the record count and the
 * records together.  The copy-out is sized before the count is known, from the share of positions the previous blocks
 * wrote a record for (WGBS: every C and G, about half); only a block that writes more than that pays a second copy.
 */
/* stage == 2 ("resident"): the templates and their reads are already in ctx->d_tpl / ctx->d_seq (bsc_block_records_raw: the device
 * prepared them); tpl and seq are not looked at */
static int bsc_records_queue(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x,
                             uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                             bsc_vcf_rec *out, uint64_t out_cap, int stage, const bsc_bcf_req *bcf) {
  const int resident = stage == 2;
  if (resident) {
    stage = 0;
    tpl = NULL;
  }
  if (y < x) return bsc_fail(BSC_ERR_ARG, "accumulate: y (%u) < x (%u) (reference asserts y >= x)", y, x);
  if (nr && !resident && (!tpl || !seq)) return bsc_fail(BSC_ERR_ARG, "accumulate: NULL template or read buffer");
  if (ctx->pending_sz || ctx->rec_pending)
    return bsc_fail(BSC_ERR_ARG, "a submitted block has not been fetched (bsc_block_fetch / bsc_block_records_fetch first)");
  const uint64_t sz64 = (uint64_t)y - x + 1;
  if (sz64 > 0x0fffffffull) return bsc_fail(BSC_ERR_ARG, "bsc_block_records: block longer than 2^28 - 1 positions");
  const uint32_t sz = (uint32_t)sz64;
  BSC_ENTER(ctx);
  int rc;
  if (bcf) { /* no packed records at all: the encoder reads the chain's per-position arrays; the block's stream goes back */
    out = NULL;
    out_cap = 0;
    bsc_bcf_pool_take(ctx, (size_t)(bcf->out_cap ? bcf->out_cap : 1));
    if ((rc = bsc_reserve(&ctx->d_bcf, &ctx->cap_bcf, (size_t)(bcf->out_cap ? bcf->out_cap : 1)))) return rc;
    if ((rc = bsc_reserve(&ctx->d_btot, &ctx->cap_btot, 3 * sizeof(unsigned long long)))) return rc;
  }
  if (!resident) {
    if ((rc = bsc_reserve(&ctx->d_tpl, &ctx->cap_tpl, (size_t)(nr ? nr : 1) * sizeof(bsc_template)))) return rc;
    if ((rc = bsc_reserve(&ctx->d_seq, &ctx->cap_seq, (size_t)(seq_bytes ? seq_bytes : 1)))) return rc;
  }
  if ((rc = bsc_reserve(&ctx->d_ref, &ctx->cap_ref, (size_t)sz + 2))) return rc;
  ctx->again.valid = 0; /* (d_out is about to be rewritten) */
  if ((rc = bsc_reserve(&ctx->d_out, &ctx->cap_out, (size_t)sz * 64u))) return rc; /* the chain's aux array */
  if ((rc = bsc_reserve(&ctx->d_vout, &ctx->cap_vout, (size_t)sz * sizeof(bsc_vcf_core)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_recs, &ctx->cap_recs, (size_t)(out_cap ? out_cap : 1) * sizeof(bsc_vcf_rec)))) return rc;
  if (dbsnp && (rc = bsc_reserve(&ctx->d_vdb, &ctx->cap_vdb, (size_t)sz))) return rc;
  if (!ctx->h_cnt && hipHostMalloc((void **)&ctx->h_cnt, 8 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess)
    return bsc_fail(BSC_ERR_NOMEM, "bsc_block_records: pinned counter block");
  if (stage) { /* the caller may recycle its buffers as soon as the call returns: inputs go through the pinned staging area */
    const size_t b_tpl = resident ? 0 : (size_t)nr * sizeof(bsc_template), b_seq = resident ? 0 : (size_t)seq_bytes, b_ref = (size_t)sz + 2,
                 b_db = dbsnp ? (size_t)sz : 0;
    const size_t o_seq = (b_tpl + 63u) & ~(size_t)63u, o_ref = (o_seq + b_seq + 63u) & ~(size_t)63u, o_db = (o_ref + b_ref + 63u) & ~(size_t)63u;
    if ((rc = bsc_stage_reserve(ctx, o_db + b_db + 64u))) return rc;
    char *st = ctx->h_stage;
    if (nr && !resident) {
      memcpy(st, tpl, b_tpl);
      memcpy(st + o_seq, seq, b_seq);
      tpl = (const bsc_template *)st;
      seq = (const uint8_t *)(st + o_seq);
    }
    memcpy(st + o_ref, ref, b_ref);
    ref = (const uint8_t *)(st + o_ref);
    if (dbsnp) {
      memcpy(st + o_db, dbsnp, b_db);
      dbsnp = (const uint8_t *)(st + o_db);
    }
  }
  hipStream_t s = ctx->stream;
  ctx->blk_tpl = tpl; /* resident: NULL — a bad template is fetched from the device for the message */
  ctx->blk_d_tpl = ctx->d_tpl;
  ctx->blk_x = x;
  ctx->mb_n = 0;
  /* (the BCF entries: 555-570 -> 780 M positions/s with two contexts alternating; the records entries, whose longer copy-out keeps
   * two contexts apart by itself, lose 3 % behind the wait — 713-724 -> 695-698 — and do not take turns) */
  const int turns = bcf != NULL && !resident;
  if (turns) bsc_h2d_turn_begin(ctx, s);
  if (nr && !resident) {
    HIP_TRY(hipMemcpyAsync(ctx->d_tpl, tpl, (size_t)nr * sizeof(bsc_template), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->d_seq, seq, (size_t)seq_bytes, hipMemcpyHostToDevice, s));
  }
  if (!(resident && ctx->ref_resident)) HIP_TRY(hipMemcpyAsync(ctx->d_ref, ref, (size_t)sz + 2, hipMemcpyHostToDevice, s));
  if (dbsnp) HIP_TRY(hipMemcpyAsync(ctx->d_vdb, dbsnp, (size_t)sz, hipMemcpyHostToDevice, s));
  if (bcf && (rc = bsc_names_upload(ctx, "bsc_block_bcf", bcf->names, s))) return rc;
  const bsc_bcf_names *const names_ready = ctx->names_up;
  ctx->names_up = NULL;
  if (turns) bsc_h2d_turn_end(ctx, s);
  void *d_db = dbsnp ? ctx->d_vdb : NULL;
  /* the chain leaves the records' emit flags once more as a byte per position: the packing pass behind it then counts from 64 bytes a tile
   * and fetches the records that are written and nothing of the others (20 M positions: 1.08 -> 0.80 ms) */
  void *d_emit = NULL; /* the packing pass counts and gathers by them; the encoder's SIZE pass reads the record lengths they hold since
                        * round 6 (its write kernel still starts from the record's first 16 bytes: 5 % slower behind a flag byte,
                        * profiles/r05_ab_emit_bytes.txt).  BSC_NO_EMIT_BYTES in the environment: without them (the A/B of tools/bench_tail.py) */
  if (!ctx->no_emit_bytes) {
    if ((rc = bsc_reserve(&ctx->d_emit, &ctx->cap_emit, (size_t)sz + 64u))) return rc;
    HIP_TRY(hipMemsetAsync(ctx->d_emit, 0, (size_t)sz + 64u, s));
    d_emit = ctx->d_emit;
  }
  if ((rc = bsc_reads_chain_queue(ctx, ctx->d_tpl, nr, ctx->d_seq, seq_bytes, x, y, ctx->d_ref, d_db, params, with_stats, ctx->d_vout,
                                  ctx->d_out, d_emit, s)))
    return rc;
  unsigned long long *d_total = ctx->d_counters + BSC_CNT_RECORDS;
  ctx->emit_hint = d_emit;
  rc = bcf ? BSC_OK : bsc_vcf_compact_device(ctx, ctx->d_vout, ctx->d_out, 0, d_db, sz, ctx->d_recs, out_cap, d_total, s);
  ctx->emit_hint = NULL;
  if (rc) return rc;
  /* INEXACT, ERR, RECORDS are consecutive counter words: the verdict and the count in one small copy */
  HIP_TRY(hipMemcpyAsync(ctx->h_cnt, ctx->d_counters + BSC_CNT_INEXACT, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  ctx->rec_out = out;
  ctx->rec_cap = out_cap;
  ctx->rec_sz = sz;
  ctx->bcf_out = NULL;
  if (bcf) { /* the encoder takes the records where the chain left them (no packing pass); {length, refused, records} come back behind the verdict */
    ctx->emit_hint = d_emit;
    ctx->again.sz = sz;
    ctx->again.rid = bcf->rid;
    ctx->again.ids = *bcf->ids;
    ctx->again.emit = d_emit;
    ctx->again.have_names = names_ready != NULL;
    ctx->again.n_names = ctx->names_up_n;
    ctx->again.name_bytes = ctx->names_up_bytes;
    ctx->names_up = names_ready;
    rc = bsc_bcf_sites_device(ctx, ctx->d_vout, ctx->d_out, sz, bcf->rid, bcf->ids, bcf->names, ctx->d_bcf, bcf->out_cap, ctx->d_btot, s);
    ctx->emit_hint = NULL;
    ctx->names_up = NULL;
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_cnt + 4, ctx->d_btot, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    uint64_t guess = ctx->bcf_share > 0.0 ? (uint64_t)((double)sz * ctx->bcf_share) + 65536u : 0u;
    if (guess > bcf->out_cap) guess = bcf->out_cap;
    if (!bcf->out) guess = 0; /* the stream stays on the device: the caller reads it in pieces (bsc_bcf_stream_read) */
    if (guess) HIP_TRY(hipMemcpyAsync(bcf->out, ctx->d_bcf, (size_t)guess, hipMemcpyDeviceToHost, s));
    ctx->bcf_blk = 1;
    ctx->bcf_keep = bcf->out == NULL;
    ctx->bcf_out = bcf->out;
    ctx->bcf_cap = bcf->out_cap;
    ctx->bcf_copied = guess;
    ctx->rec_copied = 0;
    return BSC_OK;
  }
  uint64_t guess = (uint64_t)((double)sz * ctx->rec_share) + 4096u;
  if (guess > out_cap) guess = out_cap;
  if (guess > sz) guess = sz;
  if (guess) HIP_TRY(hipMemcpyAsync(out, ctx->d_recs, (size_t)guess * sizeof(bsc_vcf_rec), hipMemcpyDeviceToHost, s));
  ctx->rec_copied = guess;
  return BSC_OK;
}

/* bsc_records_finish for a block whose stream comes back (bsc_block_bcf): h_cnt[4] = its length, h_cnt[5] = records refused, h_cnt[6] = records */
static int bsc_bcf_finish(bsc_context *ctx, uint8_t *out, int inexact) {
  const unsigned long long bytes = ctx->h_cnt[4], bad = ctx->h_cnt[5];
  ctx->bcf_bytes = bytes;
  ctx->bcf_copied = ctx->bcf_copied < bytes ? ctx->bcf_copied : bytes;
  if (bad) return bsc_fail(BSC_ERR_ARG, "bsc_block_bcf: %llu records with a genotype beyond 9 or more than 6 likelihoods", bad);
  if (bytes > ctx->bcf_cap) {
    ctx->again.valid = 1; /* bsc_block_bcf_again: the encoder alone, into the room it asks for */
    ctx->again.inexact = inexact;
    return bsc_fail(BSC_ERR_ARG, "bsc_block_bcf: the block's stream has %llu bytes, out_cap is %llu", bytes, (unsigned long long)ctx->bcf_cap);
  }
  if (ctx->bcf_keep) return bsc_inexact_status(inexact);
  if (bytes > ctx->bcf_copied) { /* first block, or more bytes than the share so far suggested: the rest in a second copy */
    HIP_TRY(hipMemcpyAsync(out + ctx->bcf_copied, (const char *)ctx->d_bcf + ctx->bcf_copied, (size_t)(bytes - ctx->bcf_copied), hipMemcpyDeviceToHost,
                           ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  if (ctx->rec_sz >= 4096u) ctx->bcf_share = (double)bytes / (double)ctx->rec_sz * 1.02;
  return bsc_inexact_status(inexact);
}

int bsc_block_bcf_again(bsc_context *ctx, uint8_t *out, uint64_t out_cap, uint64_t *n_bytes, uint64_t *n_records) {
  if (!ctx || !n_bytes || !n_records || !out_cap) return bsc_fail(BSC_ERR_ARG, "bsc_block_bcf_again: NULL argument");
  *n_bytes = *n_records = 0;
  if (!ctx->again.valid || ctx->rec_pending || ctx->pending_sz)
    return bsc_fail(BSC_ERR_ARG, "bsc_block_bcf_again: the last call on this context was not a BCF block refused for its out_cap");
  BSC_ENTER(ctx);
  hipStream_t s = ctx->stream;
  int rc;
  if ((rc = bsc_reserve(&ctx->d_bcf, &ctx->cap_bcf, (size_t)out_cap))) return rc;
  ctx->again.names.n = ctx->again.n_names; /* (the table itself is in d_bnm since the block's uploads) */
  ctx->names_up = ctx->again.have_names ? &ctx->again.names : NULL;
  ctx->names_up_n = ctx->again.n_names;
  ctx->names_up_bytes = ctx->again.name_bytes;
  ctx->emit_hint = ctx->again.emit;
  rc = bsc_bcf_sites_device(ctx, ctx->d_vout, ctx->d_out, ctx->again.sz, ctx->again.rid, &ctx->again.ids, ctx->names_up, ctx->d_bcf, out_cap, ctx->d_btot, s);
  ctx->emit_hint = NULL;
  ctx->names_up = NULL;
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->h_cnt + 4, ctx->d_btot, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  const unsigned long long bytes = ctx->h_cnt[4];
  *n_bytes = ctx->bcf_bytes = bytes;
  *n_records = ctx->h_cnt[6];
  if (bytes > out_cap) return bsc_fail(BSC_ERR_ARG, "bsc_block_bcf: the block's stream has %llu bytes, out_cap is %llu", bytes, (unsigned long long)out_cap);
  if (out && bytes) {
    HIP_TRY(hipMemcpyAsync(out, ctx->d_bcf, (size_t)bytes, hipMemcpyDeviceToHost, s));
    HIP_TRY(hipStreamSynchronize(s));
  }
  ctx->bcf_keep = out == NULL;
  ctx->again.valid = 0;
  return bsc_inexact_status(ctx->again.inexact);
}

/* the one wait of a block queued by bsc_records_queue, and what is left to do after it */
static int bsc_records_finish(bsc_context *ctx, uint64_t *n_out) {
  *n_out = 0;
  uint8_t *const bcf_out = ctx->bcf_out;
  const int is_bcf = ctx->bcf_blk;
  ctx->bcf_out = NULL;
  ctx->bcf_blk = 0;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  int inexact = 0;
  int rc = bsc_verdict(ctx, ctx->h_cnt, &inexact); /* h_cnt = {INEXACT, ERR, RECORDS} */
  if (rc) return rc; /* an invalid template: the contents of `out` are unspecified */
  const unsigned long long total = is_bcf ? ctx->h_cnt[6] : ctx->h_cnt[2]; /* the encoder counts the records it writes */
  *n_out = total;
  if (is_bcf) return bsc_bcf_finish(ctx, bcf_out, inexact);
  if (total > ctx->rec_cap)
    return bsc_fail(BSC_ERR_ARG, "bsc_block_records: the block has %llu records, out_cap is %llu", total,
                    (unsigned long long)ctx->rec_cap);
  if (total > ctx->rec_copied) { /* more records than the share so far suggested: the rest in a second copy */
    HIP_TRY(hipMemcpyAsync((char *)ctx->rec_out + (size_t)ctx->rec_copied * sizeof(bsc_vcf_rec),
                           (const char *)ctx->d_recs + (size_t)ctx->rec_copied * sizeof(bsc_vcf_rec),
                           (size_t)(total - ctx->rec_copied) * sizeof(bsc_vcf_rec), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(hipStreamSynchronize(ctx->stream));
  }
  if (ctx->rec_sz >= 4096u) { /* the share the next block's copy-out is sized from: this block's, plus 2 % */
    const double share = (double)total / (double)ctx->rec_sz * 1.02;
    ctx->rec_share = share > 1.0 ? 1.0 : share;
  }
  return bsc_inexact_status(inexact);
}

static int bsc_block_records_(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                              uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params,
                              int with_stats, bsc_vcf_rec *out, uint64_t out_cap, uint64_t *n_out, const bsc_bcf_req *bcf) {
  if (!ctx || !ref || !params || !n_out || (!out && out_cap)) return bsc_fail(BSC_ERR_ARG, "bsc_block_records: NULL argument");
  *n_out = 0;
  if (ctx->rec_pending) return bsc_fail(BSC_ERR_ARG, "bsc_block_records: a submitted block has not been fetched");
  int rc = bsc_records_queue(ctx, tpl, nr, seq, seq_bytes, x, y, ref, dbsnp, params, with_stats, out, out_cap, 0, bcf);
  if (rc) {
    (void)hipStreamSynchronize(ctx->stream); /* copies already queued read the caller's buffers: none may outlive the call */
    return rc;
  }
  BSC_ENTER(ctx);
  return bsc_records_finish(ctx, n_out);
}

int bsc_block_records(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                      uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params,
                      int with_stats, bsc_vcf_rec *out, uint64_t out_cap, uint64_t *n_out) {
  return bsc_block_records_(ctx, tpl, nr, seq, seq_bytes, x, y, ref, dbsnp, params, with_stats, out, out_cap, n_out, NULL);
}

static int bsc_bcf_req_check(const char *who, const bsc_bcf_ids *ids, uint8_t *out, uint64_t out_cap, uint64_t *n_bytes, uint64_t *n_records) {
  if (!ids || !n_bytes || !n_records || (!out && out_cap)) return bsc_fail(BSC_ERR_ARG, "%s: NULL argument", who);
  *n_bytes = 0;
  *n_records = 0;
  return BSC_OK;
}

/* bsc_block_records with the encoder behind the packing (bcfdev.hip): the block's BCF bytes come back instead of its records */
int bsc_block_bcf(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x, uint32_t y,
                  const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids,
                  const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap, uint64_t *n_bytes, uint64_t *n_records) {
  int rc = bsc_bcf_req_check("bsc_block_bcf", ids, out, out_cap, n_bytes, n_records);
  if (rc) return rc;
  const bsc_bcf_req req = {rid, ids, names, out, out_cap};
  if (ctx) ctx->bcf_bytes = 0;
  rc = bsc_block_records_(ctx, tpl, nr, seq, seq_bytes, x, y, ref, dbsnp, params, with_stats, NULL, 0, n_records, &req);
  if (ctx) *n_bytes = ctx->bcf_bytes;
  return rc;
}

/*
 * bsc_block_records from what the READER delivers: raw templates with their mismatch lists (bsc_read_block) — uploaded as they are,
 * prepared on the device (bsc_prepare_templates_device: trims, soft clips, mate overlap, indel normalisation), and straight on to
 * the grouping, the walk and the chain; the process thread's per-template work (src/process_template.c:36-111) never touches a
 * host core.  x .. y: the block as the reader found it (src/get_template_vector.c:141-147; x = bsc_block_start).
 */
/* the part behind the uploads: raw templates, reads and lists DEVICE-resident (the host entry's own copies, or the device reader's
 * arrays); cap = room for the prepared reads */
static double bsc_now_s(void) {
  struct timespec t;
  clock_gettime(CLOCK_MONOTONIC, &t);
  return (double)t.tv_sec + 1e-9 * (double)t.tv_nsec;
}
static int bsc_block_records_rawdev_(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_rseq, uint64_t seq_bytes, const void *d_rms,
                                     uint64_t n_misms, uint64_t cap, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref,
                                     const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, bsc_vcf_rec *out, uint64_t out_cap,
                                     uint64_t *n_out, bsc_prep_stats *prep_stats, bsc_read_profile *profile, const bsc_bcf_req *bcf) {
  int rc;
  const int timing = ctx->stage_timing; /* BSC_STAGE_TIMING at bsc_create: where a block's host time goes, to stderr */
  double t0 = timing ? bsc_now_s() : 0.0, t1;
  if ((rc = bsc_reserve(&ctx->d_tpl, &ctx->cap_tpl, (size_t)nr * sizeof(bsc_template)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_seq, &ctx->cap_seq, (size_t)cap))) return rc;
  hipStream_t s = ctx->stream;
  bsc_read_profile dp;
  ctx->ref_resident = 0;
  if (profile) { /* the block's reference codes (x .. y + 2) are what the profile reads: up they go first, once */
    const uint64_t n_ref = (uint64_t)y - x + 3;
    if ((rc = bsc_reserve(&ctx->d_ref, &ctx->cap_ref, (size_t)n_ref))) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->d_ref, ref, (size_t)n_ref, hipMemcpyHostToDevice, s));
    ctx->ref_resident = 1;
    dp = *profile;
    dp.ref = ctx->d_ref;
    dp.x = x;
    dp.n_ref = (uint32_t)n_ref;
  }
  if (ctx->profiling) { /* the device time of everything from the pre-processing to the encoder (bsc_last_raw_block_ms) */
    if (!ctx->ev_raw[0])
      for (int i = 0; i < 2; i++) HIP_TRY(hipEventCreate(&ctx->ev_raw[i]));
    HIP_TRY(hipEventRecord(ctx->ev_raw[0], s));
    ctx->ev_raw_valid = 0;
  }
  if (timing) {
    t1 = bsc_now_s();
    fprintf(stderr, "bsc stage: reserve + reference upload queued %.1f us\n", (t1 - t0) * 1e6);
    t0 = t1;
  }
  /* the pre-processing and everything behind it are queued back to back: the chain takes the room for the prepared reads as the bound of
   * its read buffer (the device checks every template against it; what the pre-processing wrote lies inside), so the prepared SIZE is not
   * waited for — one wait per block, at its end */
  if ((rc = bsc_prep_args_check(ctx, d_raw, nr, d_rseq, seq_bytes, d_rms, n_misms, prep, ctx->d_tpl, ctx->d_seq, cap, profile ? &dp : NULL))) return rc;
  if (prep_stats) memset(prep_stats, 0, sizeof *prep_stats);
  if ((rc = bsc_prep_queue(ctx, d_raw, nr, d_rseq, seq_bytes, d_rms, n_misms, prep, ctx->d_tpl, ctx->d_seq, cap, profile ? &dp : NULL, s))) {
    (void)hipStreamSynchronize(s);
    return rc;
  }
  if (timing) {
    t1 = bsc_now_s();
    fprintf(stderr, "bsc stage: pre-processing queued %.1f us\n", (t1 - t0) * 1e6);
    t0 = t1;
  }
  rc = bsc_records_queue(ctx, NULL, nr, NULL, cap, x, y, ref, dbsnp, params, with_stats, out, out_cap, 2, bcf);
  ctx->ref_resident = 0;
  if (rc) {
    (void)hipStreamSynchronize(ctx->stream);
    return rc;
  }
  if (ctx->profiling && hipEventRecord(ctx->ev_raw[1], s) == hipSuccess) ctx->ev_raw_valid = 1; /* (behind the encoder and the copy-out it queued) */
  if (timing) {
    t1 = bsc_now_s();
    fprintf(stderr, "bsc stage: bsc_records_queue (reserves, uploads, launches) %.1f us\n", (t1 - t0) * 1e6);
    t0 = t1;
  }
  const hipError_t se = hipStreamSynchronize(s);
  if (se != hipSuccess) return bsc_fail(BSC_ERR_HIP, "bsc_block_records_raw: %s", hipGetErrorString(se));
  if (timing) {
    t1 = bsc_now_s();
    fprintf(stderr, "bsc stage: the wait %.1f us\n", (t1 - t0) * 1e6);
    t0 = t1;
  }
  /* the pre-processing's verdict first: where the reference aborts on a template, the chain behind it ran over what was left of the block —
   * checked template by template on the device, its output never looked at */
  rc = bsc_prep_finish(ctx, d_raw, nr, d_rseq, d_rms, prep, NULL, prep_stats, profile ? &dp : NULL);
  if (profile) profile->used = dp.used;
  if (rc) {
    ctx->bcf_out = NULL;
    ctx->bcf_blk = 0;
    return rc;
  }
  rc = bsc_records_finish(ctx, n_out);
  if (timing) {
    t1 = bsc_now_s();
    fprintf(stderr, "bsc stage: bsc_records_finish %.1f us\n", (t1 - t0) * 1e6);
  }
  return rc;
}

static int bsc_block_records_raw_(bsc_context *ctx, const bsc_raw_template *raw, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                                  const bsc_misms *misms, uint64_t n_misms, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref,
                                  const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, bsc_vcf_rec *out, uint64_t out_cap,
                                  uint64_t *n_out, bsc_prep_stats *prep_stats, bsc_read_profile *profile, const bsc_bcf_req *bcf) {
  if (!ctx || !ref || !params || !prep || !n_out || (!out && out_cap)) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_raw: NULL argument");
  if (y < x) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_raw: y (%u) < x (%u)", y, x);
  *n_out = 0;
  if (prep_stats) memset(prep_stats, 0, sizeof *prep_stats);
  if (ctx->rec_pending || ctx->pending_sz) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_raw: a submitted block has not been fetched");
  if (nr && (!raw || (seq_bytes && !seq) || (n_misms && !misms))) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_raw: NULL input buffer");
  if (!nr) return bsc_block_records_(ctx, NULL, 0, NULL, 0, x, y, ref, dbsnp, params, with_stats, out, out_cap, n_out, bcf);
  if (y - x > 0x0ffffffeu) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_raw: block longer than 2^28 - 1 positions");
  BSC_ENTER(ctx);
  /* room for the prepared reads: the bytes handed over plus every padded deletion (a size no read could hold is damage: the
   * device refuses that list) */
  uint64_t cap = seq_bytes + 16;
  for (uint64_t z = 0; z < n_misms; z++)
    if (misms[z].type == BSC_MISMS_INS && misms[z].size <= seq_bytes) cap += misms[z].size;
  int rc;
  if ((rc = bsc_reserve(&ctx->d_raw, &ctx->cap_raw, (size_t)nr * sizeof *raw))) return rc;
  if ((rc = bsc_reserve(&ctx->d_rseq, &ctx->cap_rseq, (size_t)(seq_bytes ? seq_bytes : 1)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_rms, &ctx->cap_rms, (size_t)(n_misms ? n_misms : 1) * sizeof *misms))) return rc;
  hipStream_t s = ctx->stream;
  HIP_TRY(hipMemcpyAsync(ctx->d_raw, raw, (size_t)nr * sizeof *raw, hipMemcpyHostToDevice, s));
  if (seq_bytes) HIP_TRY(hipMemcpyAsync(ctx->d_rseq, seq, (size_t)seq_bytes, hipMemcpyHostToDevice, s));
  if (n_misms) HIP_TRY(hipMemcpyAsync(ctx->d_rms, misms, (size_t)n_misms * sizeof *misms, hipMemcpyHostToDevice, s));
  return bsc_block_records_rawdev_(ctx, ctx->d_raw, nr, ctx->d_rseq, seq_bytes, ctx->d_rms, n_misms, cap, prep, x, y, ref, dbsnp, params, with_stats, out,
                                   out_cap, n_out, prep_stats, profile, bcf);
}

/* the same from device-resident raw templates (the device reader's blocks, bsc_bamdev_next_block) */
static int bsc_block_rawdev_check(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes, const void *d_misms,
                                  uint64_t n_misms, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref, const bsc_vcf_params *params,
                                  uint64_t *n_out, const void *out, uint64_t out_cap) {
  if (!ctx || !ref || !params || !prep || !n_out || (!out && out_cap)) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_rawdev: NULL argument");
  if (y < x) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_rawdev: y (%u) < x (%u)", y, x);
  if (ctx->rec_pending || ctx->pending_sz) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_rawdev: a submitted block has not been fetched");
  if (nr && (!d_raw || (seq_bytes && !d_seq) || (n_misms && !d_misms))) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_rawdev: NULL input buffer");
  if (nr && y - x > 0x0ffffffeu) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_rawdev: block longer than 2^28 - 1 positions");
  return BSC_OK;
}

int bsc_block_records_rawdev(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes, const void *d_misms, uint64_t n_misms,
                             uint64_t ins_pad, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp,
                             const bsc_vcf_params *params, int with_stats, bsc_vcf_rec *out, uint64_t out_cap, uint64_t *n_out, bsc_prep_stats *prep_stats,
                             bsc_read_profile *profile) {
  int rc = bsc_block_rawdev_check(ctx, d_raw, nr, d_seq, seq_bytes, d_misms, n_misms, prep, x, y, ref, params, n_out, out, out_cap);
  if (rc) return rc;
  *n_out = 0;
  if (prep_stats) memset(prep_stats, 0, sizeof *prep_stats);
  if (!nr) return bsc_block_records_(ctx, NULL, 0, NULL, 0, x, y, ref, dbsnp, params, with_stats, out, out_cap, n_out, NULL);
  BSC_ENTER(ctx);
  return bsc_block_records_rawdev_(ctx, d_raw, nr, d_seq, seq_bytes, d_misms, n_misms, seq_bytes + ins_pad + 16, prep, x, y, ref, dbsnp, params, with_stats,
                                   out, out_cap, n_out, prep_stats, profile, NULL);
}

int bsc_block_bcf_rawdev(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes, const void *d_misms, uint64_t n_misms,
                         uint64_t ins_pad, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp,
                         const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out,
                         uint64_t out_cap, uint64_t *n_bytes, uint64_t *n_records, bsc_prep_stats *prep_stats, bsc_read_profile *profile) {
  int rc = bsc_bcf_req_check("bsc_block_bcf_rawdev", ids, out, out_cap, n_bytes, n_records);
  if (rc) return rc;
  if ((rc = bsc_block_rawdev_check(ctx, d_raw, nr, d_seq, seq_bytes, d_misms, n_misms, prep, x, y, ref, params, n_records, NULL, 0))) return rc;
  if (prep_stats) memset(prep_stats, 0, sizeof *prep_stats);
  const bsc_bcf_req req = {rid, ids, names, out, out_cap};
  ctx->bcf_bytes = 0;
  if (!nr) rc = bsc_block_records_(ctx, NULL, 0, NULL, 0, x, y, ref, dbsnp, params, with_stats, NULL, 0, n_records, &req);
  else {
    BSC_ENTER(ctx);
    rc = bsc_block_records_rawdev_(ctx, d_raw, nr, d_seq, seq_bytes, d_misms, n_misms, seq_bytes + ins_pad + 16, prep, x, y, ref, dbsnp, params, with_stats,
                                   NULL, 0, n_records, prep_stats, profile, &req);
  }
  *n_bytes = ctx->bcf_bytes;
  return rc;
}

int bsc_block_records_raw(bsc_context *ctx, const bsc_raw_template *raw, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                          const bsc_misms *misms, uint64_t n_misms, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref,
                          const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, bsc_vcf_rec *out, uint64_t out_cap,
                          uint64_t *n_out, bsc_prep_stats *prep_stats, bsc_read_profile *profile) {
  return bsc_block_records_raw_(ctx, raw, nr, seq, seq_bytes, misms, n_misms, prep, x, y, ref, dbsnp, params, with_stats, out, out_cap, n_out, prep_stats,
                                profile, NULL);
}

int bsc_block_bcf_raw(bsc_context *ctx, const bsc_raw_template *raw, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, const bsc_misms *misms,
                      uint64_t n_misms, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp,
                      const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out,
                      uint64_t out_cap, uint64_t *n_bytes, uint64_t *n_records, bsc_prep_stats *prep_stats, bsc_read_profile *profile) {
  int rc = bsc_bcf_req_check("bsc_block_bcf_raw", ids, out, out_cap, n_bytes, n_records);
  if (rc) return rc;
  const bsc_bcf_req req = {rid, ids, names, out, out_cap};
  if (ctx) ctx->bcf_bytes = 0;
  rc = bsc_block_records_raw_(ctx, raw, nr, seq, seq_bytes, misms, n_misms, prep, x, y, ref, dbsnp, params, with_stats, NULL, 0, n_records, prep_stats,
                              profile, &req);
  if (ctx) *n_bytes = ctx->bcf_bytes;
  return rc;
}

/* bsc_block_bcf_rawdev with the stream LEFT on the device (dev_cap bytes of room): a whole contig's stream is gigabytes, and page-locking a
 * host buffer of that size costs more than everything else in the call; the caller reads it in pieces into a small page-locked buffer
 * (bsc_bcf_stream_read) while a thread of its own writes the previous piece. */
int bsc_block_bcf_rawdev_keep(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes, const void *d_misms, uint64_t n_misms,
                              uint64_t ins_pad, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp,
                              const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint64_t dev_cap,
                              uint64_t *n_bytes, uint64_t *n_records, bsc_prep_stats *prep_stats, bsc_read_profile *profile) {
  if (!ids || !n_bytes || !n_records) return bsc_fail(BSC_ERR_ARG, "bsc_block_bcf_rawdev_keep: NULL argument");
  *n_bytes = 0;
  *n_records = 0;
  int rc;
  if ((rc = bsc_block_rawdev_check(ctx, d_raw, nr, d_seq, seq_bytes, d_misms, n_misms, prep, x, y, ref, params, n_records, NULL, 0))) return rc;
  if (prep_stats) memset(prep_stats, 0, sizeof *prep_stats);
  const bsc_bcf_req req = {rid, ids, names, NULL, dev_cap};
  ctx->bcf_bytes = 0;
  if (!nr) rc = bsc_block_records_(ctx, NULL, 0, NULL, 0, x, y, ref, dbsnp, params, with_stats, NULL, 0, n_records, &req);
  else {
    BSC_ENTER(ctx);
    rc = bsc_block_records_rawdev_(ctx, d_raw, nr, d_seq, seq_bytes, d_misms, n_misms, seq_bytes + ins_pad + 16, prep, x, y, ref, dbsnp, params, with_stats,
                                   NULL, 0, n_records, prep_stats, profile, &req);
  }
  *n_bytes = ctx->bcf_bytes;
  return rc;
}

/* ---- a block's stream handed over: the caller reads it out (on another thread, another stream) while the context goes on ---------------- */
static void bsc_bcf_pool_take(bsc_context *ctx, size_t need) { /* before d_bcf is reserved: a buffer that came back, if the context has none */
  if (ctx->d_bcf || !ctx->pool_mu_made) return;
  pthread_mutex_lock(&ctx->pool_mu);
  int best = -1;
  for (int k = 0; k < 4; k++)
    if (ctx->bcf_pool.p[k] && (best < 0 || ctx->bcf_pool.cap[k] > ctx->bcf_pool.cap[best])) best = k;
  if (best >= 0 && ctx->bcf_pool.cap[best] >= need) {
    ctx->d_bcf = ctx->bcf_pool.p[best];
    ctx->cap_bcf = ctx->bcf_pool.cap[best];
    ctx->bcf_pool.p[best] = NULL;
    ctx->bcf_pool.cap[best] = 0;
  }
  pthread_mutex_unlock(&ctx->pool_mu);
}

int bsc_bcf_stream_detach(bsc_context *ctx, void **d_stream, uint64_t *n_bytes) {
  if (!ctx || !d_stream || !n_bytes) return bsc_fail(BSC_ERR_ARG, "bsc_bcf_stream_detach: NULL argument");
  *d_stream = NULL;
  *n_bytes = 0;
  if (!ctx->bcf_keep || ctx->rec_pending || !ctx->d_bcf) return bsc_fail(BSC_ERR_ARG, "bsc_bcf_stream_detach: the last call left no stream on the device");
  BSC_ENTER(ctx);
  if (!ctx->pool_mu_made) {
    pthread_mutex_init(&ctx->pool_mu, NULL);
    ctx->pool_mu_made = 1;
  }
  if (!ctx->s_det) HIP_TRY(hipStreamCreateWithFlags(&ctx->s_det, hipStreamNonBlocking));
  *d_stream = ctx->d_bcf;
  *n_bytes = ctx->bcf_bytes;
  /* its capacity travels in front of nothing: kept in the pool's bookkeeping when it comes back — remember it beside the pointer */
  pthread_mutex_lock(&ctx->pool_mu);
  for (int k = 0; k < 4; k++)
    if (!ctx->bcf_pool.p[k] && !ctx->bcf_pool.cap[k]) { /* a slot that remembers the capacity of a buffer that is out */
      ctx->bcf_pool.cap[k] = ctx->cap_bcf | ((size_t)1 << 63);
      ctx->det_ptr[k] = ctx->d_bcf;
      break;
    }
  pthread_mutex_unlock(&ctx->pool_mu);
  ctx->d_bcf = NULL;
  ctx->cap_bcf = 0;
  ctx->bcf_keep = 0;
  ctx->bcf_bytes = 0;
  return BSC_OK;
}

int bsc_detached_read(bsc_context *ctx, const void *d_stream, uint64_t off, uint64_t n, void *dst) {
  if (!ctx || !d_stream || (!dst && n) || !ctx->s_det) return bsc_fail(BSC_ERR_ARG, "bsc_detached_read: NULL argument");
  if (!n) return BSC_OK;
  BSC_ENTER(ctx);
  HIP_TRY(hipMemcpyAsync(dst, (const char *)d_stream + off, (size_t)n, hipMemcpyDeviceToHost, ctx->s_det));
  return BSC_OK;
}

int bsc_detached_wait(bsc_context *ctx) {
  if (!ctx || !ctx->s_det) return bsc_fail(BSC_ERR_ARG, "bsc_detached_wait: nothing was detached");
  BSC_ENTER(ctx);
  HIP_TRY(hipStreamSynchronize(ctx->s_det));
  return BSC_OK;
}

int bsc_detached_free(bsc_context *ctx, void *d_stream) {
  if (!ctx || !d_stream || !ctx->pool_mu_made) return bsc_fail(BSC_ERR_ARG, "bsc_detached_free: NULL argument");
  pthread_mutex_lock(&ctx->pool_mu);
  int found = 0;
  for (int k = 0; k < 4 && !found; k++)
    if (ctx->det_ptr[k] == d_stream && (ctx->bcf_pool.cap[k] >> 63)) { /* back in its slot, for the next block to take */
      ctx->bcf_pool.p[k] = d_stream;
      ctx->bcf_pool.cap[k] &= ~((size_t)1 << 63);
      ctx->det_ptr[k] = NULL;
      found = 1;
    }
  pthread_mutex_unlock(&ctx->pool_mu);
  if (!found) { /* more than four out at once: this one was not booked — given back to the runtime (a device-wide wait) */
    BSC_ENTER(ctx);
    HIP_TRY(hipFree(d_stream));
  }
  return BSC_OK;
}

/* bytes [off, off + n) of the last block's stream -> dst (page-locked for a true DMA), queued on the context's stream; the caller waits
 * with bsc_synchronize before it touches dst */
int bsc_bcf_stream_read(bsc_context *ctx, uint64_t off, uint64_t n, void *dst) {
  if (!ctx || (!dst && n)) return bsc_fail(BSC_ERR_ARG, "bsc_bcf_stream_read: NULL argument");
  if (off + n > ctx->bcf_bytes || off + n > ctx->cap_bcf) return bsc_fail(BSC_ERR_ARG, "bsc_bcf_stream_read: beyond the stream's end");
  if (!n) return BSC_OK;
  BSC_ENTER(ctx);
  HIP_TRY(hipMemcpyAsync(dst, (const char *)ctx->d_bcf + off, (size_t)n, hipMemcpyDeviceToHost, ctx->stream));
  return BSC_OK;
}

/* The split form: queue the block and return; bsc_block_records_fetch waits and completes it.  stage != 0: the inputs go
 * through the pinned staging area, so the caller's buffers are free at once; 0: they are read where they lie and must stay
 * unchanged until the fetch (`out` must stay valid until then either way). */
static int bsc_records_submit(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x,
                              uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                              bsc_vcf_rec *out, uint64_t out_cap, int stage, const bsc_bcf_req *bcf) {
  if (!ctx || !ref || !params || (!out && out_cap)) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_submit: NULL argument");
  if (ctx->rec_pending || ctx->pending_sz) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_submit: the previous block has not been fetched");
  int rc = bsc_records_queue(ctx, tpl, nr, seq, seq_bytes, x, y, ref, dbsnp, params, with_stats, out, out_cap, stage, bcf);
  if (rc) {
    (void)hipStreamSynchronize(ctx->stream); /* nothing may still read the inputs / the staging area after a failed submit */
    return rc;
  }
  ctx->rec_pending = bcf ? 3 : 1;
  return BSC_OK;
}

int bsc_block_records_submit(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x,
                             uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                             bsc_vcf_rec *out, uint64_t out_cap) {
  return bsc_records_submit(ctx, tpl, nr, seq, seq_bytes, x, y, ref, dbsnp, params, with_stats, out, out_cap, 1, NULL);
}

int bsc_block_records_submit_inplace(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                                     uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params,
                                     int with_stats, bsc_vcf_rec *out, uint64_t out_cap) {
  return bsc_records_submit(ctx, tpl, nr, seq, seq_bytes, x, y, ref, dbsnp, params, with_stats, out, out_cap, 0, NULL);
}

/* The split form of bsc_block_bcf: the block is queued — inputs through the staging area, so the caller's buffers are free at once; the
 * names' table is uploaded by the call — and the call returns; bsc_block_bcf_fetch waits for it.  `out` should come from bsc_alloc_host
 * and must stay valid until the fetch. */
static int bsc_bcf_submit(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x, uint32_t y,
                          const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids,
                          const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap, int stage) {
  if (!ids || (!out && out_cap)) return bsc_fail(BSC_ERR_ARG, "bsc_block_bcf_submit: NULL argument");
  const bsc_bcf_req req = {rid, ids, names, out, out_cap};
  if (ctx) ctx->bcf_bytes = 0;
  return bsc_records_submit(ctx, tpl, nr, seq, seq_bytes, x, y, ref, dbsnp, params, with_stats, NULL, 0, stage, &req);
}

int bsc_block_bcf_submit(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x, uint32_t y,
                         const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids,
                         const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap) {
  return bsc_bcf_submit(ctx, tpl, nr, seq, seq_bytes, x, y, ref, dbsnp, params, with_stats, rid, ids, names, out, out_cap, 1);
}

/* ... with the inputs read where they lie (bsc_alloc_host buffers make the upload a true DMA): they must stay unchanged until the fetch */
int bsc_block_bcf_submit_inplace(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x,
                                 uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid,
                                 const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap) {
  return bsc_bcf_submit(ctx, tpl, nr, seq, seq_bytes, x, y, ref, dbsnp, params, with_stats, rid, ids, names, out, out_cap, 0);
}

int bsc_block_bcf_fetch(bsc_context *ctx, uint64_t *n_bytes, uint64_t *n_records) {
  if (!ctx || !n_bytes || !n_records) return bsc_fail(BSC_ERR_ARG, "bsc_block_bcf_fetch: NULL argument");
  *n_bytes = 0;
  *n_records = 0;
  if (ctx->rec_pending != 3) return bsc_fail(BSC_ERR_ARG, "bsc_block_bcf_fetch: no block was submitted");
  ctx->rec_pending = 0;
  BSC_ENTER(ctx);
  const int rc = bsc_records_finish(ctx, n_records);
  *n_bytes = ctx->bcf_bytes;
  return rc;
}

int bsc_block_records_fetch(bsc_context *ctx, uint64_t *n_out) {
  if (!ctx || !n_out) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_fetch: NULL argument");
  *n_out = 0;
  if (ctx->rec_pending != 1) return bsc_fail(BSC_ERR_ARG, "bsc_block_records_fetch: no block was submitted");
  ctx->rec_pending = 0;
  BSC_ENTER(ctx);
  return bsc_records_finish(ctx, n_out);
}

/*
 * Several blocks, one launch sequence (bsc_blocks_records_submit): what bsc_records_queue does for one block — H2D, template
 * checks + grouping, the reads-in chain, packing, copy-out, one wait — for n_blocks of them at once.  The blocks' positions lie
 * one block after another in the per-position arrays, each block from a multiple of 64 on (so a block's first packed record is
 * where the packing pass's offset of its first 64-position tile says); their reads are grouped in one pass, bins numbered
 * through; the chain runs one pair of launches (unguarded + guarded kernel) over the segments of all blocks — two pairs where a
 * block begins right behind its predecessor's last position, because only there can the printer's pending cytosine
 * (src/print_vcf.c:447-455) pair across blocks, and the second launch reads what the first left.
 */
static int bsc_blocks_queue(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                            uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                            bsc_vcf_rec *out, uint64_t out_cap, int stage, const bsc_bcf_req *bcf) {
  if (ctx->pending_sz || ctx->rec_pending)
    return bsc_fail(BSC_ERR_ARG, "a submitted block has not been fetched (bsc_block_fetch / bsc_block_records_fetch first)");
  if (n_blocks == 0 || n_blocks > 65536u) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_records: n_blocks must be 1 .. 65536, got %u", n_blocks);
  uint64_t nr64 = 0, pos64 = 0, ref64 = 0, sites = 0;
  for (uint32_t b = 0; b < n_blocks; b++) {
    if (blocks[b].y < blocks[b].x)
      return bsc_fail(BSC_ERR_ARG, "bsc_blocks_records: block %u has y (%u) < x (%u) (reference asserts y >= x)", b, blocks[b].y, blocks[b].x);
    const uint64_t sz = (uint64_t)blocks[b].y - blocks[b].x + 1;
    if (blocks[b].y == 0xffffffffu) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_records: positions exceed 32 bits");
    nr64 += blocks[b].nr;
    pos64 += (sz + 63u) & ~(uint64_t)63u;
    ref64 += sz + 2;
    sites += sz;
  }
  if (pos64 > 0x0fffffffull) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_records: more than 2^28 - 1 positions in one call");
  if (nr64 > 0x7fffffffull) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_records: more than 2^31 - 1 templates in one call");
  const uint32_t nr = (uint32_t)nr64, P = (uint32_t)pos64;
  if (nr && (!tpl || !seq)) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_records: NULL template or read buffer");
  BSC_ENTER(ctx);
  int rc;
  size_t scan_bytes = 0;
  if (bcf) { /* no packed records: the encoder reads the chain's per-position arrays of all the blocks, ONE stream goes back (bsc_blocks_bcf) */
    out = NULL;
    out_cap = 0;
    bsc_bcf_pool_take(ctx, (size_t)(bcf->out_cap ? bcf->out_cap : 1));
    if ((rc = bsc_reserve(&ctx->d_bcf, &ctx->cap_bcf, (size_t)(bcf->out_cap ? bcf->out_cap : 1)))) return rc;
    if ((rc = bsc_reserve(&ctx->d_btot, &ctx->cap_btot, 3 * sizeof(unsigned long long)))) return rc;
  }
  if ((rc = bsc_accumulate_reserve(ctx, nr, 1u, P, &scan_bytes))) return rc; /* P positions = P / 64 bins */
  if ((rc = bsc_reserve(&ctx->d_fscr, &ctx->cap_fscr, bsc_dev_chain_scratch_bytes(ctx->num_cus)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_tpl, &ctx->cap_tpl, (size_t)(nr ? nr : 1) * sizeof(bsc_template)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_seq, &ctx->cap_seq, (size_t)(seq_bytes ? seq_bytes : 1)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_ref, &ctx->cap_ref, (size_t)ref64))) return rc;
  ctx->again.valid = 0; /* (d_out is about to be rewritten) */
  if ((rc = bsc_reserve(&ctx->d_out, &ctx->cap_out, (size_t)P * 64u))) return rc; /* the chain's aux array */
  if ((rc = bsc_reserve(&ctx->d_vout, &ctx->cap_vout, (size_t)P * sizeof(bsc_vcf_core)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_recs, &ctx->cap_recs, (size_t)(out_cap ? out_cap : 1) * sizeof(bsc_vcf_rec)))) return rc;
  if (dbsnp && (rc = bsc_reserve(&ctx->d_vdb, &ctx->cap_vdb, (size_t)P))) return rc;
  const size_t tab_bytes = bsc_dev_chain_multi_table_bytes(n_blocks);
  if ((rc = bsc_reserve(&ctx->d_mblk, &ctx->cap_mblk, (size_t)n_blocks * sizeof(bsc_chain_mblock)))) return rc;
  if ((rc = bsc_reserve(&ctx->d_mtab, &ctx->cap_mtab, tab_bytes))) return rc;
  if (!ctx->h_cnt && hipHostMalloc((void **)&ctx->h_cnt, 8 * sizeof(unsigned long long), hipHostMallocDefault) != hipSuccess)
    return bsc_fail(BSC_ERR_NOMEM, "bsc_blocks_records: pinned counter block");
  /* the pinned staging area: inputs (stage != 0: the caller's buffers are free when the call returns; 0: templates, reads and
   * reference codes are read where they lie), the dbSNP flags in the device's padded layout, the tables, and the per-tile record
   * offsets on their way back */
  const size_t b_tpl = stage ? (size_t)nr * sizeof(bsc_template) : 0, b_seq = stage ? (size_t)seq_bytes : 0, b_ref = stage ? (size_t)ref64 : 0,
               b_db = dbsnp ? (size_t)P : 0;
  const size_t b_blk = (size_t)n_blocks * sizeof(bsc_chain_mblock), b_toff = ((size_t)(P >> 6) + 1u) * 4u;
#define AL64(v) (((v) + 63u) & ~(size_t)63u)
  const size_t o_seq = AL64(b_tpl), o_ref = AL64(o_seq + b_seq), o_db = AL64(o_ref + b_ref), o_blk = AL64(o_db + b_db),
               o_tab = AL64(o_blk + b_blk), o_toff = AL64(o_tab + tab_bytes);
#undef AL64
  if ((rc = bsc_stage_reserve(ctx, o_toff + b_toff + 64u))) return rc;
  char *st = ctx->h_stage;
  if (stage) {
    if (nr) {
      memcpy(st, tpl, b_tpl);
      memcpy(st + o_seq, seq, b_seq);
      tpl = (const bsc_template *)st;
      seq = (const uint8_t *)(st + o_seq);
    }
    memcpy(st + o_ref, ref, b_ref);
    ref = (const uint8_t *)(st + o_ref);
  }
  bsc_chain_mblock *mb = (bsc_chain_mblock *)(st + o_blk);
  {
    uint32_t t_end = 0, r_off = 0, p_off = 0;
    uint64_t d_in = 0;
    for (uint32_t b = 0; b < n_blocks; b++) {
      const uint32_t sz = blocks[b].y - blocks[b].x + 1u;
      t_end += blocks[b].nr;
      mb[b].x = blocks[b].x;
      mb[b].n = sz;
      mb[b].tpl_end = t_end;
      mb[b].ref_off = r_off;
      mb[b].pos_off = p_off;
      mb[b].bin0 = p_off >> 6;
      mb[b].bin_end = (p_off >> 6) + bsc_dev_n_bins(sz);
      mb[b].ref_in = 0;
      if (dbsnp) { /* the caller's flags are packed block after block; on the device every block starts on a multiple of 64 */
        memcpy(st + o_db + p_off, dbsnp + d_in, sz);
        memset(st + o_db + p_off + sz, 0, ((sz + 63u) & ~63u) - sz);
        d_in += sz;
      }
      r_off += sz + 2u;
      p_off += (sz + 63u) & ~63u;
    }
  }
  hipStream_t s = ctx->stream;
  ctx->blk_tpl = tpl;
  ctx->blk_d_tpl = ctx->d_tpl;
  ctx->blk_x = blocks[0].x;
  ctx->mb_tab = mb;
  ctx->mb_n = n_blocks;
  if (nr) {
    HIP_TRY(hipMemcpyAsync(ctx->d_tpl, tpl, (size_t)nr * sizeof(bsc_template), hipMemcpyHostToDevice, s));
    HIP_TRY(hipMemcpyAsync(ctx->d_seq, seq, (size_t)seq_bytes, hipMemcpyHostToDevice, s));
  }
  HIP_TRY(hipMemcpyAsync(ctx->d_ref, ref, (size_t)ref64, hipMemcpyHostToDevice, s));
  if (dbsnp) HIP_TRY(hipMemcpyAsync(ctx->d_vdb, st + o_db, b_db, hipMemcpyHostToDevice, s));
  HIP_TRY(hipMemcpyAsync(ctx->d_mblk, mb, b_blk, hipMemcpyHostToDevice, s));
  if (bcf && (rc = bsc_names_upload(ctx, "bsc_blocks_bcf", bcf->names, s))) return rc;
  const bsc_bcf_names *const names_ready = ctx->names_up;
  ctx->names_up = NULL;
  void *d_db = dbsnp ? ctx->d_vdb : NULL;
  /* template checks + grouping of all blocks' reads (SPAN, INEXACT = 0 and ERR = all ones first, as bsc_reads_prepare) */
  HIP_TRY(hipMemsetAsync(ctx->d_counters + BSC_CNT_SPAN, 0, 2 * sizeof(unsigned long long), s));
  HIP_TRY(hipMemsetAsync(ctx->d_counters + BSC_CNT_ERR, 0xff, sizeof(unsigned long long), s));
  int e = bsc_dev_launch_bin_reads_multi(ctx->d_tpl, nr, ctx->d_seq, seq_bytes, ctx->d_mblk, n_blocks, P >> 6, ctx->d_tflag, ctx->d_bcnt,
                                         ctx->d_boff, ctx->d_bcur, ctx->d_bscan, scan_bytes, ctx->d_rd, ctx->d_counters, s);
  if (e) return bsc_fail(BSC_ERR_HIP, "read grouping launch failed: %s", hipGetErrorString((hipError_t)e));
  /* the records' emit flags once more as a byte per position, for the packing pass (bsc_records_queue) */
  void *d_emit = NULL;
  if (!ctx->no_emit_bytes) {
    if ((rc = bsc_reserve(&ctx->d_emit, &ctx->cap_emit, (size_t)P + 64u))) return rc;
    HIP_TRY(hipMemsetAsync(ctx->d_emit, 0, (size_t)P + 64u, s));
    d_emit = ctx->d_emit;
  }
  /* the chain: one launch group per stretch of blocks none of which begins right behind its predecessor */
  bsc_window w = {1u, P, 0u, P};
  size_t cursor = 0;
  for (uint32_t b0 = 0; b0 < n_blocks;) {
    uint32_t b1 = b0;
    while (b1 + 1u < n_blocks && blocks[b1 + 1u].x != blocks[b1].y + 1u) b1++;
    bsc_chain_launch L;
    if ((rc = bsc_chain_fill(ctx, &L, &w, params, with_stats, 1, ctx->d_ref, d_db, ctx->d_vout, ctx->d_out, s))) return rc;
    L.emit_out = d_emit;
    L.rd = ctx->d_rd;
    L.bin_off = ctx->d_boff;
    L.seq = ctx->d_seq;
    L.f_scratch = ctx->d_fscr;
    L.n_bins = P >> 6;
    L.min_qual = (uint32_t)ctx->params.min_qual;
    e = bsc_dev_launch_chain_multi(&L, mb, b0, b1, st + o_tab, ctx->d_mtab, &cursor);
    if (e) return bsc_fail(BSC_ERR_HIP, "reads chain launch failed: %s", hipGetErrorString((hipError_t)e));
    if (with_stats) ctx->carry_slot ^= 1u;
    b0 = b1 + 1u;
  }
  ctx->sites += sites;
  unsigned long long *d_total = ctx->d_counters + BSC_CNT_RECORDS;
  if (bcf) { /* as bsc_records_queue's tail, over the P positions of all the blocks (the positions between two blocks write no record) */
    HIP_TRY(hipMemcpyAsync(ctx->h_cnt, ctx->d_counters + BSC_CNT_INEXACT, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    ctx->rec_out = NULL;
    ctx->rec_cap = 0;
    ctx->rec_sz = sites > 0xffffffffull ? 0xffffffffu : (uint32_t)sites;
    ctx->emit_hint = d_emit;
    ctx->again.sz = P;
    ctx->again.rid = bcf->rid;
    ctx->again.ids = *bcf->ids;
    ctx->again.emit = d_emit;
    ctx->again.have_names = names_ready != NULL;
    ctx->again.n_names = ctx->names_up_n;
    ctx->again.name_bytes = ctx->names_up_bytes;
    ctx->names_up = names_ready;
    rc = bsc_bcf_sites_device(ctx, ctx->d_vout, ctx->d_out, P, bcf->rid, bcf->ids, bcf->names, ctx->d_bcf, bcf->out_cap, ctx->d_btot, s);
    ctx->emit_hint = NULL;
    ctx->names_up = NULL;
    if (rc) return rc;
    HIP_TRY(hipMemcpyAsync(ctx->h_cnt + 4, ctx->d_btot, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
    uint64_t guess = ctx->bcf_share > 0.0 ? (uint64_t)((double)sites * ctx->bcf_share) + 65536u : 0u;
    if (guess > bcf->out_cap) guess = bcf->out_cap;
    if (!bcf->out) guess = 0;
    if (guess) HIP_TRY(hipMemcpyAsync(bcf->out, ctx->d_bcf, (size_t)guess, hipMemcpyDeviceToHost, s));
    ctx->bcf_blk = 1;
    ctx->bcf_keep = bcf->out == NULL;
    ctx->bcf_out = bcf->out;
    ctx->bcf_cap = bcf->out_cap;
    ctx->bcf_copied = guess;
    ctx->rec_copied = 0;
    ctx->mb_toff = NULL;
    return BSC_OK;
  }
  ctx->emit_hint = d_emit;
  rc = bsc_vcf_compact_device(ctx, ctx->d_vout, ctx->d_out, 0, d_db, P, ctx->d_recs, out_cap, d_total, s);
  ctx->emit_hint = NULL;
  if (rc) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->h_cnt, ctx->d_counters + BSC_CNT_INEXACT, 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipMemcpyAsync(st + o_toff, ctx->d_toff, (size_t)(P >> 6) * 4u, hipMemcpyDeviceToHost, s));
  ctx->mb_toff = (const uint32_t *)(st + o_toff);
  uint64_t guess = (uint64_t)((double)sites * ctx->rec_share) + 4096u;
  if (guess > out_cap) guess = out_cap;
  if (guess > sites) guess = sites;
  if (guess) HIP_TRY(hipMemcpyAsync(out, ctx->d_recs, (size_t)guess * sizeof(bsc_vcf_rec), hipMemcpyDeviceToHost, s));
  ctx->rec_out = out;
  ctx->rec_cap = out_cap;
  ctx->rec_copied = guess;
  ctx->rec_sz = sites > 0xffffffffull ? 0xffffffffu : (uint32_t)sites;
  return BSC_OK;
}

static int bsc_blocks_submit(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                             uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                             bsc_vcf_rec *out, uint64_t out_cap, int stage, const bsc_bcf_req *bcf) {
  if (!ctx || !blocks || !ref || !params || (!out && out_cap)) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_records_submit: NULL argument");
  int rc = bsc_blocks_queue(ctx, blocks, n_blocks, tpl, seq, seq_bytes, ref, dbsnp, params, with_stats, out, out_cap, stage, bcf);
  if (rc) {
    (void)hipStreamSynchronize(ctx->stream); /* nothing may still read the inputs / the staging area after a failed submit */
    ctx->bcf_blk = 0;
    return rc;
  }
  ctx->rec_pending = 2;
  return BSC_OK;
}

int bsc_blocks_records_submit(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                              uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                              bsc_vcf_rec *out, uint64_t out_cap) {
  return bsc_blocks_submit(ctx, blocks, n_blocks, tpl, seq, seq_bytes, ref, dbsnp, params, with_stats, out, out_cap, 1, NULL);
}

int bsc_blocks_records_submit_inplace(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl,
                                      const uint8_t *seq, uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp,
                                      const bsc_vcf_params *params, int with_stats, bsc_vcf_rec *out, uint64_t out_cap) {
  return bsc_blocks_submit(ctx, blocks, n_blocks, tpl, seq, seq_bytes, ref, dbsnp, params, with_stats, out, out_cap, 0, NULL);
}

int bsc_blocks_records_fetch(bsc_context *ctx, uint64_t *n_out, uint64_t *block_counts) {
  if (!ctx || !n_out) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_records_fetch: NULL argument");
  *n_out = 0;
  if (ctx->rec_pending != 2 || ctx->bcf_blk) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_records_fetch: no blocks were submitted by bsc_blocks_records_submit");
  ctx->rec_pending = 0;
  BSC_ENTER(ctx);
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  int rc = bsc_records_finish(ctx, n_out);
  if (rc < 0) return rc;
  if (block_counts) { /* a block's records start where the packing pass put its first 64-position tile */
    const uint64_t total = *n_out;
    for (uint32_t b = 0; b < ctx->mb_n; b++) {
      const uint64_t lo = ctx->mb_toff[ctx->mb_tab[b].pos_off >> 6];
      const uint64_t hi = b + 1u < ctx->mb_n ? ctx->mb_toff[ctx->mb_tab[b + 1u].pos_off >> 6] : total;
      block_counts[b] = hi - lo;
    }
  }
  return rc;
}

int bsc_blocks_records(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                       uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                       bsc_vcf_rec *out, uint64_t out_cap, uint64_t *n_out, uint64_t *block_counts) {
  if (!n_out) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_records: NULL argument");
  *n_out = 0;
  int rc = bsc_blocks_records_submit(ctx, blocks, n_blocks, tpl, seq, seq_bytes, ref, dbsnp, params, with_stats, out, out_cap);
  if (rc) return rc;
  return bsc_blocks_records_fetch(ctx, n_out, block_counts);
}

/* Several blocks in one launch sequence, their BCF bytes back as ONE stream (the blocks' records in the blocks' order: what bsc_block_bcf
 * gives block after block, concatenated): bsc_blocks_records' launch sequence with the encoder over the chain's arrays instead of the
 * packing pass.  All blocks lie on one contig (rid), in genome order; a names table lists the flagged positions of all of them. */
static int bsc_blocks_bcf_submit_(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                                  uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid,
                                  const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap, int stage) {
  if (!ids || (!out && out_cap)) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_bcf_submit: NULL argument");
  const bsc_bcf_req req = {rid, ids, names, out, out_cap};
  if (ctx) ctx->bcf_bytes = 0;
  return bsc_blocks_submit(ctx, blocks, n_blocks, tpl, seq, seq_bytes, ref, dbsnp, params, with_stats, NULL, 0, stage, &req);
}
int bsc_blocks_bcf_submit(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                          uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid,
                          const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap) {
  return bsc_blocks_bcf_submit_(ctx, blocks, n_blocks, tpl, seq, seq_bytes, ref, dbsnp, params, with_stats, rid, ids, names, out, out_cap, 1);
}
int bsc_blocks_bcf_submit_inplace(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                                  uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                                  int32_t rid, const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap) {
  return bsc_blocks_bcf_submit_(ctx, blocks, n_blocks, tpl, seq, seq_bytes, ref, dbsnp, params, with_stats, rid, ids, names, out, out_cap, 0);
}
int bsc_blocks_bcf_fetch(bsc_context *ctx, uint64_t *n_bytes, uint64_t *n_records) {
  if (!ctx || !n_bytes || !n_records) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_bcf_fetch: NULL argument");
  *n_bytes = *n_records = 0;
  if (ctx->rec_pending != 2 || !ctx->bcf_blk) return bsc_fail(BSC_ERR_ARG, "bsc_blocks_bcf_fetch: no blocks were submitted by bsc_blocks_bcf_submit");
  ctx->rec_pending = 0;
  BSC_ENTER(ctx);
  const int rc = bsc_records_finish(ctx, n_records);
  *n_bytes = ctx->bcf_bytes;
  return rc;
}

/* ---- site statistics -------------------------------------------------------------------------------------- */
static int bsc_sstats_init(bsc_context *ctx) {
  if (ctx->d_sstats) return BSC_OK;
  double logp[100];
  for (int i = 0; i < 100; i++) logp[i] = log(0.01 * (double)(i + 1)); /* src/init_param.c:56, libm like the reference */
  if (hipMalloc(&ctx->d_sstats, sizeof(bsc_site_stats)) != hipSuccess ||
      hipMalloc(&ctx->d_pairs, BSC_PAIR_BYTES) != hipSuccess || hipMemset(ctx->d_pairs, 0, BSC_PAIR_BYTES) != hipSuccess ||
      hipMalloc((void **)&ctx->d_carry, 4 * sizeof(uint32_t)) != hipSuccess ||
      hipMalloc(&ctx->d_ovf, (size_t)BSC_OVF_CAP * 8u) != hipSuccess ||
      hipMalloc((void **)&ctx->d_logp, sizeof logp) != hipSuccess ||
      hipMemset(ctx->d_sstats, 0, sizeof(bsc_site_stats)) != hipSuccess ||
      hipMemset(ctx->d_carry, 0, 4 * sizeof(uint32_t)) != hipSuccess ||
      hipMemcpy(ctx->d_logp, logp, sizeof logp, hipMemcpyHostToDevice) != hipSuccess) {
    hipFree(ctx->d_sstats);
    hipFree(ctx->d_pairs);
    hipFree(ctx->d_carry);
    hipFree(ctx->d_logp);
    hipFree(ctx->d_ovf);
    ctx->d_ovf = NULL;
    ctx->d_pairs = NULL;
    ctx->d_sstats = NULL;
    ctx->d_carry = NULL;
    ctx->d_logp = NULL;
    return bsc_fail(BSC_ERR_NOMEM, "site statistics: device allocation failed");
  }
  ctx->carry_slot = 0;
  return BSC_OK;
}

int bsc_vcf_stats_device(bsc_context *ctx, const void *d_core, const void *d_gtm, uint32_t gtm_stride,
                         const void *d_dbsnp, uint32_t n, void *stream) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_vcf_stats_device: ctx is NULL");
  int rc = bsc_check_stride(gtm_stride);
  if (rc) return rc;
  if (n == 0) return BSC_OK;
  if (!d_core || !d_gtm) return bsc_fail(BSC_ERR_ARG, "bsc_vcf_stats_device: NULL buffer");
  if (((uintptr_t)d_core & 15u) || ((uintptr_t)d_gtm & 7u))
    return bsc_fail(BSC_ERR_ARG, "bsc_vcf_stats_device: d_core must be 16-byte and d_gtm 8-byte aligned");
  BSC_ENTER(ctx);
  if ((rc = bsc_sstats_init(ctx))) return rc;
  const unsigned in = ctx->carry_slot, out = in ^ 1u;
  int e = bsc_dev_launch_site_stats(d_core, d_gtm, gtm_stride, d_dbsnp, n, ctx->d_tables, ctx->d_logp,
                                    ctx->d_carry + 2 * in, ctx->d_carry + 2 * out, ctx->d_sstats, ctx->d_pairs, ctx->num_cus,
                                    stream);
  if (e) return bsc_fail(BSC_ERR_HIP, "site statistics launch failed: %s", hipGetErrorString((hipError_t)e));
  if (ctx->d_gc_bins) { /* GC content by coverage, when the contig's bins are set (bsc_set_gc_bins) */
    e = bsc_dev_launch_gc_cov_gtm(d_core, d_gtm, gtm_stride, n, ctx->d_gc_bins, ctx->gc_n_bins, ctx->gc_start_pos, ctx->d_gc_table,
                                  ctx->num_cus, stream);
    if (e) return bsc_fail(BSC_ERR_HIP, "GC statistics launch failed: %s", hipGetErrorString((hipError_t)e));
  }
  ctx->carry_slot = out;
  return BSC_OK;
}

int bsc_vcf_stats(bsc_context *ctx, const bsc_vcf_core *core, const void *gtm, uint32_t gtm_stride, const uint8_t *dbsnp,
                  uint32_t n) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_vcf_stats: ctx is NULL");
  int rc = bsc_check_stride(gtm_stride);
  if (rc) return rc;
  if (n == 0) return BSC_OK;
  if (!core || !gtm) return bsc_fail(BSC_ERR_ARG, "bsc_vcf_stats: NULL buffer");
  BSC_ENTER(ctx);
  ctx->again.valid = 0; /* (d_out is about to be rewritten) */
  if ((rc = bsc_reserve(&ctx->d_out, &ctx->cap_out, (size_t)n * gtm_stride))) return rc;
  if ((rc = bsc_reserve(&ctx->d_vout, &ctx->cap_vout, (size_t)n * sizeof(bsc_vcf_core)))) return rc;
  if (dbsnp && (rc = bsc_reserve(&ctx->d_vdb, &ctx->cap_vdb, (size_t)n))) return rc;
  HIP_TRY(hipMemcpyAsync(ctx->d_out, gtm, (size_t)n * gtm_stride, hipMemcpyHostToDevice, ctx->stream));
  HIP_TRY(hipMemcpyAsync(ctx->d_vout, core, (size_t)n * sizeof(bsc_vcf_core), hipMemcpyHostToDevice, ctx->stream));
  if (dbsnp) HIP_TRY(hipMemcpyAsync(ctx->d_vdb, dbsnp, (size_t)n, hipMemcpyHostToDevice, ctx->stream));
  if ((rc = bsc_vcf_stats_device(ctx, ctx->d_vout, ctx->d_out, gtm_stride, dbsnp ? ctx->d_vdb : NULL, n, ctx->stream)))
    return rc;
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  return BSC_OK;
}

int bsc_get_site_stats(bsc_context *ctx, bsc_site_stats *out) {
  if (!ctx || !out) return bsc_fail(BSC_ERR_ARG, "bsc_get_site_stats: NULL argument");
  BSC_ENTER(ctx);
  if (!ctx->d_sstats) {
    memset(out, 0, sizeof *out);
    return BSC_OK;
  }
  HIP_TRY(hipDeviceSynchronize()); /* the statistics kernels run on whatever stream the caller chose */
  int e = bsc_dev_launch_meth_eval(ctx->d_pairs, ctx->d_tables, ctx->d_logp, ctx->d_sstats, ctx->d_ovf, ctx->d_counters,
                                   ctx->num_cus, ctx->stream);
  if (e) return bsc_fail(BSC_ERR_HIP, "methylation profile launch failed: %s", hipGetErrorString((hipError_t)e));
  HIP_TRY(hipStreamSynchronize(ctx->stream));
  unsigned long long n_ovf = 0;
  HIP_TRY(hipMemcpy(&n_ovf, ctx->d_counters + BSC_CNT_OVF, sizeof n_ovf, hipMemcpyDeviceToHost));
  HIP_TRY(hipMemset(ctx->d_counters + BSC_CNT_OVF, 0, sizeof n_ovf));
  HIP_TRY(hipMemcpy(out, ctx->d_sstats, sizeof *out, hipMemcpyDeviceToHost)); /* (synchronous, behind the memset on the null stream: the word is zero when this returns) */
  if (n_ovf > BSC_OVF_CAP)
    return bsc_fail(BSC_ERR_RANGE, "bsc_get_site_stats: %llu CpG cytosines with 512 or more informative reads of one kind since the "
                                   "statistics were last read, %u can be listed: their share of the methylation profiles is missing",
                    n_ovf, BSC_OVF_CAP);
  return BSC_OK;
}

#define BSC_GC_BYTES ((size_t)BSC_COV_CAP * 101u * sizeof(uint64_t))

int bsc_set_gc_bins(bsc_context *ctx, const void *d_gc, uint32_t n_bins, uint32_t start_pos) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_set_gc_bins: ctx is NULL");
  BSC_ENTER(ctx);
  if (d_gc && !ctx->d_gc_table) {
    if (hipMalloc(&ctx->d_gc_table, BSC_GC_BYTES) != hipSuccess) {
      ctx->d_gc_table = NULL;
      return bsc_fail(BSC_ERR_NOMEM, "bsc_set_gc_bins: device allocation failed");
    }
    HIP_TRY(hipMemset(ctx->d_gc_table, 0, BSC_GC_BYTES));
    HIP_TRY(hipStreamSynchronize(NULL)); /* the memset is done before any stream adds to the table (see bsc_create) */
  }
  ctx->d_gc_bins = d_gc;
  ctx->gc_n_bins = d_gc ? n_bins : 0;
  ctx->gc_start_pos = start_pos;
  return BSC_OK;
}

int bsc_set_gc_bins_host(bsc_context *ctx, const uint8_t *gc, uint32_t n_bins, uint32_t start_pos) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_set_gc_bins_host: ctx is NULL");
  if (!gc || !n_bins) return bsc_set_gc_bins(ctx, NULL, 0, 0);
  BSC_ENTER(ctx);
  HIP_TRY(hipDeviceSynchronize()); /* earlier launches may still read the previous contig's bins */
  int rc = bsc_reserve(&ctx->d_gc_own, &ctx->cap_gc_own, n_bins);
  if (rc) return rc;
  HIP_TRY(hipMemcpy(ctx->d_gc_own, gc, n_bins, hipMemcpyHostToDevice));
  return bsc_set_gc_bins(ctx, ctx->d_gc_own, n_bins, start_pos);
}

int bsc_get_gc_stats(bsc_context *ctx, uint64_t *out) {
  if (!ctx || !out) return bsc_fail(BSC_ERR_ARG, "bsc_get_gc_stats: NULL argument");
  BSC_ENTER(ctx);
  if (!ctx->d_gc_table) {
    memset(out, 0, BSC_GC_BYTES);
    return BSC_OK;
  }
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, ctx->d_gc_table, BSC_GC_BYTES, hipMemcpyDeviceToHost));
  return BSC_OK;
}

/* ctg_stats->gc of load_sequence (src/read_reference.c:44-131): the contig starts at its first A/C/G/T (*start_pos, 1-based);
 * from there every 100 bases make a bin = their G+C count, or 255 if one of them is not A/C/G/T; a last, shorter bin is
 * dropped.  codes[n] = reference codes 0..4 of positions 1 .. n. */
int bsc_gc_bins(const uint8_t *codes, uint64_t n, uint32_t *start_pos, uint8_t *out, uint64_t out_cap, uint64_t *n_bins) {
  if ((n && !codes) || !start_pos || !n_bins || (out_cap && !out)) return bsc_fail(BSC_ERR_ARG, "bsc_gc_bins: NULL argument");
  uint64_t k = 0;
  while (k < n && (codes[k] < 1 || codes[k] > 4)) k++;
  *start_pos = (uint32_t)(k + 1);
  *n_bins = 0;
  uint64_t nb = 0;
  unsigned in_bin = 0, valid = 0, gc = 0;
  for (; k < n; k++) {
    const uint8_t b = codes[k];
    if (b >= 1 && b <= 4) {
      valid++;
      gc += (b == 2 || b == 3);
    }
    if (++in_bin == 100) {
      if (nb >= out_cap) return bsc_fail(BSC_ERR_ARG, "bsc_gc_bins: out holds %llu bins, more are needed", (unsigned long long)out_cap);
      out[nb++] = valid == 100 ? (uint8_t)gc : 255;
      in_bin = valid = gc = 0;
    }
  }
  *n_bins = nb;
  return BSC_OK;
}

int bsc_get_site_totals(bsc_context *ctx, uint64_t out[14]) {
  if (!ctx || !out) return bsc_fail(BSC_ERR_ARG, "bsc_get_site_totals: NULL argument");
  BSC_ENTER(ctx);
  memset(out, 0, 14 * sizeof(uint64_t));
  if (!ctx->d_sstats) return BSC_OK;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, ctx->d_sstats, 14 * sizeof(uint64_t), hipMemcpyDeviceToHost));
  return BSC_OK;
}

int bsc_reset_site_stats(bsc_context *ctx) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_reset_site_stats: ctx is NULL");
  BSC_ENTER(ctx);
  if (!ctx->d_sstats) return BSC_OK;
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(ctx->d_sstats, 0, sizeof(bsc_site_stats)));
  HIP_TRY(hipMemset(ctx->d_pairs, 0, BSC_PAIR_BYTES));
  HIP_TRY(hipMemset(ctx->d_counters + BSC_CNT_OVF, 0, sizeof(unsigned long long)));
  if (ctx->d_gc_table) HIP_TRY(hipMemset(ctx->d_gc_table, 0, BSC_GC_BYTES));
  HIP_TRY(hipMemset(ctx->d_carry, 0, 4 * sizeof(uint32_t)));
  HIP_TRY(hipStreamSynchronize(NULL)); /* the memsets are done before the next block's kernels add to what they cleared (see bsc_create) */
  ctx->carry_slot = 0;
  return BSC_OK;
}

int bsc_get_stats(bsc_context *ctx, bsc_stats *out) {
  if (!ctx || !out) return bsc_fail(BSC_ERR_ARG, "bsc_get_stats: NULL argument");
  BSC_ENTER(ctx);
  HIP_TRY(hipDeviceSynchronize());
  unsigned long long c[BSC_CNT_WORDS];
  HIP_TRY(hipMemcpy(c, ctx->d_counters, sizeof c, hipMemcpyDeviceToHost));
  memset(out, 0, sizeof *out);
  out->sites = ctx->sites;
  out->covered = c[BSC_CNT_COVERED];
  for (int g = 0; g < 10; g++) out->gt_hist[g] = c[BSC_CNT_COVERED + 1 + g];
  out->het_calls = c[BSC_CNT_COVERED + 11];
  return BSC_OK;
}

int bsc_reset_stats(bsc_context *ctx) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_reset_stats: ctx is NULL");
  BSC_ENTER(ctx);
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemset(ctx->d_counters, 0, BSC_CNT_WORDS * sizeof(unsigned long long)));
  HIP_TRY(hipStreamSynchronize(NULL)); /* (see bsc_create) */
  ctx->sites = 0;
  return BSC_OK;
}

int bsc_synth_pileup_device(bsc_context *ctx, uint64_t seed, uint64_t first_site, uint64_t n, uint32_t coverage,
                            uint32_t flags, void *d_cts, void *d_ref, void *stream) {
  if (!ctx) return bsc_fail(BSC_ERR_ARG, "bsc_synth_pileup_device: ctx is NULL");
  if (n && (!d_cts || !d_ref)) return bsc_fail(BSC_ERR_ARG, "bsc_synth_pileup_device: NULL buffer");
  if (coverage > 4000) return bsc_fail(BSC_ERR_ARG, "bsc_synth_pileup_device: coverage %u > 4000", coverage);
  BSC_ENTER(ctx);
  int e = bsc_dev_launch_synth(seed, first_site, n, coverage, flags, d_cts, d_ref, ctx->num_cus, stream);
  if (e) return bsc_fail(BSC_ERR_HIP, "synth launch failed: %s", hipGetErrorString((hipError_t)e));
  return BSC_OK;
}

int bsc_synth_pileup_host(uint64_t seed, uint64_t first_site, uint64_t n, uint32_t coverage, uint32_t flags,
                          bsc_pileup *cts, uint8_t *ref) {
  if (n && (!cts || !ref)) return bsc_fail(BSC_ERR_ARG, "bsc_synth_pileup_host: NULL buffer");
  if (coverage > 4000) return bsc_fail(BSC_ERR_ARG, "bsc_synth_pileup_host: coverage %u > 4000", coverage);
  for (uint64_t i = 0; i < n; i++) {
    uint32_t rf;
    syn_site(seed, first_site + i, coverage, flags, &cts[i].counts[0][0], &cts[i].n, cts[i].quality, &cts[i].mapq2, &rf);
    ref[i] = (uint8_t)rf;
  }
  return BSC_OK;
}
