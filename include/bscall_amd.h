/*
 * bscall_amd.h — C ABI of libbscall_amd.so: bs_call's per-site genotype + methylation calling path
 * (pile-up -> gt_meth) as hand-written gfx950 (MI355X) HIP kernels.
 *
 * Plain C, plain pointers and sizes; no C++/torch types.  Every entry point names the piece of the
 * reference (heathsc/bs_call v2.1.7) it stands in for; INTEGRATION.md shows the replacement
 * src/call_genotypes.c a maintainer would compile inside the reference tree against this header.
 *
 * The record layouts are the reference's own, byte for byte, so host arrays of the reference's
 * `pileup` / `gt_meth` can be passed without conversion.
 *
 * All functions return BSC_OK (0) or a negative BSC_ERR_* code; bsc_last_error() gives the text (per thread).
 * Threading: a context is used by one thread at a time, like the reference's calc pool is driven by its single
 * process thread; calls on one context must be ordered on one stream (the context owns a heterozygous-site list and
 * counters that successive launches reuse).  Independent work = independent contexts (e.g. one per GPU).
 * There is NO CPU fallback: without a usable gfx950 device bsc_create() fails with BSC_ERR_NO_DEVICE.
 *
 * MAP OF THE ENTRIES.  A caller picks ONE level — where a block enters the library and what comes back; the other ~120 names are the same
 * computation cut at another place, fed from another kind of memory, or the readers / writers / counters around it.
 *
 *   level                          in -> out                                            entry                                   replaces (reference)
 *   1   the calc threads           pileup[] -> gt_meth[] / gt_vcf[]                      bsc_call_sites                          call_thread's loop (src/call_genotypes.c:43-115)
 *   2a  + the accumulate loop      templates + reads -> gt_vcf[]                         bsc_call_block                          call_genotypes_ML (:155-273)
 *       ... the printer's records  templates + reads -> written records, packed          bsc_block_records                       + _print_vcf_entry up to the encoding (src/print_vcf.c:32-594)
 *   2b  + the print thread         templates + reads -> the block's BCF bytes            bsc_block_bcf                           + bcf_enc_* / bcf_write (:160-222,267-378)
 *   3   + the process thread       raw templates + mismatch lists -> BCF bytes           bsc_block_bcf_raw                       + process_template (src/process_template.c:36-111)
 *   4   + the reader thread        a BAM file -> blocks in HBM -> BCF bytes              bsc_bamdev_next_block + bsc_block_bcf_rawdev[_keep]   + read_input (src/get_template_vector.c:49-389)
 *
 *   The suffixes say WHERE THE BUFFERS LIVE and WHEN THE CALL RETURNS, nothing else: (none) the caller's ordinary memory, staged by the
 *   library, results on return; _inplace: the caller's page-locked memory (bsc_alloc_host), no staging copy; _submit / _fetch: the call in two
 *   halves, the host thread returns while the GPU works (one block in flight per context, as the reference hands a block to its calc
 *   threads); _to: results into caller-provided page-locked memory; _device: everything HBM-resident, asynchronous on the caller's stream;
 *   bsc_blocks_*: several small blocks in one launch sequence (records, gt_vcf images, or — bsc_blocks_bcf_* — one BCF stream).
 *   Below the levels: their device-resident pieces for callers that keep whole contigs in HBM (bsc_accumulate_device, bsc_chain_device,
 *   bsc_reads_chain[_len]_device, bsc_vcf_*_device, bsc_bcf_*_device: what bench.py and bs_call_amd/genome.py drive); host-side readers and
 *   writers (bsc_bam_*, bsc_bamstream_*, bsc_dbsnp_*, bsc_fasta_contig, bsc_prepare_templates, bsc_bcf_record, bsc_vcf_format,
 *   bsc_report_json); statistics and inspection (bsc_get_*, bsc_reset_*, bsc_last_*_ms).  INTEGRATION.md section 1 lists every name.
 */
#ifndef BSCALL_AMD_H
#define BSCALL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BSC_ABI_VERSION 2

#define BSC_OK 0
#define BSC_ERR_ARG (-1)       /* bad argument (the reference would assert: src/call_genotypes.c:158,186,188) */
#define BSC_ERR_HIP (-2)       /* a HIP runtime call failed */
#define BSC_ERR_NOMEM (-3)     /* host or device allocation failed */
#define BSC_ERR_NO_DEVICE (-4) /* no gfx950 device / kernels not loadable */
#define BSC_ERR_RANGE (-5)     /* an input beyond what a fixed-size device table holds: the result would be incomplete */
#define BSC_WARN_INEXACT 1     /* results written, but some pile-up sums left the range where float sums are exact */

/* `pileup`, include/bs_call.h:174-182 — 104 bytes */
typedef struct {
  uint32_t counts[2][8]; /* [orientation][class]; classes 0-3 non-informative ACGT, 4-7 informative ACGT */
  uint32_t n;            /* bases that passed the quality filter */
  float quality[8];      /* sum of base qualities per class */
  float mapq2;           /* sum of MAPQ^2 */
} bsc_pileup;

/* `gt_meth`, include/bs_call.h:152-160 — 200 bytes */
typedef struct {
  uint64_t counts[8];
  int32_t qual[8];      /* rounded mean quality per class */
  double gt_prob[10];   /* log10 posterior, order AA AC AG AT CC CG CT GG GT TT */
  double fisher_strand; /* log10 Fisher strand-bias p (0 for homozygous calls) */
  int32_t mq;           /* RMS mapping quality */
  int32_t aq;           /* mean base quality */
  uint8_t max_gt;       /* argmax of gt_prob (first maximum) */
} bsc_gt_meth;

/* One template (read pair) flattened out of `align_details`, include/bs_call.h:64-73 — 40 bytes.
 * Read k holds len[k] bytes of base|qual<<2 (GET_BASE/GET_QUAL, include/bs_call.h:41-42) at seq+off[k],
 * already trimmed/clipped/normalised by the host so that byte j sits at genome position pos[k]+j. */
typedef struct {
  uint32_t pos[2];     /* forward_position, reverse_position; 0 = none */
  uint32_t len[2];     /* 0 = read absent */
  uint64_t off[2];     /* byte offsets into the block's concatenated read buffer */
  uint8_t mapq[2];
  uint8_t orientation; /* gt_strand: 0 FORWARD, 1 REVERSE */
  uint8_t bs_strand;   /* gt_bs_strand: 0 NON_CONVERTED, 1 STRAND_C2T, 2 STRAND_G2A */
  uint32_t flags;      /* 0, or BSC_TPL_* (ABI 2; the field was padding, required to be 0, before); any other bit: BSC_ERR_ARG */
} bsc_template;

/*
 * bsc_template.flags.  HOT LOOP A flips the orientation for read 1 only if read 0 "was walked": if it holds a base whose
 * quality is neither 0 nor 63 (the scan of src/call_genotypes.c:198-211 and the `continue`s in front of :224).  A host that
 * writes the read bytes knows that bit and can hand it over; otherwise (flags = 0) the device finds it out itself, which costs
 * it a 128-byte memory line per template for the one byte that usually decides.  BSC_TPL_WALK_KNOWN without a correct
 * BSC_TPL_WALKED0 gives a pile-up the reference would not: use bsc_template_walk_flags().
 */
#define BSC_TPL_WALK_KNOWN 1u /* BSC_TPL_WALKED0 is valid */
#define BSC_TPL_WALKED0 2u    /* read 0 has a base with 0 < quality < 63 */
uint32_t bsc_template_walk_flags(const uint8_t *read0, uint32_t len0); /* host C (csrc/prep.c): stops at the first countable base */

/* Model parameters: sr_param.under_conv/over_conv/ref_bias/min_qual (include/bs_call.h:320-324;
 * defaults src/init_param.c:26-31 = 0.01, 0.05, 2, 20; min_qual is clamped to [1,43] as in
 * src/parse_args.c:170-171). */
typedef struct {
  double under_conv;
  double over_conv;
  double ref_bias;
  int32_t min_qual;
  int32_t device; /* HIP device ordinal; -1 = current device */
} bsc_params;

/* Per-context counters, summed over every call since creation / bsc_reset_stats().  These are the
 * fixed-size blocks that ranks all-reduce at the end of a sharded run. */
#define BSC_STATS_WORDS 16
typedef struct {
  uint64_t sites;      /* positions processed */
  uint64_t covered;    /* positions with n > 0 (not skipped) */
  uint64_t gt_hist[10];/* called genotype histogram over covered positions */
  uint64_t het_calls;  /* covered positions whose call is heterozygous (Fisher test evaluated) */
  uint64_t reserved[3];
} bsc_stats;

typedef struct bsc_context bsc_context;

int bsc_abi_version(void);
const char *bsc_last_error(void);
void bsc_params_default(bsc_params *p);

/*
 * bsc_create: replaces fill_base_prob_table() + init_calc_threads() (src/process.c:165-166;
 * src/genotype_model.c:10-21; src/call_genotypes.c:124-138) and the lfact_store / gt_het tables of
 * init_param() (src/init_param.c:16,52-55).  Builds the tables on the host with libm exactly as the
 * reference does, uploads them verbatim, creates the context's HIP stream and workspaces.
 */
int bsc_create(const bsc_params *params, bsc_context **out);

/* bsc_destroy: replaces join_calc_threads() (src/call_genotypes.c:140-153). */
int bsc_destroy(bsc_context *ctx);

/* Copies the host-built tables out: q_prob as [44][5] doubles (e,k,ln_k,ln_k_half,ln_k_one;
 * include/bs_call.h:148-150) and lfact_store[256] (src/stats_utils.c:14-21). */
int bsc_get_tables(const bsc_context *ctx, double *q_prob_44x5, double *lfact_256);

/*
 * The QUAL staircase the fused chain reads instead of evaluating src/print_vcf.c:140-148's log: phred = (int)(-10 *
 * log(om) / LOG10) capped at 255 is, as a function of om = 1 - exp(LOG10 * gt_prob[max_gt]), a staircase; for the binade
 * e = 1023 - exponent(om) (0 .. 63) of om, phred = base_64[e] + the number of thr_64x4[e][0..3] that om does not exceed.
 * Built on the host by bisection with the operations the kernels' log path runs (a context builds the same table and fails
 * if the staircase is not monotone).  No GPU involved: what tests/test_phred_table.py checks against libm double by double
 * around every step.  Returns BSC_OK or BSC_ERR_ARG.
 */
int bsc_phred_table(double *thr_64x4, unsigned char *base_64);

/*
 * bsc_call_sites: the call_thread() loop over one block (src/call_genotypes.c:43-115): per site the
 * quality/MAPQ summary, calc_gt_prob(), the strand table + fisher() for heterozygous calls.
 *   cts[n]  pile-ups (host memory)
 *   ref[n]  reference codes 0..4 = N,A,C,G,T for the same positions (src/get_sequence.c:20-54)
 *   out     n records of `out_stride` bytes each, out_stride = 200 (a gt_meth array) or 208 (a gt_vcf array,
 *           include/bs_call.h:162-166: gt_meth, then ready = 0 at byte 200 and skip at byte 201); the first 200
 *           bytes of record i are the gt_meth of site i (all zero for a skipped site).
 *   skip[n] 1 where n == 0 (gt_vcf.skip), else 0
 * Synchronous: returns when `out` and `skip` are filled.
 */
int bsc_call_sites(bsc_context *ctx, const bsc_pileup *cts, const uint8_t *ref, uint64_t n, void *out,
                   uint32_t out_stride, uint8_t *skip);

/* Pinned (page-locked) host memory for buffers passed to the host-buffer entries: with it the copies of
 * bsc_call_sites are true DMA transfers that overlap the kernels (three-stage pipeline over 1 Mi-site chunks);
 * pageable memory works too but is staged by the runtime.  E.g. allocate work->vcf with it. */
void *bsc_alloc_host(uint64_t bytes);
void bsc_free_host(void *p);

/*
 * Same computation on device-resident buffers; asynchronous on `stream` (a hipStream_t; NULL = HIP's default
 * stream, e.g. what torch.cuda.current_stream().cuda_stream returns for PyTorch's default stream).  The caller
 * orders its own work against that stream.  d_cts/d_out must be 16-byte aligned.  Used by bench.py and by
 * pipelines that keep pile-ups in HBM.
 */
int bsc_call_sites_device(bsc_context *ctx, const void *d_cts, const void *d_ref, uint64_t n, void *d_out,
                          uint32_t out_stride, void *d_skip, void *stream);

/*
 * bsc_accumulate: HOT LOOP A of call_genotypes_ML (src/call_genotypes.c:178-226) on the device: scatter the
 * block's reads into the pile-up of positions x..y (inclusive, 1-based genome positions as in the reference).
 *   tpl[nr]   templates in any order, like the reference's align_list (the sums do not depend on the order; the
 *             device orders them by leftmost position for its own tiling)
 *   seq       the concatenated read bytes the templates point into (seq_bytes long)
 *   out       y - x + 1 pile-ups (host memory), zero where nothing is covered
 * Returns BSC_ERR_ARG where the reference asserts (y < x; a template starting left of x; orientation > 1) or
 * where a template points outside seq — the templates are checked by the kernel that reads them, before anything
 * they point at is touched, and nothing is written to `out` for a bad block; BSC_WARN_INEXACT if a position's quality or MAPQ^2 sum exceeded 2^24
 * (the reference's float sums are order-dependent there).  Uses ctx min_qual.
 */
int bsc_accumulate(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                   uint32_t x, uint32_t y, bsc_pileup *out);

/*
 * bsc_accumulate_device: the same stage on device-resident inputs — d_tpl[nr] (bsc_template, 8-byte aligned), d_seq
 * (seq_bytes read bytes) -> d_cts, the pile-ups of x .. y (16-byte aligned; room for y - x + 1 rounded up to a whole
 * number of 64-position tiles, x 104 bytes: the last tile is written whole).  Asynchronous on `stream`.  The templates
 * are validated by the kernel that reads them; bsc_block_status() waits for `stream` and returns the verdict of the
 * block queued last: BSC_OK, BSC_WARN_INEXACT or BSC_ERR_ARG naming the first template that breaks one of the
 * reference's asserts (src/call_genotypes.c:186-188) — such a template contributes nothing to the pile-up.
 * bsc_last_accumulate_ms (with bsc_set_profiling): device time of the stage's launches (template checks + read
 * descriptors, ordering, tile search, accumulate) of the most recent call, from HIP events on its stream.
 */
int bsc_accumulate_device(bsc_context *ctx, const void *d_tpl, uint32_t nr, const void *d_seq, uint64_t seq_bytes, uint32_t x,
                          uint32_t y, void *d_cts, void *stream);
int bsc_block_status(bsc_context *ctx, void *stream);
int bsc_last_accumulate_ms(bsc_context *ctx, float *ms);

/*
 * bsc_call_block: the compute of one call_genotypes_ML() invocation (src/call_genotypes.c:155-273) without its
 * thread hand-offs: accumulate (above) followed by the per-site calling of bsc_call_sites(), the pile-up never
 * leaving the device.  ref[i] is the reference code of position x + i; out / out_stride / skip as in
 * bsc_call_sites().  INTEGRATION.md shows the replacement call_genotypes_ML() built on this.
 */
int bsc_call_block(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                   uint32_t x, uint32_t y, const uint8_t *ref, void *out, uint32_t out_stride, uint8_t *skip);

/*
 * Asynchronous form of bsc_call_block, mirroring how call_genotypes_ML() dispatches a block to the calc threads and
 * returns at once (src/call_genotypes.c:260-272) so that the process thread can prepare the next block meanwhile:
 *   bsc_block_submit  copies the block's inputs into pinned staging (tpl / seq / ref may be recycled when it
 *                     returns, as the reference's align_list is), queues copy + accumulate + call, returns;
 *   bsc_block_fetch   waits for that block and copies its records into out / skip (sizes as submitted); returns
 *                     BSC_WARN_INEXACT like bsc_accumulate.  The templates are checked on the device (see
 *                     bsc_accumulate), so a block that breaks one of the reference's asserts is reported here,
 *                     BSC_ERR_ARG, with nothing written — not by bsc_block_submit.
 * One block in flight per context, as in the reference (src/call_genotypes.c:161-168).
 */
int bsc_block_submit(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                     uint32_t x, uint32_t y, const uint8_t *ref, uint32_t out_stride);
int bsc_block_fetch(bsc_context *ctx, void *out, uint8_t *skip);
/* bsc_block_submit with the destination named up front: the copy-out is queued right behind the kernels, so the records
 * travel to the host while the caller prepares its next block, and bsc_block_fetch(ctx, NULL, NULL) only waits and
 * reports.  out / skip should come from bsc_alloc_host (with pageable memory the call waits for the block instead of
 * returning at once); their contents are undefined until the fetch has returned, and after a fetch that failed. */
int bsc_block_submit_to(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                        uint32_t x, uint32_t y, const uint8_t *ref, void *out, uint32_t out_stride, uint8_t *skip);

/*
 * VCF record formation: what the reference's print thread derives from a block's gt_meth records before it hands a
 * record to htslib (_print_vcf_entry, src/print_vcf.c:32-381, with the 5-site window of print_vcf_entry /
 * flush_vcf_entries, :529-594).  One 64-byte bsc_vcf_core per position; a host formatter turns the records with
 * emit = 1 into VCF/BCF lines.  Not covered: dbSNP names (pass rs_found per position in `dbsnp`, or NULL), the JSON
 * statistics, the header.
 */
typedef struct {
  uint32_t pos;     /* 1-based position; 0 in an all-zero record = nothing called here */
  uint8_t emit;     /* 1: the reference would write a record (not hom-ref AA/TT unless all_positions or dbSNP, in region) */
  uint8_t gt;       /* called genotype 0..9 (first-max argmax of gt_prob, src/print_vcf.c:584-591) */
  uint8_t ref_code; /* REF base code 0..4 as the printer sees it (an N up to two positions before blanks it, :570-577) */
  uint8_t gt_enc;   /* FORMAT GT: two BCF allele codes ((allele+1)<<1), high and low nibble (reference gt_int table) */
  uint8_t flt;      /* 1 q20, 2 qd2, 4 fs60, 8 mq40 (FILTER fail / FT names), 128 mac1; 0 = PASS */
  uint8_t phred;    /* QUAL and FORMAT GQ; 0 in a record with emit = 0 (the printer computes it for every position,
                       src/print_vcf.c:140-148, and reads it behind `skip` only, :185-217,382-398; round 5) */
  uint8_t n_gl;     /* number of FORMAT GL values */
  char cg;          /* FORMAT CG: 'C' (= "CG"), 'H', 'N', '?', '.' */
  char alt[2];      /* ALT alleles, 0-padded */
  char cx_ref[5];   /* INFO CX: reference context */
  char cx_gt[5];    /* FORMAT CX: IUPAC context of the called genotypes */
  int32_t fs;       /* FORMAT FS (the reference writes it for heterozygous genotypes only) */
  uint32_t qd;      /* FORMAT QD; 0 where emit = 0, like phred */
  uint32_t dp;      /* FORMAT DP (non-informative depth) */
  float gl[6];      /* FORMAT GL */
  uint32_t _pad;
} bsc_vcf_core;

typedef struct {
  int32_t all_positions; /* sr_param.all_positions (-A) */
  uint32_t reg_start;    /* emit only reg_start <= position <= reg_stop: ctg->curr_reg, or 1 .. ctg->end_pos */
  uint32_t reg_stop;
} bsc_vcf_params;

/* Host buffers: gtm = n records of gtm_stride bytes (200 or 208) for positions x .. x+n-1, skip[n], ref[n+2] = the
 * reference codes of x .. x+n+1 (work->ref), dbsnp[n] = rs_found (0/1/3) per position or NULL; out[n]. */
int bsc_vcf_records(bsc_context *ctx, const void *gtm, uint32_t gtm_stride, const uint8_t *skip, const uint8_t *ref,
                    const uint8_t *dbsnp, uint32_t n, uint32_t x, const bsc_vcf_params *params, bsc_vcf_core *out);
/* Same on device-resident buffers, asynchronous on `stream` (chains behind bsc_call_sites_device). */
int bsc_vcf_records_device(bsc_context *ctx, const void *d_gtm, uint32_t gtm_stride, const void *d_skip,
                           const void *d_ref, const void *d_dbsnp, uint32_t n, uint32_t x,
                           const bsc_vcf_params *params, void *d_out, void *stream);

/*
 * Written records only, packed — what a host that renders VCF/BCF needs from a block: for every position with
 * emit = 1, in position order, its bsc_vcf_core and the gt_meth fields the encoder still reads (MC8 = counts,
 * AMQ = qual, MQ; src/print_vcf.c:306-359).  64 bytes per position on average on WGBS data instead of the 201 of a
 * gt_vcf entry, which is what bounds the host-buffer paths (PCIe).
 */
typedef struct {
  bsc_vcf_core core;  /* 64 bytes */
  uint32_t counts[8]; /* gt_meth.counts (a count beyond 2^32 - 1 saturates) */
  uint8_t qual[8];    /* gt_meth.qual, 0..43 */
  int32_t mq, aq;     /* gt_meth.mq, gt_meth.aq */
  uint8_t max_gt;     /* gt_meth.max_gt */
  uint8_t rs_found;   /* the dbSNP flag that was passed for the position (0 without dbSNP) */
  uint8_t _pad[14];
} bsc_vcf_rec;

/* d_core[n] / d_gtm[n] (device) -> d_out[min(*count, out_cap)] packed records and the number of written records of the
 * block in *d_count (a device u64; records beyond out_cap are counted, not stored).  Asynchronous on `stream`.
 * gtm_stride = 0: d_gtm is the d_aux array of bsc_reads_chain_device (64 bytes per position, already the second half of
 * the packed record; d_dbsnp is then unused: the flag is in it). */
int bsc_vcf_compact_device(bsc_context *ctx, const void *d_core, const void *d_gtm, uint32_t gtm_stride,
                           const void *d_dbsnp, uint32_t n, void *d_out, uint64_t out_cap, void *d_count, void *stream);

/*
 * One block from reads to written records, nothing else crossing PCIe on the way back: what call_genotypes_ML hands to
 * the calc threads (src/call_genotypes.c:180-226 -> :260-272) and the print thread makes of their output
 * (src/process.c:87-104 -> src/print_vcf.c:32-594), on the reads-in chain (bsc_reads_chain_device: neither the pile-up nor
 * gt_meth in HBM), then the packing — every launch and copy queued back to back, ONE wait for the templates' verdict, the
 * record count and the records together.  ref[y - x + 3] = reference codes of x .. y + 2 (work->ref1 as the reference fills
 * it, src/process_template.c:29-30); dbsnp = rs_found per position or NULL; with_stats != 0: the site statistics too.
 * out[out_cap]: pinned or pageable; *n_out = records written.  BSC_ERR_ARG if out_cap is too small (*n_out then holds
 * the number needed) or a template breaks one of the reference's asserts (the contents of out are then unspecified);
 * BSC_WARN_INEXACT as bsc_accumulate.
 * The reference aborts on such a template (asserts enabled in release builds, src/call_genotypes.c:186-188); here the block is
 * queued before its verdict is known (one wait per block), so when BSC_ERR_ARG names a template the context's site statistics,
 * CpG carry and position counters already hold what the block's other reads gave: they are undefined from there on — a
 * caller that goes on after the error calls bsc_reset_site_stats() (and bsc_reset_stats()) first.
 * One block in flight per context across ALL host-buffer block entries (bsc_accumulate, bsc_call_block, bsc_block_submit[_to],
 * bsc_block_records[_submit[_inplace]]): they share the staging area, the device workspaces and the verdict counters, and
 * each refuses with BSC_ERR_ARG while a submitted block has not been fetched.
 *
 * bsc_block_records_submit / _fetch: the same split where the reference splits it — submit returns once the block is queued,
 * as call_genotypes_ML returns once its calc threads are dispatched (src/call_genotypes.c:260-272), its inputs copied to a
 * pinned staging area (the caller's buffers are free at once); fetch is the wait at the top of the next call (:161-168) and
 * completes the block into the `out` named at submit (which must stay valid until then).  One block in flight per context.
 * bsc_block_records_submit_inplace: no staging copy — the inputs are read where they lie (page-locked buffers from
 * bsc_alloc_host make the upload a true DMA) and must stay unchanged until the fetch, as the reference's align_list stays
 * untouched until the next hand-off (src/process.c:61,68).  A host that flattens block k + 1 into a second set of buffers
 * while block k is in flight on another context keeps both PCIe directions busy (tools/bench_two_contexts.py).
 */
int bsc_block_records(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                      uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params,
                      int with_stats, bsc_vcf_rec *out, uint64_t out_cap, uint64_t *n_out);

int bsc_block_records_submit(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x,
                             uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                             bsc_vcf_rec *out, uint64_t out_cap);
int bsc_block_records_submit_inplace(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                                     uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params,
                                     int with_stats, bsc_vcf_rec *out, uint64_t out_cap);
int bsc_block_records_fetch(bsc_context *ctx, uint64_t *n_out);

/*
 * Several blocks in ONE launch sequence.  The reference calls call_genotypes_ML once per maximal run of overlapping templates
 * (src/get_template_vector.c:141-147: 10^2 .. 10^7 positions a call) and its calc threads cost nothing to start; a GPU block
 * costs a dozen launches, four copies and a wait whatever its size, which a 10 000-position block does not repay.  A host that
 * holds blocks back (integration/call_genotypes_amd_overlap.c: until 1 M positions or 65 536 blocks are pending, or the run ends;
 * a batch may mix contigs — every block carries its own)
 * hands them over together: one upload, one grouping pass over all their reads, the reads-in chain over all their tiles, one
 * packing pass, one copy-out, one wait.  Every block is still a block of its own — its printer state flushed at its end
 * (src/print_vcf.c:529-546), its first two and last two positions without the neighbours' context — so the records and the
 * statistics are those of bsc_block_records called on the blocks one after another, in order.
 *   blocks[n_blocks]  x .. y and the number of templates of every block, in the order their records are wanted (genome order
 *                     on one contig: the CpG pairing across blocks, src/print_vcf.c:447-455, follows this order)
 *   tpl, seq          the templates of all blocks, block after block (off[] index the one read buffer `seq`)
 *   ref               the blocks' reference codes one block after another: y - x + 3 codes each (x .. y + 2)
 *   dbsnp             NULL, or rs_found per position, block after block (y - x + 1 each)
 *   out[out_cap]      the written records of all blocks, block after block; block_counts[n_blocks] (optional) = how many each
 *                     block wrote
 * At most 2^28 - 1 positions (each block rounded up to a multiple of 64) and 65 536 blocks a call.  Errors as
 * bsc_block_records (a bad template is named by its index among the call's templates); one call in flight per context,
 * shared with the single-block entries.  bsc_blocks_records_submit copies the inputs to the pinned staging area: the caller's
 * buffers are free when it returns; `out` must stay valid until the fetch.
 */
typedef struct {
  uint32_t x, y; /* first and last position of the block */
  uint32_t nr;   /* its templates: the next nr of tpl[] */
  uint32_t _pad;
} bsc_block_desc;
int bsc_blocks_records_submit(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl,
                              const uint8_t *seq, uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp,
                              const bsc_vcf_params *params, int with_stats, bsc_vcf_rec *out, uint64_t out_cap);
/* no staging copy of templates, reads and reference codes: read where they lie (page-locked buffers from bsc_alloc_host make the
 * upload a true DMA), unchanged until the fetch — the form for a host that flattens its blocks straight into such buffers */
int bsc_blocks_records_submit_inplace(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl,
                                      const uint8_t *seq, uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp,
                                      const bsc_vcf_params *params, int with_stats, bsc_vcf_rec *out, uint64_t out_cap);
int bsc_blocks_records_fetch(bsc_context *ctx, uint64_t *n_out, uint64_t *block_counts);
int bsc_blocks_records(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                       uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                       bsc_vcf_rec *out, uint64_t out_cap, uint64_t *n_out, uint64_t *block_counts);

/*
 * The gt_meth / gt_vcf form: bsc_block_submit_to for several blocks in ONE launch sequence — same block list, same joined
 * template / read / reference arrays (ref: y - x + 3 codes per block, of which the first y - x + 1 are used).  One upload, one
 * grouping pass over all the blocks' reads, one accumulate launch and one launch of the calling kernel over all positions, one
 * copy-out.  In `out` (page-locked; out_stride = 200 or 208) and `skip` every block starts on a multiple of 64 positions:
 * block b's images from image block_off[b] on (returned; each block takes its positions rounded up to 64, and out / skip
 * must hold the sum); the positions in between are images of nothing (skip = 1).  bsc_block_fetch(ctx, NULL, NULL) completes it;
 * counters and errors as for one block (a bad template is named by its index among the call's templates).  The drop-in glue
 * (integration/amd_overlap_protocol.h) holds small blocks back and hands them over through this entry.
 */
int bsc_blocks_submit_to(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                         uint64_t seq_bytes, const uint8_t *ref, void *out, uint32_t out_stride, uint8_t *skip, uint64_t *block_off);
/* The same without the staging copy (round 5): templates, reads and reference codes are uploaded from where they lie — page-locked
 * buffers from bsc_alloc_host make that a true DMA — and must stay unchanged until bsc_block_fetch, as the reference's align_list
 * stays untouched until the next hand-off (src/process.c:61,68).  The form for a host that builds its batch straight in such
 * buffers (integration/amd_overlap_protocol.h since round 5): the staged form from ordinary memory runs at a third of its rate
 * (profiles/r05_small_blocks.txt). */
int bsc_blocks_submit_to_inplace(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl,
                                 const uint8_t *seq, uint64_t seq_bytes, const uint8_t *ref, void *out, uint32_t out_stride, uint8_t *skip,
                                 uint64_t *block_off);

/* bsc_vcf_format for a packed record. */
int bsc_vcf_format_rec(const bsc_vcf_rec *r, const char *contig, const char *id, char *buf, size_t cap);

/*
 * One written record as a BCF2 record (host C; csrc/bcf.c): the bytes bcf_write() emits for the record the reference
 * assembles with htslib's typed-value encoders (src/print_vcf.c:160-222, :267-378) — 32 bytes of fixed fields, the shared
 * block (ID, REF, ALT, FILTER, INFO CX) and the per-sample block (GT FT DP MQ GQ QD GL MC8 [AMQ] CS CG CX [FS]).
 * `ids` = the header dictionary indices of the keys (bsc_bcf_default_ids: the header print_vcf_header writes,
 * src/print_vcf.c:712-731); rid = the contig's index among the header's ##contig lines; id / id_len = the dbSNP name as
 * bsc_dbsnp_name returns it (id_len 0: no ID).  Returns the record's length (> cap: buf too small, nothing usable
 * written), 0 for a record that is not written (emit == 0), -1 on a bad argument.  The typed-value rules are the BCF2
 * specification's; byte parity with htslib itself is not pinned in this image.
 */
typedef struct {
  int32_t pass, fail, mac1, info_cx;
  int32_t fmt_gt, fmt_ft, fmt_gl, fmt_gq, fmt_dp, fmt_mq, fmt_qd, fmt_mc8, fmt_amq, fmt_cs, fmt_cg, fmt_cx, fmt_fs;
} bsc_bcf_ids;
void bsc_bcf_default_ids(bsc_bcf_ids *ids);
long bsc_bcf_record(const bsc_vcf_rec *r, int32_t rid, const char *id, size_t id_len, const bsc_bcf_ids *ids, uint8_t *buf,
                    size_t cap);
/* The written records of a block one after the other, named from the dbSNP index where rs_found is set (db may be NULL);
 * stops in front of the first record that does not fit: *n_done = records consumed.  Returns the bytes written. */
struct bsc_dbsnp;
long bsc_bcf_block(const bsc_vcf_rec *recs, uint64_t n, int32_t rid, const bsc_bcf_ids *ids, const struct bsc_dbsnp *db, uint8_t *buf,
                   size_t cap, uint64_t *n_done);

/*
 * Site statistics: the sums the reference's printer adds to bs_stats for every position that reaches
 * _print_vcf_entry (src/print_vcf.c:382-526; types include/bs_call.h:75-95,120-146) — the payload of the one
 * collective of a sharded run (every field is a sum, so shards add).  Computed on the device
 * from the bsc_vcf_core records of a block and the gt_meth records they were made from, accumulated in the context.
 * Faithful to the reference including its quirks: by the time the statistics are taken `alt` has been walked to its
 * terminator (:177-181), so every written record counts as a SNP (`alt[0] != '.'`, `alt[1] != ','`) and none as
 * multi-allelic; a CpG is counted when a '-' strand call directly follows a '+' strand call (prev_cpg_x, which
 * persists from block to block — pass the blocks of a contig in order).  Not covered: the per-coverage GC
 * histogram (gt_cov_stats.gc_pcent needs the reference's GC bins), indels (the path calls none), the coverage
 * table beyond BSC_COV_CAP - 1 (deeper positions share the last row), negative FS values (the reference indexes
 * its fs_stats vector with them: undefined behaviour there, ignored here).
 * Integer fields are exact; the methylation profiles are sums of doubles whose order differs from the reference's
 * position order (relative differences ~1e-15).
 */
#define BSC_COV_CAP 4096
typedef struct {
  uint64_t snps[2], indels[2], multi[2], dbSNP_sites[2], dbSNP_var[2], CpG_ref[2], CpG_nonref[2]; /* [all, passed] */
  uint64_t mut_counts[12][2], dbSNP_mut_counts[12][2]; /* stats_mut order AC AG AT CA CG CT GA GC GT TA TC TG */
  uint64_t qual[4][256];          /* qual_cat: all sites, variant sites, CpG ref, CpG non-ref; index = phred */
  uint64_t filter_counts[2][32];  /* [het genotype][flt & 31] */
  uint64_t qd_stats[256][2], fs_stats[256][2], mq_stats[256][2]; /* index = QD / FS / MQ value; [hom, het] */
  uint64_t cov[BSC_COV_CAP][6];   /* gt_cov_stats by coverage: all, var, CpG[2] (index = total depth), CpG_inf[2]
                                     (index = informative depth) */
  double CpG_ref_meth[2][101], CpG_nonref_meth[2][101]; /* [all, passed][methylation %] */
} bsc_site_stats;

/* Adds the statistics of n positions to the context's block: d_core[n] = the bsc_vcf_core records of the block (as
 * written by bsc_vcf_records_device), d_gtm the gt_meth records they came from, d_dbsnp = rs_found per position or
 * NULL.  Asynchronous on `stream`. */
int bsc_vcf_stats_device(bsc_context *ctx, const void *d_core, const void *d_gtm, uint32_t gtm_stride,
                         const void *d_dbsnp, uint32_t n, void *stream);
/* Host-buffer form (copies in, runs, returns when done). */
int bsc_vcf_stats(bsc_context *ctx, const bsc_vcf_core *core, const void *gtm, uint32_t gtm_stride, const uint8_t *dbsnp,
                  uint32_t n);
/* Device -> host (synchronises the device first) / back to zero, the CpG carry included. */
int bsc_get_site_stats(bsc_context *ctx, bsc_site_stats *out);
int bsc_reset_site_stats(bsc_context *ctx);

/*
 * GC content by coverage (gt_cov_stats.gc_pcent, src/print_vcf.c:394-398; the report's "GC" object): with the bins of the
 * contig being walked set, every bsc_chain_device(with_stats) call also adds its positions that reached the printer to
 * table[total depth][G+C count of the position's 100-base bin].  The bins are the reference's ctg_stats->gc
 * (src/read_reference.c:44-131): bsc_gc_bins computes them on the host from the contig's reference codes.
 *   bsc_set_gc_bins   d_gc[n_bins] in device memory (the caller keeps it alive while set), start_pos = the contig's first
 *                     A/C/G/T position; d_gc NULL switches the table off.  Call at every contig change.
 *   bsc_get_gc_stats  out[BSC_COV_CAP][101] (synchronises the device); zeroed by bsc_reset_site_stats.
 * bsc_vcf_stats_device / bsc_vcf_stats / bsc_block_records(with_stats) feed the same table (depth = the sum of the gt_meth counts).
 */
int bsc_gc_bins(const uint8_t *codes, uint64_t n, uint32_t *start_pos, uint8_t *out, uint64_t out_cap, uint64_t *n_bins);
int bsc_set_gc_bins(bsc_context *ctx, const void *d_gc, uint32_t n_bins, uint32_t start_pos);
/* the same from host memory: the context keeps its own device copy (synchronises the device) */
int bsc_set_gc_bins_host(bsc_context *ctx, const uint8_t *gc, uint32_t n_bins, uint32_t start_pos);
int bsc_get_gc_stats(bsc_context *ctx, uint64_t *out);

/* The first 14 words of the statistics block — snps, indels, multi, dbSNP_sites, dbSNP_var, CpG_ref, CpG_nonref, each
 * [all, passed] — as they stand now (synchronises the device).  The reference keeps a copy of these seven pairs per contig
 * (gt_ctg_stats, include/bs_call.h:124-135): a caller that walks contig after contig takes the difference of two reads. */
int bsc_get_site_totals(bsc_context *ctx, uint64_t out[14]);

/*
 * The JSON report of a run (host C; csrc/report.c): the text output_stats() writes at the end of a run
 * (src/stats.c:19-298), byte for byte, from the statistics this library accumulates and the counters of the stages in
 * front of it.  bsc_report_json returns the length of the full text (like snprintf: a return >= cap means buf was too
 * small; call with cap = 0 to size it), -1 on a NULL argument.
 */
typedef struct { /* gt_ctg_stats (include/bs_call.h:124-135) without the GC bins */
  const char *name;
  uint64_t snps[2], indels[2], multi[2], dbSNP_sites[2], dbSNP_var[2], CpG_ref[2], CpG_nonref[2];
} bsc_contig_totals;
typedef struct {
  double under_conv, over_conv;  /* the "source" line (src/stats.c:29) */
  int32_t mapq_thresh, min_qual;
  int32_t day, month, year;      /* the "date" line; year 0 = today (local time, as the reference) */
  int32_t have_dbsnp;            /* a dbSNP index was loaded: the dbSNP totals are reported */
  uint64_t filter_cts[15], filter_bases[15]; /* reads / bases by the reader's verdict, gt_filter_reason order
                                                (include/bs_call.h:50; [14] = "PairNotFound"); [0] = passed */
  uint64_t base_filter[5];       /* base_filter_types order: passed, trimmed, clipped, overlapping, low quality */
  const bsc_site_stats *total;
  const uint64_t *gc;            /* [BSC_COV_CAP][101]: positions by coverage and GC percent of their 100-base bin
                                    (gt_cov_stats.gc_pcent), or NULL = all zero */
  const uint64_t *read_profile;  /* [n_read_profile][4]: bs_stats.meth_profile, element 0 included (never reported) */
  uint32_t n_read_profile;
  uint32_t n_contigs;
  const bsc_contig_totals *contigs;
} bsc_report;
long bsc_report_json(const bsc_report *r, char *buf, size_t cap);

/*
 * The fused chain — pile-up -> call -> VCF record -> site statistics in ONE pass, the 200-byte gt_meth records never
 * reaching HBM: what the print thread consumes from a block (src/process.c:87-104 feeding src/print_vcf.c:32-594),
 * computed from what the process thread produces (src/call_genotypes.c:178-226).  Same bytes as
 * bsc_call_sites_device -> bsc_vcf_records_device (-> bsc_vcf_stats_device) over the whole block.
 *
 * A call handles one WINDOW of a block (the reference's unit: x .. y of one call_genotypes_ML, whose printer state is
 * flushed at its end), so that a long block — a whole contig — can be walked in fixed windows with HBM-resident
 * inputs; the windows of a block, passed in order, give exactly the block's records and statistics.  The printer's
 * record of a position looks at the called genotypes of 2 positions and the reference bases of up to 4 / 2 positions
 * either side, so the window's buffers carry that much context where the block has it:
 *   d_cts    pile-ups of block positions first - lc .. first + n + rc - 1, lc = min(2, first),
 *            rc = min(2, n_block - first - n)                                   [(lc + n + rc) x 104 bytes]
 *   d_ref    reference codes of block positions first - lr .. first + n + 1, lr = min(4, first)  (a block's
 *            reference has n_block + 2 codes: x .. y + 2, src/process_template.c:29-30)
 *   d_dbsnp  rs_found (0/1/3) of block positions first .. first + n - 1, or NULL
 *   d_core   n bsc_vcf_core records (16-byte aligned)
 * Asynchronous on `stream`; counters (bsc_get_stats) and, with_stats != 0, the site statistics accumulate in the context.
 */
typedef struct {
  uint32_t x;       /* genome position (1-based) of the block's first position */
  uint32_t n_block; /* positions in the block (y - x + 1) */
  uint32_t first;   /* block-relative index of the window's first position */
  uint32_t n;       /* positions in the window */
} bsc_window;
int bsc_chain_device(bsc_context *ctx, const void *d_cts, const void *d_ref, const void *d_dbsnp, const bsc_window *w,
                     const bsc_vcf_params *params, int with_stats, void *d_core, void *stream);
/*
 * bsc_reads_chain_device: one whole block from reads to records in the chain's single pass — the block's templates are
 * checked, described and ordered like bsc_accumulate_device's, then every 60-position tile is piled up in the LDS of the
 * wave that calls it: HOT LOOP A (src/call_genotypes.c:180-226), the calc threads (:43-115) and what the print thread
 * derives from their output (src/process.c:87-104 -> src/print_vcf.c:32-594) with neither the pile-up nor gt_meth in HBM.
 *   d_tpl[nr], d_seq   device-resident templates (8-byte aligned) and read bytes of the block x .. y
 *   d_ref              reference codes of x .. y + 2 (y - x + 3 bytes: work->ref1, src/process_template.c:29-30)
 *   d_dbsnp            rs_found per position or NULL
 *   d_core             y - x + 1 bsc_vcf_core records (16-byte aligned): the bytes bsc_accumulate_device + bsc_chain_device give
 *   d_aux              NULL, or 64 bytes per position (16-byte aligned): the second half of a bsc_vcf_rec — counts[8], qual[8],
 *                      mq, aq, max_gt, rs_found — of every position with a record, zero elsewhere
 * Asynchronous on `stream`; bsc_block_status() returns the templates' verdict (an invalid template contributes nothing);
 * counters and, with_stats != 0, the site statistics accumulate in the context.  bsc_last_reads_chain_ms (with
 * bsc_set_profiling): device time of all of its launches.
 */
int bsc_reads_chain_device(bsc_context *ctx, const void *d_tpl, uint32_t nr, const void *d_seq, uint64_t seq_bytes, uint32_t x,
                           uint32_t y, const void *d_ref, const void *d_dbsnp, const bsc_vcf_params *params, int with_stats,
                           void *d_core, void *d_aux, void *stream);
int bsc_last_reads_chain_ms(bsc_context *ctx, float *ms);
/* How bsc_reads_chain_device / bsc_block_records get from reads to records.  0 (default): TWO kernels — the accumulate kernel's
 * summary form leaves 48 bytes per position in HBM (per class its count and the forward-strand part of it, 16 bits each, and the
 * per-site summary of src/call_genotypes.c:44-59; the context's own workspace; a block with a position deeper than 65 535 reads
 * of one class is flagged on the device and done by the reads-in kernel queued behind as a stand-in) and the chain kernel's summary-in form starts from them: the faster form (the walk alone runs at 24
 * waves to a CU and hides its byte loads — inside the 128-register chain kernel it waits for them — and the summary's arithmetic
 * runs where there are issue slots to spare), taken whenever that workspace can be allocated.  1: always the ONE-kernel form
 * (reads-in chain: nothing per position in HBM but the records).  Same records and statistics either way.  (bsc_blocks_records
 * always runs the one-kernel form: small blocks are bound by launches and PCIe, not by the walk.)  Memory: the two-kernel form
 * keeps 48 bytes per position of the largest block seen so far in the context (grow-only; 2.4 GB for a 50 M-position block, 13 GB
 * for a maximal one of 2^28) — a context that must stay lean sets 1. */
int bsc_set_reads_fused(bsc_context *ctx, int fused);
/* Test hook (tests/test_gpu_reads_chain.py): on != 0 makes the summaries' allocation of the two-kernel form fail for real — an absurd
 * request the runtime refuses — so that the quiet fall-back to the one-kernel form can be exercised.  Off in every new context; nothing
 * in the environment switches it (the switches that are read from the environment — BSC_NO_H2D_TURNS, BSC_NO_EMIT_BYTES,
 * BSC_STAGE_TIMING, BSC_MAX_LAUNCH_SITES: A/B measurements — are read ONCE, by bsc_create). */
int bsc_debug_fail_summary_alloc(bsc_context *ctx, int on);
/* Window sizes for a caller that cuts a resident contig into windows of its own choosing (the reference's blocks are data
 * dependent, src/process_template.c:24-28; SURVEY.md 8d fixes 4 Mi).  bsc_chain_window_size: the largest window <= limit in
 * which every resident wave runs the same number of tiles and the main launch covers the window exactly — CUs x waves per
 * workgroup x (60 + 62 k): the first tile of a wave's run forms 60 records, every further one 62.  bsc_chain_window_quantum:
 * the smallest such window (k = 0).  0 if ctx is NULL. */
uint32_t bsc_chain_window_size(const bsc_context *ctx, uint32_t limit);
uint32_t bsc_chain_window_quantum(const bsc_context *ctx);
/* with bsc_set_profiling: device time of the most recent bsc_chain_device call (all of its launches) */
int bsc_last_chain_ms(bsc_context *ctx, float *ms);

/*
 * Read pre-processing (host C; csrc/prep.c): from the templates the reader hands over to the templates bsc_accumulate /
 * bsc_call_block / bsc_block_records consume — what process_template_vector does to every template of a block before it
 * calls call_genotypes_ML (src/process_template.c:36-111): the fixed trims (trim_read, src/read_utils.c:13-26), the
 * soft clips (trim_soft_clips, src/al_utils.c:122-162), the overlap of the two mates (handle_overlap, :164-318) and the
 * normalisation of indels (a deletion from the reference becomes bytes 0, an insertion is removed).
 *
 * A raw template is `align_details` (include/bs_call.h:64-73) with its gt_vectors flattened: read k = len[k] bytes
 * base|qual<<2 at seq + off[k]; its mismatch list = n_misms[k] entries at misms + misms_off[k], as get_bam_misms builds
 * it from the CIGAR (src/input_sam.c:90-136; note the reference's naming: CIGAR D -> INS, CIGAR I -> DEL).
 */
#define BSC_MISMS_MISMS 0u /* gt_misms_t, include/bs_call.h:54 */
#define BSC_MISMS_INS 1u
#define BSC_MISMS_DEL 2u
#define BSC_MISMS_SOFT 3u
typedef struct {
  uint32_t type;     /* BSC_MISMS_* */
  uint32_t position; /* offset in the read */
  uint32_t size;
} bsc_misms; /* gt_misms, include/bs_call.h:55-62 */
typedef struct {
  uint32_t pos[2];            /* forward_position, reverse_position; 0 = none */
  uint32_t reference_span[2];
  uint32_t len[2];            /* 0 = read absent */
  uint32_t n_misms[2];
  uint64_t off[2];            /* byte offsets of the reads in seq */
  uint64_t misms_off[2];      /* index of each read's first entry in misms */
  uint8_t mapq[2];
  uint8_t orientation;        /* gt_strand: 0 FORWARD, 1 REVERSE */
  uint8_t bs_strand;          /* gt_bs_strand */
  uint32_t _pad;
} bsc_raw_template;
typedef struct {
  int32_t left_trim[2], right_trim[2]; /* sr_param.left_trim / right_trim: [0] read 1, [1] read 2 (-L / -R) */
  int32_t min_qual;                    /* only for the base counters below */
} bsc_prep_params;
typedef struct { /* the base_filter / filter_cts counters of bs_stats this stage feeds (src/process_template.c:50-59) */
  uint64_t base_none, base_trim, base_clip, base_overlap, base_lowqual; /* base_filter[] */
  uint64_t reads, read_bases;                                           /* filter_cts / filter_bases[gt_flt_none] */
} bsc_prep_stats;
/* tpl_out[nr] and the prepared read bytes in seq_out (capacity seq_out_cap: the input bytes plus the padded deletions
 * always fit in seq_bytes + sum of INS sizes); *seq_out_used = bytes written.  BSC_ERR_ARG where the reference aborts
 * (a soft clip that is not at the end of its read or swallows it), naming the template. */
int bsc_prepare_templates(const bsc_raw_template *raw, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                          const bsc_misms *misms, uint64_t n_misms, const bsc_prep_params *par, bsc_template *tpl_out,
                          uint8_t *seq_out, uint64_t seq_out_cap, uint64_t *seq_out_used, bsc_prep_stats *stats);
/* The same, and the non-CpG read profile of the templates (meth_profile, src/meth_profile.c:48-77: by position in the
 * original read, C / G of the reference outside a CpG seen converted or not — bs_stats.meth_profile, the report's
 * "NonCpGreadProfile").  ref = reference codes 0..4 of genome positions x .. x + n_ref - 1, covering every template
 * and one base either side (the block's work->ref1: x .. y + 2); counts[cap][4] and used persist from call to call
 * (used = the reference vector's length: elements 1 .. used - 1 are reported, element i + 1 = read position i). */
typedef struct {
  const uint8_t *ref;
  uint32_t x, n_ref;
  uint64_t *counts;
  uint32_t cap, used;
} bsc_read_profile;
int bsc_prepare_templates_profile(const bsc_raw_template *raw, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                                  const bsc_misms *misms, uint64_t n_misms, const bsc_prep_params *par, bsc_template *tpl_out,
                                  uint8_t *seq_out, uint64_t seq_out_cap, uint64_t *seq_out_used, bsc_prep_stats *stats,
                                  bsc_read_profile *profile);
/*
 * bsc_prepare_templates ON THE DEVICE (round 5; csrc/prepdev.hip): the same bytes — prepared templates incl. their flags, prepared
 * reads in template order, the statistics — from device-resident inputs into device-resident outputs, ready for
 * bsc_accumulate_device / bsc_reads_chain_device without a host pass: one thread per template plans the trims, clips, the mate
 * overlap and the indel normalisation on the mismatch lists alone, a prefix sum places the reads, a wave per 64 reads writes
 * them (a dword per lane).  d_raw (bsc_raw_template[nr], 8-byte aligned), d_seq, d_misms (bsc_misms[n_misms]); d_tpl_out (bsc_template[nr]) and
 * d_seq_out (seq_out_cap bytes: seq_bytes + the sizes of all BSC_MISMS_INS entries always suffice).  Queued on `stream`, then
 * waited for: *seq_out_used, *stats (may be NULL) and the verdict come back with the call — BSC_ERR_ARG where the host form
 * fails, naming the lowest offending template with the host form's own message.  Workspaces (88 bytes per read + the lists)
 * stay with the context.  profile (may be NULL): the non-CpG read profile as bsc_prepare_templates_profile makes it — here
 * profile->ref is a DEVICE pointer to the codes of x .. x + n_ref - 1, counts / cap / used are the host's (counts[cap][4] receives
 * this call's counts, used grows, with the host form's clearing of what lies behind the old end).
 */
int bsc_prepare_templates_device(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes,
                                 const void *d_misms, uint64_t n_misms, const bsc_prep_params *par, void *d_tpl_out, void *d_seq_out,
                                 uint64_t seq_out_cap, uint64_t *seq_out_used, bsc_prep_stats *stats, bsc_read_profile *profile,
                                 void *stream);
/*
 * bsc_block_records from what the READER delivers (bsc_read_block: raw templates, their reads and mismatch lists, host buffers):
 * uploaded as they are, prepared on the device, then grouped, walked and called like bsc_block_records — the process thread's
 * per-template work (src/process_template.c:36-111) on no host core.  x .. y = the block as the reader found it (x =
 * bsc_block_start(raw), y = bsc_read_block.y); ref = the codes of x .. y + 2.  prep_stats (may be NULL) = the base counters the
 * pre-processing feeds; profile (may be NULL) = the read profile, counts / cap / used as in bsc_read_profile (ref, x, n_ref are
 * taken from the block).  Same records as bsc_prepare_templates[_profile] + bsc_block_records; two host waits (the prepared
 * size, the records) instead of one.
 */
int bsc_block_records_raw(bsc_context *ctx, const bsc_raw_template *raw, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                          const bsc_misms *misms, uint64_t n_misms, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref,
                          const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, bsc_vcf_rec *out, uint64_t out_cap,
                          uint64_t *n_out, bsc_prep_stats *prep_stats, bsc_read_profile *profile);
/*
 * The BCF stream of a block, encoded ON THE DEVICE (round 5, csrc/bcfdev.hip): what the output thread hands to bcf_write for the block's
 * written records — the typed values of src/print_vcf.c:160-222,267-378 behind bcf_write's fixed fields, byte for byte what
 * bsc_bcf_block (host C, the checker) makes of the same packed records — so that the record formation's tail never touches a host core
 * and ~113 bytes per written record cross PCIe instead of 128.
 *   bsc_bcf_names            the dbSNP names of the block's flagged positions (host arrays; the block entries — bsc_block_bcf*, the split
 *                            forms included — copy them into a page-locked area of the context's and upload them with the block's other
 *                            inputs, ahead of its kernels: the arrays are the caller's again when the call returns; the device-level
 *                            entries bsc_bcf_block_device / bsc_bcf_sites[_len]_device queue the upload from the caller's arrays on the
 *                            caller's stream: keep them unchanged until that stream has passed the call, page-locked for a true DMA):
 *                            pos[n] ascending
 *                            1-based positions, off[n + 1] offsets into bytes; bsc_dbsnp_names fills one from the loaded contig.  A
 *                            record whose rs_found flag is set and whose position the table lists carries that ID (at most 63 bytes of it)
 *   bsc_bcf_block_device     d_recs[<= max_recs] packed records in HBM, *d_n_recs of them (a device u64: the count bsc_vcf_compact_device
 *                            left) -> d_out[<= out_cap] bytes; d_totals = three device u64 {length of the stream, records bsc_bcf_record
 *                            refuses (gt > 9 or n_gl > 6: counted, written with the values clamped), records written}; a stream longer
 *                            than out_cap is cut at a 64-record boundary, its full length still in d_totals[0].  Asynchronous on `stream`.
 *                            d_out must be 16-byte aligned (BSC_ERR_ARG otherwise): the write kernel owns whole 16-byte pieces of the stream
 *   bsc_bcf_sites_device     the same from the per-position arrays bsc_reads_chain_device leaves — d_core[n] and d_aux[n] (64 bytes each
 *                            per position, 16-byte aligned) — with no packing pass in between: a position without a record costs 16 bytes
 *   bsc_block_bcf[_raw]      bsc_block_records[_raw] with the encoder in the packing's place: out[out_cap] receives the block's BCF bytes,
 *                            *n_bytes their number (BSC_ERR_ARG and the number needed when out_cap is too small), *n_records the records
 *                            in them; one wait per block as before
 */
typedef struct {
  const uint32_t *pos;
  const uint32_t *off;
  const char *bytes;
  uint32_t n;
} bsc_bcf_names;
/* the names of the flagged positions of x0 .. x0 + n - 1 of the loaded contig, as bsc_dbsnp_name returns them (length = *rs_len);
 * pos / off / bytes may be NULL to ask for the sizes: *n_names entries, *n_bytes bytes.  BSC_ERR_ARG if a capacity is too small. */
int bsc_dbsnp_names(const struct bsc_dbsnp *db, uint32_t x0, uint32_t n, uint32_t *pos, uint32_t *off, char *bytes, uint32_t cap_names,
                    uint64_t cap_bytes, uint32_t *n_names, uint64_t *n_bytes);
int bsc_bcf_block_device(bsc_context *ctx, const void *d_recs, const void *d_n_recs, uint64_t max_recs, int32_t rid, const bsc_bcf_ids *ids,
                         const bsc_bcf_names *names, void *d_out, uint64_t out_cap, void *d_totals, void *stream);
int bsc_bcf_sites_device(bsc_context *ctx, const void *d_core, const void *d_aux, uint32_t n, int32_t rid, const bsc_bcf_ids *ids,
                         const bsc_bcf_names *names, void *d_out, uint64_t out_cap, void *d_totals, void *stream);
/* Round 6: the chain kernel leaves every written record's BCF2 LENGTH in a byte per position (0: no record; 255: heterozygous, dbSNP-flagged or
 * longer than 254 bytes — ask the record), so that the encoder's size pass reads a byte per position and the records are read once, by the
 * write kernel (the block entries above do this between themselves).  d_len: y - x + 1 bytes, zeroed by the chain's call. */
int bsc_reads_chain_len_device(bsc_context *ctx, const void *d_tpl, uint32_t nr, const void *d_seq, uint64_t seq_bytes, uint32_t x, uint32_t y,
                               const void *d_ref, const void *d_dbsnp, const bsc_vcf_params *params, int with_stats, void *d_core, void *d_aux,
                               void *d_len, void *stream);
int bsc_bcf_sites_len_device(bsc_context *ctx, const void *d_core, const void *d_aux, const void *d_len, uint32_t n, int32_t rid, const bsc_bcf_ids *ids,
                             const bsc_bcf_names *names, void *d_out, uint64_t out_cap, void *d_totals, void *stream);
int bsc_block_bcf(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x, uint32_t y,
                  const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids,
                  const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap, uint64_t *n_bytes, uint64_t *n_records);
/* the split form (as bsc_block_records_submit / _submit_inplace / _fetch; one block in flight per context): queue and return, then wait.
 * _submit copies the inputs into the context's staging area (the caller's buffers are free at once); _submit_inplace reads them where
 * they lie (page-locked buffers: a true DMA) and they must stay unchanged until the fetch */
int bsc_block_bcf_submit(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x, uint32_t y,
                         const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids,
                         const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap);
int bsc_block_bcf_submit_inplace(bsc_context *ctx, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, uint32_t x,
                                 uint32_t y, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid,
                                 const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap);
int bsc_block_bcf_fetch(bsc_context *ctx, uint64_t *n_bytes, uint64_t *n_records);
/* SEVERAL blocks in one launch sequence, their BCF bytes back as ONE stream (round 6; bsc_blocks_records' block list and joined arrays, see
 * there): the records of the blocks in the blocks' order — what bsc_block_bcf gives block after block, concatenated — for a caller whose blocks
 * are small (the reference's unit is a run of overlapping templates, 10^2 .. 10^7 positions; a launch sequence costs ~0.2 ms whatever its size).
 * The blocks lie on ONE contig (rid) in genome order; `names` lists the flagged positions of all of them.  _submit stages the inputs (the
 * caller's buffers are free on return), _submit_inplace reads them where they lie (page-locked: bsc_alloc_host) until the fetch; one submission in
 * flight per context.  A stream longer than out_cap: BSC_ERR_ARG with *n_bytes = the room needed, then bsc_block_bcf_again. */
int bsc_blocks_bcf_submit(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                          uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid,
                          const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap);
int bsc_blocks_bcf_submit_inplace(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                                  uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats,
                                  int32_t rid, const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap);
int bsc_blocks_bcf_fetch(bsc_context *ctx, uint64_t *n_bytes, uint64_t *n_records);
/* After a BCF block entry (bsc_block_bcf, _raw, _rawdev[_keep], bsc_block_bcf_fetch) has answered BSC_ERR_ARG with *n_bytes > out_cap — the
 * block's stream is longer than the room given — and before anything else is asked of the context: the ENCODER alone once more, from the
 * per-position arrays the block left in HBM, into out[out_cap] (out == NULL: the stream stays on the device, bsc_bcf_stream_read, out_cap
 * its room there).  Nothing of the block is computed, counted (bsc_get_stats, the site statistics, the read profile) or uploaded a second
 * time; the status the refused call would have returned (BSC_OK / BSC_WARN_INEXACT) comes back. */
int bsc_block_bcf_again(bsc_context *ctx, uint8_t *out, uint64_t out_cap, uint64_t *n_bytes, uint64_t *n_records);
/* A stream left on the device (bsc_block_bcf_rawdev_keep) handed over to the caller, so that it can be read out and written WHILE the context
 * calls the next block: _detach (on the thread that drives the context, after the _keep call has returned) gives the buffer and its length
 * and the context forgets it; _read queues a copy of bytes [off, off + n) to dst (page-locked for a true DMA) on a stream of the context's
 * that nothing else uses, _wait waits for the copies queued so far, _free gives the buffer back — to a small pool the next blocks take
 * their buffers from, so a run in its steady state neither allocates nor frees device memory (both are device-wide waits).  _read / _wait /
 * _free may be called from ONE other thread, concurrently with the driving thread's calls; at most four streams out at a time are pooled. */
int bsc_bcf_stream_detach(bsc_context *ctx, void **d_stream, uint64_t *n_bytes);
int bsc_detached_read(bsc_context *ctx, const void *d_stream, uint64_t off, uint64_t n, void *dst);
int bsc_detached_wait(bsc_context *ctx);
int bsc_detached_free(bsc_context *ctx, void *d_stream);
int bsc_block_bcf_raw(bsc_context *ctx, const bsc_raw_template *raw, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, const bsc_misms *misms,
                      uint64_t n_misms, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp,
                      const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out,
                      uint64_t out_cap, uint64_t *n_bytes, uint64_t *n_records, bsc_prep_stats *prep_stats, bsc_read_profile *profile);
/* x of the block a template list starts: the first template's start - 2, at least 1 (src/process_template.c:22-28) */
uint32_t bsc_block_start(const bsc_raw_template *first);
/* get_al_qual (src/al_utils.c:19-35): the score duplicate resolution compares, with the reference's sq[k] indexing */
uint32_t bsc_template_qual(const bsc_raw_template *t, const uint8_t *seq);

/*
 * The reference sequence of a block (host C + zlib; csrc/refseq.c).
 *   bsc_fasta_contig     one contig of a FASTA file (plain / gzip / bgzip, read sequentially) as reference codes 0 = N,
 *                        1..4 = ACGT, position 1 first: load_sequence (src/read_reference.c:44-131).  *len = its length
 *                        (BSC_ERR_ARG with *len set when cap is too small: call again with a larger buffer).
 *   bsc_block_reference  get_sequence_string (src/get_sequence.c:20-54): out[sz] = codes of positions x .. x + sz - 1
 *                        (a block's work->ref1: sz = y - x + 3); positions at or beyond the contig's last one read 0, as
 *                        in the reference (its walk stops in front of end_pos).
 */
int bsc_fasta_contig(const char *path, const char *name, uint8_t *codes, uint64_t cap, uint64_t *len);
int bsc_block_reference(const uint8_t *codes, uint64_t contig_len, uint32_t x, uint32_t sz, uint8_t *out);

/*
 * BAM in, blocks of templates out (host C + zlib, no htslib; csrc/bamio.c): the reader thread of the reference —
 * read_input (src/get_template_vector.c:49-389) over get_next_align_details (src/input_sam.c:222-312).  A block is what
 * read_input queues for process_template_vector: the templates of one stretch of overlapping alignments of one contig
 * (mates joined, duplicates resolved), ready for bsc_prepare_templates -> bsc_accumulate / bsc_block_records, with
 * y = the rightmost covered position (x = bsc_block_start of the first template).
 *   bsc_bam_open[_threads]  a coordinate-sorted BAM or SAM file; the header text and the @SQ list are available at once
 *   bsc_bam_next_block  1 = *blk filled (valid until the next call), 0 = end of input, < 0 = error (bsc_last_error)
 *   bsc_bam_filter_counts  bs_stats.filter_cts / filter_bases as the reader leaves them: reads and bases by verdict,
 *                       gt_filter_reason order, [14] = "PairNotFound" ([0], the passed reads, is counted by
 *                       bsc_prepare_templates: bsc_prep_stats.reads / read_bases)
 * SAM text (plain or BGZF-compressed) is accepted as well: the kind is found out from the first bytes.
 * Not covered: CRAM input, seeking through a .bai index (a region is served by scanning), contig include / exclude lists.
 */
typedef struct bsc_bam bsc_bam;
typedef struct {
  uint32_t mapq_thresh;       /* sr_param.mapq_thresh (20) */
  uint64_t max_template_len;  /* sr_param.max_template_len (1000) */
  int32_t keep_unmatched, ignore_duplicates, keep_duplicates; /* -u / -d / -k */
  /* One region (the reference's -r contig:start-stop, region_t): only the alignments that overlap positions
   * region_start .. region_stop (1-based, inclusive) of contig region_tid reach the reader, as with the reference's index
   * query sam_itr_queryi(idx, tid, start - 1, stop) (src/get_template_vector.c:69-74) — found here by scanning, no index.
   * region_stop = 0: no region.  Pass the same bounds to the record formation (bsc_vcf_params.reg_start / reg_stop). */
  int32_t region_tid;
  uint32_t region_start, region_stop;
} bsc_reader_params;
typedef struct {
  int32_t tid;  /* index of the contig in the BAM header */
  uint32_t y;   /* rightmost position covered by the block's alignments */
  uint32_t nr;
  const bsc_raw_template *tpl;
  const uint8_t *seq;
  uint64_t seq_bytes;
  const bsc_misms *misms;
  uint64_t n_misms;
} bsc_read_block;
int bsc_bam_open(const char *path, bsc_bam **out); /* helper threads from the environment: BSC_BAM_THREADS (default 0) */
/* n_threads helpers (0 .. 16) inflate the BGZF blocks ahead of the parser: blocks are independent gzip members */
int bsc_bam_open_threads(const char *path, int n_threads, bsc_bam **out);
void bsc_bam_close(bsc_bam *b);
int bsc_bam_n_refs(const bsc_bam *b);
const char *bsc_bam_ref_name(const bsc_bam *b, int i);
uint32_t bsc_bam_ref_len(const bsc_bam *b, int i);
const char *bsc_bam_header_text(const bsc_bam *b);
int bsc_bam_next_block(bsc_bam *b, const bsc_reader_params *par, bsc_read_block *blk);
void bsc_bam_filter_counts(const bsc_bam *b, uint64_t cts[15], uint64_t bases[15]);
/* BAM records the reader dropped because their CIGAR does not cover l_seq query bases (htslib hands such a record over as it
 * is and the reference walks outside the read; SAM text with the same defect is an error, as in htslib's parser): not part of
 * the reference's counters — a caller should tell its user when this is not 0 (integration/bam2bcf.c does) */
uint64_t bsc_bam_malformed(const bsc_bam *b);

/*
 * The reader ON THE DEVICE (round 6; csrc/bamstream.c + csrc/bamdev.hip): the same blocks as bsc_bam_next_block, formed in HBM from the
 * inflated bytes of a BAM file — what the reference's reader thread does with a record (get_next_align_details, src/input_sam.c:222-312;
 * read_input, src/get_template_vector.c:49-389) runs as kernels, the host only inflates.
 *
 *   bsc_bamstream_*        the host half, usable on its own: the file as a stream of INFLATED bytes in page-locked slabs, with the offset
 *                          of every alignment record (hts_open + bgzf_read's inflate + the block_size chain of bam_read1).  n_threads
 *                          helpers (<= 0: one per core this process may run on, at most 64) inflate straight into the slabs;
 *                          slab_bytes / n_slabs 0 = 16 MiB x 6.  bsc_bamstream_next: 1 = *out filled (valid until it is released),
 *                          0 = end of the stream, < 0 = error.  BAM only.
 *   bsc_bamdev_open        a reader over `path` bound to ctx's device and stream
 *   bsc_bamdev_next_block  1 = *blk describes the next block, its templates / reads / lists DEVICE-resident (valid until the next call)
 *                          and laid out as bsc_prepare_templates_device / bsc_block_bcf_rawdev take them; 0 = end of input; < 0 = error.
 *                          Byte for byte the templates, reads, lists, y and filter counters of bsc_bam_next_block (tests/test_gpu_bamdev.py),
 *                          read offsets apart (a template's reads lie where their records came, block-relative).  Input the parallel
 *                          kernels are not exact for (re-used read names, unsorted records ...) goes through a one-lane replay of the
 *                          reference's loop on the device: slow, same bytes.  Divergences from csrc/bamio.c: a record of one base that a
 *                          duplicate comparison looks at reads quality 0 for the byte behind it (bamio.c reads its buffer's next byte).
 *   bsc_bamdev_fetch_block the block's arrays on the host (bsc_read_block's view)
 *   bsc_block_bcf_rawdev / bsc_block_records_rawdev    bsc_block_bcf_raw / bsc_block_records_raw from device-resident raw templates
 *                          (d_raw 8-byte, d_misms 4-byte aligned; ins_pad >= the sizes of all BSC_MISMS_INS entries: room for the
 *                          padded deletions).  ref / dbsnp / names are host arrays as before.
 */
/* csrc/inflate_fast.c: raw DEFLATE of one whole block (in -> exactly out_len bytes; 0, or -1 for an invalid stream) and zlib's CRC-32 */
int bsc_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);
uint32_t bsc_crc32(const uint8_t *p, size_t n);
typedef struct bsc_bamstream bsc_bamstream;
typedef struct {
  const uint8_t *bytes;     /* page-locked */
  uint64_t stream_off;      /* offset of bytes[0] in the inflated file */
  uint32_t n_bytes;
  const uint32_t *rec_off;  /* page-locked: where the records that START in this slab start, relative to bytes */
  uint32_t n_recs;
  int32_t last;             /* the stream ends with this slab */
  uint64_t seq;
} bsc_bam_slab;
int bsc_bamstream_open(const char *path, int n_threads, uint64_t slab_bytes, int n_slabs, bsc_bamstream **out);
/* the stream of a SELECTION of contigs (tids of the header's list; -1 = the unplaced reads at the file's end) — what the reference reads through
 * sam_index_load + sam_itr_queryi per region (src/process.c:125, src/get_template_vector.c:69-99), without an index file: the file is sorted, so a
 * binary search over its BGZF blocks (one block inflated per probe) finds the stretches that hold the selected contigs' records; records of other
 * contigs inside a stretch are for the record parser's contig filter (bsc_bamdev_open_contigs applies it).  Blocks need not start at a record
 * (htsjdk's writer cuts records where a block is full): the first record start of a probed block is found by a chain of checked record headers, a
 * stretch begins there and ends with the tail of its last record in the following block(s).  n_tids = 0: an empty stream. */
int bsc_bamstream_open_contigs(const char *path, int n_threads, uint64_t slab_bytes, int n_slabs, const int32_t *tids, int n_tids, bsc_bamstream **out);
void bsc_bamstream_close(bsc_bamstream *b);
int bsc_bamstream_next(bsc_bamstream *b, bsc_bam_slab *out);
int bsc_bamstream_release(bsc_bamstream *b, const bsc_bam_slab *slab);
int bsc_bamstream_n_refs(const bsc_bamstream *b);
const char *bsc_bamstream_ref_name(const bsc_bamstream *b, int i);
uint32_t bsc_bamstream_ref_len(const bsc_bamstream *b, int i);
const char *bsc_bamstream_header_text(const bsc_bamstream *b);
uint64_t bsc_bamstream_first_record(const bsc_bamstream *b); /* stream offset of the first alignment record */
int bsc_bamstream_threads(const bsc_bamstream *b);
int bsc_bamstream_default_threads(void);

typedef struct bsc_bamdev bsc_bamdev;
typedef struct {
  int32_t tid;
  uint32_t x, y;           /* the block as the process thread sees it: x = bsc_block_start of the first template */
  uint32_t nr;
  const void *d_tpl;       /* bsc_raw_template[nr] */
  const void *d_seq;
  uint64_t seq_bytes;
  const void *d_misms;     /* bsc_misms[n_misms] */
  uint64_t n_misms;
  uint64_t ins_pad;
} bsc_dev_read_block;
int bsc_bamdev_open(bsc_context *ctx, const char *path, int n_threads, bsc_bamdev **out);
/* the reader of a SELECTION of contigs (bsc_bamstream_open_contigs + the record parser's contig filter): what one rank of a sharded run reads —
 * its blocks and filter counters are exactly the whole file's reader's for those contigs, so the ranks' outputs concatenate and their counters add */
int bsc_bamdev_open_contigs(bsc_context *ctx, const char *path, int n_threads, const int32_t *tids, int n_tids, bsc_bamdev **out);
void bsc_bamdev_close(bsc_bamdev *r);
int bsc_bamdev_n_refs(const bsc_bamdev *r);
const char *bsc_bamdev_ref_name(const bsc_bamdev *r, int i);
uint32_t bsc_bamdev_ref_len(const bsc_bamdev *r, int i);
const char *bsc_bamdev_header_text(const bsc_bamdev *r);
int bsc_bamdev_next_block(bsc_bamdev *r, const bsc_reader_params *par, bsc_dev_read_block *blk);
int bsc_bamdev_fetch_block(bsc_bamdev *r, const bsc_dev_read_block *blk, bsc_raw_template *tpl, uint8_t *seq, bsc_misms *misms);
int bsc_bamdev_filter_counts(bsc_bamdev *r, uint64_t cts[15], uint64_t bases[15]);
uint64_t bsc_bamdev_malformed(bsc_bamdev *r);
/* counts: {device passes, passes the one-lane replay decided, records parsed, bytes uploaded}; seconds: {waiting for inflated slabs, in device passes} */
void bsc_bamdev_run_stats(const bsc_bamdev *r, uint64_t counts[4], double seconds[2]);
int bsc_block_records_rawdev(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes, const void *d_misms, uint64_t n_misms,
                             uint64_t ins_pad, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp,
                             const bsc_vcf_params *params, int with_stats, bsc_vcf_rec *out, uint64_t out_cap, uint64_t *n_out, bsc_prep_stats *prep_stats,
                             bsc_read_profile *profile);
int bsc_block_bcf_rawdev(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes, const void *d_misms, uint64_t n_misms,
                         uint64_t ins_pad, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp,
                         const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out,
                         uint64_t out_cap, uint64_t *n_bytes, uint64_t *n_records, bsc_prep_stats *prep_stats, bsc_read_profile *profile);
/* bsc_block_bcf_rawdev with the stream left on the device (room: dev_cap bytes; BSC_ERR_ARG and the length needed in *n_bytes when it does
 * not fit), and bytes [off, off + n) of that stream -> dst, queued on the context's stream (bsc_synchronize before dst is read): a contig-sized
 * block's stream is gigabytes — read in pieces through a small page-locked buffer it costs no page-locking of its own */
int bsc_block_bcf_rawdev_keep(bsc_context *ctx, const void *d_raw, uint32_t nr, const void *d_seq, uint64_t seq_bytes, const void *d_misms, uint64_t n_misms,
                              uint64_t ins_pad, const bsc_prep_params *prep, uint32_t x, uint32_t y, const uint8_t *ref, const uint8_t *dbsnp,
                              const bsc_vcf_params *params, int with_stats, int32_t rid, const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint64_t dev_cap,
                              uint64_t *n_bytes, uint64_t *n_records, bsc_prep_stats *prep_stats, bsc_read_profile *profile);
int bsc_bcf_stream_read(bsc_context *ctx, uint64_t off, uint64_t n, void *dst);
/* with bsc_set_profiling: device time (HIP events on the context's stream) of the most recent raw block — bsc_block_records_raw[dev],
 * bsc_block_bcf_raw[dev][_keep] — from its first pre-processing launch to the last launch it queued, the host's wait for the prepared size included */
int bsc_last_raw_block_ms(bsc_context *ctx, float *ms);

/*
 * dbSNP index (host C + zlib; csrc/dbsnp.c): the reader of the compressed index bin/dbSNP_idx writes.  In the reference
 * the index never touches the likelihoods: an entry names the record (VCF ID), forces the AA / TT homozygous-reference
 * record of a site flagged in its `fq_mask` to be written (rs_found & 2, src/print_vcf.c:139) and feeds the dbSNP
 * counters of the statistics (:426-441).
 *   bsc_dbsnp_open          load_dbSNP_header   src/dbSNP.c:27-141
 *   bsc_dbsnp_load_contig   load_dbSNP_ctg      src/dbSNP.c:157-304 (the previous contig is dropped, as print_vcf_entry
 *                                               does at a contig change, src/print_vcf.c:553-562); a contig the index
 *                                               does not list loads as "nothing flagged"; *n_snps = entries loaded
 *   bsc_dbsnp_flags         rs_found (0 / 1 / 3) of positions x0 .. x0 + n - 1 (1-based) of the loaded contig: the
 *                           `dbsnp` array of bsc_chain_device / bsc_block_records / bsc_vcf_records
 *   bsc_dbsnp_name          dbSNP_lookup_name   src/dbSNP.c:306-350: returns rs_found, the name in rs (NUL-terminated) and
 *                           in *rs_len the length the reference hands to htslib (an odd number of digits counts its
 *                           filler byte); negative on error
 */
typedef struct bsc_dbsnp bsc_dbsnp;
int bsc_dbsnp_open(const char *path, bsc_dbsnp **out);
void bsc_dbsnp_close(bsc_dbsnp *db);
int bsc_dbsnp_n_contigs(const bsc_dbsnp *db);
const char *bsc_dbsnp_contig_name(const bsc_dbsnp *db, int i);
const char *bsc_dbsnp_header(const bsc_dbsnp *db);
int bsc_dbsnp_load_contig(bsc_dbsnp *db, const char *name, uint64_t *n_snps);
int bsc_dbsnp_flags(const bsc_dbsnp *db, uint32_t x0, uint32_t n, uint8_t *out);
int bsc_dbsnp_name(const bsc_dbsnp *db, uint32_t x, char *rs, size_t cap, size_t *rs_len);

/* Host-side text rendering of one record as a VCF data line ("CHROM POS ID REF ALT QUAL FILTER INFO FORMAT SAMPLE",
 * tab separated, no newline): the field layout of the record the reference hands to htslib (src/print_vcf.c:160-380).
 * Returns the length written, 0 when c->emit == 0, -1 when buf is too small.  `id` NULL/"" prints ".". */
int bsc_vcf_format(const bsc_vcf_core *c, const bsc_gt_meth *g, const char *contig, const char *id, char *buf,
                   size_t cap);

/* Per-launch kernel timing with HIP events recorded on the launch stream (measurement support):
 * after bsc_set_profiling(ctx, 1), bsc_last_kernel_ms() returns the device time of the most recent
 * calling kernel and of its Fisher pass (it waits for that launch to finish).  With n > 2^31 sites per
 * call only the last sub-launch is reported. */
int bsc_set_profiling(bsc_context *ctx, int enable);
int bsc_last_kernel_ms(bsc_context *ctx, float *call_ms, float *fisher_ms);
/* The same for the launch `age` launches back (0 = the most recent; the last 32 are kept), so that a timed loop can run
 * without waiting on events and read its launches' device times afterwards. */
int bsc_kernel_ms_history(bsc_context *ctx, uint32_t age, float *call_ms, float *fisher_ms);

/* Measurement support: the time (ms, HIP events on `stream`, average of `reps` launches queued back to back behind three untimed ones) of a kernel that moves exactly
 * the bytes of bsc_call_sites_device(ctx, d_cts, d_ref, n, d_out, 200, d_skip) — 104 + 1 in, 200 + 1 out per position,
 * same tile shape, no arithmetic: the practical memory ceiling for that call.  OVERWRITES d_out / d_skip with junk;
 * n is rounded down to a multiple of 64. */
int bsc_stream_probe_ms(bsc_context *ctx, const void *d_cts, const void *d_ref, uint64_t n, void *d_out, void *d_skip,
                        int reps, void *stream, float *ms);

/* Blocks until everything queued on the context's own stream (the host-buffer entries) has finished. */
int bsc_synchronize(bsc_context *ctx);

/* Counters (device -> host; synchronises the context's stream). */
int bsc_get_stats(bsc_context *ctx, bsc_stats *out);
int bsc_reset_stats(bsc_context *ctx);

/*
 * Synthetic 'L-pileup' generator: fills d_cts[n] and d_ref[n] on
 * the device for absolute site indices first_site .. first_site+n-1 of a synthetic contig.  Deterministic
 * in (seed, site index) only, so any window can be regenerated independently.  Bench/test support.
 */
#define BSC_SYNTH_NRUNS 1u /* flags: 1 % of 10-kb runs are N (reference code 0) with no reads */
int bsc_synth_pileup_device(bsc_context *ctx, uint64_t seed, uint64_t first_site, uint64_t n, uint32_t coverage,
                            uint32_t flags, void *d_cts, void *d_ref, void *stream);
/* Host generator of synthetic read pairs over positions x .. x+n_sites-1 ('L-reads'; see
 * bs_call_amd/csrc/synth_reads.c): fills tpl[] (in the order of the pairs' start positions — a template that lost its
 * forward read is therefore out of leftmost-position order, as in real align_lists) and seq[]; returns the number
 * of templates, or -1 if a buffer is too small.  Reference bases come from the same synthetic genome as the
 * L-pileup generator (site index = position). */
int64_t bsc_synth_reads_host(uint64_t seed, uint32_t x, uint32_t n_sites, uint32_t coverage, uint32_t flags,
                             bsc_template *tpl, uint64_t max_templates, uint8_t *seq, uint64_t seq_cap,
                             uint64_t *seq_used);
/* Host twin of the generator (same bits), for feeding the same inputs to a CPU checker. */
int bsc_synth_pileup_host(uint64_t seed, uint64_t first_site, uint64_t n, uint32_t coverage, uint32_t flags,
                          bsc_pileup *cts, uint8_t *ref);

#ifdef __cplusplus
}
#endif
#endif /* BSCALL_AMD_H */
