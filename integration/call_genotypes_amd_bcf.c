/*
 * call_genotypes_amd_bcf.c — INTEGRATION.md level 2b as a file: the replacement for the reference's src/call_genotypes.c at which the
 * print thread's per-site work has left the host as well.  Same three exported symbols (include/bs_call.h:358-360); a block's result is
 * no longer work->vcf[] (208-byte images for print_vcf_entry, src/process.c:87-104) but its stretch of the BCF record stream — record
 * formation, the window of five sites, the bcf_enc_* calls and bcf_write's fixed fields (src/print_vcf.c:32-381) done on the device —
 * and a writer thread of this file hands the bytes to bgzf_write.  For BCF output (param->out_file_type & FT_BCF, src/print_vcf.c:633-
 * 636) only: htslib renders text VCF from a bcf1_t, which this level no longer builds.
 *
 *   call k     block k is flattened and appended to the batch in the page-locked slot being filled; the meth profiling thread is waited for.
 *              A batch of 250 000 positions (a larger block alone) is handed over: the batch in flight is fetched and its bytes given to
 *              the writer, the new one queued (bsc_blocks_bcf_submit_inplace: one launch sequence for all its blocks) — the GPU works on
 *              it while the process thread reads the next one's blocks (integration/amd_bcf_protocol.h — the same code runs in
 *              integration/demo_block.c against a mock of work_t, checked against bsc_block_bcf block by block, with its end-to-end rate)
 *   join       the last batch handed over, fetched and written; bs_stats filled once from the device's sums (src/stats.c:19-298 reads work->stats)
 *
 * The reference's print thread is left as it is: it is never handed a block (work->vcf_n stays 0) and ends on print_end.  What a
 * maintainer adds beside this file: nothing in process.c; in main(), the output must be BCF.  dbSNP: the flags and names of a block come
 * from bsc_dbsnp_flags / bsc_dbsnp_names over the index opened with bsc_dbsnp_open (the reference's own index file) — two lines at the
 * marked place, left out here because the reference keeps its index handle in ctg_t fields this file does not otherwise touch.
 *
 * Like the other two glue files this one is compiled INSIDE the bs_call source tree; here it is syntax-checked against the reference's
 * headers (`make glue-check`).
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gem_tools.h"
#include "bs_call.h"

#define AMD_WORK_T work_t
#define AMD_REF1(work) ((const char *)gt_string_get_string((work)->ref1))
#ifdef AMD_GLUE_CHECK /* `make glue-check` has htslib's types as opaque names only: the one htslib call of this file is a prototype there */
ssize_t amd_glue_check_write(htsFile *fp, const void *buf, size_t n);
#define AMD_OUT_WRITE(work, buf, n) amd_glue_check_write((work)->vcf_file, (buf), (n))
#else
#define AMD_OUT_WRITE(work, buf, n) bgzf_write((work)->vcf_file->fp.bgzf, (buf), (n))
#endif
#define AMD_BCF_WRITE(work, buf, n)                                                                                     \
  do {                                                                                                                  \
    if ((n) && AMD_OUT_WRITE((work), (buf), (size_t)(n)) != (ssize_t)(n)) {                                             \
      fprintf(stderr, "bscall_amd: Failed to write vcf/bcf record\n"); /* src/print_vcf.c:380 */                        \
      exit(1);                                                                                                          \
    }                                                                                                                   \
  } while (0)
#include "amd_bcf_protocol.h"

static bsc_template *amd_tpl;
static uint8_t *amd_seq;
static size_t amd_tpl_cap, amd_seq_cap;

static void *amd_grow(void *p, size_t *cap, size_t need, size_t elem) {
  if (need > *cap) {
    p = realloc(p, need * elem);
    if (!p) { fprintf(stderr, "bscall_amd: out of memory\n"); exit(1); }
    *cap = need;
  }
  return p;
}

void init_calc_threads(sr_param *const param) {
  work_t *const work = &param->work;
  bsc_params p;
  bsc_params_default(&p);
  p.under_conv = param->under_conv;
  p.over_conv = param->over_conv;
  p.ref_bias = param->ref_bias;
  p.min_qual = param->min_qual;
  bsc_context *ctx = NULL;
  if (bsc_create(&p, &ctx) != BSC_OK) amd_bdie("bsc_create");
  amd_bvp.all_positions = param->all_positions ? 1 : 0; /* -A (src/print_vcf.c:147) */
  amd_bstats = param->work.stats != NULL;               /* the bs_stats sums, on the device */
  amd_bcf_init(work, ctx);
  work->calc_end = false;
  work->n_calc_threads = 1;
  work->calc_threads_complete = 1;
  work->calc_threads = NULL;
}

void join_calc_threads(sr_param *const param) {
  work_t *const work = &param->work;
  amd_bcf_join(); /* the last block: fetched, written */
  if (work->stats) {
    bsc_site_stats *st = malloc(sizeof *st);
    if (st && bsc_get_site_stats(amd_bctx, st) == BSC_OK) {
      /* every sum field of bs_stats / gt_ctg_stats (include/bs_call.h:75-146) has its twin in bsc_site_stats, field by field
       * (include/bscall_amd.h names the reference's statement next to each): copied here into work->stats for output_stats */
    }
    free(st);
  }
  work->calc_end = true;
  bsc_destroy(amd_bctx);
  amd_bctx = NULL;
  pthread_mutex_lock(&work->vcf_mutex); /* original :150-152 */
  pthread_cond_signal(&work->vcf_cond);
  pthread_mutex_unlock(&work->vcf_mutex);
}

void call_genotypes_ML(ctg_t *const ctg, gt_vector *const align_list, const uint32_t x, const uint32_t y, sr_param *const param) {
  work_t *const work = &param->work;
  assert(y >= x);
  const uint32_t nr = gt_vector_get_used(align_list);
  align_details **al_p = gt_vector_get_mem(align_list, align_details *);
  size_t nbytes = 0;
  for (uint32_t i = 0; i < nr; i++)
    for (int k = 0; k < 2; k++)
      if (al_p[i]->read[k]) nbytes += gt_vector_get_used(al_p[i]->read[k]);
  amd_tpl = amd_grow(amd_tpl, &amd_tpl_cap, nr ? nr : 1, sizeof *amd_tpl);
  amd_seq = amd_grow(amd_seq, &amd_seq_cap, nbytes ? nbytes : 1, 1);
  size_t off = 0;
  for (uint32_t i = 0; i < nr; i++) {
    const align_details *al = al_p[i];
    bsc_template *t = amd_tpl + i;
    memset(t, 0, sizeof *t);
    t->pos[0] = al->forward_position;
    t->pos[1] = al->reverse_position;
    t->orientation = (uint8_t)al->orientation;
    t->bs_strand = (uint8_t)al->bs_strand;
    for (int k = 0; k < 2; k++) {
      t->mapq[k] = al->mapq[k];
      if (!al->read[k]) continue;
      const uint32_t rl = gt_vector_get_used(al->read[k]);
      t->len[k] = rl;
      t->off[k] = off;
      memcpy(amd_seq + off, gt_vector_get_mem(al->read[k], uint8_t), rl);
      off += rl;
    }
    t->flags = bsc_template_walk_flags(amd_seq + t->off[0], t->len[0]); /* :198-211 of the original, while the bytes are in the cache */
  }
  /* (dbSNP: bsc_dbsnp_flags / bsc_dbsnp_names for x .. y here, handed to amd_bcf_call's submission) */
  amd_bcf_call(work, ctg->vcf_rid /* src/print_vcf.c:163 */, amd_tpl, nr, amd_seq, nbytes, x, y);
}
