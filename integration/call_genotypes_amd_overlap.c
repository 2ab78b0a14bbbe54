/*
 * call_genotypes_amd_overlap.c — the OVERLAPPED drop-in replacement for the reference's src/call_genotypes.c: same
 * three exported symbols (include/bs_call.h:358-360) and the same publish protocol towards the print thread as
 * call_genotypes_amd.c, but the call returns as soon as the block is queued on the GPU, as the original returns as soon
 * as its calc threads are dispatched (src/call_genotypes.c:260-272):
 *
 *   call k     the block is flattened and appended, with a private copy of its reference codes (the caller overwrites
 *              work->ref1 for the next block, src/process_template.c:29-30), to the BATCH being filled; the meth profiling
 *              thread is waited for; if the batch now holds AMD_BATCH_POSITIONS positions (1 M) it is FLUSHED:
 *   flush      the batch submitted before (complete by now) is fetched — bsc_block_fetch waits for its kernels and for its
 *              images, which bsc_blocks_submit_to had queued straight into a page-locked gt_vcf[] array — and PUBLISHED
 *              block by block, in order: the print thread is waited for (vcf_n == 0, print_cond2), work->vcf / vcf_x /
 *              vcf_ctg are pointed at the block's images, work->ref receives its reference codes, the `ready` flags are
 *              set, the print thread is woken (original :228-258 and :110-114); then the batch being filled is submitted
 *              as ONE launch sequence into the OTHER gt_vcf[] array.
 *   join       what is still held back is submitted, everything published, then the context is destroyed.
 * So the GPU computes batch k and copies it out while the process thread prepares batch k+1 and the print thread writes
 * batch k-1 — the overlap the original has between its calc threads and its process thread — and a block
 * of a few thousand positions (the reference's common case, src/get_template_vector.c:141-147) no longer pays a launch
 * sequence of its own: held back, 10 000-position blocks go through the library at over 400 M positions/s instead of 43 M
 * (profiles/r04_small_blocks.txt).  work->vcf alternates between the two arrays; an array is written again two flushes
 * after it was published, after the print thread has been waited for in between.
 *
 * The thread protocol itself lives in integration/amd_overlap_protocol.h, shared with integration/demo_block.c, which runs
 * the very same code against a mock of the work_t fields it touches — on the GPU, and as a CPU-only build with stub
 * bsc_* entries under ThreadSanitizer (tests/test_glue_tsan.py).  Round 3: the meth profiling thread is waited for in EVERY
 * call, block pending or not, as the original does (src/call_genotypes.c:244-251) — round 2's version skipped the wait in
 * the first call, after which process_template_vector overwrites work->ref1 under the profiler (src/meth_profile.c:51).
 *
 * Like call_genotypes_amd.c this file is compiled INSIDE the bs_call source tree (it includes the reference's own
 * headers); here it is only syntax-checked (`make glue-check`).
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gem_tools.h"
#include "bs_call.h"

#define AMD_WORK_T work_t
#define AMD_GT_VCF_T gt_vcf
#define AMD_CTG_T ctg_t
#define AMD_SET_REF(work, src, sz)                                            \
  do {                                                                        \
    gt_string_resize((work)->ref, (sz) + 3);                                  \
    memcpy(gt_string_get_string((work)->ref), (src), (size_t)(sz) + 3);       \
    gt_string_set_length((work)->ref, (sz) + 2);                              \
  } while (0)
#define AMD_REF1(work) ((const char *)gt_string_get_string((work)->ref1))
#include "amd_overlap_protocol.h"

static bsc_template *amd_tpl;
static uint8_t *amd_seq;
static size_t amd_tpl_cap, amd_seq_cap;
static gt_vcf *amd_user_vcf; /* what work->vcf held before the first call (restored at join, it is the caller's to free) */
static int amd_user_vcf_size;

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
  if (bsc_create(&p, &amd_ctx) != BSC_OK) amd_die("bsc_create");
  amd_overlap_init();
  work->calc_end = false;
  work->n_calc_threads = 1;
  work->calc_threads_complete = 1; /* the next call never waits on calc_cond2: it fetches instead */
  work->calc_threads = NULL;
  amd_user_vcf = work->vcf;
  amd_user_vcf_size = work->vcf_size;
}

void join_calc_threads(sr_param *const param) {
  work_t *const work = &param->work;
  amd_overlap_join(work); /* the last block: published, drained by the print thread, the arrays freed */
  work->calc_end = true;
  work->vcf = amd_user_vcf;
  work->vcf_size = amd_user_vcf_size;
  bsc_destroy(amd_ctx);
  amd_ctx = NULL;
  pthread_mutex_lock(&work->vcf_mutex); /* original :150-152 */
  pthread_cond_signal(&work->vcf_cond);
  pthread_mutex_unlock(&work->vcf_mutex);
}

void call_genotypes_ML(ctg_t *const ctg, gt_vector *const align_list, const uint32_t x, const uint32_t y,
                       sr_param *const param) {
  work_t *const work = &param->work;
  assert(y >= x);
  const uint32_t nr = gt_vector_get_used(align_list);
  /* flatten align_details -> bsc_template + one read buffer (the library copies them into its own staging area, so the
   * reader may recycle the align_list as soon as the call returns) */
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
    /* was read 0 "walked" (:198-211 of the original)?  Its first bytes are in the cache right now */
    t->flags = bsc_template_walk_flags(amd_seq + t->off[0], t->len[0]);
  }
  /* previous block published, this one submitted, the meth profiling thread waited for (amd_overlap_protocol.h) */
  amd_overlap_call(work, ctg, amd_tpl, nr, amd_seq, nbytes, x, y);
}
