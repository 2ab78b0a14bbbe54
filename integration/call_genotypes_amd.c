/*
 * call_genotypes_amd.c — drop-in replacement for the reference's src/call_genotypes.c, to be compiled INSIDE the
 * bs_call source tree (it includes the reference's own bs_call.h / gem_tools.h, which need htslib's headers) and
 * linked with -lbscall_amd.  It is NOT built in this repository (htslib is not in this image); INTEGRATION.md
 * explains the build line.  It exports the three symbols the rest of bs_call calls
 * (include/bs_call.h:358-360) and keeps the publish protocol towards the print thread
 * (src/call_genotypes.c:228-258 and :110-118 in the original; src/process.c:74-110 is the consumer).
 *
 * Differences from the original, by design:
 *   - one block is computed by bsc_call_block() (accumulate + call on the MI355X) instead of a pthread pool;
 *   - the block is published to the print thread at once (all `ready` flags, one signal), which the print thread's
 *     in-order spin/wait loop accepts unchanged;
 *   - the call is synchronous: call_genotypes_ML returns when the block is published, so calc_threads_complete is
 *     always "complete" and the next call never waits on calc_cond2.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gem_tools.h"
#include "bs_call.h"

#include <bscall_amd.h>

static bsc_context *amd_ctx;
static bsc_template *amd_tpl;
static uint8_t *amd_seq, *amd_skip;
static size_t amd_tpl_cap, amd_seq_cap, amd_skip_cap;

static void amd_die(const char *what) {
  fprintf(stderr, "bscall_amd: %s: %s\n", what, bsc_last_error());
  exit(1); /* the code base's convention for fatal errors (gt_fatal_error_msg) */
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
  work->calc_end = false;
  work->n_calc_threads = 1;
  work->calc_threads_complete = 1; /* nothing in flight */
  work->calc_threads = NULL;
}

void join_calc_threads(sr_param *const param) {
  work_t *const work = &param->work;
  work->calc_end = true;
  bsc_destroy(amd_ctx);
  amd_ctx = NULL;
  pthread_mutex_lock(&work->vcf_mutex); /* original :150-152 */
  pthread_cond_signal(&work->vcf_cond);
  pthread_mutex_unlock(&work->vcf_mutex);
}

static void *amd_grow(void *p, size_t *cap, size_t need, size_t elem) {
  if (need > *cap) {
    p = realloc(p, need * elem);
    if (!p) { fprintf(stderr, "bscall_amd: out of memory\n"); exit(1); }
    *cap = need;
  }
  return p;
}

void call_genotypes_ML(ctg_t *const ctg, gt_vector *const align_list, const uint32_t x, const uint32_t y,
                       sr_param *const param) {
  work_t *const work = &param->work;
  assert(y >= x);
  const uint32_t sz = y - x + 1;
  const uint32_t nr = gt_vector_get_used(align_list);

  /* flatten align_details -> bsc_template + one read buffer (borrowed data: valid until the next hand-off) */
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

  /* wait for the print thread to have drained the previous block, then own work->vcf (original :228-242) */
  pthread_mutex_lock(&work->print_mutex);
  while (param->work.vcf_n) {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += 5;
    pthread_cond_timedwait(&work->print_cond2, &work->print_mutex, &ts);
  }
  pthread_mutex_unlock(&work->print_mutex);
  if ((int)sz > work->vcf_size) {
    work->vcf = realloc(work->vcf, sizeof(gt_vcf) * sz);
    work->vcf_size = sz;
  }
  amd_skip = amd_grow(amd_skip, &amd_skip_cap, sz, 1);

  /* the block: reads -> pile-up -> gt_meth, written as gt_vcf images (stride 208: gtm, ready = 0, skip).
   * work->ref1 holds the reference codes of x .. y+2 (src/process_template.c:29-30). */
  const uint8_t *ref_codes = (const uint8_t *)gt_string_get_string(work->ref1);
  int rc = bsc_call_block(amd_ctx, amd_tpl, nr, amd_seq, nbytes, x, y, ref_codes, work->vcf, (uint32_t)sizeof(gt_vcf),
                          amd_skip);
  if (rc < 0) amd_die("bsc_call_block"); /* BSC_ERR_ARG here = one of the original's asserts (:158,186,188) */
  if (rc == BSC_WARN_INEXACT) fprintf(stderr, "bscall_amd: %s\n", bsc_last_error());

  work->vcf_x = x;
  work->vcf_ctg = ctg;
  /* meth profiling must be finished before the reference strings are swapped (original :244-254) */
  pthread_mutex_lock(&work->mprof_mutex);
  while (work->mprof_read_idx != work->mprof_write_idx) {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += 5;
    pthread_cond_timedwait(&work->mprof_cond2, &work->mprof_mutex, &ts);
  }
  pthread_mutex_unlock(&work->mprof_mutex);
  gt_string *tp = work->ref;
  work->ref = work->ref1;
  work->ref1 = tp;

  /* publish the whole block: records first, then the flags, then wake the print thread (original :110-114,255-258) */
  pthread_mutex_lock(&work->vcf_mutex);
  for (uint32_t i = 0; i < sz; i++) work->vcf[i].ready = true;
  pthread_mutex_unlock(&work->vcf_mutex);
  work->vcf_n = sz;
  pthread_mutex_lock(&work->print_mutex);
  pthread_cond_signal(&work->print_cond1);
  pthread_mutex_unlock(&work->print_mutex);
  pthread_mutex_lock(&work->vcf_mutex);
  pthread_cond_signal(&work->vcf_cond);
  pthread_mutex_unlock(&work->vcf_mutex);
}
