/*
 * call_genotypes_amd_overlap.c — the OVERLAPPED drop-in replacement for the reference's src/call_genotypes.c: same
 * three exported symbols (include/bs_call.h:358-360) and the same publish protocol towards the print thread as
 * call_genotypes_amd.c, but the call returns as soon as the block is queued on the GPU, as the original returns as soon
 * as its calc threads are dispatched (src/call_genotypes.c:260-272):
 *
 *   call k     1. block k-1 (if any) is fetched — bsc_block_fetch waits for its kernels and for its records, which
 *                 bsc_block_submit_to had queued straight into a page-locked gt_vcf[] array — and PUBLISHED: the print
 *                 thread is waited for (vcf_n == 0, print_cond2), work->vcf / vcf_x / vcf_ctg are pointed at block k-1,
 *                 the mprof thread is waited for, work->ref receives block k-1's reference codes, the `ready` flags are
 *                 set, the print thread is woken (original :228-258 and :110-114);
 *              2. block k is flattened, its reference codes are copied (the caller overwrites work->ref1 for the next
 *                 block, src/process_template.c:29-30), and it is submitted into the OTHER gt_vcf[] array.
 *   join       the last block is fetched and published, then the context is destroyed.
 * So the GPU computes block k and copies it out while the process thread prepares block k+1 and the print thread
 * writes block k-1 — the overlap the original has between its calc threads and its process thread,
 * which the synchronous glue gives up.  work->vcf alternates between the two arrays; a block's array is reused two
 * calls later, after the print thread has been waited for twice.
 *
 * Like call_genotypes_amd.c this file is compiled INSIDE the bs_call source tree (it includes the reference's own
 * headers); here it is only syntax-checked (`make glue-check`) and its protocol is exercised by integration/demo_block.c
 * against a mock of the work_t fields it touches.
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
static uint8_t *amd_seq;
static size_t amd_tpl_cap, amd_seq_cap;
/* the two blocks that can be alive at once: one being printed, one being computed */
static struct amd_slot {
  gt_vcf *vcf;     /* page-locked (bsc_alloc_host): the copy-out is a true DMA behind the kernels */
  uint8_t *skip;   /* bsc_block_submit_to's skip array (the gt_vcf images carry the flag too) */
  char *ref;       /* private copy of the block's reference codes, x .. y + 2, NUL-terminated */
  size_t cap, ref_cap;
  uint32_t x, sz;
  ctg_t *ctg;
} amd_slot[2];
static int amd_cur = -1; /* slot of the block in flight, -1 = none */
static int amd_next;     /* slot the next block goes into */
static gt_vcf *amd_user_vcf; /* what work->vcf held before the first call (restored at join, it is the caller's to free) */
static int amd_user_vcf_size;

static void amd_die(const char *what) {
  fprintf(stderr, "bscall_amd: %s: %s\n", what, bsc_last_error());
  exit(1); /* the code base's convention for fatal errors (gt_fatal_error_msg) */
}

static void *amd_grow(void *p, size_t *cap, size_t need, size_t elem) {
  if (need > *cap) {
    p = realloc(p, need * elem);
    if (!p) { fprintf(stderr, "bscall_amd: out of memory\n"); exit(1); }
    *cap = need;
  }
  return p;
}

static void amd_timed_wait(pthread_cond_t *c, pthread_mutex_t *m) { /* the original's 5-second re-check waits */
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  ts.tv_sec += 5;
  pthread_cond_timedwait(c, m, &ts);
}

/* block in flight -> complete -> handed to the print thread */
static void amd_publish_pending(sr_param *const param) {
  work_t *const work = &param->work;
  if (amd_cur < 0) return;
  struct amd_slot *s = &amd_slot[amd_cur];
  const int rc = bsc_block_fetch(amd_ctx, NULL, NULL); /* waits for the kernels and the copy-out */
  if (rc < 0) amd_die("bsc_block_fetch"); /* BSC_ERR_ARG here = one of the original's asserts (:186,188) on that block */
  if (rc == BSC_WARN_INEXACT) fprintf(stderr, "bscall_amd: %s\n", bsc_last_error());
  /* the print thread must have drained the block before (original :228-235) */
  pthread_mutex_lock(&work->print_mutex);
  while (work->vcf_n) amd_timed_wait(&work->print_cond2, &work->print_mutex);
  pthread_mutex_unlock(&work->print_mutex);
  work->vcf = s->vcf;
  work->vcf_size = (int)s->cap;
  work->vcf_x = s->x;
  work->vcf_ctg = s->ctg;
  /* meth profiling reads work->ref: it must be idle before the codes change (original :244-254) */
  pthread_mutex_lock(&work->mprof_mutex);
  while (work->mprof_read_idx != work->mprof_write_idx) amd_timed_wait(&work->mprof_cond2, &work->mprof_mutex);
  pthread_mutex_unlock(&work->mprof_mutex);
  gt_string_resize(work->ref, s->sz + 3);
  memcpy(gt_string_get_string(work->ref), s->ref, (size_t)s->sz + 3);
  gt_string_set_length(work->ref, s->sz + 2);
  /* records are complete: flags, then wake the print thread (original :110-114, :255-258) */
  pthread_mutex_lock(&work->vcf_mutex);
  for (uint32_t i = 0; i < s->sz; i++) s->vcf[i].ready = true;
  pthread_mutex_unlock(&work->vcf_mutex);
  work->vcf_n = (int)s->sz;
  pthread_mutex_lock(&work->print_mutex);
  pthread_cond_signal(&work->print_cond1);
  pthread_mutex_unlock(&work->print_mutex);
  pthread_mutex_lock(&work->vcf_mutex);
  pthread_cond_signal(&work->vcf_cond);
  pthread_mutex_unlock(&work->vcf_mutex);
  amd_cur = -1;
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
  work->calc_threads_complete = 1; /* the next call never waits on calc_cond2: it fetches instead */
  work->calc_threads = NULL;
  amd_user_vcf = work->vcf;
  amd_user_vcf_size = work->vcf_size;
}

void join_calc_threads(sr_param *const param) {
  work_t *const work = &param->work;
  amd_publish_pending(param); /* the last block */
  /* the print thread still reads the published array: wait until it is drained before the arrays go */
  pthread_mutex_lock(&work->print_mutex);
  while (work->vcf_n) amd_timed_wait(&work->print_cond2, &work->print_mutex);
  pthread_mutex_unlock(&work->print_mutex);
  work->calc_end = true;
  work->vcf = amd_user_vcf;
  work->vcf_size = amd_user_vcf_size;
  for (int k = 0; k < 2; k++) {
    bsc_free_host(amd_slot[k].vcf);
    bsc_free_host(amd_slot[k].skip);
    free(amd_slot[k].ref);
    memset(&amd_slot[k], 0, sizeof amd_slot[k]);
  }
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
  const uint32_t sz = y - x + 1;
  const uint32_t nr = gt_vector_get_used(align_list);

  /* 1. the previous block: complete it and hand it to the print thread (the original waits for its calc threads
   *    here, :161-168) */
  amd_publish_pending(param);

  /* 2. this block: flatten align_details -> bsc_template + one read buffer (the library copies them into its own
   *    staging area, so the reader may recycle the align_list as soon as the call returns) */
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
  }
  struct amd_slot *s = &amd_slot[amd_next];
  if (sz > s->cap) { /* the slot was published two calls ago and the print thread waited for since: free to regrow */
    bsc_free_host(s->vcf);
    bsc_free_host(s->skip);
    s->cap = (size_t)sz + sz / 4;
    s->vcf = bsc_alloc_host((uint64_t)s->cap * sizeof(gt_vcf));
    s->skip = bsc_alloc_host((uint64_t)s->cap);
    if (!s->vcf || !s->skip) amd_die("bsc_alloc_host");
  }
  s->ref = amd_grow(s->ref, &s->ref_cap, (size_t)sz + 3, 1);
  memcpy(s->ref, gt_string_get_string(work->ref1), (size_t)sz + 2); /* work->ref1: codes of x .. y+2 */
  s->ref[sz + 2] = 0;
  s->x = x;
  s->sz = sz;
  s->ctg = ctg;
  /* records land as gt_vcf images (stride 208: gtm, ready = 0, skip) right behind the kernels */
  if (bsc_block_submit_to(amd_ctx, amd_tpl, nr, amd_seq, nbytes, x, y, (const uint8_t *)s->ref, s->vcf, (uint32_t)sizeof(gt_vcf),
                          s->skip) < 0)
    amd_die("bsc_block_submit_to");
  amd_cur = amd_next;
  amd_next ^= 1;
  /* returns at once, like the original after its dispatch (:260-272) */
}
