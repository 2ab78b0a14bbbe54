/*
 * amd_overlap_protocol.h — the thread protocol of the OVERLAPPED replacement call_genotypes_ML, written once and included
 * by both of its users: integration/call_genotypes_amd_overlap.c (inside the bs_call tree, against the reference's work_t)
 * and integration/demo_block.c (against a mock of the work_t fields it touches — on the GPU, and as a CPU-only build with
 * stub bsc_* entries under ThreadSanitizer: tests/test_glue_tsan.py).  What it restates of the original
 * (src/call_genotypes.c): the wait for the print thread before work->vcf changes hands (:228-235), the wait for the meth
 * profiling thread IN EVERY CALL before the caller may overwrite work->ref1 (:244-251), the publication of a complete
 * block — `ready` flags, vcf_n, the two signals (:110-114, :255-258).
 *
 * Before including, define:
 *   AMD_WORK_T                  the work_t type; fields used: vcf, vcf_size, vcf_n, vcf_x, vcf_ctg, print_mutex, print_cond1,
 *                               print_cond2, vcf_mutex, vcf_cond, mprof_mutex, mprof_cond2, mprof_read_idx, mprof_write_idx
 *   AMD_GT_VCF_T                gt_vcf (208 bytes: gt_meth, ready, skip)
 *   AMD_CTG_T                   ctg_t
 *   AMD_SET_REF(work, src, sz)  work->ref := the sz + 2 reference codes at src (+ terminator)
 *   AMD_REF1(work)              const char *: the codes of x .. y + 2 of the block being handed over (work->ref1)
 */
#ifndef AMD_OVERLAP_PROTOCOL_H
#define AMD_OVERLAP_PROTOCOL_H

#include <pthread.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <bscall_amd.h>

static bsc_context *amd_ctx;
/* the two blocks that can be alive at once: one being printed, one being computed */
static struct amd_slot {
  AMD_GT_VCF_T *vcf; /* page-locked (bsc_alloc_host): the copy-out is a true DMA behind the kernels */
  uint8_t *skip;     /* bsc_block_submit_to's skip array (the gt_vcf images carry the flag too) */
  char *ref;         /* private copy of the block's reference codes, x .. y + 2, NUL-terminated */
  size_t cap, ref_cap;
  uint32_t x, sz;
  AMD_CTG_T *ctg;
} amd_slot[2];
static int amd_cur = -1; /* slot of the block in flight, -1 = none */
static int amd_next;     /* slot the next block goes into */

static void amd_die(const char *what) {
  fprintf(stderr, "bscall_amd: %s: %s\n", what, bsc_last_error());
  exit(1); /* the code base's convention for fatal errors (gt_fatal_error_msg) */
}

static void amd_timed_wait(pthread_cond_t *c, pthread_mutex_t *m) { /* the original's 5-second re-check waits */
  struct timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  ts.tv_sec += 5;
  pthread_cond_timedwait(c, m, &ts);
}

/* The meth profiling thread reads work->ref1 for the templates of the block being handed over (src/meth_profile.c:51,
 * queued by process_template_vector, src/process_template.c:116-124); the caller overwrites ref1 as soon as the call
 * returns (:29-30).  The original waits for the queue to drain in EVERY call (:244-251) — so must this one, block pending
 * or not. */
static void amd_wait_mprof(AMD_WORK_T *const work) {
  pthread_mutex_lock(&work->mprof_mutex);
  while (work->mprof_read_idx != work->mprof_write_idx) amd_timed_wait(&work->mprof_cond2, &work->mprof_mutex);
  pthread_mutex_unlock(&work->mprof_mutex);
}

/* block in flight -> complete -> handed to the print thread */
static void amd_publish_pending(AMD_WORK_T *const work) {
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
  AMD_SET_REF(work, s->ref, s->sz); /* the print thread is idle (vcf_n == 0): nobody reads work->ref now */
  /* records are complete: flags, then wake the print thread (original :110-114, :255-258) */
  pthread_mutex_lock(&work->vcf_mutex);
  for (uint32_t i = 0; i < s->sz; i++) __atomic_store_n(&s->vcf[i].ready, true, __ATOMIC_RELEASE);
  pthread_mutex_unlock(&work->vcf_mutex);
  pthread_mutex_lock(&work->print_mutex);
  work->vcf_n = (int)s->sz;
  pthread_cond_signal(&work->print_cond1);
  pthread_mutex_unlock(&work->print_mutex);
  pthread_mutex_lock(&work->vcf_mutex);
  pthread_cond_signal(&work->vcf_cond);
  pthread_mutex_unlock(&work->vcf_mutex);
  amd_cur = -1;
}

/*
 * One call_genotypes_ML: (1) the previous block is completed and handed to the print thread (the original waits for its
 * calc threads here, :161-168); (2) this block — already flattened into tpl / seq by the caller — is submitted into the
 * other gt_vcf[] array with a private copy of its reference codes; (3) the meth profiling thread is waited for, so that
 * the caller may overwrite work->ref1.  Returns with the block in flight, like the original after its dispatch (:260-272).
 */
static void amd_overlap_call(AMD_WORK_T *const work, AMD_CTG_T *const ctg, const bsc_template *tpl, uint32_t nr, const uint8_t *seq,
                             uint64_t nbytes, uint32_t x, uint32_t y) {
  const uint32_t sz = y - x + 1;
#ifdef AMD_TEST_ROUND2_BUG
  const int had_pending = amd_cur >= 0;
#endif
  amd_publish_pending(work);
  struct amd_slot *s = &amd_slot[amd_next];
  if (sz > s->cap) { /* the slot was published two calls ago and the print thread waited for since: free to regrow */
    bsc_free_host(s->vcf);
    bsc_free_host(s->skip);
    s->cap = (size_t)sz + sz / 4;
    s->vcf = bsc_alloc_host((uint64_t)s->cap * sizeof(AMD_GT_VCF_T));
    s->skip = bsc_alloc_host((uint64_t)s->cap);
    if (!s->vcf || !s->skip) amd_die("bsc_alloc_host");
  }
  if ((size_t)sz + 3 > s->ref_cap) {
    s->ref = realloc(s->ref, (size_t)sz + 3);
    if (!s->ref) { fprintf(stderr, "bscall_amd: out of memory\n"); exit(1); }
    s->ref_cap = (size_t)sz + 3;
  }
  memcpy(s->ref, AMD_REF1(work), (size_t)sz + 2); /* work->ref1: codes of x .. y + 2 */
  s->ref[sz + 2] = 0;
  s->x = x;
  s->sz = sz;
  s->ctg = ctg;
  /* records land as gt_vcf images (stride 208: gtm, ready = 0, skip) right behind the kernels */
  if (bsc_block_submit_to(amd_ctx, tpl, nr, seq, nbytes, x, y, (const uint8_t *)s->ref, s->vcf, (uint32_t)sizeof(AMD_GT_VCF_T),
                          s->skip) < 0)
    amd_die("bsc_block_submit_to");
  amd_cur = amd_next;
  amd_next ^= 1;
#ifdef AMD_TEST_ROUND2_BUG /* tests only: round 2's behaviour, no wait in a call that found no block pending — the harness must catch it */
  if (had_pending)
#endif
    amd_wait_mprof(work); /* in EVERY call, block pending or not (original :244-251) */
}

/* join_calc_threads: the last block is published, the print thread drains it, the arrays go */
static void amd_overlap_join(AMD_WORK_T *const work) {
  amd_publish_pending(work);
  pthread_mutex_lock(&work->print_mutex);
  while (work->vcf_n) amd_timed_wait(&work->print_cond2, &work->print_mutex);
  pthread_mutex_unlock(&work->print_mutex);
  for (int k = 0; k < 2; k++) {
    bsc_free_host(amd_slot[k].vcf);
    bsc_free_host(amd_slot[k].skip);
    free(amd_slot[k].ref);
    memset(&amd_slot[k], 0, sizeof amd_slot[k]);
  }
  amd_cur = -1;
  amd_next = 0;
}

#endif /* AMD_OVERLAP_PROTOCOL_H */
