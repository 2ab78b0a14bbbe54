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
/*
 * Round 4: small blocks are HELD BACK.  The reference calls call_genotypes_ML once per maximal run of overlapping templates
 * (src/get_template_vector.c:141-147: 10^2 .. 10^7 positions), and its calc threads cost nothing to start; a GPU block costs a
 * dozen launches, four copies and a wait whatever its size (a 10 000-position block: 43 M positions/s, under the host's own
 * cores).  So a call appends its block to the BATCH being filled and returns; when the batch holds AMD_BATCH_POSITIONS positions
 * (or at join) it is submitted as one launch sequence (bsc_blocks_submit_to) and the batch submitted before it — complete by
 * then — is handed to the print thread block by block, in order, exactly as single blocks were.  Two batches are alive at once:
 * one being filled / in flight, one being printed.
 */
#ifndef AMD_BATCH_POSITIONS
#define AMD_BATCH_POSITIONS 1000000u
#endif
static struct amd_batch {
  /* inputs, joined as bsc_blocks_submit_to wants them (plain memory: the library copies them to its pinned staging area) */
  bsc_template *tpl;
  uint8_t *seq;
  char *ref;            /* the blocks' reference codes, x .. y + 2 each, one block after another */
  bsc_block_desc *desc;
  AMD_CTG_T **ctg;
  uint64_t *off;        /* where each block's images start in vcf[] (bsc_blocks_submit_to) */
  size_t n_tpl, cap_tpl, n_seq, cap_seq, n_ref, cap_ref, n_blk, cap_blk;
  uint64_t positions, padded; /* positions of the blocks; the same with every block rounded up to 64 (the images needed) */
  /* outputs: page-locked (bsc_alloc_host), the copy-out is a true DMA behind the kernels */
  AMD_GT_VCF_T *vcf;
  uint8_t *skip;
  size_t cap_out;
} amd_batch[2];
static int amd_flight = -1; /* batch in flight, -1 = none */
static int amd_fill;        /* batch being filled */
static uint64_t amd_threshold = AMD_BATCH_POSITIONS;

/* BSCALL_AMD_BATCH_POSITIONS in the environment overrides the threshold (0: every block is submitted by its own call, as in
 * round 3); called once, before the first block */
static void amd_overlap_init(void) {
  const char *e = getenv("BSCALL_AMD_BATCH_POSITIONS");
  if (e && *e) amd_threshold = strtoull(e, NULL, 10);
}

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

static void *amd_grow_to(void *p, size_t *cap, size_t need, size_t elem) {
  if (need > *cap) {
    const size_t n = need + need / 2 + 64;
    p = realloc(p, n * elem);
    if (!p) { fprintf(stderr, "bscall_amd: out of memory\n"); exit(1); }
    *cap = n;
  }
  return p;
}

/* batch in flight -> complete -> its blocks handed to the print thread, one after another */
static void amd_publish_flight(AMD_WORK_T *const work) {
  if (amd_flight < 0) return;
  struct amd_batch *q = &amd_batch[amd_flight];
  const int rc = bsc_block_fetch(amd_ctx, NULL, NULL); /* waits for the kernels and the copy-out */
  if (rc < 0) amd_die("bsc_block_fetch"); /* BSC_ERR_ARG here = one of the original's asserts (:186,188) on a block of the batch */
  if (rc == BSC_WARN_INEXACT) fprintf(stderr, "bscall_amd: %s\n", bsc_last_error());
  size_t ref_at = 0;
  for (size_t b = 0; b < q->n_blk; b++) {
    const uint32_t sz = q->desc[b].y - q->desc[b].x + 1u;
    AMD_GT_VCF_T *const v = q->vcf + q->off[b];
    /* the print thread must have drained the block before (original :228-235) */
    pthread_mutex_lock(&work->print_mutex);
    while (work->vcf_n) amd_timed_wait(&work->print_cond2, &work->print_mutex);
    pthread_mutex_unlock(&work->print_mutex);
    work->vcf = v;
    work->vcf_size = (int)(q->cap_out - q->off[b]);
    work->vcf_x = q->desc[b].x;
    work->vcf_ctg = q->ctg[b];
    AMD_SET_REF(work, q->ref + ref_at, sz); /* the print thread is idle (vcf_n == 0): nobody reads work->ref now */
    ref_at += (size_t)sz + 2;
    /* records are complete: flags, then wake the print thread (original :110-114, :255-258) */
    pthread_mutex_lock(&work->vcf_mutex);
    for (uint32_t i = 0; i < sz; i++) __atomic_store_n(&v[i].ready, true, __ATOMIC_RELEASE);
    pthread_mutex_unlock(&work->vcf_mutex);
    pthread_mutex_lock(&work->print_mutex);
    work->vcf_n = (int)sz;
    pthread_cond_signal(&work->print_cond1);
    pthread_mutex_unlock(&work->print_mutex);
    pthread_mutex_lock(&work->vcf_mutex);
    pthread_cond_signal(&work->vcf_cond);
    pthread_mutex_unlock(&work->vcf_mutex);
  }
  amd_flight = -1;
}

/* the batch being filled goes to the GPU; the one submitted before it is published first (its images are complete by now, and
 * only then is the print thread done with the arrays of the batch before that — the ones about to be written again) */
static void amd_flush(AMD_WORK_T *const work) {
  amd_publish_flight(work);
  struct amd_batch *q = &amd_batch[amd_fill];
  if (q->n_blk == 0) return;
  if (q->padded > q->cap_out) { /* this batch's arrays were published two flushes ago and the print thread waited for since */
    bsc_free_host(q->vcf);
    bsc_free_host(q->skip);
    q->cap_out = (size_t)(q->padded + q->padded / 4);
    q->vcf = bsc_alloc_host((uint64_t)q->cap_out * sizeof(AMD_GT_VCF_T));
    q->skip = bsc_alloc_host((uint64_t)q->cap_out);
    if (!q->vcf || !q->skip) amd_die("bsc_alloc_host");
  }
  /* records land as gt_vcf images (stride 208: gtm, ready = 0, skip) right behind the kernels */
  if (bsc_blocks_submit_to(amd_ctx, q->desc, (uint32_t)q->n_blk, q->tpl, q->seq, q->n_seq, (const uint8_t *)q->ref, q->vcf,
                           (uint32_t)sizeof(AMD_GT_VCF_T), q->skip, q->off) < 0)
    amd_die("bsc_blocks_submit_to");
  amd_flight = amd_fill;
  amd_fill ^= 1;
  q = &amd_batch[amd_fill]; /* published in this very flush: its inputs are free to be overwritten, its images are the print thread's */
  q->n_tpl = q->n_seq = q->n_ref = q->n_blk = 0;
  q->positions = q->padded = 0;
}

/*
 * One call_genotypes_ML: the block — already flattened into tpl / seq by the caller — joins the batch being filled, with a
 * private copy of its reference codes; the meth profiling thread is waited for, so that the caller may overwrite work->ref1;
 * a full batch is submitted (and the batch before it published).  Returns with the block pending or in flight, like the
 * original after its dispatch (:260-272).
 */
static void amd_overlap_call(AMD_WORK_T *const work, AMD_CTG_T *const ctg, const bsc_template *tpl, uint32_t nr, const uint8_t *seq,
                             uint64_t nbytes, uint32_t x, uint32_t y) {
  const uint32_t sz = y - x + 1;
#ifdef AMD_TEST_ROUND2_BUG
  const int had_pending = amd_flight >= 0 || amd_batch[amd_fill].n_blk != 0;
#endif
  /* a batch may hold 2^28 - 1 image slots and 2^31 - 1 templates (bsc_blocks_submit_to): one that this block would push past
   * either goes first */
  if (amd_batch[amd_fill].n_blk && (amd_batch[amd_fill].padded + sz + 64u > 0x0fffffffull || amd_batch[amd_fill].n_tpl + nr > 0x7fffffffull))
    amd_flush(work);
  struct amd_batch *q = &amd_batch[amd_fill];
  q->tpl = amd_grow_to(q->tpl, &q->cap_tpl, q->n_tpl + nr, sizeof *q->tpl);
  q->seq = amd_grow_to(q->seq, &q->cap_seq, q->n_seq + (size_t)nbytes, 1);
  q->ref = amd_grow_to(q->ref, &q->cap_ref, q->n_ref + (size_t)sz + 3, 1);
  {
    size_t cap = q->cap_blk;
    q->desc = amd_grow_to(q->desc, &cap, q->n_blk + 1, sizeof *q->desc);
    cap = q->cap_blk;
    q->ctg = amd_grow_to(q->ctg, &cap, q->n_blk + 1, sizeof *q->ctg);
    q->off = amd_grow_to(q->off, &q->cap_blk, q->n_blk + 1, sizeof *q->off);
  }
  for (uint32_t i = 0; i < nr; i++) { /* the read offsets become offsets into the batch's read buffer */
    bsc_template t = tpl[i];
    t.off[0] += q->n_seq;
    t.off[1] += q->n_seq;
    q->tpl[q->n_tpl + i] = t;
  }
  memcpy(q->seq + q->n_seq, seq, (size_t)nbytes);
  memcpy(q->ref + q->n_ref, AMD_REF1(work), (size_t)sz + 2); /* work->ref1: codes of x .. y + 2 */
  q->ref[q->n_ref + sz + 2] = 0;
  q->desc[q->n_blk].x = x;
  q->desc[q->n_blk].y = y;
  q->desc[q->n_blk].nr = nr;
  q->desc[q->n_blk]._pad = 0;
  q->ctg[q->n_blk] = ctg;
  q->n_tpl += nr;
  q->n_seq += (size_t)nbytes;
  q->n_ref += (size_t)sz + 2;
  q->n_blk++;
  q->positions += sz;
  q->padded += ((uint64_t)sz + 63u) & ~(uint64_t)63u;
  if (q->positions >= amd_threshold) amd_flush(work);
#ifdef AMD_TEST_ROUND2_BUG /* tests only: round 2's behaviour, no wait in a call that found nothing pending — the harness must catch it */
  if (had_pending)
#endif
    amd_wait_mprof(work); /* in EVERY call, whatever is pending (original :244-251): the reference codes were copied above */
}

/* join_calc_threads: what is still held back is submitted, everything is published, the print thread drains it, the arrays go */
static void amd_overlap_join(AMD_WORK_T *const work) {
  amd_flush(work);         /* publishes the batch in flight, submits the one being filled */
  amd_publish_flight(work); /* ... and publishes that one */
  pthread_mutex_lock(&work->print_mutex);
  while (work->vcf_n) amd_timed_wait(&work->print_cond2, &work->print_mutex);
  pthread_mutex_unlock(&work->print_mutex);
  for (int k = 0; k < 2; k++) {
    struct amd_batch *q = &amd_batch[k];
    bsc_free_host(q->vcf);
    bsc_free_host(q->skip);
    free(q->tpl);
    free(q->seq);
    free(q->ref);
    free(q->desc);
    free(q->ctg);
    free(q->off);
    memset(q, 0, sizeof *q);
  }
  amd_flight = -1;
  amd_fill = 0;
}

#endif /* AMD_OVERLAP_PROTOCOL_H */
