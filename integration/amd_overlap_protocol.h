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
 * (or 65 536 blocks, or at join) it is submitted as one launch sequence and its blocks reach the print thread one by one, in
 * order, exactly as single blocks did.
 *
 * Round 5: three things stay busy at once — the GPU on batch k + 1, the print thread on the blocks of batch k, the process thread
 * on filling batch k + 2 — so three batches are alive (filled / in flight / being handed over):
 *   - the batch is built straight in page-locked buffers and submitted through bsc_blocks_submit_to_inplace (no staging copy:
 *     the staged form from ordinary memory ran at the speed of the host's own cores, profiles/r05_small_blocks.txt);
 *   - a flush fetches the batch in flight, makes sure the batch handed over BEFORE it is through (its images are the ones the
 *     new submission overwrites: two output arrays alternate), submits the filled batch, and only then starts handing the
 *     fetched batch's blocks to the print thread — round 4 handed over every block, waiting for the print thread each time,
 *     BEFORE it submitted, with the GPU idle;
 *   - blocks are handed over as the print thread becomes free: one look at the top of every call (never waiting), the rest
 *     at the next flush.
 */
#ifndef AMD_BATCH_POSITIONS
#define AMD_BATCH_POSITIONS 1000000u
#endif
#define AMD_BATCH_BLOCKS 65536u /* bsc_blocks_submit_to's limit on the blocks of one call */
static struct amd_batch {
  /* inputs, joined as bsc_blocks_submit_to_inplace wants them; tpl / seq / ref in page-locked memory (bsc_alloc_host): the upload
   * is a DMA straight out of them, and they stay untouched until the batch has been fetched */
  bsc_template *tpl;
  uint8_t *seq;
  char *ref;            /* the blocks' reference codes, x .. y + 2 each, one block after another */
  bsc_block_desc *desc;
  AMD_CTG_T **ctg;
  uint64_t *off;        /* where each block's images start in its output array (returned by the submission) */
  size_t n_tpl, cap_tpl, n_seq, cap_seq, n_ref, cap_ref, n_blk, cap_blk;
  uint64_t positions, padded; /* positions of the blocks; the same with every block rounded up to 64 (the images needed) */
  int out_ix;           /* which of the two output arrays holds its images (set when it is submitted) */
} amd_batch[3];
/* outputs: page-locked (bsc_alloc_host), the copy-out is a true DMA behind the kernels; submission n writes array n & 1 */
static struct amd_outbuf {
  AMD_GT_VCF_T *vcf;
  uint8_t *skip;
  size_t cap;
} amd_out[2];
static int amd_fill;        /* batch being filled */
static int amd_flight = -1; /* batch in flight, -1 = none */
static int amd_pub = -1;    /* batch whose blocks are being handed to the print thread, -1 = none */
static size_t amd_pub_next, amd_pub_ref_at; /* its next block, and where that block's reference codes start */
static unsigned amd_submits;
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

/* the same for a page-locked array holding `used` elements.  Page-locking memory is slow (milliseconds per call), and batches
 * resemble each other: an array grows to at least what the largest array of its kind has needed so far (*hint), so that after the
 * first batch the three slots are sized in one step each */
static void *amd_grow_pinned(void *p, size_t *cap, size_t used, size_t need, size_t elem, size_t *hint) {
  if (need > *hint) *hint = need;
  if (need > *cap) {
    size_t n = need + need / 2 + 4096;
    if (n < *hint + *hint / 8) n = *hint + *hint / 8;
    void *q = bsc_alloc_host((uint64_t)n * elem);
    if (!q) amd_die("bsc_alloc_host");
    if (used) memcpy(q, p, used * elem);
    bsc_free_host(p);
    p = q;
    *cap = n;
  }
  return p;
}

/* the print thread must have drained the block handed over before (original :228-235) */
static void amd_drain(AMD_WORK_T *const work) {
  pthread_mutex_lock(&work->print_mutex);
  while (work->vcf_n) amd_timed_wait(&work->print_cond2, &work->print_mutex);
  pthread_mutex_unlock(&work->print_mutex);
}

/* Blocks of the batch being handed over -> the print thread, one after another, each when the print thread has drained the one
 * before.  wait == false: as many as it is ready for right now (none, or one: it is busy again the moment it has a block). */
static void amd_publish_some(AMD_WORK_T *const work, bool wait) {
  if (amd_pub < 0) return;
  struct amd_batch *q = &amd_batch[amd_pub];
  struct amd_outbuf *o = &amd_out[q->out_ix];
  while (amd_pub_next < q->n_blk) {
    pthread_mutex_lock(&work->print_mutex);
    while (work->vcf_n) {
      if (!wait) {
        pthread_mutex_unlock(&work->print_mutex);
        return;
      }
      amd_timed_wait(&work->print_cond2, &work->print_mutex);
    }
    pthread_mutex_unlock(&work->print_mutex);
    const size_t b = amd_pub_next++;
    const uint32_t sz = q->desc[b].y - q->desc[b].x + 1u;
    AMD_GT_VCF_T *const v = o->vcf + q->off[b];
    work->vcf = v;
    work->vcf_size = (int)(o->cap - q->off[b]);
    work->vcf_x = q->desc[b].x;
    work->vcf_ctg = q->ctg[b];
    AMD_SET_REF(work, q->ref + amd_pub_ref_at, sz); /* the print thread is idle (vcf_n == 0): nobody reads work->ref now */
    amd_pub_ref_at += (size_t)sz + 2;
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
}

/*
 * The batch being filled goes to the GPU.  Order: (1) the batch in flight is fetched — the one wait for the GPU; (2) the batch
 * handed over before it must be through: every block with the print thread and the last one drained, because its output array
 * is the one this submission writes and its slot the next one to be filled; (3) submit; (4) the fetched batch starts to be
 * handed over.  The print thread is never waited for with the GPU idle unless it is a whole batch behind.
 */
static void amd_flush(AMD_WORK_T *const work) {
  const int done = amd_flight;
  if (done >= 0) {
    const int rc = bsc_block_fetch(amd_ctx, NULL, NULL); /* waits for the kernels and the copy-out */
    if (rc < 0) amd_die("bsc_block_fetch"); /* BSC_ERR_ARG here = one of the original's asserts (:186,188) on a block of the batch */
    if (rc == BSC_WARN_INEXACT) fprintf(stderr, "bscall_amd: %s\n", bsc_last_error());
  }
  amd_publish_some(work, true);
  if (amd_pub >= 0) amd_drain(work);
  amd_pub = -1;
  struct amd_batch *q = &amd_batch[amd_fill];
  amd_flight = -1;
  if (q->n_blk) {
    struct amd_outbuf *o = &amd_out[amd_submits & 1u];
    if (q->padded > o->cap) { /* its last user was handed over and drained above, or two flushes ago */
      bsc_free_host(o->vcf);
      bsc_free_host(o->skip);
      o->cap = (size_t)(q->padded + q->padded / 4);
      o->vcf = bsc_alloc_host((uint64_t)o->cap * sizeof(AMD_GT_VCF_T));
      o->skip = bsc_alloc_host((uint64_t)o->cap);
      if (!o->vcf || !o->skip) amd_die("bsc_alloc_host");
    }
    /* records land as gt_vcf images (stride 208: gtm, ready = 0, skip) right behind the kernels */
    if (bsc_blocks_submit_to_inplace(amd_ctx, q->desc, (uint32_t)q->n_blk, q->tpl, q->seq, q->n_seq, (const uint8_t *)q->ref, o->vcf,
                                     (uint32_t)sizeof(AMD_GT_VCF_T), o->skip, q->off) < 0)
      amd_die("bsc_blocks_submit_to_inplace");
    q->out_ix = (int)(amd_submits & 1u);
    amd_submits++;
    amd_flight = amd_fill;
  }
  amd_pub = done;
  amd_pub_next = amd_pub_ref_at = 0;
  amd_publish_some(work, false);
  for (int k = 0; k < 3; k++) /* the next batch is filled in the slot that is neither in flight nor being handed over */
    if (k != amd_flight && k != amd_pub) {
      amd_fill = k;
      break;
    }
  q = &amd_batch[amd_fill];
  q->n_tpl = q->n_seq = q->n_ref = q->n_blk = 0;
  q->positions = q->padded = 0;
}

/*
 * One call_genotypes_ML: the block — already flattened into tpl / seq by the caller — joins the batch being filled, with a
 * private copy of its reference codes; the meth profiling thread is waited for, so that the caller may overwrite work->ref1;
 * a full batch is submitted.  Returns with the block pending or in flight, like the original after its dispatch (:260-272).
 */
static void amd_overlap_call(AMD_WORK_T *const work, AMD_CTG_T *const ctg, const bsc_template *tpl, uint32_t nr, const uint8_t *seq,
                             uint64_t nbytes, uint32_t x, uint32_t y) {
  const uint32_t sz = y - x + 1;
#ifdef AMD_TEST_ROUND2_BUG
  const int had_pending = amd_flight >= 0 || amd_pub >= 0 || amd_batch[amd_fill].n_blk != 0;
#endif
  amd_publish_some(work, false); /* the print thread may have become free since the last look */
  /* a batch may hold 2^28 - 1 image slots, 2^31 - 1 templates and 65 536 blocks (bsc_blocks_submit_to): one that this block would
   * push past any of them goes first */
  if (amd_batch[amd_fill].n_blk &&
      (amd_batch[amd_fill].padded + sz + 64u > 0x0fffffffull || amd_batch[amd_fill].n_tpl + nr > 0x7fffffffull ||
       amd_batch[amd_fill].n_blk >= AMD_BATCH_BLOCKS))
    amd_flush(work);
  struct amd_batch *q = &amd_batch[amd_fill];
  static size_t hint_tpl, hint_seq, hint_ref;
  q->tpl = amd_grow_pinned(q->tpl, &q->cap_tpl, q->n_tpl, q->n_tpl + nr, sizeof *q->tpl, &hint_tpl);
  q->seq = amd_grow_pinned(q->seq, &q->cap_seq, q->n_seq, q->n_seq + (size_t)nbytes, 1, &hint_seq);
  q->ref = amd_grow_pinned(q->ref, &q->cap_ref, q->n_ref, q->n_ref + (size_t)sz + 3, 1, &hint_ref);
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

/* join_calc_threads: what is still held back is submitted, everything is handed over, the print thread drains it, the arrays go */
static void amd_overlap_join(AMD_WORK_T *const work) {
  amd_flush(work); /* fetches the batch in flight, submits the one being filled */
  amd_flush(work); /* fetches that one; nothing left to submit */
  amd_publish_some(work, true);
  amd_drain(work);
  amd_pub = -1;
  for (int k = 0; k < 3; k++) {
    struct amd_batch *q = &amd_batch[k];
    bsc_free_host(q->tpl);
    bsc_free_host(q->seq);
    bsc_free_host(q->ref);
    free(q->desc);
    free(q->ctg);
    free(q->off);
    memset(q, 0, sizeof *q);
  }
  for (int k = 0; k < 2; k++) {
    bsc_free_host(amd_out[k].vcf);
    bsc_free_host(amd_out[k].skip);
    memset(&amd_out[k], 0, sizeof amd_out[k]);
  }
  amd_flight = -1;
  amd_fill = 0;
  amd_submits = 0;
}

#endif /* AMD_OVERLAP_PROTOCOL_H */
