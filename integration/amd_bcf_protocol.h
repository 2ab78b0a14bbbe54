/*
 * amd_bcf_protocol.h — the thread protocol of INTEGRATION.md level 2b, written once and included by its two users:
 * integration/call_genotypes_amd_bcf.c (inside the bs_call tree, against the reference's work_t) and integration/demo_block.c (against
 * the mock of integration/mock_work.h, on the GPU).  At this level the per-site work of the reference's print thread
 * (src/process.c:87-104 -> src/print_vcf.c:32-381: record formation, the window of five sites, the bcf_enc_* calls, bcf_write's fixed
 * fields, the bs_stats sums) has left the host: what comes back from a block is its stretch of the BCF record stream, and what is left
 * to a host thread is bgzf_write.  So the hand-over is of BYTES, to a writer thread of this file's own; the reference's print thread is
 * never given a block (work->vcf_n stays 0) and ends on print_end as it does after an empty input.
 *
 *   call k   the block — flattened into tpl / seq by the caller — is copied into page-locked input slot k & 1 with its reference codes
 *            (the caller overwrites work->ref1 the moment the call returns, src/process_template.c:29-30); the slot's previous user,
 *            block k - 2, was fetched during call k - 1 and its bytes must have been written (the one wait for the writer); block k - 1,
 *            in flight since the last call, is fetched (bsc_block_bcf_fetch: the one wait for the GPU) and its bytes go to the writer;
 *            block k is queued (bsc_block_bcf_submit_inplace: uploads, the reads -> records chain, the encoder and the copy-out behind
 *            each other on the context's stream) and the meth profiling thread is waited for, in EVERY call, as the original does
 *            (src/call_genotypes.c:244-251).  The call returns with block k in flight: the GPU works on it while the process thread
 *            reads and pre-processes block k + 1 and the writer writes block k - 1.
 *   join     the last block is fetched and written; the writer ends.
 *
 * One block = one launch sequence here (the gt_vcf form holds small blocks back and submits them together, amd_overlap_protocol.h; the
 * bytes form has no entry for several blocks yet — blocks cannot simply be merged, the printer's window of five sites starts afresh
 * with every block).
 *
 * Before including, define:
 *   AMD_WORK_T                     the work_t type; fields used: mprof_mutex, mprof_cond2, mprof_read_idx, mprof_write_idx
 *   AMD_REF1(work)                 const char *: the codes of x .. y + 2 of the block being handed over (work->ref1)
 *   AMD_BCF_WRITE(work, buf, n)    the records' bytes in order -> the output (bgzf_write on work->vcf_file in the reference)
 */
#ifndef AMD_BCF_PROTOCOL_H
#define AMD_BCF_PROTOCOL_H

#include <pthread.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <bscall_amd.h>

static bsc_context *amd_bctx;
static bsc_bcf_ids amd_bids;
static struct amd_bslot {
  bsc_template *tpl; /* inputs, page-locked: the upload is a DMA straight out of them, untouched until the fetch */
  uint8_t *seq, *ref;
  uint8_t *out;      /* the block's stretch of the stream, page-locked */
  size_t cap_tpl, cap_seq, cap_ref, cap_out;
  uint64_t n_bytes, n_rec;
  int32_t rid;
  int to_write;      /* handed to the writer, not written yet (under amd_bmu) */
} amd_bslot[2];
static pthread_mutex_t amd_bmu = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t amd_bcv = PTHREAD_COND_INITIALIZER;
static pthread_t amd_bwriter;
static int amd_bwrite_next; /* the slot the writer takes next (blocks are written in order: slots alternate) */
static bool amd_bend;
static void *amd_bwork;
static unsigned amd_bcalls;
static int amd_bflight = -1; /* slot of the block in flight, -1 = none */
static uint64_t amd_brecords, amd_bbytes;
static bsc_vcf_params amd_bvp = {0, 1, 0xffffffffu};
static int amd_bstats = 1;

static void amd_bdie(const char *what) {
  fprintf(stderr, "bscall_amd: %s: %s\n", what, bsc_last_error());
  exit(1); /* the code base's convention for fatal errors (gt_fatal_error_msg) */
}

static void *amd_bwriter_main(void *arg) {
  (void)arg;
  for (;;) {
    pthread_mutex_lock(&amd_bmu);
    while (!amd_bslot[amd_bwrite_next].to_write && !amd_bend) pthread_cond_wait(&amd_bcv, &amd_bmu);
    struct amd_bslot *s = &amd_bslot[amd_bwrite_next];
    if (!s->to_write) { /* the end, and nothing left */
      pthread_mutex_unlock(&amd_bmu);
      return NULL;
    }
    pthread_mutex_unlock(&amd_bmu);
    AMD_BCF_WRITE((AMD_WORK_T *)amd_bwork, s->out, s->n_bytes); /* src/print_vcf.c:379-380, a block at a time */
    pthread_mutex_lock(&amd_bmu);
    s->to_write = 0;
    amd_bwrite_next ^= 1;
    pthread_cond_broadcast(&amd_bcv);
    pthread_mutex_unlock(&amd_bmu);
  }
}

static void amd_bcf_init(AMD_WORK_T *const work, bsc_context *ctx) {
  amd_bctx = ctx;
  amd_bwork = work;
  bsc_bcf_default_ids(&amd_bids); /* the dictionary indices print_vcf_header's header yields (src/print_vcf.c:621-745) */
  amd_bend = false;
  amd_bwrite_next = 0;
  amd_bcalls = 0;
  amd_bflight = -1;
  amd_brecords = amd_bbytes = 0;
  if (pthread_create(&amd_bwriter, NULL, amd_bwriter_main, NULL)) {
    fprintf(stderr, "bscall_amd: no writer thread\n");
    exit(1);
  }
}

static void *amd_bgrow(void *p, size_t *cap, size_t need) {
  if (need > *cap) {
    bsc_free_host(p);
    *cap = need + need / 2 + 4096;
    p = bsc_alloc_host((uint64_t)*cap);
    if (!p) amd_bdie("bsc_alloc_host");
  }
  return p;
}

static void amd_bwait_mprof(AMD_WORK_T *const work) { /* as amd_overlap_protocol.h: in EVERY call (src/call_genotypes.c:244-251) */
  pthread_mutex_lock(&work->mprof_mutex);
  while (work->mprof_read_idx != work->mprof_write_idx) {
    struct timespec ts;
    clock_gettime(CLOCK_REALTIME, &ts);
    ts.tv_sec += 5;
    pthread_cond_timedwait(&work->mprof_cond2, &work->mprof_mutex, &ts);
  }
  pthread_mutex_unlock(&work->mprof_mutex);
}

/* the block in flight: waited for, its bytes to the writer */
static void amd_bcf_collect(void) {
  if (amd_bflight < 0) return;
  struct amd_bslot *s = &amd_bslot[amd_bflight];
  int rc = bsc_block_bcf_fetch(amd_bctx, &s->n_bytes, &s->n_rec);
  if (rc == BSC_ERR_ARG && s->n_bytes > s->cap_out) { /* a block of long records (names, -A over multi-allelic sites): the encoder alone once
                                                          more, with the room it asks for */
    s->out = amd_bgrow(s->out, &s->cap_out, (size_t)s->n_bytes + 4096);
    rc = bsc_block_bcf_again(amd_bctx, s->out, s->cap_out, &s->n_bytes, &s->n_rec);
  }
  if (rc < 0) amd_bdie("bsc_block_bcf_fetch"); /* BSC_ERR_ARG here = one of the original's asserts (:186,188) on the block */
  if (rc == BSC_WARN_INEXACT) fprintf(stderr, "bscall_amd: %s\n", bsc_last_error());
  amd_brecords += s->n_rec;
  amd_bbytes += s->n_bytes;
  pthread_mutex_lock(&amd_bmu);
  s->to_write = 1;
  pthread_cond_broadcast(&amd_bcv);
  pthread_mutex_unlock(&amd_bmu);
  amd_bflight = -1;
}

/* One call_genotypes_ML at level 2b.  rid: the contig's index in the header (ctg->vcf_rid, src/print_vcf.c:163). */
static void amd_bcf_call(AMD_WORK_T *const work, int32_t rid, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t nbytes, uint32_t x,
                         uint32_t y) {
  const uint32_t sz = y - x + 1;
  struct amd_bslot *s = &amd_bslot[amd_bcalls & 1u];
  pthread_mutex_lock(&amd_bmu); /* block k - 2 used this slot: fetched in the last call; written by now? */
  while (s->to_write) pthread_cond_wait(&amd_bcv, &amd_bmu);
  pthread_mutex_unlock(&amd_bmu);
  s->tpl = amd_bgrow(s->tpl, &s->cap_tpl, (size_t)(nr ? nr : 1) * sizeof *s->tpl);
  s->seq = amd_bgrow(s->seq, &s->cap_seq, (size_t)nbytes + 16);
  s->ref = amd_bgrow(s->ref, &s->cap_ref, (size_t)sz + 3);
  s->out = amd_bgrow(s->out, &s->cap_out, (size_t)sz * 128u + 4096u);
  memcpy(s->tpl, tpl, (size_t)nr * sizeof *tpl);
  memcpy(s->seq, seq, (size_t)nbytes);
  memcpy(s->ref, AMD_REF1(work), (size_t)sz + 2); /* work->ref1: codes of x .. y + 2 */
  s->ref[sz + 2] = 0;
  s->rid = rid;
  amd_bcf_collect(); /* block k - 1 */
  if (bsc_block_bcf_submit_inplace(amd_bctx, s->tpl, nr, s->seq, nbytes, x, y, s->ref, NULL /* dbSNP flags: bsc_dbsnp_flags */, &amd_bvp, amd_bstats, rid,
                                   &amd_bids, NULL /* names: bsc_dbsnp_names */, s->out, s->cap_out) < 0)
    amd_bdie("bsc_block_bcf_submit_inplace");
  amd_bflight = (int)(amd_bcalls & 1u);
  amd_bcalls++;
  amd_bwait_mprof(work);
}

/* join_calc_threads: the last block is fetched and written, the writer ends, the buffers go */
static void amd_bcf_join(void) {
  amd_bcf_collect();
  pthread_mutex_lock(&amd_bmu);
  amd_bend = true;
  pthread_cond_broadcast(&amd_bcv);
  pthread_mutex_unlock(&amd_bmu);
  pthread_join(amd_bwriter, NULL);
  for (int k = 0; k < 2; k++) {
    struct amd_bslot *s = &amd_bslot[k];
    bsc_free_host(s->tpl);
    bsc_free_host(s->seq);
    bsc_free_host(s->ref);
    bsc_free_host(s->out);
    memset(s, 0, sizeof *s);
  }
}

#endif /* AMD_BCF_PROTOCOL_H */
