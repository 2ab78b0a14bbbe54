/*
 * amd_bcf_protocol.h — the thread protocol of INTEGRATION.md level 2b, written once and included by its two users:
 * integration/call_genotypes_amd_bcf.c (inside the bs_call tree, against the reference's work_t) and integration/demo_block.c (against
 * the mock of integration/mock_work.h, on the GPU).  At this level the per-site work of the reference's print thread
 * (src/process.c:87-104 -> src/print_vcf.c:32-381: record formation, the window of five sites, the bcf_enc_* calls, bcf_write's fixed
 * fields, the bs_stats sums) has left the host: what comes back from a block is its stretch of the BCF record stream, and what is left
 * to a host thread is bgzf_write.  So the hand-over is of BYTES, to a writer thread of this file's own; the reference's print thread is
 * never given a block (work->vcf_n stays 0) and ends on print_end as it does after an empty input.
 *
 *   call k   the block — flattened into tpl / seq by the caller — is APPENDED to the batch in the page-locked input slot being filled, with its
 *            reference codes (the caller overwrites work->ref1 the moment the call returns, src/process_template.c:29-30), and the meth
 *            profiling thread is waited for, in EVERY call, as the original does (src/call_genotypes.c:244-251).  A batch is handed over when
 *            it holds AMD_BCF_BATCH_POSITIONS positions (a block that large goes alone), at a change of contig, at 65 536 blocks:
 *   hand-over  the batch in flight (the other slot) is fetched (bsc_blocks_bcf_fetch: the one wait for the GPU) and its bytes go to the
 *            writer; this one is queued (bsc_blocks_bcf_submit_inplace: ONE launch sequence for all its blocks — uploads, the reads ->
 *            records chain, the encoder and the copy-out behind each other on the context's stream; a launch sequence costs ~0.2 ms
 *            whatever its size, and the reference's blocks are runs of overlapping templates, 10^2 .. 10^7 positions) and the other slot is
 *            filled next, once the writer has written what it held.  The GPU works on a batch while the process thread reads and
 *            pre-processes the next one's blocks and the writer writes the one before.
 *   join     the last batch is handed over, fetched and written; the writer ends.
 *
 * The blocks of a batch stay blocks: the printer's window of five sites starts afresh with every block (bsc_blocks_bcf: the stream of a
 * batch is the streams of its blocks one after another).
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
#ifndef AMD_BCF_BATCH_POSITIONS
#define AMD_BCF_BATCH_POSITIONS 250000u /* a batch is handed over when it holds this many positions: a launch sequence's ~0.2 ms are then under a
                                          tenth of the batch's time on the device, and the page-locked slots stay small (page-locking costs ~1 ms a MB) */
#endif
static struct amd_bslot {
  bsc_template *tpl; /* inputs of the batch's blocks one after another, page-locked: the upload is a DMA straight out of them, untouched until the fetch */
  uint8_t *seq, *ref;
  bsc_block_desc *desc;
  uint8_t *out;      /* the batch's stretch of the stream, page-locked */
  size_t cap_tpl, cap_seq, cap_ref, cap_desc, cap_out; /* bytes */
  uint32_t n_blk, n_tpl;
  uint64_t n_seq, n_ref, n_pos, n_pad; /* read bytes, reference codes, positions, positions with every block rounded up to 64 */
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
static unsigned amd_bcalls, amd_bbatches;
static int amd_bfill;        /* the slot being filled */
static int amd_bflight = -1; /* slot of the batch in flight, -1 = none */
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
  amd_bcalls = amd_bbatches = 0;
  amd_bfill = 0;
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

/* what the batch holds so far moves along; pos: the positions the batch holds with this block — the new room is what the batch will need when it
 * is full at this rate, so that a slot is page-locked once, not once per growth step */
static void *amd_bgrow_keep(void *p, size_t *cap, size_t need, size_t used, uint64_t pos) {
  if (need > *cap) {
    size_t nc = need + need / 2 + 4096;
    if (pos && pos < AMD_BCF_BATCH_POSITIONS) {
      const double full = (double)need * (double)AMD_BCF_BATCH_POSITIONS / (double)pos * 1.15;
      if (full > (double)nc && full < 4e9) nc = (size_t)full;
    }
    void *q = bsc_alloc_host((uint64_t)nc);
    if (!q) amd_bdie("bsc_alloc_host");
    if (used) memcpy(q, p, used);
    bsc_free_host(p);
    *cap = nc;
    p = q;
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

/* the batch in flight: waited for, its bytes to the writer, its input slot free again */
static void amd_bcf_collect(void) {
  if (amd_bflight < 0) return;
  struct amd_bslot *s = &amd_bslot[amd_bflight];
  int rc = bsc_blocks_bcf_fetch(amd_bctx, &s->n_bytes, &s->n_rec);
  if (rc == BSC_ERR_ARG && s->n_bytes > s->cap_out) { /* blocks of long records (names, -A over multi-allelic sites): the encoder alone once
                                                          more, with the room it asks for */
    s->out = amd_bgrow(s->out, &s->cap_out, (size_t)s->n_bytes + 4096);
    rc = bsc_block_bcf_again(amd_bctx, s->out, s->cap_out, &s->n_bytes, &s->n_rec);
  }
  if (rc < 0) amd_bdie("bsc_blocks_bcf_fetch"); /* BSC_ERR_ARG here = one of the original's asserts (:186,188) on a block */
  if (rc == BSC_WARN_INEXACT) fprintf(stderr, "bscall_amd: %s\n", bsc_last_error());
  amd_brecords += s->n_rec;
  amd_bbytes += s->n_bytes;
  s->n_blk = s->n_tpl = 0;
  s->n_seq = s->n_ref = s->n_pos = s->n_pad = 0;
  pthread_mutex_lock(&amd_bmu);
  s->to_write = 1;
  pthread_cond_broadcast(&amd_bcv);
  pthread_mutex_unlock(&amd_bmu);
  amd_bflight = -1;
}

/* the batch being filled is handed over (behind the one in flight, which is fetched first: one submission per context) */
static void amd_bcf_flush(void) {
  struct amd_bslot *s = &amd_bslot[amd_bfill];
  if (!s->n_blk) return;
  amd_bcf_collect();
  s->out = amd_bgrow(s->out, &s->cap_out, (size_t)s->n_pos * 128u + 4096u);
  if (bsc_blocks_bcf_submit_inplace(amd_bctx, s->desc, s->n_blk, s->tpl, s->seq, s->n_seq, s->ref, NULL /* dbSNP flags: bsc_dbsnp_flags */, &amd_bvp, amd_bstats,
                                    s->rid, &amd_bids, NULL /* names: bsc_dbsnp_names */, s->out, s->cap_out) < 0)
    amd_bdie("bsc_blocks_bcf_submit_inplace");
  amd_bflight = amd_bfill;
  amd_bfill ^= 1;
  amd_bbatches++;
}

/* One call_genotypes_ML at level 2b.  rid: the contig's index in the header (ctg->vcf_rid, src/print_vcf.c:163). */
static void amd_bcf_call(AMD_WORK_T *const work, int32_t rid, const bsc_template *tpl, uint32_t nr, const uint8_t *seq, uint64_t nbytes, uint32_t x,
                         uint32_t y) {
  const uint32_t sz = y - x + 1;
  const uint64_t pad = ((uint64_t)sz + 63u) & ~(uint64_t)63u;
  struct amd_bslot *s = &amd_bslot[amd_bfill];
  if (s->n_blk && (rid != s->rid || s->n_blk == 65536u || s->n_pad + pad > 0x0f000000ull || (uint64_t)s->n_tpl + nr > 0x7fffffffull)) {
    amd_bcf_flush();
    s = &amd_bslot[amd_bfill];
  }
  if (!s->n_blk) { /* the slot's previous batch was fetched at the last hand-over; written by now? (its `out` is about to be sized for this one) */
    pthread_mutex_lock(&amd_bmu);
    while (s->to_write) pthread_cond_wait(&amd_bcv, &amd_bmu);
    pthread_mutex_unlock(&amd_bmu);
    s->rid = rid;
  }
  const uint64_t pos_with = s->n_pos + sz;
  s->tpl = amd_bgrow_keep(s->tpl, &s->cap_tpl, ((size_t)s->n_tpl + (nr ? nr : 1)) * sizeof *s->tpl, (size_t)s->n_tpl * sizeof *s->tpl, pos_with);
  s->seq = amd_bgrow_keep(s->seq, &s->cap_seq, (size_t)(s->n_seq + nbytes) + 16, (size_t)s->n_seq, pos_with);
  s->ref = amd_bgrow_keep(s->ref, &s->cap_ref, (size_t)(s->n_ref + sz) + 3, (size_t)s->n_ref, pos_with);
  s->desc = amd_bgrow_keep(s->desc, &s->cap_desc, ((size_t)s->n_blk + 1) * sizeof *s->desc, (size_t)s->n_blk * sizeof *s->desc, pos_with);
  memcpy(s->tpl + s->n_tpl, tpl, (size_t)nr * sizeof *tpl);
  for (uint32_t i = 0; i < nr; i++) { /* the reads' places in the batch's joined read buffer */
    s->tpl[s->n_tpl + i].off[0] += s->n_seq;
    s->tpl[s->n_tpl + i].off[1] += s->n_seq;
  }
  memcpy(s->seq + s->n_seq, seq, (size_t)nbytes);
  memcpy(s->ref + s->n_ref, AMD_REF1(work), (size_t)sz + 2); /* work->ref1: codes of x .. y + 2 */
  memset(&s->desc[s->n_blk], 0, sizeof *s->desc);
  s->desc[s->n_blk].x = x;
  s->desc[s->n_blk].y = y;
  s->desc[s->n_blk].nr = nr;
  s->n_blk++;
  s->n_tpl += nr;
  s->n_seq += nbytes;
  s->n_ref += (uint64_t)sz + 2u;
  s->n_pos += sz;
  s->n_pad += pad;
  amd_bcalls++;
  if (s->n_pos >= AMD_BCF_BATCH_POSITIONS) amd_bcf_flush();
  amd_bwait_mprof(work);
}

/* join_calc_threads: the last batch is handed over, fetched and written, the writer ends, the buffers go */
static void amd_bcf_join(void) {
  amd_bcf_flush();
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
    bsc_free_host(s->desc);
    bsc_free_host(s->out);
    memset(s, 0, sizeof *s);
  }
}

#endif /* AMD_BCF_PROTOCOL_H */
