/*
 * overlap_tsan.c — the overlapped glue's thread protocol (integration/amd_overlap_protocol.h) on the CPU alone, for
 * ThreadSanitizer: the same protocol code as integration/call_genotypes_amd_overlap.c, the mock work_t and the three
 * reference-shaped threads of integration/mock_work.h, and STUB bsc_* entries in this file — bsc_blocks_submit_to_inplace starts a
 * thread that fills the gt_vcf[] images of the batch's blocks a little later (the GPU's kernels and copy-out), reading the batch's
 * inputs where they lie as the real entry does, bsc_block_fetch joins it.  Blocks are held back until a batch holds argv[2]
 * positions (default 9 000: a handful of blocks; 0: none held); argv[3] = the largest block (default 5 000 positions; with 6 and
 * 70 000 blocks under a huge threshold a batch reaches the library's 65 536-block limit and must be flushed by the glue itself).
 *
 *   gcc -O1 -g -fsanitize=thread -Iinclude integration/overlap_tsan.c -o /tmp/overlap_tsan -lpthread && /tmp/overlap_tsan
 *
 * The process thread of this program does what the reference's does around each call (src/process_template.c:29-30,116-124,
 * src/process.c:61-68): writes the block's reference codes into work->ref1, queues profiling jobs that read them, calls the
 * glue — and, the moment the call returns, goes on to the next block, i.e. overwrites ref1.  Exit status 0 = every record
 * and every reference code arrived where the print thread expected them, no profiling job saw ref1 change under it (and,
 * under -fsanitize=thread, no data race was reported).  -DAMD_TEST_ROUND2_BUG builds round 2's protocol (no wait for the
 * profiling thread in a call that found no block pending): this program must then fail (tests/test_glue_tsan.py).
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "mock_work.h"
#include "amd_overlap_protocol.h"

/* ---- stub library: just enough of include/bscall_amd.h for the protocol ---- */
struct bsc_context {
  pthread_t th;
  int busy;
  const bsc_block_desc *desc; /* the glue's batch: stays valid until the fetch */
  uint32_t n_blocks;
  const uint8_t *ref;
  gt_vcf *out;
  uint8_t *skip;
  uint64_t *off;
};
static struct bsc_context stub_ctx;
const char *bsc_last_error(void) { return "stub"; }
void *bsc_alloc_host(uint64_t bytes) { return malloc(bytes ? bytes : 1); }
void bsc_free_host(void *p) { free(p); }

static void stub_record(gt_vcf *v, uint32_t pos, uint8_t code) { /* deterministic in (position, reference code) */
  memset(v, 0, sizeof *v);
  v->gtm.counts[0] = pos;
  v->gtm.counts[1] = code;
  v->gtm.mq = (int32_t)(pos % 61u);
  v->gtm.max_gt = (uint8_t)(pos % 10u);
  v->skip = (pos % 17u) == 0;
}

static void *stub_worker(void *arg) {
  struct bsc_context *c = arg;
  usleep(2000); /* the kernels run while the caller prepares the next blocks */
  size_t ref_at = 0;
  for (uint32_t b = 0; b < c->n_blocks; b++) {
    const uint32_t x = c->desc[b].x, sz = c->desc[b].y - x + 1u;
    gt_vcf *v = c->out + c->off[b];
    for (uint32_t i = 0; i < sz; i++) stub_record(v + i, x + i, c->ref[ref_at + i]);
    for (uint32_t i = 0; i < sz; i++) c->skip[c->off[b] + i] = v[i].skip;
    ref_at += (size_t)sz + 2;
  }
  return NULL;
}

/* bsc_blocks_submit_to_inplace: every block's images from a multiple of 64 on, the offsets returned */
int bsc_blocks_submit_to_inplace(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                         uint64_t seq_bytes, const uint8_t *ref, void *out, uint32_t out_stride, uint8_t *skip, uint64_t *block_off) {
  (void)tpl; (void)seq; (void)seq_bytes;
  if (ctx->busy || out_stride != sizeof(gt_vcf) || n_blocks == 0 || n_blocks > 65536u) return BSC_ERR_ARG; /* the real entry's limits */
  uint64_t p = 0;
  for (uint32_t b = 0; b < n_blocks; b++) {
    block_off[b] = p;
    p += ((uint64_t)(blocks[b].y - blocks[b].x + 1u) + 63u) & ~(uint64_t)63u;
  }
  ctx->desc = blocks;
  ctx->n_blocks = n_blocks;
  ctx->ref = ref;
  ctx->out = out;
  ctx->skip = skip;
  ctx->off = block_off;
  ctx->busy = 1;
  return pthread_create(&ctx->th, NULL, stub_worker, ctx) ? BSC_ERR_HIP : BSC_OK;
}

int bsc_block_fetch(bsc_context *ctx, void *out, uint8_t *skip) {
  (void)out; (void)skip;
  if (!ctx->busy) return BSC_ERR_ARG;
  pthread_join(ctx->th, NULL);
  ctx->busy = 0;
  return BSC_OK;
}

int main(int argc, char **argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 40;
  const uint32_t max_sz = argc > 3 ? (uint32_t)atoi(argv[3]) : 5000, min_sz = max_sz > 400 ? 200 : 2;
  amd_overlap_init();
  amd_threshold = argc > 2 ? (uint64_t)atoll(argv[2]) : 9000u; /* a handful of blocks per batch; 0: every block its own batch */
  work_t w;
  mock_work_init(&w);
  amd_ctx = &stub_ctx;
  pthread_t pt, mt;
  pthread_create(&pt, NULL, mock_print_thread, &w);
  pthread_create(&mt, NULL, mock_mprof_thread, &w);
  uint8_t *ref = malloc(max_sz + 3);
  bsc_template tpl[4];
  uint8_t seq[16];
  memset(tpl, 0, sizeof tpl);
  memset(seq, 0, sizeof seq);
  /* what the print thread must have seen at the end */
  work_t expect;
  mock_work_init(&expect);
  uint32_t x = 1000;
  uint64_t s = 88172645463325252ull;
  for (int k = 0; k < nblk; k++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const uint32_t sz = min_sz + (uint32_t)(s % (max_sz - min_sz));
    for (uint32_t i = 0; i < sz + 2; i++) ref[i] = (uint8_t)(1 + ((s >> (i % 40)) + i * 7 + (uint32_t)k) % 4);
    /* the process thread: reference codes of the block into work->ref1, one profiling job per template */
    mock_prepare_block(&w, ref, sz, max_sz > 400 ? 30 + (int)(s % 300) : 1 + (int)(s % 3));
    amd_overlap_call(&w, NULL, tpl, 4, seq, sizeof seq, x, x + sz - 1);
    /* ... and straight on to the next block: ref1 is overwritten at the top of the next iteration */
    expect.ref_hash = mock_fnv(expect.ref_hash, ref, (size_t)sz + 2);
    for (uint32_t i = 0; i < sz; i++) {
      gt_vcf v;
      stub_record(&v, x + i, ref[i]);
      mock_consume(&expect, &v);
    }
    x += sz + 50;
  }
  amd_overlap_join(&w);
  pthread_mutex_lock(&w.print_mutex);
  w.print_end = true;
  pthread_cond_signal(&w.print_cond1);
  pthread_mutex_unlock(&w.print_mutex);
  pthread_mutex_lock(&w.mprof_mutex);
  w.mprof_end = true;
  pthread_cond_signal(&w.mprof_cond1);
  pthread_mutex_unlock(&w.mprof_mutex);
  pthread_join(pt, NULL);
  pthread_join(mt, NULL);
  const int ok = w.hash == expect.hash && w.records == expect.records && w.ref_hash == expect.ref_hash && w.mprof_bad == 0;
  printf("%d blocks: %llu records (expected %llu), record hash %s, reference hash %s, %llu profiling jobs, %llu saw ref1 change under them\n",
         nblk, (unsigned long long)w.records, (unsigned long long)expect.records, w.hash == expect.hash ? "ok" : "DIFFERENT",
         w.ref_hash == expect.ref_hash ? "ok" : "DIFFERENT", (unsigned long long)w.mprof_jobs, (unsigned long long)w.mprof_bad);
  free(ref);
  return ok ? 0 : 1;
}
