/*
 * bcf_tsan.c — the bytes form's thread protocol (integration/amd_bcf_protocol.h: the code integration/call_genotypes_amd_bcf.c is made
 * of) on the CPU alone, for ThreadSanitizer: the mock work_t and its profiling thread (integration/mock_work.h), and STUB bsc_* entries
 * in this file — bsc_blocks_bcf_submit_inplace starts a thread that, a little later (the GPU's kernels and copy-out), reads the batch's
 * inputs WHERE THEY LIE, as the real entry does, and writes a "stream" that depends on every input byte into the caller's buffer;
 * bsc_blocks_bcf_fetch joins it.  Every third submission answers that its stream is longer than the buffer (the glue must then ask for
 * the encoder once more, with the room asked for).  The process thread overwrites work->ref1 and its own template / read buffers the moment a
 * call returns.  Exit status 0 = the writer saw the expected bytes in order, no profiling job saw ref1 change under it.
 *
 *   gcc -O1 -g -fsanitize=thread -Iinclude -Iintegration integration/bcf_tsan.c -o /tmp/bcf_tsan -lpthread && /tmp/bcf_tsan
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>

#include "mock_work.h"
#define AMD_BCF_WRITE(work, buf, n)                                    \
  do {                                                                 \
    (work)->bcf_hash = mock_fnv((work)->bcf_hash, (buf), (size_t)(n)); \
    (work)->bcf_bytes += (n);                                          \
  } while (0)
#define AMD_BCF_BATCH_POSITIONS 9000u /* blocks of 200 .. 5 000 positions: batches of one and of several blocks */
#include "amd_bcf_protocol.h"

struct bsc_context {
  pthread_t th;
  int busy, overflow;
  const bsc_block_desc *desc;
  uint32_t n_blocks;
  const bsc_template *tpl;
  const uint8_t *seq, *ref;
  uint64_t seq_bytes, cap, bytes, recs;
  uint8_t *out;
};
static struct bsc_context stub_ctx;
static unsigned stub_submits;
const char *bsc_last_error(void) { return "stub"; }
void *bsc_alloc_host(uint64_t bytes) { return malloc(bytes ? bytes : 1); }
void bsc_free_host(void *p) { free(p); }
void bsc_bcf_default_ids(bsc_bcf_ids *ids) { memset(ids, 0, sizeof *ids); }

/* a block's "stream": 3 bytes per position from (position, reference code, a digest of its templates' positions and of its reads' bytes — found
 * through the templates' offsets into the joined read buffer, as the library finds them); a long batch: 200 bytes per position (longer than the
 * glue's first guess of 128) */
static uint64_t stub_len(uint32_t x, uint32_t y, int long_one) { return (uint64_t)(y - x + 1) * (long_one ? 200u : 3u); }
static void stub_stream(const bsc_template *tpl, uint32_t nr, const uint8_t *seq, const uint8_t *ref, uint32_t x, uint32_t y, int long_one, uint8_t *out) {
  uint64_t d = 1469598103934665603ull;
  for (uint32_t i = 0; i < nr; i++) {
    d = mock_fnv(d, tpl[i].pos, sizeof tpl[i].pos);
    d = mock_fnv(d, seq + tpl[i].off[0], tpl[i].len[0]);
  }
  const uint32_t per = long_one ? 200u : 3u;
  for (uint32_t i = 0; i <= y - x; i++)
    for (uint32_t k = 0; k < per; k++) out[(size_t)i * per + k] = (uint8_t)((x + i) * 31u + ref[i] * 7u + k + (uint32_t)(d >> (k % 56)));
}
/* the batch's stream: its blocks' streams one after another (bsc_blocks_bcf) */
static uint64_t stub_batch(const struct bsc_context *c, int long_one, uint8_t *out) {
  uint64_t o = 0, r = 0;
  uint32_t t = 0;
  for (uint32_t b = 0; b < c->n_blocks; b++) {
    const bsc_block_desc *k = &c->desc[b];
    if (out) stub_stream(c->tpl + t, k->nr, c->seq, c->ref + r, k->x, k->y, long_one, out + o);
    o += stub_len(k->x, k->y, long_one);
    r += (uint64_t)(k->y - k->x + 1) + 2u;
    t += k->nr;
  }
  return o;
}

static void *stub_worker(void *arg) {
  struct bsc_context *c = arg;
  usleep(1500);
  c->bytes = stub_batch(c, c->overflow, NULL);
  c->recs = 0;
  for (uint32_t b = 0; b < c->n_blocks; b++) c->recs += c->desc[b].y - c->desc[b].x + 1;
  if (c->bytes <= c->cap) (void)stub_batch(c, c->overflow, c->out);
  return NULL;
}

int bsc_blocks_bcf_submit_inplace(bsc_context *ctx, const bsc_block_desc *blocks, uint32_t n_blocks, const bsc_template *tpl, const uint8_t *seq,
                                  uint64_t seq_bytes, const uint8_t *ref, const uint8_t *dbsnp, const bsc_vcf_params *params, int with_stats, int32_t rid,
                                  const bsc_bcf_ids *ids, const bsc_bcf_names *names, uint8_t *out, uint64_t out_cap) {
  (void)dbsnp; (void)params; (void)with_stats; (void)rid; (void)ids; (void)names;
  if (ctx->busy) return BSC_ERR_ARG; /* one submission in flight per context */
  ctx->desc = blocks; ctx->n_blocks = n_blocks; ctx->tpl = tpl; ctx->seq = seq; ctx->seq_bytes = seq_bytes; ctx->ref = ref;
  ctx->out = out; ctx->cap = out_cap;
  ctx->overflow = (stub_submits++ % 3u) == 2u;
  ctx->busy = 1;
  return pthread_create(&ctx->th, NULL, stub_worker, ctx) ? BSC_ERR_HIP : BSC_OK;
}

int bsc_blocks_bcf_fetch(bsc_context *ctx, uint64_t *n_bytes, uint64_t *n_records) {
  if (!ctx->busy) return BSC_ERR_ARG;
  pthread_join(ctx->th, NULL);
  ctx->busy = 0;
  *n_bytes = ctx->bytes;
  *n_records = ctx->recs;
  return ctx->bytes > ctx->cap ? BSC_ERR_ARG : BSC_OK;
}

int bsc_block_bcf_again(bsc_context *ctx, uint8_t *out, uint64_t out_cap, uint64_t *n_bytes, uint64_t *n_records) {
  if (ctx->busy || ctx->bytes <= ctx->cap) return BSC_ERR_ARG; /* only after a refusal */
  *n_bytes = ctx->bytes;
  *n_records = ctx->recs;
  if (*n_bytes > out_cap) return BSC_ERR_ARG;
  (void)stub_batch(ctx, 1, out); /* (the real one: from what the batch left in HBM) */
  return BSC_OK;
}

/* The expected stream: the stub makes every third SUBMISSION long, so the harness mirrors the glue's batching rule (a batch is handed over when
 * it holds AMD_BCF_BATCH_POSITIONS positions — set small below, so that batches of several blocks, of one block, and blocks larger than a batch
 * all occur) to know which blocks are written long. */
int main(int argc, char **argv) {
  const int nblk = argc > 1 ? atoi(argv[1]) : 40;
  const uint32_t max_sz = 5000, min_sz = 200;
  work_t w, expect;
  mock_work_init(&w);
  mock_work_init(&expect);
  pthread_t mt;
  pthread_create(&mt, NULL, mock_mprof_thread, &w);
  amd_bcf_init(&w, &stub_ctx);
  uint8_t *ref = malloc(max_sz + 3), *tmp = malloc((size_t)max_sz * 200u);
  bsc_template tpl[4];
  uint8_t seq[64];
  uint32_t x = 1000;
  uint64_t s = 88172645463325252ull;
  /* pass 1: the blocks' sizes, to know the batches; pass 2 (same generator): the calls and the expected bytes */
  uint32_t *sizes = malloc((size_t)nblk * sizeof *sizes);
  int *long_blk = malloc((size_t)nblk * sizeof *long_blk);
  {
    uint64_t g = s, pos = 0;
    unsigned sub = 0;
    int first = 0;
    for (int k = 0; k < nblk; k++) {
      g ^= g << 13; g ^= g >> 7; g ^= g << 17;
      sizes[k] = min_sz + (uint32_t)(g % (max_sz - min_sz));
      pos += sizes[k];
      if (pos >= AMD_BCF_BATCH_POSITIONS || k + 1 == nblk) {
        for (int j = first; j <= k; j++) long_blk[j] = (sub % 3u) == 2u;
        sub++;
        first = k + 1;
        pos = 0;
      }
    }
  }
  for (int k = 0; k < nblk; k++) {
    s ^= s << 13; s ^= s >> 7; s ^= s << 17;
    const uint32_t sz = sizes[k];
    for (uint32_t i = 0; i < sz + 2; i++) ref[i] = (uint8_t)(1 + ((s >> (i % 40)) + i * 7 + (uint32_t)k) % 4);
    memset(tpl, 0, sizeof tpl); /* the caller's own buffers: rewritten for every block, as the glue's flatten step does */
    for (int i = 0; i < 4; i++) {
      tpl[i].pos[0] = x + (uint32_t)i * 10u + (uint32_t)(s % 7u);
      tpl[i].len[0] = 16;
      tpl[i].off[0] = 16u * (uint32_t)i;
    }
    for (size_t i = 0; i < sizeof seq; i++) seq[i] = (uint8_t)(s >> (i % 57));
    mock_prepare_block(&w, ref, sz, 30 + (int)(s % 300));
    stub_stream(tpl, 4, seq, ref, x, x + sz - 1, long_blk[k], tmp);
    AMD_BCF_WRITE(&expect, tmp, stub_len(x, x + sz - 1, long_blk[k]));
    amd_bcf_call(&w, 0, tpl, 4, seq, sizeof seq, x, x + sz - 1);
    x += sz + 50;
  }
  amd_bcf_join();
  pthread_mutex_lock(&w.mprof_mutex);
  w.mprof_end = true;
  pthread_cond_signal(&w.mprof_cond1);
  pthread_mutex_unlock(&w.mprof_mutex);
  pthread_join(mt, NULL);
  const int ok = w.bcf_hash == expect.bcf_hash && w.bcf_bytes == expect.bcf_bytes && w.mprof_bad == 0;
  printf("%d blocks in %u batches: %llu bytes (expected %llu), stream hash %s, %llu profiling jobs, %llu saw ref1 change under them\n", nblk, amd_bbatches,
         (unsigned long long)w.bcf_bytes, (unsigned long long)expect.bcf_bytes, w.bcf_hash == expect.bcf_hash ? "ok" : "DIFFERENT",
         (unsigned long long)w.mprof_jobs, (unsigned long long)w.mprof_bad);
  free(ref);
  free(tmp);
  free(sizes);
  free(long_blk);
  return ok ? 0 : 1;
}
