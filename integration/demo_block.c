/*
 * demo_block.c — plain-C host program against libbscall_amd.so (gcc only: no hipcc, no Python): what a C host such as
 * bs_call does for a run of blocks, in the two forms of the replacement call_genotypes_ML:
 *
 *   synchronous   per block: bsc_prepare_templates (the process thread's read pre-processing) -> bsc_call_block into a
 *                 gt_vcf[] array -> the records are consumed                      (integration/call_genotypes_amd.c)
 *   overlapped    block k is submitted (bsc_block_submit_to, into one of two pinned gt_vcf[] arrays) and the call
 *                 returns; at the start of the next call (and at the end of the run) block k is fetched and PUBLISHED
 *                 to a consumer thread that drains work->vcf[] in index order on the `ready` flags exactly as the
 *                 reference's print thread does (src/process.c:87-104) — so block k's copy-out and consumption overlap
 *                 block k+1's preparation and submission               (integration/call_genotypes_amd_overlap.c)
 *
 * Both forms must deliver the same bytes: the program compares a running hash of every gt_vcf record in consumption
 * order and fails if they differ.  Then it forms VCF records of the last block and the run statistics, as round 1's
 * demo did.  Synthetic reads stand in for what input_sam.c delivers.
 *
 *   make demo && bs_call_amd/lib/demo_block [positions per block] [coverage] [blocks]
 */
#include <pthread.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include <bscall_amd.h>

#define CHECK(call)                                                          \
  do {                                                                       \
    int rc_ = (call);                                                        \
    if (rc_ < 0) {                                                           \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, bsc_last_error()); \
      exit(1);                                                               \
    }                                                                        \
  } while (0)

static const char *GT_NAME[10] = {"AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT"};

/* `gt_vcf`, include/bs_call.h:162-166: what the calc side publishes per position (208 bytes) */
typedef struct {
  bsc_gt_meth gtm;
  bool ready;
  bool skip;
} gt_vcf;
_Static_assert(sizeof(gt_vcf) == 208, "gt_vcf is 208 bytes: the out_stride of the gt_vcf[] form");

/* the fields of the reference's work_t the publish protocol uses (include/bs_call.h:230-282) */
typedef struct {
  gt_vcf *vcf;
  int vcf_n;
  uint32_t vcf_x;
  bool print_end;
  pthread_mutex_t print_mutex, vcf_mutex;
  pthread_cond_t print_cond1, print_cond2, vcf_cond;
  /* what this demo's "printer" keeps */
  uint64_t hash, records, covered;
} work_t;

static void consume(work_t *w, const gt_vcf *v) { /* stands in for print_vcf_entry: every byte the printer would read */
  uint64_t q[sizeof v->gtm / 8], h = w->hash;
  memcpy(q, &v->gtm, sizeof q);
  for (size_t i = 0; i < sizeof q / sizeof q[0]; i++) h = (h ^ q[i]) * 1099511628211ull;
  h = (h ^ (unsigned)v->skip) * 1099511628211ull;
  w->hash = h;
  w->records++;
  w->covered += !v->skip;
}

/* print_thread, src/process.c:74-110: wait for a block (vcf_n > 0), take its positions in index order as their `ready`
 * flags appear, then declare the block drained (vcf_n = 0, print_cond2) */
static void *print_thread(void *arg) {
  work_t *w = arg;
  for (;;) {
    pthread_mutex_lock(&w->print_mutex);
    while (!w->vcf_n && !w->print_end) pthread_cond_wait(&w->print_cond1, &w->print_mutex);
    const int n = w->vcf_n;
    pthread_mutex_unlock(&w->print_mutex);
    if (!n) break;
    for (int i = 0; i < n; i++) {
      gt_vcf *v = w->vcf + i;
      if (!__atomic_load_n(&v->ready, __ATOMIC_ACQUIRE)) {
        pthread_mutex_lock(&w->vcf_mutex);
        while (!__atomic_load_n(&v->ready, __ATOMIC_ACQUIRE)) pthread_cond_wait(&w->vcf_cond, &w->vcf_mutex);
        pthread_mutex_unlock(&w->vcf_mutex);
      }
      consume(w, v);
    }
    pthread_mutex_lock(&w->print_mutex);
    w->vcf_n = 0;
    pthread_cond_signal(&w->print_cond2);
    pthread_mutex_unlock(&w->print_mutex);
  }
  return NULL;
}

/* the publish protocol of call_genotypes_ML (src/call_genotypes.c:228-258 and :110-114): wait until the printer has
 * drained the previous block, hand it the array, set the flags, wake it */
static void publish(work_t *w, gt_vcf *arr, uint32_t sz, uint32_t x) {
  pthread_mutex_lock(&w->print_mutex);
  while (w->vcf_n) pthread_cond_wait(&w->print_cond2, &w->print_mutex);
  w->vcf = arr;
  w->vcf_x = x;
  pthread_mutex_unlock(&w->print_mutex);
  for (uint32_t i = 0; i < sz; i++) __atomic_store_n(&arr[i].ready, true, __ATOMIC_RELEASE);
  pthread_mutex_lock(&w->print_mutex);
  w->vcf_n = (int)sz;
  pthread_cond_signal(&w->print_cond1);
  pthread_mutex_unlock(&w->print_mutex);
  pthread_mutex_lock(&w->vcf_mutex);
  pthread_cond_signal(&w->vcf_cond);
  pthread_mutex_unlock(&w->vcf_mutex);
}

/* one block of the run: raw templates as the reader delivers them, prepared templates, extent, reference codes */
typedef struct {
  bsc_raw_template *raw;
  bsc_template *tpl;
  uint8_t *seq_raw, *seq;
  uint64_t seq_raw_used, seq_used;
  uint32_t nt, x, y, sz;
  uint8_t *ref;
} block_t;

static void make_block(block_t *b, uint64_t seed, uint32_t first, uint32_t n, uint32_t cov) {
  const uint64_t max_t = (uint64_t)n * cov / 150u + 64u, seq_cap = max_t * 200u + 1024u;
  bsc_template *t0 = malloc(max_t * sizeof *t0);
  b->seq_raw = malloc(seq_cap);
  const int64_t nt = bsc_synth_reads_host(seed, first, n, cov, 0, t0, max_t, b->seq_raw, seq_cap, &b->seq_raw_used);
  if (nt <= 0) { fprintf(stderr, "no reads generated\n"); exit(1); }
  b->nt = (uint32_t)nt;
  b->raw = calloc((size_t)nt, sizeof *b->raw);
  for (int64_t i = 0; i < nt; i++) { /* as get_next_align_details leaves them: no indels here, the span is the length */
    for (int k = 0; k < 2; k++) {
      b->raw[i].pos[k] = t0[i].pos[k];
      b->raw[i].len[k] = b->raw[i].reference_span[k] = t0[i].len[k];
      b->raw[i].off[k] = t0[i].off[k];
      b->raw[i].mapq[k] = t0[i].mapq[k];
    }
    b->raw[i].orientation = t0[i].orientation;
    b->raw[i].bs_strand = t0[i].bs_strand;
  }
  free(t0);
  /* the process thread's pre-processing (src/process_template.c:36-111): trims, soft clips, mate overlap, indels */
  b->tpl = malloc((size_t)nt * sizeof *b->tpl);
  b->seq = malloc(b->seq_raw_used + 16);
  const bsc_prep_params pp = {{0, 0}, {0, 0}, 20};
  CHECK(bsc_prepare_templates(b->raw, b->nt, b->seq_raw, b->seq_raw_used, NULL, 0, &pp, b->tpl, b->seq, b->seq_raw_used + 16,
                              &b->seq_used, NULL));
  b->x = bsc_block_start(b->raw);
  b->y = b->x;
  for (uint32_t i = 0; i < b->nt; i++)
    for (int k = 0; k < 2; k++)
      if (b->tpl[i].len[k] && b->tpl[i].pos[k] + b->tpl[i].len[k] - 1 > b->y) b->y = b->tpl[i].pos[k] + b->tpl[i].len[k] - 1;
  b->sz = b->y - b->x + 1;
  /* reference codes of x .. y+2 (work->ref1): the synthetic genome of the generators */
  bsc_pileup *scratch = malloc((size_t)(b->sz + 2) * sizeof *scratch);
  b->ref = malloc(b->sz + 2);
  CHECK(bsc_synth_pileup_host(seed, b->x, b->sz + 2, 0, 0, scratch, b->ref));
  free(scratch);
}

static double now(void) {
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 200000u;
  const uint32_t cov = argc > 2 ? (uint32_t)atol(argv[2]) : 30u;
  const int nblk = argc > 3 ? atoi(argv[3]) : 3;
  const uint64_t seed = 88172645463325252ull;
  block_t *blk = calloc((size_t)nblk, sizeof *blk);
  uint32_t max_sz = 0;
  for (int k = 0; k < nblk; k++) {
    make_block(&blk[k], seed, 10000u + (uint32_t)k * (n + 5000u), n, cov); /* blocks a coverage gap apart */
    if (blk[k].sz > max_sz) max_sz = blk[k].sz;
  }

  bsc_context *ctx = NULL;
  bsc_params par;
  bsc_params_default(&par);
  CHECK(bsc_create(&par, &ctx)); /* init_calc_threads + fill_base_prob_table */

  /* two page-locked gt_vcf arrays: work->vcf alternates between them in the overlapped form */
  gt_vcf *arr[2] = {bsc_alloc_host((uint64_t)max_sz * sizeof(gt_vcf)), bsc_alloc_host((uint64_t)max_sz * sizeof(gt_vcf))};
  uint8_t *skp[2] = {bsc_alloc_host(max_sz), bsc_alloc_host(max_sz)};
  if (!arr[0] || !arr[1] || !skp[0] || !skp[1]) { fprintf(stderr, "%s\n", bsc_last_error()); return 1; }

  /* ---- synchronous form: call, then consume ---- */
  work_t ws;
  memset(&ws, 0, sizeof ws);
  ws.hash = 1469598103934665603ull;
  double t0 = now();
  for (int k = 0; k < nblk; k++) {
    const block_t *b = &blk[k];
    CHECK(bsc_call_block(ctx, b->tpl, b->nt, b->seq, b->seq_used, b->x, b->y, b->ref, arr[0], sizeof(gt_vcf), skp[0]));
    for (uint32_t i = 0; i < b->sz; i++) consume(&ws, arr[0] + i);
  }
  const double t_sync = now() - t0;

  /* ---- overlapped form: submit block k, publish block k-1 to the consumer thread meanwhile ---- */
  work_t wo;
  memset(&wo, 0, sizeof wo);
  wo.hash = 1469598103934665603ull;
  pthread_mutex_init(&wo.print_mutex, NULL);
  pthread_mutex_init(&wo.vcf_mutex, NULL);
  pthread_cond_init(&wo.print_cond1, NULL);
  pthread_cond_init(&wo.print_cond2, NULL);
  pthread_cond_init(&wo.vcf_cond, NULL);
  pthread_t pt;
  pthread_create(&pt, NULL, print_thread, &wo);
  t0 = now();
  int in_flight = -1; /* block whose records are on their way into arr[in_flight & 1] */
  for (int k = 0; k <= nblk; k++) {
    if (in_flight >= 0) { /* start of call k (or join_calc_threads): block k-1 has to be complete before it is published */
      CHECK(bsc_block_fetch(ctx, NULL, NULL));
      publish(&wo, arr[in_flight & 1], blk[in_flight].sz, blk[in_flight].x);
      in_flight = -1;
    }
    if (k == nblk) break;
    const block_t *b = &blk[k];
    /* arr[k & 1] was published two calls ago: the printer has drained it before the previous publish returned */
    for (uint32_t i = 0; i < b->sz; i++) arr[k & 1][i].ready = false;
    CHECK(bsc_block_submit_to(ctx, b->tpl, b->nt, b->seq, b->seq_used, b->x, b->y, b->ref, arr[k & 1], sizeof(gt_vcf), skp[k & 1]));
    in_flight = k; /* returns at once: the process thread goes on to prepare block k+1 */
  }
  pthread_mutex_lock(&wo.print_mutex);
  while (wo.vcf_n) pthread_cond_wait(&wo.print_cond2, &wo.print_mutex);
  wo.print_end = true;
  pthread_cond_signal(&wo.print_cond1);
  pthread_mutex_unlock(&wo.print_mutex);
  pthread_join(pt, NULL);
  const double t_over = now() - t0;
  printf("%d blocks, %llu positions: synchronous %.1f ms, overlapped %.1f ms; consumer saw %llu / %llu records, hash %016llx / %016llx\n",
         nblk, (unsigned long long)ws.records, t_sync * 1e3, t_over * 1e3, (unsigned long long)ws.records,
         (unsigned long long)wo.records, (unsigned long long)ws.hash, (unsigned long long)wo.hash);
  if (ws.hash != wo.hash || ws.records != wo.records || ws.covered != wo.covered) {
    fprintf(stderr, "the overlapped form delivered different records\n");
    return 1;
  }

  /* ---- the last block once more: gt_meth, VCF record fields, run statistics (as the print thread derives them) ---- */
  const block_t *b = &blk[nblk - 1];
  const uint32_t sz = b->sz;
  bsc_gt_meth *gtm = bsc_alloc_host((uint64_t)sz * sizeof *gtm);
  uint8_t *skip = bsc_alloc_host(sz);
  bsc_vcf_core *vcf = bsc_alloc_host((uint64_t)sz * sizeof *vcf);
  if (!gtm || !skip || !vcf) { fprintf(stderr, "%s\n", bsc_last_error()); return 1; }
  CHECK(bsc_reset_stats(ctx));
  CHECK(bsc_call_block(ctx, b->tpl, b->nt, b->seq, b->seq_used, b->x, b->y, b->ref, gtm, sizeof *gtm, skip)); /* call_genotypes_ML */
  bsc_vcf_params vp = {0, 1, 0xffffffffu};
  CHECK(bsc_vcf_records(ctx, gtm, sizeof *gtm, skip, b->ref, NULL, sz, b->x, &vp, vcf)); /* _print_vcf_entry, up to htslib */
  uint64_t emitted = 0, hets = 0;
  int shown = 0;
  for (uint32_t i = 0; i < sz; i++) {
    const bsc_vcf_core *c = vcf + i;
    if (!c->emit) continue;
    emitted++;
    const int het = GT_NAME[c->gt][0] != GT_NAME[c->gt][1];
    hets += het;
    if (het && shown < 5) {
      char line[1024];
      if (bsc_vcf_format(c, gtm + i, "chrS", NULL, line, sizeof line) > 0) puts(line);
      shown++;
    }
  }
  /* the printer's run statistics (bs_stats): accumulated on the device block by block, read once at the end */
  CHECK(bsc_vcf_stats(ctx, vcf, gtm, sizeof *gtm, NULL, sz));
  bsc_site_stats *ss = malloc(sizeof *ss);
  CHECK(bsc_get_site_stats(ctx, ss));
  double meth_mean = 0.0, meth_n = 0.0;
  for (int i = 0; i < 101; i++) {
    meth_mean += i * (ss->CpG_ref_meth[0][i] + ss->CpG_nonref_meth[0][i]);
    meth_n += ss->CpG_ref_meth[0][i] + ss->CpG_nonref_meth[0][i];
  }
  printf("statistics: %llu records (%llu PASS), %llu reference CpGs, %llu non-reference CpGs, mean CpG methylation %.1f %%\n",
         (unsigned long long)ss->snps[0], (unsigned long long)ss->snps[1], (unsigned long long)ss->CpG_ref[0],
         (unsigned long long)ss->CpG_nonref[0], meth_n > 0 ? meth_mean / meth_n : 0.0);
  if (ss->snps[0] != emitted) { fprintf(stderr, "statistics disagree with the records\n"); return 1; }
  free(ss);
  bsc_stats st;
  CHECK(bsc_get_stats(ctx, &st));
  printf("block %u..%u: %u templates, %llu bases -> %llu positions called (%llu covered), %llu VCF records, %llu het\n", b->x, b->y,
         b->nt, (unsigned long long)b->seq_used, (unsigned long long)st.sites, (unsigned long long)st.covered,
         (unsigned long long)emitted, (unsigned long long)hets);
  bsc_free_host(gtm);
  bsc_free_host(skip);
  bsc_free_host(vcf);
  for (int k = 0; k < 2; k++) {
    bsc_free_host(arr[k]);
    bsc_free_host(skp[k]);
  }
  bsc_destroy(ctx); /* join_calc_threads */
  return emitted > 0 ? 0 : 1;
}
