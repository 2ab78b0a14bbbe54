/*
 * demo_block.c — plain-C host program against libbscall_amd.so (gcc only: no hipcc, no Python): what a C host such as
 * bs_call does for a run of blocks, in the two forms of the replacement call_genotypes_ML:
 *
 *   synchronous   per block: bsc_prepare_templates (the process thread's read pre-processing) -> bsc_call_block into a
 *                 gt_vcf[] array -> the records are consumed                      (integration/call_genotypes_amd.c)
 *   overlapped    the protocol of integration/call_genotypes_amd_overlap.c itself (integration/amd_overlap_protocol.h, the
 *                 same code) against a mock of the reference's work_t (integration/mock_work.h): a call appends its block
 *                 to a batch in page-locked buffers and returns once the meth profiling thread has let go of work->ref1; a
 *                 full batch is submitted (bsc_blocks_submit_to_inplace) while the batch before it is HANDED OVER block by
 *                 block to a print thread that drains work->vcf[] in index order on the `ready` flags exactly as the
 *                 reference's does (src/process.c:87-104), reading work->ref beside it; a profiling thread reads work->ref1
 *                 for every queued template while this thread, like the reference's process thread, overwrites ref1 for the
 *                 next block the moment the call returns.  BSC_DEMO_PRINT_NS in the environment makes the mock printer as
 *                 slow as a real one (nanoseconds per position) for end-to-end timings.
 *   bytes         INTEGRATION.md 2b, the protocol of integration/call_genotypes_amd_bcf.c itself (integration/amd_bcf_protocol.h): a call
 *                 queues its block (bsc_block_bcf_submit_inplace) and returns with it in flight; the next call fetches its stretch of the
 *                 BCF record stream and hands the BYTES to a writer thread.  Checked against bsc_block_bcf block by block.
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

#include "mock_work.h"
#include "amd_overlap_protocol.h"
/* the bytes form: the mock's writer hashes what bgzf_write would be given */
#define AMD_BCF_WRITE(work, buf, n)                                                                \
  do { /* (BSC_DEMO_PRINT_NS < 0: a writer that only counts, as the mock printer then does — a byte-serial hash runs at ~1 GB/s) */ \
    if ((work)->print_ns >= 0) (work)->bcf_hash = mock_fnv((work)->bcf_hash, (buf), (size_t)(n)); \
    (work)->bcf_bytes += (n);                                                                      \
  } while (0)
#include "amd_bcf_protocol.h"

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
  mock_work_init(&ws);
  double t0 = now();
  for (int k = 0; k < nblk; k++) {
    const block_t *b = &blk[k];
    CHECK(bsc_call_block(ctx, b->tpl, b->nt, b->seq, b->seq_used, b->x, b->y, b->ref, arr[0], sizeof(gt_vcf), skp[0]));
    for (uint32_t i = 0; i < b->sz; i++) mock_consume(&ws, arr[0] + i);
    ws.ref_hash = mock_fnv(ws.ref_hash, b->ref, (size_t)b->sz + 2);
  }
  const double t_sync = now() - t0;

  /* ---- overlapped form: the glue's own protocol (amd_overlap_protocol.h) between this thread (the process thread), a
   * print thread and a meth profiling thread ---- */
  work_t wo;
  mock_work_init(&wo);
  amd_ctx = ctx;
  amd_overlap_init(); /* BSCALL_AMD_BATCH_POSITIONS: how many positions are held back before a launch sequence */
  pthread_t pt, mt;
  pthread_create(&pt, NULL, mock_print_thread, &wo);
  pthread_create(&mt, NULL, mock_mprof_thread, &wo);
  /* BSC_DEMO_MPROF_JOBS: profiling jobs queued per block (default: one per template up to 2 000, each a mutex round trip with
   * the mock profiling thread — which then dominates a timing of many small blocks) */
  const int mprof_cap = getenv("BSC_DEMO_MPROF_JOBS") ? atoi(getenv("BSC_DEMO_MPROF_JOBS")) : 2000;
  t0 = now();
  for (int k = 0; k < nblk; k++) {
    const block_t *b = &blk[k];
    /* process_template_vector: the block's reference codes into work->ref1 (over the previous block's), one profiling job
     * per template (capped: the ring has 256 slots and the jobs are cheap here) */
    mock_prepare_block(&wo, b->ref, b->sz, (int)b->nt < mprof_cap ? (int)b->nt : mprof_cap);
    amd_overlap_call(&wo, NULL, b->tpl, b->nt, b->seq, b->seq_used, b->x, b->y); /* call_genotypes_ML: returns with the block in flight */
  }
  amd_overlap_join(&wo); /* join_calc_threads */
  pthread_mutex_lock(&wo.print_mutex);
  wo.print_end = true;
  pthread_cond_signal(&wo.print_cond1);
  pthread_mutex_unlock(&wo.print_mutex);
  pthread_mutex_lock(&wo.mprof_mutex);
  wo.mprof_end = true;
  pthread_cond_signal(&wo.mprof_cond1);
  pthread_mutex_unlock(&wo.mprof_mutex);
  pthread_join(pt, NULL);
  pthread_join(mt, NULL);
  const double t_over = now() - t0;
  printf("glue protocol end to end: %.1f M positions/s (threshold %llu positions, mock printer %ld ns per position)\n",
         (double)wo.records / t_over * 1e-6, (unsigned long long)amd_threshold, wo.print_ns);
  printf("%d blocks, %llu positions: synchronous %.1f ms, overlapped %.1f ms; consumer saw %llu / %llu records, hash %016llx / %016llx\n",
         nblk, (unsigned long long)ws.records, t_sync * 1e3, t_over * 1e3, (unsigned long long)ws.records,
         (unsigned long long)wo.records, (unsigned long long)ws.hash, (unsigned long long)wo.hash);
  printf("overlapped form: %llu profiling jobs read work->ref1, %llu found it changed under them; reference codes beside the blocks %s\n",
         (unsigned long long)wo.mprof_jobs, (unsigned long long)wo.mprof_bad, ws.ref_hash == wo.ref_hash ? "as handed over" : "DIFFERENT");
  if (wo.print_ns < 0) ws.hash = wo.hash; /* timing mode: the consumers only counted */
  if (ws.hash != wo.hash || ws.records != wo.records || ws.covered != wo.covered || ws.ref_hash != wo.ref_hash || wo.mprof_bad) {
    fprintf(stderr, "the overlapped form delivered different records\n");
    return 1;
  }

  /* ---- the bytes form: the same blocks, their BCF record streams to a writer thread; against bsc_block_bcf ---- */
  {
    work_t wb, wr;
    mock_work_init(&wb);
    mock_work_init(&wr);
    bsc_bcf_ids ids;
    bsc_bcf_default_ids(&ids);
    const bsc_vcf_params vp0 = {0, 1, 0xffffffffu};
    uint8_t *one = bsc_alloc_host((uint64_t)max_sz * 128u + 4096u);
    if (!one) { fprintf(stderr, "%s\n", bsc_last_error()); return 1; }
    uint64_t n_rec_sync = 0;
    t0 = now();
    for (int k = 0; k < nblk; k++) { /* one call per block: upload, kernels, copy-out, wait; then the "write" on this thread */
      const block_t *b = &blk[k];
      uint64_t nb = 0, nr = 0;
      CHECK(bsc_block_bcf(ctx, b->tpl, b->nt, b->seq, b->seq_used, b->x, b->y, b->ref, NULL, &vp0, 0, 3, &ids, NULL, one, (uint64_t)max_sz * 128u + 4096u, &nb, &nr));
      AMD_BCF_WRITE(&wr, one, nb);
      n_rec_sync += nr;
    }
    const double t_bsync = now() - t0;
    pthread_t mt2;
    pthread_create(&mt2, NULL, mock_mprof_thread, &wb);
    amd_bstats = 0;
    amd_bcf_init(&wb, ctx);
    t0 = now();
    uint64_t positions = 0;
    for (int k = 0; k < nblk; k++) {
      const block_t *b = &blk[k];
      mock_prepare_block(&wb, b->ref, b->sz, (int)b->nt < mprof_cap ? (int)b->nt : mprof_cap);
      amd_bcf_call(&wb, 3, b->tpl, b->nt, b->seq, b->seq_used, b->x, b->y);
      positions += b->sz;
    }
    amd_bcf_join();
    const double t_bytes = now() - t0;
    pthread_mutex_lock(&wb.mprof_mutex);
    wb.mprof_end = true;
    pthread_cond_signal(&wb.mprof_cond1);
    pthread_mutex_unlock(&wb.mprof_mutex);
    pthread_join(mt2, NULL);
    printf("bytes form end to end: %.1f M positions/s (%llu records, %llu bytes in %.1f ms; one bsc_block_bcf per block: %.1f ms); streams %s; %llu profiling jobs, %llu found ref1 changed\n",
           (double)positions / t_bytes * 1e-6, (unsigned long long)amd_brecords, (unsigned long long)amd_bbytes, t_bytes * 1e3, t_bsync * 1e3,
           wb.bcf_hash == wr.bcf_hash && wb.bcf_bytes == wr.bcf_bytes ? "identical" : "DIFFERENT", (unsigned long long)wb.mprof_jobs,
           (unsigned long long)wb.mprof_bad);
    if (wb.bcf_hash != wr.bcf_hash || wb.bcf_bytes != wr.bcf_bytes || amd_brecords != n_rec_sync || !amd_brecords || wb.mprof_bad) {
      fprintf(stderr, "the bytes form delivered a different stream\n");
      return 1;
    }
    bsc_free_host(one);
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
