/*
 * demo_block.c — plain-C caller of libbscall_amd.so (built with gcc, no hipcc, no Python): the calls a C host such
 * as bs_call makes for one block — reads in, gt_meth out, VCF record fields out.  Synthetic reads stand in for what
 * input_sam.c / process_template.c deliver.  Prints a few VCF data lines and a per-block summary.
 *
 *   make demo && bs_call_amd/lib/demo_block [positions] [coverage]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <bscall_amd.h>

#define CHECK(call)                                                        \
  do {                                                                     \
    int rc_ = (call);                                                      \
    if (rc_ < 0) {                                                         \
      fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, bsc_last_error()); \
      return 1;                                                            \
    }                                                                      \
  } while (0)

static const char *GT_NAME[10] = {"AA", "AC", "AG", "AT", "CC", "CG", "CT", "GG", "GT", "TT"};

int main(int argc, char **argv) {
  const uint32_t n = argc > 1 ? (uint32_t)atol(argv[1]) : 200000u;
  const uint32_t cov = argc > 2 ? (uint32_t)atol(argv[2]) : 30u;
  const uint64_t seed = 88172645463325252ull;
  const uint32_t first = 10000u;

  /* the block's reads (what process_template_vector hands to call_genotypes_ML) */
  const uint64_t max_t = (uint64_t)n * cov / 150u + 64u, seq_cap = max_t * 200u + 1024u;
  bsc_template *tpl = malloc(max_t * sizeof *tpl);
  uint8_t *seq = malloc(seq_cap);
  uint64_t seq_used = 0;
  const int64_t nt = bsc_synth_reads_host(seed, first, n, cov, 0, tpl, max_t, seq, seq_cap, &seq_used);
  if (nt <= 0) { fprintf(stderr, "no reads generated\n"); return 1; }
  uint32_t x = first - 2, y = first;
  for (int64_t i = 0; i < nt; i++)
    for (int k = 0; k < 2; k++)
      if (tpl[i].len[k] && tpl[i].pos[k] + tpl[i].len[k] - 1 > y) y = tpl[i].pos[k] + tpl[i].len[k] - 1;
  const uint32_t sz = y - x + 1;

  /* reference codes of x .. y+2 (work->ref1): the synthetic genome of the generators */
  bsc_pileup *scratch = malloc((size_t)(sz + 2) * sizeof *scratch);
  uint8_t *ref = malloc(sz + 2);
  CHECK(bsc_synth_pileup_host(seed, x, sz + 2, 0, 0, scratch, ref));
  free(scratch);

  bsc_context *ctx = NULL;
  bsc_params par;
  bsc_params_default(&par);
  CHECK(bsc_create(&par, &ctx)); /* init_calc_threads + fill_base_prob_table */

  /* page-locked result arrays, as the glue would allocate work->vcf */
  bsc_gt_meth *gtm = bsc_alloc_host((uint64_t)sz * sizeof *gtm);
  uint8_t *skip = bsc_alloc_host(sz);
  bsc_vcf_core *vcf = bsc_alloc_host((uint64_t)sz * sizeof *vcf);
  if (!gtm || !skip || !vcf) { fprintf(stderr, "%s\n", bsc_last_error()); return 1; }

  CHECK(bsc_call_block(ctx, tpl, (uint32_t)nt, seq, seq_used, x, y, ref, gtm, sizeof *gtm, skip)); /* call_genotypes_ML */
  bsc_vcf_params vp = {0, 1, 0xffffffffu};
  CHECK(bsc_vcf_records(ctx, gtm, sizeof *gtm, skip, ref, NULL, sz, x, &vp, vcf)); /* _print_vcf_entry, up to htslib */

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
  printf("block %u..%u: %lld templates, %llu bases -> %llu positions called (%llu covered), %llu VCF records, %llu het\n", x, y,
         (long long)nt, (unsigned long long)seq_used, (unsigned long long)st.sites, (unsigned long long)st.covered,
         (unsigned long long)emitted, (unsigned long long)hets);
  bsc_free_host(gtm);
  bsc_free_host(skip);
  bsc_free_host(vcf);
  bsc_destroy(ctx); /* join_calc_threads */
  free(tpl);
  free(seq);
  free(ref);
  return emitted > 0 ? 0 : 1;
}
