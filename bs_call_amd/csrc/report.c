/*
 * report.c — the JSON report of a run (host C): the text output_stats() writes from bs_stats and the per-contig
 * gt_ctg_stats at the end of a run (src/stats.c:19-298; types include/bs_call.h:75-146), byte for byte, from the
 * statistics block this library accumulates on the device (bsc_site_stats), the read-level counters of the stages in
 * front of it (bsc_prep_stats; the reader's filter counts) and the per-contig totals.
 *
 * Faithful to the reference's text including its accidents: no line break between the QCDistributions and
 * VCFFilterStats objects (:91), a coverage / QC object without any entry loses its opening brace (:113-124, :60-69 print
 * the brace only in front of the first entry or when there is none at all), "bq_thread" for the base-quality threshold
 * (:29), the methylation profiles with "%.8g".
 *
 * The reference keeps gt_cov_stats in a hash keyed by coverage and sorts it before printing (:109); here the table is
 * dense (bsc_site_stats.cov, bsc_report.gc), walked in index order: the same rows in the same order.
 */
#include <inttypes.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <time.h>

#include "../../include/bscall_amd.h"

/* bounded appender: counts what it would have written, like snprintf */
typedef struct {
  char *buf;
  size_t cap, len;
} rep_out;

static void rep_puts(rep_out *o, const char *s) {
  const size_t n = strlen(s);
  if (o->len < o->cap) {
    const size_t room = o->cap - o->len;
    memcpy(o->buf + o->len, s, n < room ? n : room);
  }
  o->len += n;
}

static void rep_printf(rep_out *o, const char *fmt, ...) {
  char tmp[512];
  va_list ap;
  va_start(ap, fmt);
  const int n = vsnprintf(tmp, sizeof tmp, fmt, ap);
  va_end(ap);
  if (n > 0) rep_puts(o, tmp); /* every format used here yields far fewer than 512 bytes */
}

/* {"All": a, "Passed": p} spread over five lines at the given depth of tabs, under a key */
static void rep_all_passed(rep_out *o, const char *tabs, const char *key, const uint64_t v[2], const char *tail) {
  rep_printf(o, "%s\"%s\": {\n%s\t\"All\": %" PRIu64 ",\n%s\t\"Passed\": %" PRIu64 "\n%s}%s", tabs, key, tabs, v[0], tabs, v[1],
             tabs, tail);
}

/* QD / MQ vectors: every index with a count, as {"NonVariant": hom, "Variant": het} (src/stats.c:70-90) */
static void rep_hom_het_vector(rep_out *o, const char *key, const uint64_t v[256][2]) {
  rep_printf(o, "\t\t\t\"%s\": ", key);
  char open = '{';
  for (int i = 0; i < 256; i++)
    if (v[i][0] + v[i][1] > 0) {
      rep_printf(o, "%c\n\t\t\t\t\"%d\": {\"NonVariant\": %" PRIu64 ", \"Variant\": %" PRIu64 "}", open, i, v[i][0], v[i][1]);
      open = ',';
    }
  if (open == '{') rep_puts(o, "{");
}

/* one column of the coverage table, twelve entries to a line (src/stats.c:111-188) */
static void rep_cov_column(rep_out *o, const uint64_t cov[BSC_COV_CAP][6], int col) {
  int on_line = 0;
  char open = '{';
  for (uint32_t c = 0; c < BSC_COV_CAP; c++) {
    if (!cov[c][col]) continue;
    if (on_line == 0) {
      rep_printf(o, "%c\n\t\t\t\t", open);
      open = ',';
    } else
      rep_puts(o, ", ");
    rep_printf(o, "\"%" PRIu32 "\": %" PRIu64, c, cov[c][col]);
    on_line = (on_line + 1) % 12;
  }
}

/* 256 phred counts, sixteen to a line; the first block of the reference separates with ", " and ends its lines with a
 * blank, the others with "," then blank or line break (src/stats.c:203-229) */
static void rep_qual_row(rep_out *o, const char *key, const uint64_t q[256], int first_style, const char *tail) {
  rep_printf(o, "\t\t\t\"%s\": [\n\t\t\t\t", key);
  for (int i = 0; i < 255; i++) {
    if (first_style) {
      rep_printf(o, "%" PRIu64 ", ", q[i]);
      if ((i & 15) == 15) rep_puts(o, "\n\t\t\t\t");
    } else {
      rep_printf(o, "%" PRIu64 ",", q[i]);
      rep_puts(o, (i & 15) == 15 ? "\n\t\t\t\t" : " ");
    }
  }
  rep_printf(o, "%" PRIu64 "\n\t\t\t]%s", q[255], tail);
}

static void rep_meth_row(rep_out *o, const char *key, const double m[101], const char *tail) {
  rep_printf(o, "\t\t\t\"%s\": [\n\t\t\t\t", key);
  for (int i = 0; i < 100; i++) {
    rep_printf(o, "%.8g, ", m[i]);
    if ((i & 15) == 15) rep_puts(o, "\n\t\t\t\t");
  }
  rep_printf(o, "%.8g\n\t\t\t]%s", m[100], tail);
}

long bsc_report_json(const bsc_report *r, char *buf, size_t cap) {
  static const char *const mut_names[12] = {"A>C", "A>G", "A>T", "C>A", "C>G", "C>T", "G>A", "G>C", "G>T", "T>A", "T>C", "T>G"};
  static const char *const read_filters[15] = {"Passed",         "Unmapped",    "QC_Flags",       "SecondaryAlignment",
                                               "MateUnmapped",   "Duplicate",   "NoPosition",     "NoMatePosition",
                                               "MismatchContig", "BadOrientation", "LargeInsertSize", "NoSequence",
                                               "LowMAPQ",        "NotCorrectlyAligned", "PairNotFound"};
  static const char *const base_filters[5] = {"Passed", "Trimmed", "Clipped", "Overlapping", "LowQuality"};
  static const char *const vcf_filters[4] = {"q20", "qd2", "fs60", "mq40"}; /* src/init_param.c:15 */
  if (!r || !r->total || (!buf && cap)) return -1;
  const bsc_site_stats *st = r->total;
  rep_out out = {buf, cap, 0};
  rep_out *o = &out;

  rep_printf(o, "{\n\t\"source\": \"bs_call_v2.1, under_conversion=%g, over_conversion=%g, mapq_thresh=%d, bq_thread=%d\",\n",
             r->under_conv, r->over_conv, r->mapq_thresh, r->min_qual);
  int day = r->day, month = r->month, year = r->year;
  if (year == 0) {
    const time_t now = time(NULL);
    struct tm tmv;
    localtime_r(&now, &tmv);
    day = tmv.tm_mday;
    month = tmv.tm_mon + 1;
    year = tmv.tm_year + 1900;
  }
  rep_printf(o, "\t\"date\": \"%02d/%02d/%04d\",\n", day, month, year);

  /* filterStats: reads and bases by the reader's verdict, bases by what pre-processing did to them */
  rep_puts(o, "\t\"filterStats\": {\n\t\t\"ReadLevel\": {\n");
  for (int i = 0; i < 15; i++) {
    if (i && !r->filter_cts[i]) continue;
    rep_printf(o, "%s\t\t\t\"%s\": {\n\t\t\t\t\"Reads\": %" PRIu64 ",\n\t\t\t\t\"Bases\": %" PRIu64 "\n\t\t\t}", i ? ",\n" : "",
               read_filters[i], r->filter_cts[i], r->filter_bases[i]);
  }
  rep_puts(o, "\n\t\t},\n\t\t\"BaseLevel\": {\n");
  for (int i = 0; i < 5; i++) {
    if (i && !r->base_filter[i]) continue;
    rep_printf(o, "%s\t\t\t\"%s\": %" PRIu64, i ? ",\n" : "", base_filters[i], r->base_filter[i]);
  }
  rep_puts(o, "\n\t\t}\n\t},\n\t\"totalStats\": {\n");

  rep_all_passed(o, "\t\t", "SNPS", st->snps, ",\n");
  rep_all_passed(o, "\t\t", "Indels", st->indels, ",\n");
  rep_all_passed(o, "\t\t", "Multiallelic", st->multi, ",\n");
  if (r->have_dbsnp) {
    rep_all_passed(o, "\t\t", "dbSNPSites", st->dbSNP_sites, ",\n");
    rep_all_passed(o, "\t\t", "dbSNPVariantSites", st->dbSNP_var, ",\n");
  }
  rep_all_passed(o, "\t\t", "RefCpG", st->CpG_ref, ",\n");
  rep_all_passed(o, "\t\t", "NonRefCpG", st->CpG_nonref, ",\n");

  rep_puts(o, "\t\t\"QCDistributions\": {\n\t\t\t\"FisherStrand\": ");
  { /* only the heterozygous column is reported for FS (src/stats.c:61-67) */
    char open = '{';
    for (int i = 0; i < 256; i++)
      if (st->fs_stats[i][1] > 0) {
        rep_printf(o, "%c\n\t\t\t\t\"%d\": %" PRIu64, open, i, st->fs_stats[i][1]);
        open = ',';
      }
    if (open == '{') rep_puts(o, "{");
  }
  rep_puts(o, "\n\t\t\t},\n");
  rep_hom_het_vector(o, "QualityByDepth", st->qd_stats);
  rep_puts(o, "\n\t\t\t},\n");
  rep_hom_het_vector(o, "RMSMappingQuality", st->mq_stats);
  rep_puts(o, "\n\t\t\t}\n\t\t},\t\t\"VCFFilterStats\": {\n");
  rep_printf(o, "\t\t\t\"PASS\": {\"NonVariant\": %" PRIu64 ", \"Variant\": %" PRIu64 "}", st->filter_counts[0][0],
             st->filter_counts[1][0]);
  for (int bits = 1; bits < 16; bits++) {
    rep_puts(o, ",\n\t\t\t");
    char sep = '"';
    for (int f = 0; f < 4; f++)
      if (bits >> f & 1) {
        rep_printf(o, "%c%s", sep, vcf_filters[f]);
        sep = ',';
      }
    rep_printf(o, "\": {\"NonVariant\": %" PRIu64 ", \"Variant\": %" PRIu64 "}", st->filter_counts[0][bits],
               st->filter_counts[1][bits]);
  }
  rep_puts(o, "\n\t\t},\n\t\t\"coverage\": {\n");

  static const struct {
    const char *key;
    int col;
  } columns[6] = {{"All", 0}, {"Variant", 1}, {"RefCpG", 2}, {"RefCpGInf", 4}, {"NonRefCpG", 3}, {"NonRefCpGInf", 5}};
  for (int k = 0; k < 6; k++) {
    rep_printf(o, "\t\t\t\"%s\": ", columns[k].key);
    rep_cov_column(o, st->cov, columns[k].col);
    rep_puts(o, "\n\t\t\t},\n");
  }
  rep_puts(o, "\t\t\t\"GC\": ");
  {
    char open = '{';
    for (uint32_t c = 0; c < BSC_COV_CAP; c++) {
      if (!st->cov[c][0]) continue;
      rep_printf(o, "%c\n\t\t\t\t\"%" PRIu32 "\": [\n\t\t\t\t\t", open, c);
      open = ',';
      const uint64_t *g = r->gc ? r->gc + (size_t)c * 101u : NULL;
      for (int i = 0; i < 100; i++) {
        rep_printf(o, "%" PRIu64 ",", g ? g[i] : (uint64_t)0);
        rep_puts(o, (i & 15) == 15 ? "\n\t\t\t\t\t" : " ");
      }
      rep_printf(o, "%" PRIu64 "\n\t\t\t\t]", g ? g[100] : (uint64_t)0);
    }
  }
  rep_puts(o, "\n\t\t\t}\n\t\t},\n\t\t\"quality\": {\n");
  rep_qual_row(o, "All", st->qual[0], 1, ",\n");
  rep_qual_row(o, "Variant", st->qual[1], 0, ",\n");
  rep_qual_row(o, "RefCpG", st->qual[2], 0, ",\n");
  rep_qual_row(o, "NonRefCpG", st->qual[3], 0, "\n");

  rep_puts(o, "\t\t},\n\t\t\"mutations\": {\n");
  for (int m = 0; m < 12; m++)
    rep_printf(o, "\t\t\t\"%s\": { \"All\": %" PRIu64 ", \"Passed\": %" PRIu64 ", \"dbSNPAll\": %" PRIu64 ", \"dbSNPPassed\": %" PRIu64 " }%s\n",
               mut_names[m], st->mut_counts[m][0], st->mut_counts[m][1], st->dbSNP_mut_counts[m][0], st->dbSNP_mut_counts[m][1],
               m < 11 ? "," : "");

  rep_puts(o, "\t\t},\n\t\t\"methylation\": {\n");
  rep_meth_row(o, "AllRefCpg", st->CpG_ref_meth[0], ",\n");
  rep_meth_row(o, "PassedRefCpg", st->CpG_ref_meth[1], ",\n");
  rep_meth_row(o, "AllNonRefCpg", st->CpG_nonref_meth[0], ",\n");
  rep_meth_row(o, "PassedNonRefCpg", st->CpG_nonref_meth[1], "");
  if (r->n_read_profile && r->read_profile) { /* element 0 of the reference's vector is never reported (:272) */
    rep_puts(o, ",\n\t\t\t\"NonCpGreadProfile\": ");
    char open = '[';
    for (uint32_t i = 1; i < r->n_read_profile; i++) {
      const uint64_t *c = r->read_profile + (size_t)i * 4u;
      rep_printf(o, "%c\n\t\t\t\t[ %" PRIu64 ", %" PRIu64 ", %" PRIu64 ", %" PRIu64 " ]", open, c[0], c[1], c[2], c[3]);
      open = ',';
    }
    rep_puts(o, "\n\t\t\t]");
  }

  rep_puts(o, "\n\t\t}\n\t},\n\t\"contigStats\": ");
  {
    char open = '{';
    for (uint32_t i = 0; i < r->n_contigs; i++) {
      const bsc_contig_totals *c = &r->contigs[i];
      if (!c->snps[0]) continue; /* a contig without a single written record is not listed (:284) */
      rep_printf(o, "%c\n\t\t\"%s\": {\n", open, c->name ? c->name : "");
      open = ',';
      rep_all_passed(o, "\t\t\t", "SNPS", c->snps, ",\n");
      rep_all_passed(o, "\t\t\t", "Indels", c->indels, ",\n");
      rep_all_passed(o, "\t\t\t", "Multiallelic", c->multi, ",\n");
      if (r->have_dbsnp) {
        rep_all_passed(o, "\t\t\t", "dbSNPSites", c->dbSNP_sites, ",\n");
        rep_all_passed(o, "\t\t\t", "dbSNPVariantSites", c->dbSNP_var, ",\n");
      }
      rep_all_passed(o, "\t\t\t", "RefCpG", c->CpG_ref, ",\n");
      rep_all_passed(o, "\t\t\t", "NonRefCpG", c->CpG_nonref, "\n\t\t}");
    }
  }
  rep_puts(o, "\n\t}\n}\n");
  if (cap) buf[out.len < cap ? out.len : cap - 1] = '\0';
  return (long)out.len;
}
