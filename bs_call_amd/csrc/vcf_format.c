/*
 * vcf_format.c — host-side text rendering of one bsc_vcf_core record (+ its gt_meth) as a VCF data line: the layout
 * htslib's VCF writer gives the fields the reference encodes in _print_vcf_entry (src/print_vcf.c:160-380).  No
 * computation: every number comes from the device records.  Integer fields are exact by construction; the text form of
 * the GL floats ("%g", six significant digits, as htslib prints floats) is not pinned against htslib here.
 */
#include <stdio.h>
#include <string.h>

#include "../../include/bscall_amd.h"

static const char *const FLT_NAMES[4] = {"q20", "qd2", "fs60", "mq40"}; /* src/init_param.c:15 */
static const char GT_A[10] = {'A', 'A', 'A', 'A', 'C', 'C', 'C', 'G', 'G', 'T'};
static const char GT_B[10] = {'A', 'C', 'G', 'T', 'C', 'G', 'T', 'G', 'T', 'T'};

#define PUT(...)                                              \
  do {                                                        \
    int w_ = snprintf(buf + len, cap - len, __VA_ARGS__);     \
    if (w_ < 0 || (size_t)w_ >= cap - len) return -1;         \
    len += (size_t)w_;                                        \
  } while (0)

/* Writes "CHROM POS ID REF ALT QUAL FILTER INFO FORMAT SAMPLE" (tab separated, no newline) into buf.
 * Returns the length, 0 if the record is not emitted (emit == 0), -1 if buf is too small. */
int bsc_vcf_format(const bsc_vcf_core *c, const bsc_gt_meth *g, const char *contig, const char *id, char *buf,
                   size_t cap) {
  if (!c || !g || !buf || cap == 0) return -1;
  if (!c->emit) return 0;
  size_t len = 0;
  const int gt = c->gt < 10 ? c->gt : 0;
  const int het = GT_A[gt] != GT_B[gt];
  PUT("%s\t%u\t%s\t%c\t", contig ? contig : ".", c->pos, (id && *id) ? id : ".", c->cx_ref[2]);
  if (c->alt[0]) {
    PUT("%c", c->alt[0]);
    if (c->alt[1]) PUT(",%c", c->alt[1]);
  } else PUT(".");
  PUT("\t%u\t%s\tCX=%.5s\t", c->phred, c->flt == 0 ? "PASS" : ((c->flt & 128) ? "mac1" : "fail"), c->cx_ref);
  /* FORMAT keys (src/print_vcf.c:268-378): AMQ only when some class is covered, FS only for heterozygous calls */
  int n_amq = 0;
  for (int i = 0; i < 8; i++) n_amq += g->counts[i] > 0;
  PUT("GT:FT:DP:MQ:GQ:QD:GL:MC8%s:CS:CG:CX%s\t", n_amq ? ":AMQ" : "", het ? ":FS" : "");
  PUT("%d/%d:", ((c->gt_enc >> 4) >> 1) - 1, ((c->gt_enc & 15) >> 1) - 1);
  if (c->flt & 15) {
    /* The reference's copy loop leaves every name's terminator in the buffer (`while((*p++ = *p1++));`,
     * src/print_vcf.c:289-293): the BCF field is "q20\0;qd2\0" (csrc/bcf.c writes exactly that) and a VCF writer, which
     * ends a per-sample string at its first NUL, shows the FIRST failed filter only. */
    for (int i = 0; i < 4; i++)
      if (c->flt >> i & 1) {
        PUT("%s", FLT_NAMES[i]);
        break;
      }
  } else PUT("PASS");
  PUT(":%u:%d:%u:%u:", c->dp, g->mq, c->phred, c->qd);
  for (int i = 0; i < c->n_gl && i < 6; i++) PUT("%s%g", i ? "," : "", (double)c->gl[i]);
  PUT(":");
  for (int i = 0; i < 8; i++) PUT("%s%llu", i ? "," : "", (unsigned long long)g->counts[i]);
  if (n_amq) {
    PUT(":");
    int first = 1;
    for (int i = 0; i < 8; i++)
      if (g->counts[i] > 0) {
        PUT("%s%d", first ? "" : ",", g->qual[i]);
        first = 0;
      }
  }
  {
    const int has_c = GT_A[gt] == 'C' || GT_B[gt] == 'C', has_g = GT_A[gt] == 'G' || GT_B[gt] == 'G';
    PUT(":%s%s%s:%c:%.5s", has_c ? "+" : "", has_g ? "-" : "", (has_c || has_g) ? "" : "NA", c->cg, c->cx_gt);
  }
  if (het) PUT(":%d", c->fs);
  return (int)len;
}

int bsc_vcf_format_rec(const bsc_vcf_rec *r, const char *contig, const char *id, char *buf, size_t cap) {
  bsc_gt_meth g;
  memset(&g, 0, sizeof g);
  for (int i = 0; i < 8; i++) {
    g.counts[i] = r->counts[i];
    g.qual[i] = r->qual[i];
  }
  g.mq = r->mq;
  g.aq = r->aq;
  g.max_gt = r->max_gt;
  return bsc_vcf_format(&r->core, &g, contig, id, buf, cap);
}
