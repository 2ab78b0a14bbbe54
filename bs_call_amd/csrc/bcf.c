/*
 * bcf.c — one written record as a BCF2 record (host C): the bytes bcf_write() puts into an uncompressed BCF stream for
 * the record _print_vcf_entry assembles (src/print_vcf.c:160-222 the shared block, :267-378 the per-sample block).
 *
 * The reference does not hand htslib fields to format: it encodes the typed values itself with htslib's inline encoders
 * (bcf_enc_size / bcf_enc_int1 / bcf_enc_vint / bcf_enc_vfloat / bcf_enc_vchar) and bcf_write() prefixes the fixed
 * fields.  htslib is an un-vendored dependency of the reference (no version pinned; README: "htslib 1.10" / "1.11");
 * those encoders implement the typed-value rules of the BCF2 specification (hts-specs VCFv4.3 section 6.3), restated
 * here:
 *   descriptor byte   = length << 4 | type (INT8 1, INT16 2, INT32 3, FLOAT 5, CHAR 7); a length >= 15 is written as 15
 *                       followed by the true length as a typed integer
 *   a single integer  = the smallest of INT8 [-120, 127], INT16 [-32760, 32767], INT32 that holds it
 *   an integer vector = all elements in the smallest such type that holds its minimum and maximum; one element = as a
 *                       single integer
 *   record            = l_shared (shared block + 24), l_indiv, CHROM, POS (0-based), rlen, QUAL (float),
 *                       n_allele << 16 | n_info, n_fmt << 24 | n_sample, shared block, per-sample block; little endian
 * Parity with htslib's own bytes is NOT pinned in this image (no htslib here): tests decode the record with an
 * independent reader written from the specification and compare an independent Python encoder.
 *
 * The dictionary indices of the FILTER / INFO / FORMAT keys are those the header of print_vcf_header() yields
 * (src/print_vcf.c:712-731; htslib numbers the keys in order of first appearance, PASS = 0).
 */
#include <string.h>

#include "../../include/bscall_amd.h"

enum { BT_INT8 = 1, BT_INT16 = 2, BT_INT32 = 3, BT_FLOAT = 5, BT_CHAR = 7 };

typedef struct {
  uint8_t *p;
  size_t cap, len;
} bcf_buf;

static void put_bytes(bcf_buf *b, const void *src, size_t n) {
  if (n && b->len + n <= b->cap) memcpy(b->p + b->len, src, n);
  b->len += n;
}
static void put_u8(bcf_buf *b, unsigned v) {
  const uint8_t c = (uint8_t)v;
  put_bytes(b, &c, 1);
}
static void put_le(bcf_buf *b, uint32_t v, int bytes) {
  uint8_t c[4] = {(uint8_t)v, (uint8_t)(v >> 8), (uint8_t)(v >> 16), (uint8_t)(v >> 24)};
  put_bytes(b, c, (size_t)bytes);
}

static int int_type(int32_t lo, int32_t hi) {
  if (hi <= 127 && lo >= -120) return BT_INT8;
  if (hi <= 32767 && lo >= -32760) return BT_INT16;
  return BT_INT32;
}
static int type_bytes(int t) { return t == BT_INT8 ? 1 : (t == BT_INT16 ? 2 : 4); }

static void put_int(bcf_buf *b, int32_t v);

static void put_descriptor(bcf_buf *b, uint32_t n, int type) {
  if (n >= 15) {
    put_u8(b, 15u << 4 | (unsigned)type);
    put_int(b, (int32_t)n);
  } else
    put_u8(b, n << 4 | (unsigned)type);
}

static void put_int(bcf_buf *b, int32_t v) { /* a typed single integer */
  const int t = int_type(v, v);
  put_descriptor(b, 1, t);
  put_le(b, (uint32_t)v, type_bytes(t));
}

static void put_ints(bcf_buf *b, const int32_t *v, int n) { /* a typed integer vector */
  if (n == 1) {
    put_int(b, v[0]);
    return;
  }
  int32_t lo = v[0], hi = v[0];
  for (int i = 1; i < n; i++) {
    if (v[i] < lo) lo = v[i];
    if (v[i] > hi) hi = v[i];
  }
  const int t = int_type(lo, hi);
  put_descriptor(b, (uint32_t)n, t);
  for (int i = 0; i < n; i++) put_le(b, (uint32_t)v[i], type_bytes(t));
}

static void put_chars(bcf_buf *b, const char *s, uint32_t n) {
  put_descriptor(b, n, BT_CHAR);
  put_bytes(b, s, n);
}

void bsc_bcf_default_ids(bsc_bcf_ids *ids) { /* the order print_vcf_header appends the header lines in */
  if (!ids) return;
  ids->pass = 0;
  ids->info_cx = ids->fmt_cx = 1; /* INFO CX and FORMAT CX share the key */
  ids->fail = 2;                   /* q20 3, qd2 4, fs60 5, mq40 6: never referenced by a record (FT carries their names) */
  ids->mac1 = 7;
  ids->fmt_gt = 8;
  ids->fmt_ft = 9;
  ids->fmt_gl = 10;
  ids->fmt_gq = 11;
  ids->fmt_dp = 12;
  ids->fmt_mq = 13;
  ids->fmt_qd = 14;
  ids->fmt_mc8 = 15;
  ids->fmt_amq = 16;
  ids->fmt_cs = 17;
  ids->fmt_cg = 18;
  ids->fmt_fs = 19;
}

long bsc_bcf_record(const bsc_vcf_rec *r, int32_t rid, const char *id, size_t id_len, const bsc_bcf_ids *ids, uint8_t *buf, size_t cap) {
  static const char *const flt_name[4] = {"q20", "qd2", "fs60", "mq40"};                                  /* src/init_param.c:15 */
  static const char *const cs_str[10] = {"NA", "+", "-", "NA", "+", "+-", "+", "-", "-", "NA"};         /* src/print_vcf.c:58-59 */
  static const uint8_t gt_het[10] = {0, 1, 1, 1, 0, 1, 1, 0, 1, 0};                                        /* src/init_param.c:16 */
  if (!r || !ids || (!buf && cap) || (id_len && !id)) return -1;
  const bsc_vcf_core *c = &r->core;
  if (!c->emit) return 0;
  if (c->gt > 9 || c->n_gl > 6) return -1;
  bcf_buf sh = {NULL, 0, 0}, in = {NULL, 0, 0};
  /* two passes over the same statements: sizes first, then bytes behind the 32-byte fixed part */
  long total = 0;
  for (int pass = 0; pass < 2; pass++) {
    if (pass) {
      const size_t l_shared = sh.len, l_indiv = in.len;
      total = (long)(32 + l_shared + l_indiv);
      if ((size_t)total > cap) return total;
      bcf_buf fx = {buf, 32, 0};
      uint32_t n_allele = 1u + (c->alt[0] ? 1u : 0u) + (c->alt[0] && c->alt[1] ? 1u : 0u);
      int n_amq = 0;
      for (int i = 0; i < 8; i++) n_amq += r->counts[i] > 0;
      const uint32_t n_fmt = 11u + (n_amq ? 1u : 0u) + (gt_het[c->gt] ? 1u : 0u);
      const float qual = (float)c->phred;
      uint32_t qbits;
      memcpy(&qbits, &qual, 4);
      put_le(&fx, (uint32_t)(l_shared + 24), 4);
      put_le(&fx, (uint32_t)l_indiv, 4);
      put_le(&fx, (uint32_t)rid, 4);
      put_le(&fx, c->pos - 1u, 4);
      put_le(&fx, 1u, 4); /* rlen */
      put_le(&fx, qbits, 4);
      put_le(&fx, n_allele << 16 | 1u, 4); /* one INFO field */
      put_le(&fx, n_fmt << 24 | 1u, 4);    /* one sample */
      sh = (bcf_buf){buf + 32, l_shared, 0};
      in = (bcf_buf){buf + 32 + l_shared, l_indiv, 0};
    }
    /* ---- shared: ID, REF, ALT, FILTER, INFO CX (:165-221) ---- */
    put_chars(&sh, id, (uint32_t)id_len);
    put_chars(&sh, &c->cx_ref[2], 1);
    if (c->alt[0]) {
      put_chars(&sh, &c->alt[0], 1);
      if (c->alt[1]) put_chars(&sh, &c->alt[1], 1);
    }
    const int32_t fid = c->flt == 0 ? ids->pass : ((c->flt & 128) ? ids->mac1 : ids->fail);
    put_ints(&sh, &fid, 1);
    put_int(&sh, ids->info_cx);
    put_chars(&sh, c->cx_ref, 5);
    /* ---- per sample: GT FT DP MQ GQ QD GL MC8 [AMQ] CS CG CX [FS] (:267-378) ---- */
    int32_t x[8];
    x[0] = c->gt_enc >> 4;
    x[1] = c->gt_enc & 15;
    put_int(&in, ids->fmt_gt);
    put_ints(&in, x, 2);
    char ft[24];
    uint32_t ft_len = 0;
    if (c->flt & 15) { /* each name WITH its terminator, ';' between: the reference's copy loop (:283-296), "q20\0;qd2\0" */
      for (int f = 0; f < 4; f++)
        if (c->flt >> f & 1) {
          if (ft_len) ft[ft_len++] = ';';
          const size_t l = strlen(flt_name[f]) + 1;
          memcpy(ft + ft_len, flt_name[f], l);
          ft_len += (uint32_t)l;
        }
    } else {
      memcpy(ft, "PASS", 4);
      ft_len = 4;
    }
    put_int(&in, ids->fmt_ft);
    put_chars(&in, ft, ft_len);
    put_int(&in, ids->fmt_dp);
    put_int(&in, (int32_t)c->dp);
    put_int(&in, ids->fmt_mq);
    put_int(&in, r->mq);
    put_int(&in, ids->fmt_gq);
    put_int(&in, c->phred);
    put_int(&in, ids->fmt_qd);
    put_int(&in, (int32_t)c->qd);
    put_int(&in, ids->fmt_gl);
    put_descriptor(&in, c->n_gl, BT_FLOAT);
    for (int i = 0; i < c->n_gl; i++) {
      uint32_t bits;
      memcpy(&bits, &c->gl[i], 4);
      put_le(&in, bits, 4);
    }
    put_int(&in, ids->fmt_mc8);
    for (int i = 0; i < 8; i++) x[i] = (int32_t)r->counts[i];
    put_ints(&in, x, 8);
    int k = 0;
    for (int i = 0; i < 8; i++)
      if (r->counts[i] > 0) x[k++] = r->qual[i];
    if (k) {
      put_int(&in, ids->fmt_amq);
      put_ints(&in, x, k);
    }
    put_int(&in, ids->fmt_cs);
    put_chars(&in, cs_str[c->gt], (uint32_t)strlen(cs_str[c->gt]));
    put_int(&in, ids->fmt_cg);
    put_chars(&in, &c->cg, 1);
    put_int(&in, ids->fmt_cx);
    put_chars(&in, c->cx_gt, 5);
    if (gt_het[c->gt]) {
      put_int(&in, ids->fmt_fs);
      put_int(&in, c->fs);
    }
  }
  return total;
}

/* The written records of a block, one after the other (what the print thread hands bcf_write for a block): recs[n] in
 * position order; db != NULL names the records whose rs_found flag is set (bsc_dbsnp_name: the loaded contig must be the
 * block's).  Stops in front of the first record that does not fit; *n_done = records consumed (written or not emitted).
 * Returns the bytes written, -1 on a bad argument. */
long bsc_bcf_block(const bsc_vcf_rec *recs, uint64_t n, int32_t rid, const bsc_bcf_ids *ids, const bsc_dbsnp *db, uint8_t *buf,
                   size_t cap, uint64_t *n_done) {
  if ((n && !recs) || !ids || !buf || !n_done) return -1;
  size_t len = 0;
  uint64_t i = 0;
  for (; i < n; i++) {
    char rs[64];
    size_t rs_len = 0;
    if (db && recs[i].rs_found) {
      if (bsc_dbsnp_name(db, recs[i].core.pos, rs, sizeof rs, &rs_len) < 0) return -1;
    }
    const long w = bsc_bcf_record(recs + i, rid, rs_len ? rs : NULL, rs_len, ids, buf + len, cap - len);
    if (w < 0) return -1;
    if ((size_t)w > cap - len) break;
    len += (size_t)w;
  }
  *n_done = i;
  return (long)len;
}
