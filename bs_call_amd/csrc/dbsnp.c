/*
 * dbsnp.c — reader of bs_call's compressed dbSNP index (the file bin/dbSNP_idx writes), host C + zlib.
 *
 * What the reference does with it: nothing touches the likelihoods; a position's index entry
 * (1) names the record (VCF ID column, src/print_vcf.c:167-171), (2) forces the AA / TT homozygous-reference records of
 * sites flagged in the index's `fq_mask` to be written (rs_found & 2, :139), (3) feeds the dbSNP counters of the
 * statistics (:426-441).  The device side takes (2) and (3) as one byte per position — bsc_dbsnp_flags() fills exactly
 * the `dbsnp` array of bsc_chain_device / bsc_block_records / bsc_vcf_records — and the names stay on the host
 * (bsc_dbsnp_name(), for bsc_vcf_format_rec's ID column).
 *
 * Replaces, function for function:
 *   bsc_dbsnp_open          load_dbSNP_header   src/dbSNP.c:27-141   32-byte file header {magic 0xd7278434, 0, directory
 *                                                                     offset, largest uncompressed block, compressed
 *                                                                     directory size}; zlib directory {version, 0,
 *                                                                     n_prefixes u16, n_contigs u32, "track ..." header,
 *                                                                     prefixes, per contig {min_bin, max_bin, offset,
 *                                                                     name}}; trailing magic
 *   bsc_dbsnp_load_contig   load_dbSNP_ctg      src/dbSNP.c:157-304  zlib blocks {u64 size, data} up to a zero size; a
 *                                                                     block = bins: bin increment (1 / 2 / 3 / 5 bytes),
 *                                                                     then entries {position in bin | prefix << 6,
 *                                                                     [2 prefix bytes], name digits, terminator: bit 0
 *                                                                     = last of the bin, bit 1 = fq_mask}
 *   bsc_dbsnp_name          dbSNP_lookup_name   src/dbSNP.c:306-350  bin = x >> 6, bit = x & 63; 0 / 1 / 3
 * The on-disk format is the writer's (src/dbSNP_output.c:139-182,202-299); tools/make_dbsnp_index.py writes it.
 * Error handling: the reference prints and returns NULL / false; here every malformed input is a BSC_ERR_ARG with text.
 */
#include <errno.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "../../include/bscall_amd.h"

#define DBSNP_MAGIC 0xd7278434u

typedef struct {
  uint64_t mask, fq_mask;
  int n_entries;
  uint16_t *entries; /* (digit bytes << 8) | position in bin | prefix id << 6 */
  uint8_t *name_buf;
} dbsnp_bin;

typedef struct {
  char *name;
  uint32_t min_bin, max_bin;
  uint64_t file_offset;
} dbsnp_ctg;

struct bsc_dbsnp {
  FILE *fp;
  uint64_t bufsize; /* largest uncompressed block */
  uint32_t n_prefixes, n_ctgs;
  char **prefix;
  char *header;
  dbsnp_ctg *ctgs;
  int loaded; /* index of the loaded contig, -1 = none */
  dbsnp_bin *bins;
  uint64_t bins_used; /* 1 + the highest bin of the loaded contig that holds entries */
  uint64_t n_snps;
};

/* bscall_api.c keeps the per-thread error text; this file reports through the same channel */
int bsc_set_error(int code, const char *fmt, ...);

static void dbsnp_unload(bsc_dbsnp *db) {
  if (db->bins && db->loaded >= 0) {
    const dbsnp_ctg *c = db->ctgs + db->loaded;
    uint64_t n = (uint64_t)c->max_bin - c->min_bin + 1;
    if (n > db->bins_used) n = db->bins_used; /* the bins past it were never written (and their pages never touched) */
    for (uint64_t i = 0; i < n; i++) {
      free(db->bins[i].entries);
      free(db->bins[i].name_buf);
    }
  }
  free(db->bins);
  db->bins = NULL;
  db->bins_used = 0;
  db->loaded = -1;
  db->n_snps = 0;
}

void bsc_dbsnp_close(bsc_dbsnp *db) {
  if (!db) return;
  dbsnp_unload(db);
  if (db->fp) fclose(db->fp);
  for (uint32_t i = 0; i < db->n_prefixes && db->prefix; i++) free(db->prefix[i]);
  free(db->prefix);
  free(db->header);
  for (uint32_t i = 0; i < db->n_ctgs && db->ctgs; i++) free(db->ctgs[i].name);
  free(db->ctgs);
  free(db);
}

static char *dup_str(const char *s, size_t l) {
  char *p = malloc(l + 1);
  if (p) {
    memcpy(p, s, l);
    p[l] = 0;
  }
  return p;
}

int bsc_dbsnp_open(const char *path, bsc_dbsnp **out) {
  if (!path || !out) return bsc_set_error(BSC_ERR_ARG, "bsc_dbsnp_open: NULL argument");
  *out = NULL;
  FILE *fp = fopen(path, "rb");
  if (!fp) return bsc_set_error(BSC_ERR_ARG, "bsc_dbsnp_open: cannot open %s: %s", path, strerror(errno));
  uint32_t td[2];
  uint64_t td1[3];
  if (fread(td, sizeof(uint32_t), 2, fp) != 2 || td[0] != DBSNP_MAGIC || fread(td1, sizeof(uint64_t), 3, fp) != 3) {
    fclose(fp);
    return bsc_set_error(BSC_ERR_ARG, "bsc_dbsnp_open: %s is not a dbSNP index (bad magic / short header)", path);
  }
  bsc_dbsnp *db = calloc(1, sizeof *db);
  unsigned char *ubuf = NULL, *cbuf = NULL;
  int rc = BSC_OK;
  if (!db) {
    fclose(fp);
    return bsc_set_error(BSC_ERR_NOMEM, "bsc_dbsnp_open: out of memory");
  }
  db->fp = fp;
  db->bufsize = td1[1];
  db->loaded = -1;
#define FAIL(...)                                   \
  do {                                              \
    rc = bsc_set_error(BSC_ERR_ARG, __VA_ARGS__);   \
    goto done;                                      \
  } while (0)
  if (td1[1] == 0 || td1[1] > (1ull << 32) || td1[2] == 0 || td1[2] > (1ull << 32)) FAIL("bsc_dbsnp_open: implausible block sizes");
  ubuf = malloc((size_t)td1[1]);
  cbuf = malloc((size_t)td1[2]);
  if (!ubuf || !cbuf) {
    rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_dbsnp_open: out of memory");
    goto done;
  }
  uint32_t tail = 0;
  if (fseek(fp, (long)td1[0], SEEK_SET) || fread(cbuf, 1, (size_t)td1[2], fp) != td1[2] || fread(&tail, sizeof tail, 1, fp) != 1 ||
      tail != DBSNP_MAGIC)
    FAIL("bsc_dbsnp_open: directory block unreadable or not followed by the magic");
  uLongf size = (uLongf)td1[1];
  if (uncompress(ubuf, &size, cbuf, (uLong)td1[2]) != Z_OK) FAIL("bsc_dbsnp_open: directory does not decompress");
  if (size < 16) FAIL("bsc_dbsnp_open: directory too short");
  uint16_t npre;
  uint32_t nctg;
  memcpy(&npre, ubuf + 2, 2);
  memcpy(&nctg, ubuf + 4, 4);
  /* a prefix takes at least 1 byte of the directory, a contig at least 17: counts beyond that are damage, not a reason to
   * reserve memory for them */
  if ((uint64_t)npre > size || (uint64_t)nctg > size / 17u) FAIL("bsc_dbsnp_open: directory lists more entries than it holds");
  const char *p = (const char *)ubuf + 8, *p1 = (const char *)ubuf + size;
  size_t l = strnlen(p, (size_t)(p1 - p));
  if (p + 8 >= p1 || strncmp(p, "track ", 6) || p + l >= p1) FAIL("bsc_dbsnp_open: no \"track\" header line");
  db->header = dup_str(p + 6, l - 6);
  p += l + 1;
  db->prefix = calloc(npre ? npre : 1, sizeof(char *));
  db->ctgs = calloc(nctg ? nctg : 1, sizeof(dbsnp_ctg));
  if (!db->header || !db->prefix || !db->ctgs) {
    rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_dbsnp_open: out of memory");
    goto done;
  }
  for (uint32_t i = 0; i < npre; i++) {
    if (p >= p1) FAIL("bsc_dbsnp_open: directory ends inside the prefix list");
    l = strnlen(p, (size_t)(p1 - p));
    if (p + l >= p1) FAIL("bsc_dbsnp_open: unterminated prefix");
    db->prefix[i] = dup_str(p, l);
    db->n_prefixes = i + 1;
    p += l + 1;
  }
  for (uint32_t i = 0; i < nctg; i++) {
    if (p + 16 >= p1) FAIL("bsc_dbsnp_open: directory ends inside the contig list");
    dbsnp_ctg *c = db->ctgs + i;
    memcpy(&c->min_bin, p, 4);
    memcpy(&c->max_bin, p + 4, 4);
    memcpy(&c->file_offset, p + 8, 8);
    if (c->max_bin < c->min_bin) FAIL("bsc_dbsnp_open: contig %u has max_bin < min_bin", i);
    if (c->max_bin >= (1u << 26)) FAIL("bsc_dbsnp_open: contig %u has a bin beyond position 2^32", i);
    p += 16;
    l = strnlen(p, (size_t)(p1 - p));
    if (p + l >= p1) FAIL("bsc_dbsnp_open: unterminated contig name");
    c->name = dup_str(p, l);
    db->n_ctgs = i + 1;
    for (uint32_t k = 0; k < i; k++)
      if (!strcmp(db->ctgs[k].name, c->name)) FAIL("bsc_dbsnp_open: duplicate contig %s", c->name);
    p += l + 1;
  }
done:
  free(ubuf);
  free(cbuf);
  if (rc) {
    bsc_dbsnp_close(db);
    return rc;
  }
  *out = db;
  return BSC_OK;
#undef FAIL
}

int bsc_dbsnp_n_contigs(const bsc_dbsnp *db) { return db ? (int)db->n_ctgs : 0; }
const char *bsc_dbsnp_contig_name(const bsc_dbsnp *db, int i) { return (db && i >= 0 && (uint32_t)i < db->n_ctgs) ? db->ctgs[i].name : NULL; }
const char *bsc_dbsnp_header(const bsc_dbsnp *db) { return db ? db->header : NULL; }

/* file byte -> BCD byte of the name digits: 0x21 + n -> two digits of n (00 .. 99), 0x85 + d -> digit d and a filler */
static int dbsnp_digit_byte(unsigned b) {
  if (b >= 0x21 && b <= 0x84) return (int)((((b - 0x21) / 10) << 4) | ((b - 0x21) % 10));
  if (b >= 0x85 && b <= 0x8e) return (int)(((b - 0x85) << 4) | 0xf);
  return 0xff;
}

int bsc_dbsnp_load_contig(bsc_dbsnp *db, const char *name, uint64_t *n_snps) {
  if (!db || !name) return bsc_set_error(BSC_ERR_ARG, "bsc_dbsnp_load_contig: NULL argument");
  dbsnp_unload(db);
  int ci = -1;
  for (uint32_t i = 0; i < db->n_ctgs; i++)
    if (!strcmp(db->ctgs[i].name, name)) ci = (int)i;
  if (n_snps) *n_snps = 0;
  if (ci < 0) return BSC_OK; /* a contig the index does not know: nothing is flagged (the reference: dbSNP_ctg = NULL) */
  const dbsnp_ctg *ctg = db->ctgs + ci;
  const uint64_t nb = (uint64_t)ctg->max_bin - ctg->min_bin + 1;
  db->bins = calloc((size_t)nb, sizeof(dbsnp_bin));
  unsigned char *ubuf = malloc((size_t)db->bufsize);
  size_t csz = 1 + (size_t)((double)db->bufsize * .75);
  unsigned char *cbuf = malloc(csz);
  uint16_t entries[64];
  uint8_t *name_buf = malloc(258 * 64);
  int rc = BSC_OK;
#define FAIL(...)                                   \
  do {                                              \
    rc = bsc_set_error(BSC_ERR_ARG, __VA_ARGS__);   \
    goto done;                                      \
  } while (0)
  if (!db->bins || !ubuf || !cbuf || !name_buf) {
    rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_dbsnp_load_contig: out of memory");
    goto done;
  }
  db->loaded = ci;
  if (fseek(db->fp, (long)ctg->file_offset, SEEK_SET)) FAIL("bsc_dbsnp_load_contig: seek failed");
  uint64_t curr_bin = ctg->min_bin;
  dbsnp_bin *bins = db->bins;
  uint64_t total = 0;
  for (;;) {
    uint64_t sz;
    if (fread(&sz, sizeof sz, 1, db->fp) != 1) FAIL("bsc_dbsnp_load_contig: %s: block size unreadable", name);
    if (sz == 0) break; /* end-of-contig marker */
    if (sz > (1ull << 32)) FAIL("bsc_dbsnp_load_contig: %s: implausible block size", name);
    if (csz < sz) {
      csz = (size_t)sz + (size_t)(sz / 10);
      unsigned char *t = realloc(cbuf, csz);
      if (!t) {
        rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_dbsnp_load_contig: out of memory");
        goto done;
      }
      cbuf = t;
    }
    if (fread(cbuf, 1, (size_t)sz, db->fp) != sz) FAIL("bsc_dbsnp_load_contig: %s: truncated block", name);
    uLongf size = (uLongf)db->bufsize;
    if (uncompress(ubuf, &size, cbuf, (uLong)sz) != Z_OK) FAIL("bsc_dbsnp_load_contig: %s: block does not decompress", name);
    const unsigned char *bp = ubuf, *bp_end = ubuf + size;
    int n_entries = 0, name_ptr = 0, prev_ix = -1;
    uint64_t mask[2] = {0, 0};
    while (bp < bp_end) {
      if (!n_entries) { /* distance from the previous bin (src/dbSNP.c:208-244) */
        uint32_t inc = 0;
        const unsigned x = *bp++;
        switch (x & 3u) {
          case 0: inc = x >> 2; break;
          case 1:
            if (bp >= bp_end) FAIL("bsc_dbsnp_load_contig: %s: truncated bin header", name);
            inc = *bp++;
            break;
          case 2: {
            if (bp + 1 >= bp_end) FAIL("bsc_dbsnp_load_contig: %s: truncated bin header", name);
            uint16_t k;
            memcpy(&k, bp, 2);
            bp += 2;
            inc = k;
          } break;
          default: {
            if (bp + 3 >= bp_end) FAIL("bsc_dbsnp_load_contig: %s: truncated bin header", name);
            memcpy(&inc, bp, 4);
            bp += 4;
          } break;
        }
        curr_bin += inc;
        if (curr_bin > ctg->max_bin || bp >= bp_end) break;
        bins += inc;
      }
      const unsigned x = *bp++;
      const int prefix_ix = (int)(x >> 6);
      int sl = 0;
      if (!prefix_ix) { /* an explicit two-byte prefix index */
        if (bp + 2 >= bp_end) FAIL("bsc_dbsnp_load_contig: %s: truncated entry", name);
        name_buf[name_ptr++] = *bp++;
        name_buf[name_ptr++] = *bp++;
        sl = 2;
      }
      if ((int)(x & 63u) <= prev_ix || prefix_ix > (int)db->n_prefixes) FAIL("bsc_dbsnp_load_contig: %s: entries of a bin out of order", name);
      prev_ix = (int)(x & 63u);
      int k = name_ptr;
      while (bp < bp_end && *bp > 3 && sl++ < 256) {
        const int d = dbsnp_digit_byte(*bp++);
        name_buf[name_ptr++] = (uint8_t)d;
      }
      k = name_ptr - k;
      if (bp >= bp_end || *bp > 3) FAIL("bsc_dbsnp_load_contig: %s: unterminated name", name);
      const uint64_t msk = (uint64_t)1 << prev_ix;
      mask[0] |= msk;
      const unsigned tm = *bp++;
      if (tm & 2u) mask[1] |= msk;
      if (n_entries == 64) FAIL("bsc_dbsnp_load_contig: %s: more than 64 entries in a bin", name);
      entries[n_entries++] = (uint16_t)((k << 8) | x);
      if (tm & 1u) { /* last entry of the bin */
        if (bins->n_entries) FAIL("bsc_dbsnp_load_contig: %s: bin listed twice", name);
        if ((uint64_t)(bins - db->bins) + 1 > db->bins_used) db->bins_used = (uint64_t)(bins - db->bins) + 1;
        bins->entries = malloc(sizeof(uint16_t) * (size_t)n_entries);
        bins->name_buf = malloc((size_t)(name_ptr ? name_ptr : 1));
        if (!bins->entries || !bins->name_buf) {
          rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_dbsnp_load_contig: out of memory");
          goto done;
        }
        memcpy(bins->entries, entries, sizeof(uint16_t) * (size_t)n_entries);
        memcpy(bins->name_buf, name_buf, (size_t)name_ptr);
        bins->n_entries = n_entries;
        bins->mask = mask[0];
        bins->fq_mask = mask[1];
        total += (uint64_t)n_entries;
        n_entries = 0;
        mask[0] = mask[1] = 0;
        name_ptr = 0;
        prev_ix = -1;
      }
    }
  }
  db->n_snps = total;
  if (n_snps) *n_snps = total;
done:
  free(ubuf);
  free(cbuf);
  free(name_buf);
  if (rc) dbsnp_unload(db);
  return rc;
#undef FAIL
}

/* rs_found of positions x0 .. x0 + n - 1 of the loaded contig: 0 not in dbSNP, 1 in dbSNP, 3 in dbSNP and flagged in the
 * index's fq_mask (src/dbSNP.c:306-318) — the `dbsnp` array of the device entry points */
int bsc_dbsnp_flags(const bsc_dbsnp *db, uint32_t x0, uint32_t n, uint8_t *out) {
  if (!db || (n && !out)) return bsc_set_error(BSC_ERR_ARG, "bsc_dbsnp_flags: NULL argument");
  memset(out, 0, n);
  if (db->loaded < 0 || !db->bins) return BSC_OK;
  const dbsnp_ctg *c = db->ctgs + db->loaded;
  for (uint64_t i = 0; i < n;) {
    const uint64_t x = (uint64_t)x0 + i;
    const uint64_t bn = x >> 6;
    const unsigned first = (unsigned)(x & 63u);
    uint64_t take = 64u - first;
    if (take > n - i) take = n - i;
    if (bn >= c->min_bin && bn <= c->max_bin) {
      const dbsnp_bin *b = db->bins + (bn - c->min_bin);
      if (b->mask) {
        for (uint64_t k = 0; k < take; k++) {
          const uint64_t mk = (uint64_t)1 << (first + k);
          if (b->mask & mk) out[i + k] = (b->fq_mask & mk) ? 3 : 1;
        }
      }
    }
    i += take;
  }
  return BSC_OK;
}

/* dbSNP_lookup_name (src/dbSNP.c:306-350): the flag of position x and its name in rs (NUL-terminated).  *rs_len is the
 * length the REFERENCE hands to htslib: prefix + two characters per digit byte, i.e. a name with an odd number of digits
 * counts its filler (a NUL) — the BCF ID field of the reference carries that byte. */
int bsc_dbsnp_name(const bsc_dbsnp *db, uint32_t x, char *rs, size_t cap, size_t *rs_len) {
  static const char dtab[16] = {'0', '1', '2', '3', '4', '5', '6', '7', '8', '9', 0, 0, 0, 0, 0, 0};
  if (!db || !rs || cap < 1) return bsc_set_error(BSC_ERR_ARG, "bsc_dbsnp_name: NULL argument");
  rs[0] = 0;
  if (rs_len) *rs_len = 0;
  if (db->loaded < 0 || !db->bins) return 0;
  const dbsnp_ctg *c = db->ctgs + db->loaded;
  const uint64_t bn = x >> 6;
  if (bn < c->min_bin || bn > c->max_bin) return 0;
  const dbsnp_bin *b = db->bins + (bn - c->min_bin);
  const uint64_t mk = (uint64_t)1 << (x & 63u);
  if (!(b->mask & mk)) return 0;
  const int res = (b->fq_mask & mk) ? 3 : 1;
  uint64_t mk1 = b->mask & (mk - 1);
  int i = 0, j = 0;
  while (mk1) {
    if (mk1 & 1u) {
      const uint16_t en = b->entries[i++];
      j += en >> 8;
      if (!((en >> 6) & 3)) j += 2;
    }
    mk1 >>= 1;
  }
  int prefix_id = (b->entries[i] >> 6) & 3;
  const unsigned char *tp1 = b->name_buf + j;
  if ((prefix_id--) == 0) {
    /* The writer stores an explicit prefix index (the fourth prefix onwards) as a native little-endian u16
     * (src/dbSNP_output.c:280); the reference's reader puts the two bytes together high byte first (src/dbSNP.c:337) and
     * so indexes its prefix table out of bounds for every such entry.  There is no behaviour to reproduce there; this
     * reader takes the index as written. */
    prefix_id = tp1[0] | (tp1[1] << 8);
    tp1 += 2;
  }
  if (prefix_id < 0 || (uint32_t)prefix_id >= db->n_prefixes)
    return bsc_set_error(BSC_ERR_ARG, "bsc_dbsnp_name: position %u names prefix %d of %u", x, prefix_id, db->n_prefixes);
  const char *pre = db->prefix[prefix_id];
  const int nd = b->entries[i] >> 8;
  if (strlen(pre) + 2u * (size_t)nd + 1u > cap) return bsc_set_error(BSC_ERR_ARG, "bsc_dbsnp_name: buffer too small");
  char *tp = rs;
  while (*pre) *tp++ = *pre++;
  for (int k = 0; k < nd; k++) {
    const unsigned z = *tp1++;
    *tp++ = dtab[z >> 4];
    *tp++ = dtab[z & 15];
  }
  *tp = 0;
  if (rs_len) *rs_len = (size_t)(tp - rs);
  return res;
}

/* The names of every flagged position of x0 .. x0 + n - 1 (the table the device encoder of csrc/bcfdev.hip looks a record's ID up
 * in): ascending positions, offsets, the bytes bsc_dbsnp_name returns for each (*rs_len of them, the filler of an odd digit count
 * included).  NULL arrays: the sizes only. */
int bsc_dbsnp_names(const bsc_dbsnp *db, uint32_t x0, uint32_t n, uint32_t *pos, uint32_t *off, char *bytes, uint32_t cap_names,
                    uint64_t cap_bytes, uint32_t *n_names, uint64_t *n_bytes) {
  if (!db || !n_names || !n_bytes) return bsc_set_error(BSC_ERR_ARG, "bsc_dbsnp_names: NULL argument");
  const int fill = pos && off && bytes;
  uint32_t k = 0;
  uint64_t nb = 0;
  *n_names = 0;
  *n_bytes = 0;
  if (db->loaded >= 0 && db->bins && n) {
    const dbsnp_ctg *c = db->ctgs + db->loaded;
    const uint64_t last = (uint64_t)x0 + n - 1;
    uint64_t bn0 = (uint64_t)x0 >> 6, bn1 = last >> 6;
    if (bn0 < c->min_bin) bn0 = c->min_bin;
    if (bn1 > c->max_bin) bn1 = c->max_bin;
    for (uint64_t bn = bn0; bn <= bn1 && bn >= c->min_bin; bn++) {
      if (bn - c->min_bin >= db->bins_used) break;
      uint64_t m = db->bins[bn - c->min_bin].mask;
      while (m) {
        const unsigned bit = (unsigned)__builtin_ctzll(m);
        m &= m - 1;
        const uint64_t x = bn * 64u + bit;
        if (x < x0 || x > last) continue;
        char rs[256];
        size_t l = 0;
        const int r = bsc_dbsnp_name(db, (uint32_t)x, rs, sizeof rs, &l);
        if (r < 0) return r;
        if (fill) {
          if (k >= cap_names || nb + l > cap_bytes || nb + l > 0xffffffffull)
            return bsc_set_error(BSC_ERR_ARG, "bsc_dbsnp_names: more than %u names / %llu bytes", cap_names, (unsigned long long)cap_bytes);
          pos[k] = (uint32_t)x;
          off[k] = (uint32_t)nb;
          memcpy(bytes + nb, rs, l);
        }
        k++;
        nb += l;
      }
    }
  }
  if (fill) off[k] = (uint32_t)nb;
  *n_names = k;
  *n_bytes = nb;
  return BSC_OK;
}
