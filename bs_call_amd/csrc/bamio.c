/*
 * bamio.c — BAM in, blocks of templates out (host C + zlib, no htslib): the reader thread of the reference.
 *
 *   bsc_bam_open / bsc_bam_close      hts_open + sam_hdr_read for a BAM file: BGZF blocks inflated with zlib, the header
 *                                     text and the reference sequence list (SAM specification sections 4.1, 4.2)
 *   bsc_bam_next_block                read_input (src/get_template_vector.c:49-389) turned inside out: instead of handing
 *                                     each finished block to the process thread it returns it — the templates of one block
 *                                     (mates joined, duplicates resolved, in order of their leftmost position), the contig
 *                                     and y = the rightmost covered position; per record get_next_align_details
 *                                     (src/input_sam.c:222-312): the flag filters and their reasons, orientation, the
 *                                     forward / reverse positions, get_bam_misms (:90-136), get_seq_and_qual (:61-88),
 *                                     get_bs_strand (:144-220)
 *   bsc_bam_filter_counts             bs_stats.filter_cts / filter_bases as the reader leaves them (the report's ReadLevel)
 *
 * Not covered: SAM text and CRAM input, region queries through a .bai index (the reference's -r), contig include /
 * exclude lists (every @SQ contig is processed).  The block a call returns stays valid until the next call.
 *
 * htslib is an un-vendored dependency of the reference; what this file needs from it is the BAM / BGZF layout, which the
 * SAM specification fixes.  Parity of the template stream is pinned by hand-worked scenarios and an independent Python
 * restatement (the repository's test oracle) over BAM files written by tools/make_bam.py — not against htslib, which this image lacks.
 */
#include <errno.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include "../../include/bscall_amd.h"

int bsc_set_error(int code, const char *fmt, ...);

/* ---- BGZF stream ---------------------------------------------------------------------------------------------------
 * Blocks are independent gzip members: with helper threads (bsc_bam_open_threads) a ring of blocks is read in file order
 * under one lock, inflated by whichever helper claimed the block, and consumed in order; without, the caller's thread does
 * both.  Errors found by a helper travel with the block and are raised by the consumer (the error text is per thread). */
#define BGZF_RING 64

typedef struct {
  uint8_t raw[65536 + 64]; /* one compressed block */
  uint8_t out[65536];      /* its inflated bytes */
  uint32_t clen, isize, crc;
  int state;               /* SLOT_* */
  const char *err;
} bgzf_slot;
enum { SLOT_FREE, SLOT_BUSY, SLOT_READY, SLOT_EOF, SLOT_ERR };

typedef struct {
  FILE *f;
  bgzf_slot *ring;   /* 1 slot without helpers, BGZF_RING with */
  uint8_t *cur;      /* the block being consumed */
  uint32_t n, o;     /* its inflated length, read offset */
  int eof;
  /* helpers */
  int n_threads, closing, stop_reading;
  pthread_t th[16];
  pthread_mutex_t mu;
  pthread_cond_t cv_ready, cv_free;
  uint64_t read_idx, cons_idx; /* next block to read from the file / next block the consumer takes */
  int holding;                 /* the consumer still uses ring[(cons_idx - 1) % BGZF_RING] */
} bgzf_in;

/* one block's header and payload from the file -> slot (raw, clen, crc, isize).  1 = a block, 0 = end of file, < 0 = s->err set */
static int bgzf_read_raw(FILE *f, bgzf_slot *s) {
  uint8_t h[18];
  const size_t got = fread(h, 1, 18, f);
  if (got == 0) return 0;
  s->err = NULL;
  if (got != 18 || h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) {
    s->err = "BAM: not a BGZF block (truncated file or plain gzip)";
    return -1;
  }
  const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
  /* the BC subfield is the first one in every file written by the usual tools; walk the extra field to be safe (it is
   * read into the slot's output area, which the payload's inflation overwrites later) */
  uint32_t bsize = 0;
  uint8_t *extra = s->out;
  memcpy(extra, h + 12, 6);
  if (xlen > 6 && fread(extra + 6, 1, xlen - 6, f) != xlen - 6) {
    s->err = "BAM: truncated BGZF header";
    return -1;
  }
  for (uint32_t p = 0; p + 4 <= xlen;) {
    const uint32_t sl = extra[p + 2] | (uint32_t)extra[p + 3] << 8;
    if (extra[p] == 'B' && extra[p + 1] == 'C' && sl == 2 && p + 6 <= xlen) bsize = (extra[p + 4] | (uint32_t)extra[p + 5] << 8) + 1u;
    p += 4 + sl;
  }
  if (bsize < 12 + xlen + 8) {
    s->err = "BAM: BGZF block without a valid BC field";
    return -1;
  }
  s->clen = bsize - 12 - xlen - 8;
  uint8_t tail[8];
  if (fread(s->raw, 1, s->clen, f) != s->clen || fread(tail, 1, 8, f) != 8) {
    s->err = "BAM: truncated BGZF block";
    return -1;
  }
  s->crc = tail[0] | (uint32_t)tail[1] << 8 | (uint32_t)tail[2] << 16 | (uint32_t)tail[3] << 24;
  s->isize = tail[4] | (uint32_t)tail[5] << 8 | (uint32_t)tail[6] << 16 | (uint32_t)tail[7] << 24;
  if (s->isize > 65536) {
    s->err = "BAM: BGZF block larger than 64 KiB";
    return -1;
  }
  return 1;
}

/* raw -> out; 0 or -1 with s->err set.  Touches nothing but the slot: safe on any thread. */
static int bgzf_inflate(bgzf_slot *s) {
  z_stream z;
  memset(&z, 0, sizeof z);
  if (inflateInit2(&z, -15) != Z_OK) {
    s->err = "BAM: zlib initialisation failed";
    return -1;
  }
  z.next_in = s->raw;
  z.avail_in = s->clen;
  z.next_out = s->out;
  z.avail_out = 65536;
  const int r = inflate(&z, Z_FINISH);
  const uint32_t produced = (uint32_t)z.total_out;
  inflateEnd(&z);
  if (r != Z_STREAM_END || produced != s->isize) {
    s->err = "BAM: corrupt BGZF block";
    return -1;
  }
  if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), s->out, s->isize) != s->crc) {
    s->err = "BAM: BGZF checksum mismatch";
    return -1;
  }
  return 0;
}

static void *bgzf_helper(void *arg) {
  bgzf_in *z = (bgzf_in *)arg;
  pthread_mutex_lock(&z->mu);
  for (;;) {
    bgzf_slot *s = &z->ring[z->read_idx % BGZF_RING];
    while (!z->closing && !z->stop_reading && s->state != SLOT_FREE) {
      pthread_cond_wait(&z->cv_free, &z->mu);
      s = &z->ring[z->read_idx % BGZF_RING];
    }
    if (z->closing || z->stop_reading) break;
    /* claim the next block: the file is read in order under the lock, the inflation happens outside it */
    const int r = bgzf_read_raw(z->f, s);
    z->read_idx++;
    if (r <= 0) {
      s->state = r == 0 ? SLOT_EOF : SLOT_ERR;
      z->stop_reading = 1;
      pthread_cond_broadcast(&z->cv_ready);
      pthread_cond_broadcast(&z->cv_free);
      break;
    }
    s->state = SLOT_BUSY;
    pthread_mutex_unlock(&z->mu);
    const int e = bgzf_inflate(s);
    pthread_mutex_lock(&z->mu);
    s->state = e ? SLOT_ERR : SLOT_READY;
    pthread_cond_broadcast(&z->cv_ready);
  }
  pthread_mutex_unlock(&z->mu);
  return NULL;
}

static int bgzf_fill(bgzf_in *z) { /* 1 = a block with data, 0 = end of file, < 0 error */
  if (z->n_threads == 0) {
    for (;;) {
      bgzf_slot *s = &z->ring[0];
      const int r = bgzf_read_raw(z->f, s);
      if (r == 0) {
        z->eof = 1;
        return 0;
      }
      if (r < 0 || bgzf_inflate(s)) return bsc_set_error(BSC_ERR_ARG, "%s", s->err);
      z->cur = s->out;
      z->n = s->isize;
      z->o = 0;
      if (s->isize) return 1; /* an empty block (the end-of-file marker) is skipped */
    }
  }
  pthread_mutex_lock(&z->mu);
  for (;;) {
    if (z->holding) { /* hand the block just consumed back to the helpers */
      z->ring[(z->cons_idx - 1) % BGZF_RING].state = SLOT_FREE;
      z->holding = 0;
      pthread_cond_broadcast(&z->cv_free);
    }
    bgzf_slot *s = &z->ring[z->cons_idx % BGZF_RING];
    /* the slot belongs to block cons_idx once a helper has claimed it: read_idx > cons_idx */
    while (!(z->read_idx > z->cons_idx && (s->state == SLOT_READY || s->state == SLOT_EOF || s->state == SLOT_ERR))) pthread_cond_wait(&z->cv_ready, &z->mu);
    if (s->state == SLOT_EOF) {
      pthread_mutex_unlock(&z->mu);
      z->eof = 1;
      return 0;
    }
    if (s->state == SLOT_ERR) {
      const char *msg = s->err;
      pthread_mutex_unlock(&z->mu);
      return bsc_set_error(BSC_ERR_ARG, "%s", msg ? msg : "BAM: read error");
    }
    z->cons_idx++;
    z->holding = 1;
    z->cur = s->out;
    z->n = s->isize;
    z->o = 0;
    if (s->isize) break;
  }
  pthread_mutex_unlock(&z->mu);
  return 1;
}

static int bgzf_start(bgzf_in *z, int n_threads) {
  if (n_threads < 0) n_threads = 0;
  if (n_threads > 16) n_threads = 16;
  z->ring = calloc(n_threads ? BGZF_RING : 1, sizeof(bgzf_slot));
  if (!z->ring) return -1;
  z->n_threads = 0;
  if (n_threads) {
    pthread_mutex_init(&z->mu, NULL);
    pthread_cond_init(&z->cv_ready, NULL);
    pthread_cond_init(&z->cv_free, NULL);
    for (int i = 0; i < n_threads; i++) {
      if (pthread_create(&z->th[z->n_threads], NULL, bgzf_helper, z)) break;
      z->n_threads++;
    }
    if (z->n_threads == 0) { /* no helper could be started: the caller's thread does the work, on slot 0 */
      pthread_mutex_destroy(&z->mu);
      pthread_cond_destroy(&z->cv_ready);
      pthread_cond_destroy(&z->cv_free);
    }
  }
  return 0;
}

static void bgzf_stop(bgzf_in *z) {
  if (z->n_threads) {
    pthread_mutex_lock(&z->mu);
    z->closing = 1;
    pthread_cond_broadcast(&z->cv_free);
    pthread_cond_broadcast(&z->cv_ready);
    pthread_mutex_unlock(&z->mu);
    for (int i = 0; i < z->n_threads; i++) pthread_join(z->th[i], NULL);
    pthread_mutex_destroy(&z->mu);
    pthread_cond_destroy(&z->cv_ready);
    pthread_cond_destroy(&z->cv_free);
    z->n_threads = 0;
  }
  free(z->ring);
  z->ring = NULL;
}

/* n bytes into dst; returns 1, 0 at a clean end of file (nothing read), < 0 on error / truncation */
static int bgzf_read(bgzf_in *z, void *dst, size_t n) {
  uint8_t *d = (uint8_t *)dst;
  size_t done = 0;
  while (done < n) {
    if (z->o == z->n) {
      const int r = bgzf_fill(z);
      if (r < 0) return r;
      if (r == 0) return done ? bsc_set_error(BSC_ERR_ARG, "BAM: input truncated") : 0;
    }
    const size_t take = (n - done < (size_t)(z->n - z->o)) ? n - done : (size_t)(z->n - z->o);
    memcpy(d + done, z->cur + z->o, take);
    z->o += (uint32_t)take;
    done += take;
  }
  return 1;
}

static uint32_t le32(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

/* ---- the reader's state ------------------------------------------------------------------------------------------ */
typedef struct name_node { /* a forward-facing read waiting for its mate: align_hash, include/bs_call.h:184-190 */
  struct name_node *next;
  uint32_t ix;    /* its template in the block being built */
  uint32_t flag;  /* alignment_flag */
  uint32_t len;
  char name[];
} name_node;

#define NAME_BUCKETS 4096u

typedef struct { /* the block being built or handed out */
  bsc_raw_template *tpl;
  name_node **waiting; /* per template: its hash entry or NULL (al_hash_list) */
  uint32_t n, cap;
  uint8_t *seq;
  uint64_t seq_len, seq_cap;
  bsc_misms *ms;
  uint64_t n_ms, ms_cap;
} blk_buf;

struct bsc_bam {
  bgzf_in z;
  char *text;
  uint32_t l_text;
  int32_t n_ref;
  char **ref_name;
  uint32_t *ref_len;
  uint8_t *rec; /* one BAM record */
  uint32_t rec_cap;
  /* read_input's variables (src/get_template_vector.c:53-58) */
  int32_t curr_tid, old_tid;
  uint32_t max_pos, start_pos, curr_pos, start_idx;
  blk_buf cur, out;
  name_node *bucket[NAME_BUCKETS];
  int finished;
  uint64_t filter_cts[15], filter_bases[15];
};

static uint32_t name_hash(const char *s, uint32_t n) {
  uint32_t h = 2166136261u;
  for (uint32_t i = 0; i < n; i++) h = (h ^ (uint8_t)s[i]) * 16777619u;
  return h & (NAME_BUCKETS - 1u);
}
static name_node *name_find(bsc_bam *b, const char *s, uint32_t n) {
  for (name_node *p = b->bucket[name_hash(s, n)]; p; p = p->next)
    if (p->len == n && !memcmp(p->name, s, n)) return p;
  return NULL;
}
static void name_unlink(bsc_bam *b, name_node *q) {
  name_node **pp = &b->bucket[name_hash(q->name, q->len)];
  while (*pp && *pp != q) pp = &(*pp)->next;
  if (*pp) *pp = q->next;
}
static name_node *name_add(bsc_bam *b, const char *s, uint32_t n, uint32_t flag, uint32_t ix) {
  name_node *q = malloc(sizeof *q + n);
  if (!q) return NULL;
  q->len = n;
  q->flag = flag;
  q->ix = ix;
  memcpy(q->name, s, n);
  name_node **pp = &b->bucket[name_hash(s, n)];
  q->next = *pp;
  *pp = q;
  return q;
}
static void name_clear(bsc_bam *b) {
  for (uint32_t i = 0; i < NAME_BUCKETS; i++) {
    name_node *p = b->bucket[i];
    while (p) {
      name_node *nx = p->next;
      free(p);
      p = nx;
    }
    b->bucket[i] = NULL;
  }
}

static int blk_reserve(blk_buf *k, uint32_t more_tpl, uint64_t more_seq, uint64_t more_ms) {
  if (k->n + more_tpl > k->cap) {
    const uint32_t nc = (k->n + more_tpl) * 2 + 64;
    bsc_raw_template *t = realloc(k->tpl, (size_t)nc * sizeof *t);
    if (t) k->tpl = t;
    name_node **w = t ? realloc(k->waiting, (size_t)nc * sizeof *w) : NULL;
    if (!t || !w) return bsc_set_error(BSC_ERR_NOMEM, "BAM: out of memory");
    k->waiting = w;
    k->cap = nc;
  }
  if (k->seq_len + more_seq > k->seq_cap) {
    const uint64_t nc = (k->seq_len + more_seq) * 2 + 4096;
    uint8_t *s = realloc(k->seq, nc);
    if (!s) return bsc_set_error(BSC_ERR_NOMEM, "BAM: out of memory");
    k->seq = s;
    k->seq_cap = nc;
  }
  if (k->n_ms + more_ms > k->ms_cap) {
    const uint64_t nc = (k->n_ms + more_ms) * 2 + 256;
    bsc_misms *m = realloc(k->ms, nc * sizeof *m);
    if (!m) return bsc_set_error(BSC_ERR_NOMEM, "BAM: out of memory");
    k->ms = m;
    k->ms_cap = nc;
  }
  return BSC_OK;
}

/* ---- open / close -------------------------------------------------------------------------------------------------- */
void bsc_bam_close(bsc_bam *b) {
  if (!b) return;
  bgzf_stop(&b->z); /* the helpers read the file: they are joined before it is closed */
  if (b->z.f) fclose(b->z.f);
  free(b->text);
  if (b->ref_name)
    for (int32_t i = 0; i < b->n_ref; i++) free(b->ref_name[i]);
  free(b->ref_name);
  free(b->ref_len);
  free(b->rec);
  name_clear(b);
  blk_buf *ks[2] = {&b->cur, &b->out};
  for (int i = 0; i < 2; i++) {
    free(ks[i]->tpl);
    free(ks[i]->waiting);
    free(ks[i]->seq);
    free(ks[i]->ms);
  }
  free(b);
}

int bsc_bam_open(const char *path, bsc_bam **out) {
  const char *e = getenv("BSC_BAM_THREADS");
  return bsc_bam_open_threads(path, e ? atoi(e) : 0, out);
}

int bsc_bam_open_threads(const char *path, int n_threads, bsc_bam **out) {
  if (!path || !out) return bsc_set_error(BSC_ERR_ARG, "bsc_bam_open: NULL argument");
  *out = NULL;
  bsc_bam *b = calloc(1, sizeof *b);
  if (!b) return bsc_set_error(BSC_ERR_NOMEM, "bsc_bam_open: out of memory");
  b->curr_tid = b->old_tid = -1;
  b->z.f = fopen(path, "rb");
  if (!b->z.f) {
    const int e = errno;
    bsc_bam_close(b);
    return bsc_set_error(BSC_ERR_ARG, "bsc_bam_open: cannot open '%s': %s", path, strerror(e));
  }
  setvbuf(b->z.f, NULL, _IOFBF, 1 << 20);
  if (bgzf_start(&b->z, n_threads)) {
    bsc_bam_close(b);
    return bsc_set_error(BSC_ERR_NOMEM, "bsc_bam_open: out of memory");
  }
  uint8_t h[8];
  int rc = bgzf_read(&b->z, h, 8);
  if (rc <= 0 || memcmp(h, "BAM\1", 4)) {
    bsc_bam_close(b);
    return rc < 0 ? rc : bsc_set_error(BSC_ERR_ARG, "bsc_bam_open: '%s' is not a BAM file", path);
  }
  b->l_text = le32(h + 4);
  b->text = malloc((size_t)b->l_text + 1);
  if (!b->text || (b->l_text && bgzf_read(&b->z, b->text, b->l_text) <= 0) || bgzf_read(&b->z, h, 4) <= 0) goto bad;
  b->text[b->l_text] = 0;
  b->n_ref = (int32_t)le32(h);
  if (b->n_ref < 0) goto bad;
  b->ref_name = calloc((size_t)b->n_ref + 1, sizeof *b->ref_name);
  b->ref_len = calloc((size_t)b->n_ref + 1, sizeof *b->ref_len);
  if (!b->ref_name || !b->ref_len) goto bad;
  for (int32_t i = 0; i < b->n_ref; i++) {
    if (bgzf_read(&b->z, h, 4) <= 0) goto bad;
    const uint32_t ln = le32(h);
    if (ln == 0 || ln > 65536) goto bad;
    b->ref_name[i] = malloc(ln);
    if (!b->ref_name[i] || bgzf_read(&b->z, b->ref_name[i], ln) <= 0 || bgzf_read(&b->z, h, 4) <= 0) goto bad;
    b->ref_name[i][ln - 1] = 0;
    b->ref_len[i] = le32(h);
  }
  *out = b;
  return BSC_OK;
bad:
  bsc_bam_close(b);
  return bsc_set_error(BSC_ERR_ARG, "bsc_bam_open: '%s': malformed or truncated BAM header", path);
}

int bsc_bam_n_refs(const bsc_bam *b) { return b ? b->n_ref : 0; }
const char *bsc_bam_ref_name(const bsc_bam *b, int i) { return (b && i >= 0 && i < b->n_ref) ? b->ref_name[i] : NULL; }
uint32_t bsc_bam_ref_len(const bsc_bam *b, int i) { return (b && i >= 0 && i < b->n_ref) ? b->ref_len[i] : 0; }
const char *bsc_bam_header_text(const bsc_bam *b) { return b ? b->text : NULL; }
void bsc_bam_filter_counts(const bsc_bam *b, uint64_t cts[15], uint64_t bases[15]) {
  if (!b) return;
  if (cts) memcpy(cts, b->filter_cts, sizeof b->filter_cts);
  if (bases) memcpy(bases, b->filter_bases, sizeof b->filter_bases);
}

/* ---- one alignment record ------------------------------------------------------------------------------------------ */
enum { F_PAIRED = 1, F_PROPER = 2, F_UNMAP = 4, F_MUNMAP = 8, F_REVERSE = 16, F_READ2 = 128, F_SECONDARY = 256, F_QCFAIL = 512,
       F_DUP = 1024, F_SUPP = 2048 };
enum { FLT_NONE, FLT_UNMAPPED, FLT_QC, FLT_SECONDARY, FLT_MATE_UNMAPPED, FLT_DUPLICATE, FLT_NOPOS, FLT_NOMATEPOS, FLT_MISMATCH_CHR,
       FLT_ORIENTATION, FLT_INSERT_SIZE, FLT_NOSEQ, FLT_MAPQ, FLT_NOT_ALIGNED, FLT_PAIR_NOT_FOUND };

typedef struct { /* align_details as get_next_align_details fills it for ONE record, plus what read_input reads off bam1_t */
  int32_t tid;
  uint32_t fwd, rev;      /* forward_position, reverse_position */
  uint32_t span, aln_len; /* reference_span[ix], align_length */
  uint32_t flag;          /* alignment_flag */
  uint32_t l_seq;
  uint8_t mapq, orientation, bs_strand, reverse;
  const char *name;
  uint32_t l_name;
  const uint8_t *cigar; /* n_cigar little-endian dwords, not necessarily aligned */
  uint32_t n_cigar;
  const uint8_t *seq4, *qual, *aux, *end;
} bam_rec;

/* get_bs_strand, src/input_sam.c:144-220: the aligner's conversion tag — XB:A (GEM), ZB:Z (Novoalign), XG:Z (Bowtie /
 * Bismark), ZS:Z (BSMAP), YD:Z (bwa-meth) */
static uint8_t bs_strand_of(const uint8_t *s, const uint8_t *end) {
  static const uint8_t sub_size[256] = {['A'] = 1, ['C'] = 1, ['c'] = 1, ['s'] = 2, ['S'] = 2, ['i'] = 4, ['I'] = 4, ['f'] = 4, ['d'] = 8,
                                        ['Z'] = 'Z', ['H'] = 'H', ['B'] = 'B'};
  uint8_t strand = 0;
  int ok = 1;
  while (ok && s + 4 <= end) {
    enum { UNKNOWN, GEM, BOWTIE, NOVALIGN, BSMAP, BWAMETH } al = UNKNOWN;
    if (s[0] == 'Z') al = s[1] == 'B' ? NOVALIGN : (s[1] == 'S' ? BSMAP : UNKNOWN);
    else if (s[0] == 'X') al = s[1] == 'G' ? BOWTIE : (s[1] == 'B' ? GEM : UNKNOWN);
    else if (s[0] == 'Y' && s[1] == 'D') al = BWAMETH;
    s += 2;
    const uint8_t type = *s++;
    switch (type) {
      case 'A':
        if (al == GEM) strand = *s == 'C' ? 1 : (*s == 'G' ? 2 : strand);
        s++;
        break;
      case 'C': case 'c': s++; break;
      case 'S': case 's':
        if (s + 2 <= end) s += 2; else ok = 0;
        break;
      case 'I': case 'i': case 'f':
        if (s + 4 <= end) s += 4; else ok = 0;
        break;
      case 'd':
        if (s + 8 <= end) s += 8; else ok = 0;
        break;
      case 'Z':
        if (al == BOWTIE || al == NOVALIGN) strand = *s == 'C' ? 1 : (*s == 'G' ? 2 : strand);
        else if (al == BSMAP) strand = *s == '+' ? 1 : (*s == '-' ? 2 : strand);
        else if (al == BWAMETH) strand = *s == 'f' ? 1 : (*s == 'r' ? 2 : strand);
        /* fall through */
      case 'H':
        while (s < end && *s) s++;
        if (s < end) s++; else ok = 0;
        break;
      case 'B': {
        const unsigned sz = sub_size[*s++];
        if (s + 4 <= end && sz != 0) {
          const uint32_t n = le32(s);
          s += 4;
          if ((uint64_t)n * sz <= (uint64_t)(end - s)) s += (size_t)n * sz; else ok = 0;
        } else ok = 0;
      } break;
      default: break; /* an unknown type: the reference moves on without consuming a value */
    }
  }
  return strand;
}

/* Next record -> *r.  Returns 0 = use it, 1 = filtered (counted), -1 = end of input, < -1 = error.
 * get_next_align_details, src/input_sam.c:222-312. */
static int next_record(bsc_bam *b, const bsc_reader_params *par, bam_rec *r, int *filtered) {
  uint8_t h[4];
  int rc = bgzf_read(&b->z, h, 4);
  if (rc == 0) return -1;
  if (rc < 0) return -2;
  const uint32_t bs = le32(h);
  if (bs < 32 || bs > (1u << 29)) return bsc_set_error(BSC_ERR_ARG, "BAM: implausible record size %u", bs), -2;
  if (bs > b->rec_cap) {
    uint8_t *nr = realloc(b->rec, (size_t)bs * 2);
    if (!nr) return bsc_set_error(BSC_ERR_NOMEM, "BAM: out of memory"), -2;
    b->rec = nr;
    b->rec_cap = bs * 2;
  }
  if (bgzf_read(&b->z, b->rec, bs) <= 0) return bsc_set_error(BSC_ERR_ARG, "BAM: input truncated"), -2;
  const uint8_t *p = b->rec;
  const int32_t tid = (int32_t)le32(p), pos = (int32_t)le32(p + 4), mtid = (int32_t)le32(p + 20), mpos = (int32_t)le32(p + 24);
  const int32_t isize = (int32_t)le32(p + 28);
  const uint32_t l_name = p[8], mapq = p[9], n_cigar = p[12] | (uint32_t)p[13] << 8, flag = p[14] | (uint32_t)p[15] << 8, l_seq = le32(p + 16);
  const uint64_t need = 32ull + l_name + 4ull * n_cigar + (l_seq + 1) / 2 + l_seq;
  if (need > bs || l_name == 0) return bsc_set_error(BSC_ERR_ARG, "BAM: malformed record"), -2;
  r->tid = tid;
  r->name = (const char *)p + 32;
  r->l_name = l_name; /* the reference keys the pair table on l_qname bytes, the terminator included */
  r->cigar = p + 32 + l_name;
  r->n_cigar = n_cigar;
  r->seq4 = p + 32 + l_name + 4 * n_cigar;
  r->qual = r->seq4 + (l_seq + 1) / 2;
  r->aux = r->qual + l_seq;
  r->end = p + bs;
  r->l_seq = l_seq;
  int flt = FLT_NONE;
  if ((flag & F_PAIRED) && !par->keep_unmatched) {
    if ((flag & (F_PROPER | F_UNMAP | F_MUNMAP | F_QCFAIL | F_SECONDARY | F_SUPP | F_DUP)) != F_PROPER) {
      if (flag & (F_SECONDARY | F_SUPP)) flt = FLT_SECONDARY;
      else if (flag & F_UNMAP) flt = FLT_UNMAPPED;
      else if (flag & F_MUNMAP) flt = FLT_MATE_UNMAPPED;
      else if (flag & F_QCFAIL) flt = FLT_QC;
      else if (flag & F_DUP) {
        if (!par->ignore_duplicates) flt = FLT_DUPLICATE;
      } else flt = FLT_NOT_ALIGNED;
    }
  } else if (flag & (F_UNMAP | F_QCFAIL | F_SECONDARY | F_SUPP | F_DUP)) {
    if (flag & (F_SECONDARY | F_SUPP)) flt = FLT_SECONDARY;
    else if (flag & F_UNMAP) flt = FLT_UNMAPPED;
    else if (flag & F_QCFAIL) flt = FLT_QC;
    else if (flag & F_DUP) flt = FLT_DUPLICATE;
  }
  int mis_matched = (flag & (F_MUNMAP | F_PROPER)) != F_PROPER;
  const int reverse = (flag & F_REVERSE) != 0, second = (flag & F_READ2) != 0;
  r->reverse = (uint8_t)reverse;
  r->orientation = ((second && reverse) || !(second || reverse)) ? 0 : 1;
  const int mult_seg = (flag & (F_PAIRED | F_MUNMAP)) == F_PAIRED;
  if (reverse) {
    r->fwd = (uint32_t)(mpos + 1);
    r->rev = (uint32_t)(pos + 1);
  } else {
    r->fwd = (uint32_t)(pos + 1);
    r->rev = (uint32_t)(mpos + 1);
  }
  r->mapq = (uint8_t)mapq;
  if (mapq < par->mapq_thresh && !flt) flt = FLT_MAPQ;
  uint32_t aflag = flag;
  if (mult_seg) {
    if (tid != mtid) {
      if (!flt) flt = FLT_MISMATCH_CHR;
      if (par->keep_unmatched) mis_matched = 1;
    }
    if (!flt && (uint64_t)(isize < 0 ? -(int64_t)isize : (int64_t)isize) > par->max_template_len) {
      flt = FLT_INSERT_SIZE;
      if (par->keep_unmatched) mis_matched = 1;
    }
    if (reverse) {
      if (pos < mpos) {
        if (!flt) flt = FLT_ORIENTATION;
        if (par->keep_unmatched) mis_matched = 1;
      }
      if (mis_matched) r->fwd = 0;
    } else {
      if (pos > mpos) {
        if (!flt) flt = FLT_ORIENTATION;
        if (par->keep_unmatched) mis_matched = 1;
      }
      if (mis_matched) r->rev = 0;
    }
  }
  if (!mult_seg || mis_matched) aflag &= ~(uint32_t)F_PAIRED;
  r->flag = aflag;
  *filtered = flt;
  if (flt && !(par->keep_unmatched && (flt == FLT_INSERT_SIZE || flt == FLT_MISMATCH_CHR || flt == FLT_ORIENTATION))) return 1;
  /* CIGAR -> reference span and length in the read (get_bam_misms; the list itself is built when the read is stored) */
  uint32_t span = 0, position = 0;
  for (uint32_t i = 0; i < n_cigar; i++) {
    uint32_t c;
    memcpy(&c, p + 32 + l_name + 4 * i, 4);
    const uint32_t len = c >> 4;
    switch (c & 15u) {
      case 0: case 7: case 8: position += len; span += len; break; /* M = X */
      case 6: case 4: case 1: position += len; break;                /* P S I */
      case 2: span += len; break;                                    /* D */
      default: break;                                                /* H N and the unused codes: nothing */
    }
  }
  r->span = span;
  r->aln_len = position;
  r->bs_strand = bs_strand_of(r->aux, r->end);
  return 0;
}

/* the record's read and mismatch list appended to block k as read `ix` of template t (get_seq_and_qual, get_bam_misms) */
static int store_read(blk_buf *k, bsc_raw_template *t, int ix, const bam_rec *r) {
  int rc = blk_reserve(k, 0, r->l_seq + 2, r->n_cigar);
  if (rc) return rc;
  uint8_t *sq = k->seq + k->seq_len;
  for (uint32_t i = 0; i < r->l_seq; i++) {
    const unsigned c4 = (r->seq4[i >> 1] >> ((~i & 1u) << 2)) & 15u; /* high nibble first */
    unsigned q = r->qual[i];
    if (q > 43) q = 43; /* MAX_QUAL */
    const unsigned base = c4 == 1 ? 1 : (c4 == 2 ? 2 : (c4 == 4 ? 3 : (c4 == 8 ? 4 : 0)));
    sq[i] = base ? (uint8_t)((base - 1) | (q << 2)) : 0; /* anything but A C G T is N: byte 0 */
  }
  t->off[ix] = k->seq_len;
  t->len[ix] = r->l_seq;
  k->seq_len += r->l_seq;
  t->misms_off[ix] = k->n_ms;
  uint32_t position = 0, nm = 0;
  for (uint32_t i = 0; i < r->n_cigar; i++) {
    uint32_t c;
    memcpy(&c, r->cigar + 4 * i, 4);
    const uint32_t len = c >> 4;
    bsc_misms m = {0, position, len};
    switch (c & 15u) {
      case 0: case 7: case 8: position += len; continue;
      case 6: case 4: m.type = BSC_MISMS_SOFT; position += len; break; /* padding is treated as a soft clip (:107-114) */
      case 1: m.type = BSC_MISMS_DEL; position += len; break;          /* inserted in the read */
      case 2: m.type = BSC_MISMS_INS; break;                           /* deleted from the read */
      default: continue;
    }
    k->ms[k->n_ms++] = m;
    nm++;
  }
  t->n_misms[ix] = nm;
  t->mapq[ix] = r->mapq;
  t->reference_span[ix] = r->span;
  return BSC_OK;
}

static void swap_blocks(bsc_bam *b) {
  const blk_buf t = b->out;
  b->out = b->cur;
  b->cur = t;
  b->cur.n = 0;
  b->cur.seq_len = 0;
  b->cur.n_ms = 0;
}

/* get_al_qual over a stored template (bsc_template_qual's rule; src/al_utils.c:19-35) */
static uint32_t tpl_qual(const blk_buf *k, const bsc_raw_template *t) { return bsc_template_qual(t, k->seq); }

static void count_filter(bsc_bam *b, int reason, uint64_t reads, uint64_t bases) {
  b->filter_cts[reason] += reads;
  b->filter_bases[reason] += bases;
}

/* ---- read_input as a generator --------------------------------------------------------------------------------------- */
int bsc_bam_next_block(bsc_bam *b, const bsc_reader_params *par, bsc_read_block *blk) {
  if (!b || !par || !blk) return bsc_set_error(BSC_ERR_ARG, "bsc_bam_next_block: NULL argument");
  memset(blk, 0, sizeof *blk);
  if (b->finished) return 0;
  for (;;) {
    bam_rec r;
    int filtered = 0;
    const int ret = next_record(b, par, &r, &filtered);
    if (ret < -1) return BSC_ERR_ARG;
    if (ret == -1) { /* end of input: the block in hand is the last one (handle_end_of_block, :18-45) */
      b->finished = 1;
      name_clear(b);
      if (!b->cur.n) return 0;
      blk->tid = b->curr_tid;
      blk->y = b->max_pos;
      swap_blocks(b);
      goto hand_out;
    }
    if (ret > 0) {
      count_filter(b, filtered, 1, r.l_seq);
      continue;
    }
    int new_block = 0, new_contig = 0, have_out = 0;
    if (b->curr_tid < 0 || b->curr_tid != r.tid) { /* a new contig is also the start of a new block */
      new_contig = new_block = 1;
      b->old_tid = b->curr_tid;
      b->curr_tid = r.tid;
      if (r.tid < 0 || r.tid >= b->n_ref) return bsc_set_error(BSC_ERR_ARG, "BAM: a mapped record without a valid reference id");
    }
    /* a forward-facing read (or a lone one) is inserted; the backwards-facing mate of a pair joins the stored template */
    int insert = 1;
    if (!new_contig) {
      if ((r.flag & F_PAIRED) && r.fwd > 0 && r.rev > 0) {
        if (r.fwd == r.rev) insert = name_find(b, r.name, r.l_name) == NULL;
        else if (r.reverse) insert = r.fwd > r.rev;
        else insert = r.fwd < r.rev;
      }
      if (insert && b->start_pos > 0) { /* does it still touch the pile-up in hand? a gap of more than one base ends the block */
        if (r.fwd > 0) {
          if (r.fwd > b->max_pos && (r.rev > b->max_pos || r.rev == 0) && r.fwd - b->max_pos > 1) new_block = 1;
        } else if (r.rev > b->max_pos && r.rev - b->max_pos > 1) new_block = 1;
      }
    }
    if (new_block) {
      name_clear(b); /* unmatched reads leave the table; their templates stay in the block with one read */
      if (!insert) return bsc_set_error(BSC_ERR_ARG, "BAM: input not sorted by coordinate (a mate opens a block)");
      b->curr_pos = 0;
      b->start_idx = 0;
      if (b->cur.n) {
        blk->tid = new_contig ? b->old_tid : b->curr_tid; /* the last block of the previous contig carries ITS name */
        blk->y = b->max_pos;
        swap_blocks(b);
        have_out = 1;
      }
      if (new_contig && b->old_tid >= 0) b->old_tid = -1;
      b->max_pos = b->start_pos = 0;
    }
    blk_buf *k = &b->cur;
    { /* the pile-up's extent */
      const uint32_t st = r.reverse ? r.rev : r.fwd, ml = st + r.span;
      if (ml > b->max_pos) b->max_pos = ml;
      if (b->start_pos == 0 || b->start_pos > st) b->start_pos = st;
    }
    const int ix = r.reverse ? 1 : 0;
    int rc = BSC_OK;
    int append = 0; /* store the record as a new template without a waiting entry */
    if (r.flag & F_PAIRED) {
      if (!insert) { /* the mate should be waiting */
        name_node *q = name_find(b, r.name, r.l_name);
        if (q) {
          bsc_raw_template *t = &k->tpl[q->ix];
          if (t->pos[0] != r.fwd || t->pos[1] != r.rev) return bsc_set_error(BSC_ERR_ARG, "BAM: the mates of '%.*s' disagree on their positions", (int)r.l_name, r.name);
          if ((rc = store_read(k, t, ix, &r))) return rc;
          k->waiting[q->ix] = NULL;
          name_unlink(b, q);
          free(q);
        } else {
          count_filter(b, FLT_PAIR_NOT_FOUND, 1, r.l_seq);
          /* its mate may have gone as a duplicate: then it lies inside the block and is dropped; otherwise kept only with
           * keep_unmatched (the reference warns and drops it) */
          int skip = 0;
          if (!par->keep_duplicates && (r.reverse ? r.rev : r.fwd) >= b->start_pos) skip = 1;
          if (!skip && par->keep_unmatched) {
            const uint32_t x = (r.fwd > 0 ? r.fwd : r.rev) + r.aln_len;
            if (x > b->max_pos) b->max_pos = x;
            append = 1;
          }
        }
      } else { /* forward-facing: stored, waiting for its mate; first the duplicate check among templates starting here */
        int skip = 0;
        if (!par->keep_duplicates) {
          const uint32_t pos = r.fwd > 0 ? r.fwd : r.rev;
          if (pos == b->curr_pos) {
            for (uint32_t j = b->start_idx; j < k->n; j++) {
              bsc_raw_template *t1 = &k->tpl[j];
              if (r.fwd != t1->pos[0] || r.rev != t1->pos[1] || r.bs_strand != t1->bs_strand) continue;
              /* mean MAPQ of the reads each holds; the newcomer holds one */
              int maxq1 = 0, kn1 = 0;
              for (int z = 0; z < 2; z++)
                if (t1->len[z] > 0) {
                  maxq1 += t1->mapq[z];
                  kn1++;
                }
              const int maxq = r.mapq;
              maxq1 /= kn1;
              /* score of the newcomer: as a template holding this one read */
              if ((rc = blk_reserve(k, 1, r.l_seq + 2, r.n_cigar))) return rc;
              bsc_raw_template cand;
              memset(&cand, 0, sizeof cand);
              cand.pos[0] = r.fwd;
              cand.pos[1] = r.rev;
              cand.orientation = r.orientation;
              cand.bs_strand = r.bs_strand;
              t1 = &k->tpl[j];
              const uint64_t seq_mark = k->seq_len, ms_mark = k->n_ms;
              if ((rc = store_read(k, &cand, ix, &r))) return rc;
              bsc_raw_template dropped = cand;
              if (maxq1 < maxq || (maxq == maxq1 && tpl_qual(k, t1) < tpl_qual(k, &cand))) { /* the newcomer replaces it */
                name_node *q = name_find(b, r.name, r.l_name);
                if (q && k->waiting[j]) return bsc_set_error(BSC_ERR_ARG, "BAM: duplicate read name '%.*s'", (int)r.l_name, r.name);
                if (!q) q = k->waiting[j];
                if (q) {
                  name_unlink(b, q);
                  free(q);
                }
                const int had_entry = k->waiting[j] != NULL;
                q = name_add(b, r.name, r.l_name, r.flag, j);
                if (!q) return bsc_set_error(BSC_ERR_NOMEM, "BAM: out of memory");
                dropped = *t1;
                *t1 = cand;
                /* the reference re-uses the template's entry when it has one; a fresh entry is not recorded against the
                 * template (al_hash_list keeps its NULL, :300-309) */
                k->waiting[j] = had_entry ? q : NULL;
              } else { /* the newcomer goes: its bytes are released again */
                k->seq_len = seq_mark;
                k->n_ms = ms_mark;
              }
              const uint32_t l1 = dropped.len[0], l2 = dropped.len[1];
              count_filter(b, FLT_DUPLICATE, (l1 && l2) ? 2 : 1, (uint64_t)l1 + l2);
              skip = 1;
            }
          } else {
            b->curr_pos = pos;
            b->start_idx = k->n;
          }
        }
        if (!skip) {
          if (name_find(b, r.name, r.l_name)) return bsc_set_error(BSC_ERR_ARG, "BAM: duplicate read name '%.*s'", (int)r.l_name, r.name);
          if ((rc = blk_reserve(k, 1, 0, 0))) return rc;
          name_node *q = name_add(b, r.name, r.l_name, r.flag, k->n);
          if (!q) return bsc_set_error(BSC_ERR_NOMEM, "BAM: out of memory");
          bsc_raw_template *t = &k->tpl[k->n];
          memset(t, 0, sizeof *t);
          t->pos[0] = r.fwd;
          t->pos[1] = r.rev;
          t->orientation = r.orientation;
          t->bs_strand = r.bs_strand;
          if ((rc = store_read(k, t, ix, &r))) return rc;
          k->waiting[k->n] = q;
          k->n++;
        }
      }
    } else { /* single (non-paired) reads */
      int skip = 0;
      if (!par->keep_duplicates) {
        const uint32_t pos = r.fwd > 0 ? r.fwd : r.rev;
        if (pos == b->curr_pos) {
          for (uint32_t j = b->start_idx; j < k->n; j++) {
            bsc_raw_template *t1 = &k->tpl[j];
            const name_node *w = k->waiting[j];
            if (r.fwd != t1->pos[0] || r.rev != t1->pos[1] || r.bs_strand != t1->bs_strand) continue;
            if (!(w == NULL || (w->flag & 9u) == 9u || (w->flag & 9u) == 0u)) continue;
            if ((rc = blk_reserve(k, 1, r.l_seq + 2, r.n_cigar))) return rc;
            t1 = &k->tpl[j];
            bsc_raw_template cand;
            memset(&cand, 0, sizeof cand);
            cand.pos[0] = r.fwd;
            cand.pos[1] = r.rev;
            cand.orientation = r.orientation;
            cand.bs_strand = r.bs_strand;
            const uint64_t seq_mark = k->seq_len, ms_mark = k->n_ms;
            if ((rc = store_read(k, &cand, ix, &r))) return rc;
            /* the reference compares mapq[0] of both, whichever read they hold (:357) */
            bsc_raw_template dropped = cand;
            if (t1->mapq[0] < cand.mapq[0] || (t1->mapq[0] == cand.mapq[0] && tpl_qual(k, t1) < tpl_qual(k, &cand))) {
              dropped = *t1;
              *t1 = cand;
            } else {
              k->seq_len = seq_mark;
              k->n_ms = ms_mark;
            }
            /* one duplicate read; its bases are added to the PASSED column (:361-364) */
            b->filter_cts[FLT_DUPLICATE]++;
            b->filter_bases[FLT_NONE] += dropped.len[ix];
            skip = 1;
          }
        } else {
          b->curr_pos = pos;
          b->start_idx = k->n;
        }
      }
      append = !skip;
    }
    if (append) {
      if ((rc = blk_reserve(k, 1, 0, 0))) return rc;
      bsc_raw_template *t = &k->tpl[k->n];
      memset(t, 0, sizeof *t);
      t->pos[0] = r.fwd;
      t->pos[1] = r.rev;
      t->orientation = r.orientation;
      t->bs_strand = r.bs_strand;
      if ((rc = store_read(k, t, ix, &r))) return rc;
      k->waiting[k->n] = NULL;
      k->n++;
    }
    if (have_out) goto hand_out;
  }
hand_out:
  { /* what the process thread asserts before it touches the block (src/process_template.c:24-26): 0 < x <= y */
    const uint32_t x0 = b->out.n ? (b->out.tpl[0].pos[0] ? b->out.tpl[0].pos[0] : b->out.tpl[0].pos[1]) : 0;
    if (b->out.n && (x0 == 0 || x0 > blk->y))
      return bsc_set_error(BSC_ERR_ARG, "BAM: a block whose first template starts at %u, right of the block's end %u (a mate position that is "
                                        "negative in the file, or input not sorted by coordinate)", x0, blk->y);
  }
  blk->nr = b->out.n;
  blk->tpl = b->out.tpl;
  blk->seq = b->out.seq;
  blk->seq_bytes = b->out.seq_len;
  blk->misms = b->out.ms;
  blk->n_misms = b->out.n_ms;
  return 1;
}
