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
 * SAM text (plain or BGZF-compressed) is read through the same code: every alignment line is re-encoded as a BAM record
 * (SAM specification sections 1.4 and 4.2) in front of the record decoder.
 *
 * A region (the reference's -r) is served by scanning: bsc_reader_params.region_* keeps the alignments an index query would
 * return.  Not covered: CRAM input, the .bai index itself (no seeking), contig include / exclude lists (every @SQ contig is
 * processed).  The block a call returns stays valid until the next call.
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
  /* SAM text input: the same reader behind a line parser that re-encodes every alignment line as a BAM record */
  int is_sam, sam_plain; /* sam_plain: the file is not BGZF-compressed */
  char *line;
  size_t line_cap;
  uint8_t pend[8];       /* bytes already taken from the stream while its kind was being found out */
  uint32_t n_pend, o_pend;
  int32_t last_tid;      /* reference-name lookups: consecutive records mostly share it */
  /* read_input's variables (src/get_template_vector.c:53-58) */
  int32_t curr_tid, old_tid;
  uint32_t max_pos, start_pos, curr_pos, start_idx;
  blk_buf cur, out;
  name_node *bucket[NAME_BUCKETS];
  int finished;
  uint64_t filter_cts[15], filter_bases[15];
  uint64_t malformed; /* BAM records dropped because their CIGAR does not cover l_seq query bases */
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
  free(b->line);
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

static int sam_read_header(bsc_bam *b);

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
  b->last_tid = -1;
  { /* what kind of file: gzip magic -> BGZF (BAM, or SAM text inside); anything else is read as plain SAM text */
    uint8_t m[2];
    const size_t got = fread(m, 1, 2, b->z.f);
    b->sam_plain = !(got == 2 && m[0] == 0x1f && m[1] == 0x8b);
    if (fseek(b->z.f, 0, SEEK_SET)) {
      bsc_bam_close(b);
      return bsc_set_error(BSC_ERR_ARG, "bsc_bam_open: '%s' is not seekable", path);
    }
  }
  if (bgzf_start(&b->z, b->sam_plain ? 0 : n_threads)) {
    bsc_bam_close(b);
    return bsc_set_error(BSC_ERR_NOMEM, "bsc_bam_open: out of memory");
  }
  uint8_t h[8];
  int rc = 1;
  if (!b->sam_plain) {
    rc = bgzf_read(&b->z, h, 4);
    if (rc < 0) {
      bsc_bam_close(b);
      return rc;
    }
  }
  if (b->sam_plain || rc == 0 || memcmp(h, "BAM\1", 4)) { /* SAM text: hand the bytes already taken back to the line reader */
    if (!b->sam_plain && rc > 0) {
      memcpy(b->pend, h, 4);
      b->n_pend = 4;
    }
    b->is_sam = 1;
    rc = sam_read_header(b);
    if (rc) {
      bsc_bam_close(b);
      return rc;
    }
    if (b->l_text == 0 && b->is_sam == 3 && b->line[0] != '@' && strchr(b->line, '\t') == NULL) {
      bsc_bam_close(b);
      return bsc_set_error(BSC_ERR_ARG, "bsc_bam_open: '%s' is neither a BAM file nor SAM text", path);
    }
    *out = b;
    return BSC_OK;
  }
  if (bgzf_read(&b->z, h + 4, 4) <= 0) goto bad;
  /* l_text and n_ref come from the file, and neither is believed before the bytes behind it have arrived: the text grows
   * with what the stream delivers, the reference list by doubling, so a damaged count costs an error and not gigabytes */
  {
    const uint32_t l_text = le32(h + 4);
    size_t cap = 0;
    while (b->l_text < l_text) {
      const uint32_t step = l_text - b->l_text < (1u << 20) ? l_text - b->l_text : (1u << 20);
      if ((size_t)b->l_text + step + 1 > cap) {
        cap = ((size_t)b->l_text + step) * 2 + 1;
        if (cap > (size_t)l_text + 1) cap = (size_t)l_text + 1;
        char *nt = realloc(b->text, cap);
        if (!nt) goto bad;
        b->text = nt;
      }
      if (bgzf_read(&b->z, b->text + b->l_text, step) <= 0) goto bad;
      b->l_text += step;
    }
    if (!b->text && !(b->text = malloc(1))) goto bad;
    b->text[b->l_text] = 0;
  }
  if (bgzf_read(&b->z, h, 4) <= 0) goto bad;
  const int32_t n_ref = (int32_t)le32(h);
  if (n_ref < 0) goto bad;
  size_t ref_cap = (size_t)(n_ref < 1024 ? n_ref : 1024) + 1;
  b->ref_name = calloc(ref_cap, sizeof *b->ref_name);
  b->ref_len = calloc(ref_cap, sizeof *b->ref_len);
  if (!b->ref_name || !b->ref_len) goto bad;
  for (int32_t i = 0; i < n_ref; i++) { /* b->n_ref counts the entries filled so far: what bsc_bam_close() frees */
    if ((size_t)i + 1 >= ref_cap) {
      ref_cap *= 2;
      char **nn = realloc(b->ref_name, ref_cap * sizeof *b->ref_name);
      if (nn) b->ref_name = nn;
      uint32_t *nl = realloc(b->ref_len, ref_cap * sizeof *b->ref_len);
      if (nl) b->ref_len = nl;
      if (!nn || !nl) goto bad;
    }
    if (bgzf_read(&b->z, h, 4) <= 0) goto bad;
    const uint32_t ln = le32(h);
    if (ln == 0 || ln > 65536) goto bad;
    char *nm = malloc(ln);
    if (!nm) goto bad;
    b->ref_name[i] = nm;
    b->n_ref = i + 1;
    if (bgzf_read(&b->z, nm, ln) <= 0 || bgzf_read(&b->z, h, 4) <= 0) goto bad;
    nm[ln - 1] = 0;
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

uint64_t bsc_bam_malformed(const bsc_bam *b) { return b ? b->malformed : 0; }

/* ---- byte stream under the parsers: BGZF, or the plain file for uncompressed SAM text ------------------------------ */
static int in_read(bsc_bam *b, void *dst, size_t n) { /* 1, 0 = clean end of input, < 0 = error */
  uint8_t *d = (uint8_t *)dst;
  size_t done = 0;
  while (done < n && b->o_pend < b->n_pend) d[done++] = b->pend[b->o_pend++];
  if (done == n) return 1;
  if (b->sam_plain) {
    const size_t got = fread(d + done, 1, n - done, b->z.f);
    if (got == n - done) return 1;
    return (done + got) ? bsc_set_error(BSC_ERR_ARG, "SAM: input truncated") : 0;
  }
  const int r = bgzf_read(&b->z, d + done, n - done);
  if (r == 0 && done) return bsc_set_error(BSC_ERR_ARG, "BAM: input truncated");
  return r;
}

/* one text line (without its terminator) into b->line: 1, 0 = end of input, < 0 = error */
static int in_getline(bsc_bam *b, size_t *len) {
  size_t n = 0;
  if (!b->line) { /* an empty first line must still leave a (terminated) buffer behind */
    b->line = malloc(256);
    if (!b->line) return bsc_set_error(BSC_ERR_NOMEM, "SAM: out of memory");
    b->line_cap = 256;
    b->line[0] = 0;
  }
  for (;;) {
    char c;
    int r;
    if (b->o_pend < b->n_pend) {
      c = (char)b->pend[b->o_pend++];
      r = 1;
    } else if (b->sam_plain) {
      const int ch = getc(b->z.f);
      r = ch == EOF ? 0 : 1;
      c = (char)ch;
    } else {
      if (b->z.o == b->z.n) {
        r = bgzf_fill(&b->z);
        if (r < 0) return r;
        if (r == 0) c = 0;
        else c = (char)b->z.cur[b->z.o++];
      } else {
        r = 1;
        c = (char)b->z.cur[b->z.o++];
      }
    }
    if (r == 0) {
      if (n == 0) return 0;
      break; /* a last line without a newline */
    }
    if (c == '\n') break;
    if (n + 2 > b->line_cap) {
      char *nl = realloc(b->line, (n + 2) * 2 + 256);
      if (!nl) return bsc_set_error(BSC_ERR_NOMEM, "SAM: out of memory");
      b->line = nl;
      b->line_cap = (n + 2) * 2 + 256;
    }
    b->line[n++] = c;
  }
  if (n && b->line[n - 1] == '\r') n--;
  if (b->line) b->line[n] = 0;
  *len = n;
  return 1;
}

static int32_t sam_tid(bsc_bam *b, const char *name, size_t len) {
  if (len == 1 && name[0] == '*') return -1;
  if (b->last_tid >= 0 && strlen(b->ref_name[b->last_tid]) == len && !memcmp(b->ref_name[b->last_tid], name, len)) return b->last_tid;
  for (int32_t i = 0; i < b->n_ref; i++)
    if (strlen(b->ref_name[i]) == len && !memcmp(b->ref_name[i], name, len)) return b->last_tid = i;
  return -2;
}

/* the header lines of a SAM file: the text as it stands, the @SQ SN / LN pairs as the reference list */
static int sam_read_header(bsc_bam *b) {
  size_t cap = 1 << 16, len = 0, ll;
  b->text = malloc(cap);
  if (!b->text) return bsc_set_error(BSC_ERR_NOMEM, "SAM: out of memory");
  int32_t cap_ref = 0;
  for (;;) {
    /* peek: header lines start with '@'; the first other line is an alignment and stays in b->line for next_record */
    const int r = in_getline(b, &ll);
    if (r < 0) return r;
    if (r == 0) {
      b->line_cap = b->line_cap; /* empty file body */
      if (b->line) b->line[0] = 0;
      b->is_sam = 2; /* no alignment line is waiting */
      break;
    }
    if (ll == 0) continue; /* a blank line: skipped, like the ones between alignment lines */
    if (b->line[0] != '@') {
      b->is_sam = 3; /* b->line holds the first alignment line */
      break;
    }
    if (len + ll + 2 > cap) {
      cap = (len + ll + 2) * 2;
      char *nt = realloc(b->text, cap);
      if (!nt) return bsc_set_error(BSC_ERR_NOMEM, "SAM: out of memory");
      b->text = nt;
    }
    memcpy(b->text + len, b->line, ll);
    len += ll;
    b->text[len++] = '\n';
    if (!strncmp(b->line, "@SQ\t", 4)) {
      const char *sn = NULL, *lnp = NULL;
      size_t sn_len = 0;
      for (char *f = b->line + 4; f && *f;) {
        char *e = strchr(f, '\t');
        const size_t fl = e ? (size_t)(e - f) : strlen(f);
        if (fl > 3 && f[0] == 'S' && f[1] == 'N' && f[2] == ':') {
          sn = f + 3;
          sn_len = fl - 3;
        } else if (fl > 3 && f[0] == 'L' && f[1] == 'N' && f[2] == ':')
          lnp = f + 3;
        f = e ? e + 1 : NULL;
      }
      if (!sn || !lnp) return bsc_set_error(BSC_ERR_ARG, "SAM: an @SQ line without SN or LN");
      if (b->n_ref + 1 > cap_ref) {
        cap_ref = cap_ref * 2 + 64;
        char **nn = realloc(b->ref_name, (size_t)cap_ref * sizeof *nn);
        if (nn) b->ref_name = nn;
        uint32_t *nl = nn ? realloc(b->ref_len, (size_t)cap_ref * sizeof *nl) : NULL;
        if (!nn || !nl) return bsc_set_error(BSC_ERR_NOMEM, "SAM: out of memory");
        b->ref_len = nl;
      }
      b->ref_name[b->n_ref] = malloc(sn_len + 1);
      if (!b->ref_name[b->n_ref]) return bsc_set_error(BSC_ERR_NOMEM, "SAM: out of memory");
      memcpy(b->ref_name[b->n_ref], sn, sn_len);
      b->ref_name[b->n_ref][sn_len] = 0;
      b->ref_len[b->n_ref] = (uint32_t)strtoul(lnp, NULL, 10);
      b->n_ref++;
    }
  }
  b->text[len] = 0;
  b->l_text = (uint32_t)len;
  if (!b->ref_name) { /* a header without @SQ lines: an empty list, not a NULL one */
    b->ref_name = calloc(1, sizeof *b->ref_name);
    b->ref_len = calloc(1, sizeof *b->ref_len);
    if (!b->ref_name || !b->ref_len) return bsc_set_error(BSC_ERR_NOMEM, "SAM: out of memory");
  }
  return BSC_OK;
}

/* the alignment line in b->line -> a BAM record in b->rec (SAM specification section 1.4 -> 4.2); returns its size */
static long sam_encode_line(bsc_bam *b, size_t ll) {
  char *f[12];
  int nf = 0;
  char *p = b->line;
  char *aux = NULL;
  for (; nf < 11; nf++) {
    f[nf] = p;
    char *e = strchr(p, '\t');
    if (!e) {
      nf++;
      p = NULL;
      break;
    }
    *e = 0;
    p = e + 1;
  }
  if (nf < 11) return bsc_set_error(BSC_ERR_ARG, "SAM: an alignment line with %d of the 11 mandatory fields", nf);
  aux = p; /* the optional fields, tab separated, or NULL */
  const size_t l_name = strlen(f[0]) + 1;
  if (l_name > 255) return bsc_set_error(BSC_ERR_ARG, "SAM: read name longer than 254 characters");
  const int32_t tid = sam_tid(b, f[2], strlen(f[2]));
  int32_t mtid = (f[6][0] == '=' && !f[6][1]) ? tid : sam_tid(b, f[6], strlen(f[6]));
  if (tid == -2 || mtid == -2) return bsc_set_error(BSC_ERR_ARG, "SAM: reference '%s' is not in the header", tid == -2 ? f[2] : f[6]);
  /* worst-case size: name + cigar ops + sequence + qualities + the optional fields re-encoded.  A re-encoded field can be
   * LONGER than its text (every element of a B array, 1-2 text bytes, becomes 4 bytes; an empty "XX:i:" becomes 7), so the
   * bound is 8 bytes per text byte of the optional part plus slack: tag + type + count (8) per field of >= 5 text bytes
   * and 4 bytes per remaining text byte at most. */
  const size_t l_seq = (f[9][0] == '*' && !f[9][1]) ? 0 : strlen(f[9]);
  size_t n_cig = 0;
  if (!(f[5][0] == '*' && !f[5][1]))
    for (const char *c = f[5]; *c; c++) n_cig += (*c < '0' || *c > '9');
  const size_t aux_len = aux ? ll - (size_t)(aux - b->line) : 0;
  const size_t need = 32 + l_name + 4 * n_cig + (l_seq + 1) / 2 + l_seq + 8 * aux_len + 128;
  if (need > b->rec_cap) {
    uint8_t *nr = realloc(b->rec, need * 2);
    if (!nr) return bsc_set_error(BSC_ERR_NOMEM, "SAM: out of memory");
    b->rec = nr;
    b->rec_cap = (uint32_t)(need * 2);
  }
  uint8_t *r = b->rec;
  const uint32_t flag = (uint32_t)strtoul(f[1], NULL, 10), mapq = (uint32_t)strtoul(f[4], NULL, 10);
  /* POS / PNEXT: 0 .. 2^31 - 1 (section 1.4; a BAM record holds them minus one as int32), TLEN: -2^31 + 1 .. 2^31 - 1; a value
   * outside is an error as in sam_parse1, not a wrapped number */
  const long long pos1 = strtoll(f[3], NULL, 10), mpos1 = strtoll(f[7], NULL, 10), tlen1 = strtoll(f[8], NULL, 10);
  if (pos1 < 0 || pos1 > 0x7fffffffLL || mpos1 < 0 || mpos1 > 0x7fffffffLL || tlen1 < -0x7fffffffLL || tlen1 > 0x7fffffffLL)
    return bsc_set_error(BSC_ERR_ARG, "SAM: position or template length out of range in the line of read '%s'", f[0]);
  const int32_t pos = (int32_t)(pos1 - 1), mpos = (int32_t)(mpos1 - 1), tlen = (int32_t)tlen1;
#define PUT32(off, v)                        \
  do {                                       \
    const uint32_t v_ = (uint32_t)(v);       \
    r[(off)] = (uint8_t)v_;                  \
    r[(off) + 1] = (uint8_t)(v_ >> 8);       \
    r[(off) + 2] = (uint8_t)(v_ >> 16);      \
    r[(off) + 3] = (uint8_t)(v_ >> 24);      \
  } while (0)
  PUT32(0, tid);
  PUT32(4, pos);
  r[8] = (uint8_t)l_name;
  r[9] = (uint8_t)(mapq > 255 ? 255 : mapq);
  r[10] = r[11] = 0; /* the bin: not used by this reader */
  r[12] = (uint8_t)n_cig;
  r[13] = (uint8_t)(n_cig >> 8);
  r[14] = (uint8_t)flag;
  r[15] = (uint8_t)(flag >> 8);
  PUT32(16, l_seq);
  PUT32(20, mtid);
  PUT32(24, mpos);
  PUT32(28, tlen);
  size_t o = 32;
  memcpy(r + o, f[0], l_name);
  o += l_name;
  if (n_cig) {
    if (n_cig > 65535) return bsc_set_error(BSC_ERR_ARG, "SAM: more than 65 535 CIGAR operations");
    for (const char *c = f[5]; *c;) {
      char *e;
      const unsigned long len = strtoul(c, &e, 10);
      static const char ops[] = "MIDNSHP=X";
      const char *q = (*e && e != c) ? strchr(ops, *e) : NULL;
      if (!q || len >= (1ul << 28)) return bsc_set_error(BSC_ERR_ARG, "SAM: malformed CIGAR '%s'", f[5]);
      PUT32(o, (uint32_t)len << 4 | (uint32_t)(q - ops));
      o += 4;
      c = e + 1;
    }
  }
  static const char codes[] = "=ACMGRSVTWYHKDBN";
  memset(r + o, 0, (l_seq + 1) / 2);
  for (size_t i = 0; i < l_seq; i++) {
    char ch = f[9][i];
    if (ch >= 'a' && ch <= 'z') ch = (char)(ch - 32);
    const char *q = strchr(codes, ch);
    const unsigned v = (q && ch) ? (unsigned)(q - codes) : 15u;
    r[o + (i >> 1)] |= (uint8_t)(v << ((~i & 1u) << 2));
  }
  o += (l_seq + 1) / 2;
  if (f[10][0] == '*' && !f[10][1]) memset(r + o, 0xff, l_seq);
  else {
    if (strlen(f[10]) != l_seq) return bsc_set_error(BSC_ERR_ARG, "SAM: '%s': sequence and quality lengths differ", f[0]);
    for (size_t i = 0; i < l_seq; i++) r[o + i] = (uint8_t)(f[10][i] - 33);
  }
  o += l_seq;
  /* optional fields TAG:TYPE:VALUE.  Integers are written as 32-bit ('i'): the reader only needs to step over them */
  for (char *t = aux; t && *t;) {
    char *e = strchr(t, '\t');
    if (e) *e = 0;
    const size_t tl = strlen(t);
    if (tl >= 5 && t[2] == ':' && t[4] == ':') {
      const char ty = t[3], *v = t + 5;
      r[o++] = (uint8_t)t[0];
      r[o++] = (uint8_t)t[1];
      switch (ty) {
        case 'A':
          r[o++] = 'A';
          r[o++] = (uint8_t)v[0];
          break;
        case 'i':
          r[o++] = 'i';
          PUT32(o, (int32_t)strtol(v, NULL, 10));
          o += 4;
          break;
        case 'f': {
          const float fv = strtof(v, NULL);
          uint32_t bits;
          memcpy(&bits, &fv, 4);
          r[o++] = 'f';
          PUT32(o, bits);
          o += 4;
        } break;
        case 'Z': case 'H': {
          const size_t vl = strlen(v);
          r[o++] = (uint8_t)ty;
          memcpy(r + o, v, vl + 1);
          o += vl + 1;
        } break;
        case 'B': { /* B:<type>,v,v,... -> type, count, values; only 32-bit element types are written ('i' / 'f') */
          const char sub = v[0];
          uint32_t cnt = 0;
          for (const char *c = v; *c; c++) cnt += *c == ',';
          r[o++] = 'B';
          r[o++] = (uint8_t)(sub == 'f' ? 'f' : 'i');
          PUT32(o, cnt);
          o += 4;
          const char *c = strchr(v, ',');
          while (c) {
            c++;
            if (sub == 'f') {
              const float fv = strtof(c, NULL);
              uint32_t bits;
              memcpy(&bits, &fv, 4);
              PUT32(o, bits);
            } else
              PUT32(o, (int32_t)strtol(c, NULL, 10));
            o += 4;
            c = strchr(c, ',');
          }
        } break;
        default:
          o -= 2; /* an unknown type: dropped */
          break;
      }
    }
    t = e ? e + 1 : NULL;
  }
#undef PUT32
  return (long)o;
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
  uint32_t bs;
  if (b->is_sam) {
    size_t ll = 0;
    if (b->is_sam == 3) { /* the line the header scan stopped at */
      b->is_sam = 1;
      ll = strlen(b->line);
    } else if (b->is_sam == 2)
      return -1;
    else {
      int rl;
      do {
        rl = in_getline(b, &ll);
        if (rl < 0) return -2;
        if (rl == 0) return -1;
      } while (ll == 0); /* blank lines are skipped */
    }
    const long n = sam_encode_line(b, ll);
    if (n < 0) return -2;
    bs = (uint32_t)n;
  } else {
    uint8_t h[4];
    int rc = in_read(b, h, 4);
    if (rc == 0) return -1;
    if (rc < 0) return -2;
    bs = le32(h);
    if (bs < 32 || bs > (1u << 29)) return bsc_set_error(BSC_ERR_ARG, "BAM: implausible record size %u", bs), -2;
    if (bs > b->rec_cap) {
      uint8_t *nr = realloc(b->rec, (size_t)bs * 2);
      if (!nr) return bsc_set_error(BSC_ERR_NOMEM, "BAM: out of memory"), -2;
      b->rec = nr;
      b->rec_cap = bs * 2;
    }
    if (in_read(b, b->rec, bs) <= 0) return bsc_set_error(BSC_ERR_ARG, "BAM: input truncated"), -2;
  }
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
  if (l_seq && n_cigar) {
    /* A CIGAR whose query length is not l_seq.  htslib, which the reference reads through, refuses it where it parses SAM
     * text (sam_parse1) — an error here too; a binary BAM record is handed over as it is (bam_read1 does not look), and what
     * the reference then does with it is a walk outside the read's bases: here the record is dropped before anything walks it,
     * and the file is read on. */
    uint64_t qlen = 0;
    for (uint32_t i = 0; i < n_cigar; i++) {
      uint32_t c;
      memcpy(&c, r->cigar + 4 * i, 4);
      const uint32_t op = c & 15u;
      if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen += c >> 4; /* M I S = X consume the query */
    }
    if (qlen != l_seq) {
      if (b->is_sam)
        return bsc_set_error(BSC_ERR_ARG, "read '%.*s': CIGAR covers %llu query bases, the sequence has %u", (int)l_name, r->name,
                             (unsigned long long)qlen, l_seq), -2;
      b->malformed++; /* counted apart (bsc_bam_malformed): damaged input must not change the coverage unseen */
      return 2;       /* like a record outside the region: it does not exist for the reader */
    }
  }
  if (par->region_stop) { /* an index query hands over the records that overlap the region: the others do not exist for the reader */
    uint32_t reflen = 0;
    for (uint32_t i = 0; i < n_cigar; i++) {
      uint32_t c;
      memcpy(&c, r->cigar + 4 * i, 4);
      const uint32_t op = c & 15u;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += c >> 4; /* M D N = X consume the reference */
    }
    const int64_t beg = (int64_t)par->region_start - 1, end = par->region_stop;
    const int64_t rend = (int64_t)pos + (reflen ? reflen : 1);
    if (tid != par->region_tid || pos >= end || rend <= beg) return 2;
  }
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
    r->fwd = (uint32_t)mpos + 1u;
    r->rev = (uint32_t)pos + 1u;
  } else {
    r->fwd = (uint32_t)pos + 1u;
    r->rev = (uint32_t)mpos + 1u;
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
    if (ret == 2) continue; /* outside the region */
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
