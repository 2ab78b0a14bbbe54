/*
 * bamstream.c — the host half of the device BAM reader (round 6): a BAM file as a stream of INFLATED bytes in page-locked slabs, with
 * the offset of every alignment record, and nothing else.  What is left on the host of the reference's reader thread
 * (sam_read1 under read_input, src/get_template_vector.c:49-389) is what a GPU cannot start from: the gzip members.
 *
 *   open      the BAM header (text, reference list) from the first blocks; then the INDEX of every BGZF block of the file — offset, sizes,
 *             checksum — built by all helpers at once: each takes a stretch of the file, finds the first block header in it (the 16 fixed
 *             bytes every writer emits) and follows the BSIZE chain to the stretch's end; where two stretches meet, the chain of the first
 *             must end exactly where the second began (else the file is indexed again by one thread, from its first byte).  With the
 *             index every block's place in the slab sequence is known before the first byte is inflated.
 *   helpers   take blocks with one atomic add, in file order: read the payload (pread), inflate straight into the block's place in its
 *             slab (no copy), crc32, and walk the block's records (block_size prefixes) while the bytes are in the core's own cache,
 *             from the block's first record start AS THE HELPER FINDS IT: offset 0 when a record header stands there (htslib's writer never
 *             cuts a record: bgzf_flush_try — one check), else the first offset from which a chain of checked record headers runs on
 *             (htsjdk's writer cuts records where a block is full): record starts + what hangs over.  No word is shared between helpers
 *             but the slab's completion count.
 *   consumer  bsc_bamstream_next: the slab's blocks CHECKED in order (O(1) each: did the predecessor's last record end where this one's
 *             walk began?), a block whose helper guessed wrong (or whose size field is cut in two) is walked again from the true state;
 *             bytes + record starts, ready for one hipMemcpyAsync each.
 *
 * What was measured on the way (50 Mb at 30x, 3.07 GB inflated, a 2 x 64-core host): one mutex and three broadcast condition variables
 * 3.6 GB/s with 16 helpers and 1.0 with 32; the walk in block order by the helpers behind a turn word 4.5 / 1.1 (64 helpers), behind a
 * token passed from cache line to cache line 4.4 / 2.4 — in block order every helper waits for the slowest of its predecessors; the
 * speculative walk 5.8 whatever the number of helpers: the one dispatcher's two preads per block (it parsed the headers as it went) were
 * the ceiling.  Hence the index.
 *
 * BAM only (SAM text and CRAM go through csrc/bamio.c or not at all).  Layout facts: SAM specification 4.1 (BGZF), 4.2 (BAM).
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>
#include <zlib.h>

#include <hip/hip_runtime_api.h>

#include "../../include/bscall_amd.h"

int bsc_set_error(int code, const char *fmt, ...);

#define SEQ_NONE 0xffffffffu
struct bs_blk { /* a block's place in its slab and what its helper's walk found */
  uint32_t boff, isize;
  uint32_t sp_base, sp_n;   /* its record starts: sparse[sp_base .. sp_base + sp_n), relative to the slab */
  uint32_t exit_skip;       /* bytes of its last record that lie in the following blocks */
  uint8_t exit_hdr[4], exit_hdr_n; /* ... or a size field cut in two */
  uint32_t lead;            /* where the helper's walk began: the block's first record start as the helper found it (0: the block starts at a
                               record — htslib's writer —; more: the tail of a record cut by the block boundary lies in front — htsjdk's) */
  uint8_t valid;            /* the walk from there met only plausible sizes */
  uint8_t restart;          /* the chain starts afresh here (a stretch of a contig selection): entry_skip bytes are stepped over first */
  uint32_t entry_skip;
};
struct bs_ent { /* one BGZF block of the file */
  uint64_t file_off; /* of its deflate payload */
  uint32_t clen, isize, crc;
  uint32_t seq;      /* the slab-load it belongs to (SEQ_NONE: an empty block, stepped over) */
  uint32_t boff, blk_ix, sp_base;
  uint32_t restart, entry_skip;
  uint32_t full;     /* not 0: the block inflates to this many bytes and only its first isize go into the stream (the end of a selected stretch
                        whose last record reaches into this block: the bytes up to the block's first record start) */
};
struct bs_seq { /* one slab-load of the stream */
  uint64_t stream_off;
  uint32_t n_bytes, n_ent, last;
};

typedef struct bs_slab_ {
  uint8_t *bytes;     /* page-locked, slab_bytes */
  uint32_t *rec_off;  /* page-locked: starts of the records that BEGIN in this slab, relative to bytes (dense: the consumer's) */
  uint32_t *sparse;   /* the helpers' finds, a region per block */
  struct bs_blk *blk; /* the slab's blocks in stream order */
  uint32_t n_recs;
  uint32_t done;      /* blocks inflated and walked (atomic) */
  uint64_t ready_for; /* sequence number + 1 of the slab-load that is complete in it (under mu) */
  int out;            /* handed to the consumer, not yet released */
  int allocated;      /* its buffers exist (atomic: the slabs behind the first are made by a thread of their own while the helpers start) */
} bs_slab;

struct bsc_bamstream {
  int fd;
  const uint8_t *map; /* the header's blocks only */
  size_t map_len;     /* the file's length */
  /* header */
  char *text;
  uint32_t l_text;
  int32_t n_ref;
  char **ref_name;
  uint32_t *ref_len;
  uint64_t first_rec_off;
  /* index */
  struct bs_ent *ent;
  uint64_t n_ent;
  struct bs_seq *seq;
  uint64_t n_seq;
  /* slabs */
  bs_slab *slab;
  int n_slabs;
  size_t slab_bytes;
  uint32_t rec_cap, blk_cap, sparse_cap;
  pthread_mutex_t mu;
  pthread_cond_t cv_ready;
  int closing, has_err;
  const char *err;      /* first error, raised by the consumer */
  uint64_t next_ent;    /* the next block to take (atomic) */
  uint64_t n_released;  /* slab-loads the consumer has handed back (atomic): load q may be filled once q < n_released + n_slabs */
  /* the true chain state: the consumer's */
  uint64_t w_skip;
  uint8_t w_hdr[4];
  uint32_t w_hdr_n;
  /* consumer */
  uint64_t cons_seq;
  pthread_t *th;
  int n_threads;
  uint64_t total_recs, total_bytes, n_rewalked;
  int dbg_nowalk; /* measurement only (BSC_BAMSTREAM_NOWALK at open): no record walk — the helpers' raw inflate rate */
  int unpinned;   /* no device: ordinary memory (the stream is usable without a GPU; uploads from it are staged by the runtime) */
  int idx_threads, idx_serial; /* how the index was built */
  pthread_t alloc_th;          /* makes the slabs behind the first (page-locking costs ~1 ms per MB) */
  int alloc_running, alloc_dev, sync_made;
  int32_t *sel_tid; /* a selection of contigs (ascending; -1 = the unplaced reads at the file's end), sel_n < 0: the whole file */
  int sel_n;
};

static uint32_t le32(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }

/* the BGZF block at file offset pos: payload, its length, isize, crc, the block's size.  0 = end of file, 1 = ok, -1 = *err set */
static int bgzf_parse(const uint8_t *map, size_t len, size_t pos, const uint8_t **payload, uint32_t *clen, uint32_t *isize, uint32_t *crc,
                      uint32_t *bsize, const char **err) {
  if (pos == len) return 0;
  if (len - pos < 18) {
    *err = "BAM: truncated BGZF header";
    return -1;
  }
  const uint8_t *h = map + pos;
  if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) {
    *err = "BAM: not a BGZF block (truncated file or plain gzip)";
    return -1;
  }
  const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
  if (len - pos < 12u + xlen) {
    *err = "BAM: truncated BGZF header";
    return -1;
  }
  uint32_t bs = 0;
  for (uint32_t p = 0; p + 4 <= xlen;) {
    const uint8_t *x = h + 12 + p;
    const uint32_t sl = x[2] | (uint32_t)x[3] << 8;
    if (x[0] == 'B' && x[1] == 'C' && sl == 2 && p + 6 <= xlen) bs = (x[4] | (uint32_t)x[5] << 8) + 1u;
    p += 4 + sl;
  }
  if (bs < 12 + xlen + 8) {
    *err = "BAM: BGZF block without a valid BC field";
    return -1;
  }
  if (len - pos < bs) {
    *err = "BAM: truncated BGZF block";
    return -1;
  }
  *payload = h + 12 + xlen;
  *clen = bs - 12 - xlen - 8;
  *crc = le32(h + bs - 8);
  *isize = le32(h + bs - 4);
  *bsize = bs;
  if (*isize > 65536) {
    *err = "BAM: BGZF block larger than 64 KiB";
    return -1;
  }
  return 1;
}

/* the same from the file itself, two small reads per block: header (+ extra field) and trailer.  The dispatcher's way — through a mapping
 * it would take the page faults of the whole file, one thread for all helpers (5 GB/s of inflated bytes was the ceiling whatever their number) */
static int bgzf_peek(int fd, size_t len, size_t pos, uint64_t *payload_off, uint32_t *clen, uint32_t *isize, uint32_t *crc, uint32_t *bsize, const char **err) {
  if (pos == len) return 0;
  uint8_t h[18 + 256];
  if (len - pos < 18 || pread(fd, h, 18, (off_t)pos) != 18) {
    *err = "BAM: truncated BGZF header";
    return -1;
  }
  if (h[0] != 0x1f || h[1] != 0x8b || h[2] != 8 || !(h[3] & 4)) {
    *err = "BAM: not a BGZF block (truncated file or plain gzip)";
    return -1;
  }
  const uint32_t xlen = h[10] | (uint32_t)h[11] << 8;
  uint32_t bs = 0;
  if (xlen == 6 && h[12] == 'B' && h[13] == 'C' && h[14] == 2 && h[15] == 0) bs = (h[16] | (uint32_t)h[17] << 8) + 1u; /* every writer's layout */
  else { /* walk the extra field */
    uint8_t *x = malloc((size_t)xlen + 1);
    if (!x || len - pos < 12u + xlen || pread(fd, x, xlen, (off_t)(pos + 12)) != (ssize_t)xlen) {
      free(x);
      *err = "BAM: truncated BGZF header";
      return -1;
    }
    for (uint32_t p = 0; p + 4 <= xlen;) {
      const uint32_t sl = x[p + 2] | (uint32_t)x[p + 3] << 8;
      if (x[p] == 'B' && x[p + 1] == 'C' && sl == 2 && p + 6 <= xlen) bs = (x[p + 4] | (uint32_t)x[p + 5] << 8) + 1u;
      p += 4 + sl;
    }
    free(x);
  }
  if (bs < 12 + xlen + 8) {
    *err = "BAM: BGZF block without a valid BC field";
    return -1;
  }
  uint8_t t[8];
  if (len - pos < bs || pread(fd, t, 8, (off_t)(pos + bs - 8)) != 8) {
    *err = "BAM: truncated BGZF block";
    return -1;
  }
  *payload_off = pos + 12 + xlen;
  *clen = bs - 12 - xlen - 8;
  *crc = le32(t);
  *isize = le32(t + 4);
  *bsize = bs;
  if (*isize > 65536) {
    *err = "BAM: BGZF block larger than 64 KiB";
    return -1;
  }
  return 1;
}

static int use_zlib = -1; /* BSC_BAMSTREAM_ZLIB in the environment: zlib's inflate and crc32 instead of csrc/inflate_fast.c (the A/B) */
static const char *bgzf_inflate_to(const uint8_t *payload, uint32_t clen, uint8_t *dst, uint32_t isize, uint32_t crc) {
  if (use_zlib < 0) use_zlib = getenv("BSC_BAMSTREAM_ZLIB") != NULL;
  if (!use_zlib) {
    if (bsc_inflate_raw(payload, clen, dst, isize)) return "BAM: corrupt BGZF block";
    if (bsc_crc32(dst, isize) != crc) return "BAM: BGZF checksum mismatch";
    return NULL;
  }
  z_stream z;
  memset(&z, 0, sizeof z);
  if (inflateInit2(&z, -15) != Z_OK) return "BAM: zlib initialisation failed";
  z.next_in = (Bytef *)payload;
  z.avail_in = clen;
  z.next_out = dst;
  z.avail_out = isize;
  const int r = inflate(&z, Z_FINISH);
  const uint32_t produced = (uint32_t)z.total_out;
  inflateEnd(&z);
  if (r != Z_STREAM_END || produced != isize) return "BAM: corrupt BGZF block";
  if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), dst, isize) != crc) return "BAM: BGZF checksum mismatch";
  return NULL;
}

/* the records that start in p[0 .. n) of slab s (the block lies at s->bytes + boff); the chain's state carries over */
static const char *walk_block(bsc_bamstream *b, bs_slab *s, uint32_t boff, const uint8_t *p, uint32_t n) {
  uint32_t o = 0;
  while (o < n) {
    if (b->w_skip) {
      const uint64_t take = b->w_skip < (uint64_t)(n - o) ? b->w_skip : (uint64_t)(n - o);
      o += (uint32_t)take;
      b->w_skip -= take;
      continue;
    }
    uint32_t bsz;
    if (b->w_hdr_n == 0) { /* a record starts here */
      if (s->n_recs >= b->rec_cap) return "BAM: more records in a slab than its table holds";
      s->rec_off[s->n_recs++] = boff + o;
      if (n - o >= 4) {
        bsz = le32(p + o);
        o += 4;
      } else {
        while (o < n) b->w_hdr[b->w_hdr_n++] = p[o++];
        break; /* the size's other bytes are the next block's first */
      }
    } else {
      while (b->w_hdr_n < 4 && o < n) b->w_hdr[b->w_hdr_n++] = p[o++];
      if (b->w_hdr_n < 4) break;
      bsz = le32(b->w_hdr);
      b->w_hdr_n = 0;
    }
    if (bsz < 32 || bsz > (1u << 29)) return "BAM: implausible record size";
    b->w_skip = bsz;
  }
  return NULL;
}

/* the helper's walk: the block as if a record started at its first byte */
static int64_t first_record_in(const bsc_bamstream *b, const uint8_t *buf, uint32_t n);
/* The helper's walk of its own block, without knowing what hangs over from the block before: from the block's first record start — offset 0
 * when a record header stands there (one check: every block of an htslib-written file), else the first offset from which a chain of checked
 * headers runs on (first_record_in: files whose writer cuts records at block ends) — the consumer accepts the finds if the predecessor's
 * overhang ends exactly there. */
static void walk_speculative(const bsc_bamstream *b, struct bs_blk *k, uint32_t *sparse, const uint8_t *p) {
  const uint32_t n = k->isize;
  uint32_t o = 0, cnt = 0;
  k->valid = 1;
  k->exit_skip = 0;
  k->exit_hdr_n = 0;
  k->lead = 0;
  k->sp_n = 0;
  if (n >= 36u) {
    const int64_t f = first_record_in(b, p, n);
    if (f < 0) { /* no record header inside: the middle of a long record, or damage — the consumer's walk says which */
      k->valid = 0;
      return;
    }
    k->lead = o = (uint32_t)f;
  }
  while (o < n) {
    sparse[k->sp_base + cnt++] = k->boff + o;
    if (n - o < 4) {
      while (o < n) k->exit_hdr[k->exit_hdr_n++] = p[o++];
      break;
    }
    const uint32_t bsz = le32(p + o);
    if (bsz < 32 || bsz > (1u << 29)) {
      k->valid = 0; /* not a record start, if the assumption was wrong; an error, if it was right: the consumer's walk says which */
      break;
    }
    const uint64_t end = (uint64_t)o + 4u + bsz;
    if (end > n) {
      k->exit_skip = (uint32_t)(end - n);
      break;
    }
    o = (uint32_t)end;
  }
  k->sp_n = cnt;
}

static void set_err(bsc_bamstream *b, const char *e) { /* mu held */
  if (!b->err) b->err = e;
  __atomic_store_n(&b->has_err, 1, __ATOMIC_RELEASE);
  pthread_cond_broadcast(&b->cv_ready);
}

/* ---- the index ------------------------------------------------------------------------------------------------------------------ */
typedef struct {
  bsc_bamstream *b;
  size_t from, limit; /* the stretch: blocks that START in [first block at or after from, limit) */
  int first;          /* from is a block start (the file's first stretch) */
  struct bs_ent *ent;
  size_t n, cap;
  size_t start, end;  /* where its chain began / ended (end >= limit) */
  const char *err;
  int found;
} idx_job;

static const uint8_t bgzf_magic[16] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0};

static void *idx_scan(void *a) {
  idx_job *j = (idx_job *)a;
  bsc_bamstream *b = j->b;
  size_t pos = j->from;
  j->found = 0;
  j->err = NULL;
  j->n = 0;
  if (!j->first) { /* the first header in the stretch: the fixed bytes, confirmed by the header behind the block they announce */
    const size_t win = 1u << 18;
    uint8_t *w = malloc(win + 16);
    if (!w) {
      j->err = "BAM: out of memory";
      return NULL;
    }
    int ok = 0;
    while (pos < j->limit && !ok) {
      const size_t want = b->map_len - pos < win + 15 ? b->map_len - pos : win + 15;
      const ssize_t got = pread(b->fd, w, want, (off_t)pos);
      if (got < 16) break;
      size_t o = 0;
      while (o + 16 <= (size_t)got) {
        const uint8_t *m = memmem(w + o, (size_t)got - o, bgzf_magic, 16);
        if (!m) break;
        const size_t c = pos + (size_t)(m - w);
        if (c >= j->limit) break;
        uint64_t po;
        uint32_t cl, is, cr, bs;
        const char *e = NULL;
        if (bgzf_peek(b->fd, b->map_len, c, &po, &cl, &is, &cr, &bs, &e) == 1) {
          const int r2 = c + bs == b->map_len ? 1 : bgzf_peek(b->fd, b->map_len, c + bs, &po, &cl, &is, &cr, &bs, &e);
          if (r2 == 1) {
            pos = c;
            ok = 1;
            break;
          }
        }
        o = (size_t)(m - w) + 1;
      }
      if (!ok) pos += (size_t)got - 15;
    }
    free(w);
    if (!ok) return NULL; /* no block starts in this stretch (or none this scan recognises) */
  }
  j->found = 1;
  j->start = pos;
  while (pos < j->limit) {
    uint64_t po;
    uint32_t cl, is, cr, bs;
    const char *e = NULL;
    const int r = bgzf_peek(b->fd, b->map_len, pos, &po, &cl, &is, &cr, &bs, &e);
    if (r == 0) break;
    if (r < 0) {
      j->err = e;
      break;
    }
    if (j->n == j->cap) {
      j->cap = j->cap * 2 + 1024;
      struct bs_ent *ne = realloc(j->ent, j->cap * sizeof *ne);
      if (!ne) {
        j->err = "BAM: out of memory";
        break;
      }
      j->ent = ne;
    }
    struct bs_ent *t = &j->ent[j->n++];
    t->file_off = po;
    t->clen = cl;
    t->isize = is;
    t->crc = cr;
    t->seq = SEQ_NONE;
    t->restart = t->entry_skip = t->full = 0;
    pos += bs;
  }
  j->end = pos;
  return NULL;
}

/* ---- a selection of contigs: the stretches of the file that hold their records --------------------------------------------------------
 * The file is sorted by (contig, position), so the contig of the first record that STARTS in a block grows with the block number: a binary
 * search over blocks, each probe one block inflated, finds where a contig's records begin and end — no index file.  htslib's writer starts
 * every block at a record (offset 0 is the answer at once); htsjdk's cuts records wherever a block is full, so the first record start of a
 * block is FOUND: the first offset from which a chain of record headers — each checked field by field against what a BAM record can hold —
 * runs to the block's end.  The stretch of contig c: from the last block whose first record lies before c (it may hold c's first records;
 * the bytes in front of that record belong to a record of an earlier contig and are stepped over) to the last block whose first record is
 * of c or before, plus the bytes of the following block(s) up to their first record start (the tail of c's last record, when it is cut);
 * stretches that touch are merged; records of other contigs inside a stretch are dropped by the record parser's contig filter
 * (csrc/bamdev_core.h). */
static int plausible_record(const bsc_bamstream *b, const uint8_t *p, uint32_t avail) { /* avail >= 36 bytes from the record's size field on */
  const uint32_t bs = le32(p);
  const int32_t t = (int32_t)le32(p + 4), pos = (int32_t)le32(p + 8), mt = (int32_t)le32(p + 24), mpos = (int32_t)le32(p + 28); /* refID, pos, next_refID, next_pos */
  const uint32_t l_name = p[12], n_cig = p[16] | (uint32_t)p[17] << 8, l_seq = le32(p + 20);
  if (bs < 32 || bs > (1u << 29) || t < -1 || t >= b->n_ref || mt < -1 || mt >= b->n_ref || l_name == 0 || pos < -1 || mpos < -1) return 0;
  if (t >= 0 && pos >= 0 && (uint32_t)pos > b->ref_len[t]) return 0;
  if (mt >= 0 && mpos >= 0 && (uint32_t)mpos > b->ref_len[mt]) return 0;
  if (32ull + l_name + 4ull * n_cig + ((uint64_t)l_seq + 1) / 2 + l_seq > bs) return 0;
  /* the name: printable characters and its terminator, as far as the bytes at hand show it */
  for (uint32_t i = 0; i < l_name && 36u + i < avail; i++) {
    const uint8_t c = p[36 + i];
    if (i + 1 == l_name ? c != 0 : (c < 33 || c > 126)) return 0;
  }
  return 1;
}
/* the offset of the first record that starts in buf[0 .. n), or -1 (none whose fixed fields lie inside the block: a neighbour speaks) */
static int64_t first_record_in(const bsc_bamstream *b, const uint8_t *buf, uint32_t n) {
  for (uint32_t o = 0; o + 36u <= n; o++) {
    if (!plausible_record(b, buf + o, n - o)) continue;
    /* the chain from o: every header that still lies inside the block */
    uint64_t q = (uint64_t)o + 4u + le32(buf + o);
    int ok = 1;
    for (int hops = 0; hops < 6 && ok && q + 36u <= n; hops++) {
      ok = plausible_record(b, buf + q, n - (uint32_t)q);
      q += 4ull + le32(buf + q);
    }
    if (ok) return o;
  }
  return -1;
}
/* *tid: the contig of the first record that starts in block k (-2: header bytes only, -3: no record starts in it); *at: where it starts */
static int first_tid_of(bsc_bamstream *b, uint64_t k, uint64_t k_hdr, uint32_t hdr_skip, uint8_t *raw, uint8_t *buf, int64_t *tid, uint32_t *at) {
  *at = 0;
  if (k < k_hdr) {
    *tid = -2; /* header bytes only: before every contig */
    return BSC_OK;
  }
  const struct bs_ent *e = &b->ent[k];
  if (e->isize == 0) {
    *tid = -3; /* an empty block: the caller looks at a neighbour */
    return BSC_OK;
  }
  if (pread(b->fd, raw, e->clen, (off_t)e->file_off) != (ssize_t)e->clen) return bsc_set_error(BSC_ERR_ARG, "BAM: read error");
  const char *er = bgzf_inflate_to(raw, e->clen, buf, e->isize, e->crc);
  if (er) return bsc_set_error(BSC_ERR_ARG, "%s", er);
  int64_t o;
  if (k == k_hdr) { /* the first record of the file: behind the header, wherever that ends */
    if (hdr_skip >= e->isize) {
      *tid = -2;
      return BSC_OK;
    }
    if (e->isize - hdr_skip < 36u) { /* (its fixed fields are cut: the next block's first record speaks, this block stays in front of everything) */
      *tid = -2;
      *at = hdr_skip;
      return BSC_OK;
    }
    if (!plausible_record(b, buf + hdr_skip, e->isize - hdr_skip)) return bsc_set_error(BSC_ERR_ARG, "BAM: malformed first record");
    o = hdr_skip;
  } else if ((o = first_record_in(b, buf, e->isize)) < 0) {
    *tid = -3;
    return BSC_OK;
  }
  const int32_t t = (int32_t)le32(buf + o + 4);
  *at = (uint32_t)o;
  *tid = t < 0 ? (int64_t)1 << 40 : t; /* unplaced reads come last */
  return BSC_OK;
}

static int select_contigs(bsc_bamstream *b) {
  /* the block that holds the first record, and where in it */
  uint64_t cum = 0, k_hdr = b->n_ent;
  uint32_t hdr_skip = 0;
  for (uint64_t k = 0; k < b->n_ent; k++) {
    if (cum + b->ent[k].isize > b->first_rec_off) {
      k_hdr = k;
      hdr_skip = (uint32_t)(b->first_rec_off - cum);
      break;
    }
    cum += b->ent[k].isize;
  }
  uint8_t *raw = malloc(65536 + 64), *buf = malloc(65536);
  struct {
    uint64_t lo, hi;
  } *rg = calloc((size_t)b->sel_n + 1, sizeof *rg);
  int n_rg = 0, rc = BSC_OK;
  if (!raw || !buf || !rg) rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of memory");
  for (int i = 0; i < b->sel_n && !rc && k_hdr < b->n_ent; i++) {
    const int64_t c = b->sel_tid[i] < 0 ? (int64_t)1 << 40 : b->sel_tid[i];
    /* lo: the last block whose first record is before c; hi: the last whose first record is of c or before (both >= k_hdr) */
    uint64_t bound[2];
    for (int w = 0; w < 2 && !rc; w++) {
      uint64_t a = k_hdr, z = b->n_ent; /* invariant: blocks < a satisfy the predicate (or a == k_hdr), blocks >= z do not */
      while (a + 1 < z && !rc) {
        uint64_t m = a + (z - a) / 2, mm = m;
        int64_t t = -3;
        uint32_t at_unused;
        while (mm < z && !rc) { /* empty blocks have no record to look at: the next one speaks for them */
          rc = first_tid_of(b, mm, k_hdr, hdr_skip, raw, buf, &t, &at_unused);
          if (t != -3) break;
          mm++;
        }
        if (rc) break;
        if (mm >= z || t == -3) {
          z = m;
          continue;
        }
        const int pred = w == 0 ? t < c : t <= c;
        if (pred) a = mm;
        else z = m;
      }
      bound[w] = a;
    }
    if (rc) break;
    if (bound[1] < bound[0]) bound[1] = bound[0];
    if (n_rg && bound[0] <= rg[n_rg - 1].hi + 1) { /* touches the stretch before it */
      if (bound[1] > rg[n_rg - 1].hi) rg[n_rg - 1].hi = bound[1];
    } else {
      rg[n_rg].lo = bound[0];
      rg[n_rg].hi = bound[1];
      n_rg++;
    }
  }
  if (!rc) {
    uint64_t total = 0;
    for (int i = 0; i < n_rg; i++) total += rg[i].hi - rg[i].lo + 1;
    /* (+ the blocks behind a stretch that hold the tail of its last record: as many as lie in front of the next record start — one, unless a
     * record is longer than a block) */
    uint64_t cap = total + (uint64_t)n_rg * 2u + 1u;
    struct bs_ent *ne = malloc(cap * sizeof *ne);
    if (!ne) rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of memory");
    else {
      uint64_t o = 0;
      for (int i = 0; i < n_rg && !rc; i++) {
        const uint64_t first = o;
        if (o + (rg[i].hi - rg[i].lo + 1) + 2u > cap) { /* (the tails behind the stretches before took more than was set aside) */
          cap = o + (rg[i].hi - rg[i].lo + 1) + total + 16u;
          struct bs_ent *nn = realloc(ne, cap * sizeof *ne);
          if (!nn) {
            rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of memory");
            break;
          }
          ne = nn;
        }
        for (uint64_t k = rg[i].lo; k <= rg[i].hi; k++) ne[o++] = b->ent[k];
        /* the stretch's first block with bytes restarts the chain: at the header's end, or at the block's first record start (what lies in
         * front of it is the tail of a record of an earlier contig) */
        for (uint64_t q = first; q < o && !rc; q++)
          if (ne[q].isize) {
            int64_t t;
            uint32_t at = 0;
            rc = first_tid_of(b, rg[i].lo + (q - first), k_hdr, hdr_skip, raw, buf, &t, &at);
            ne[q].restart = 1;
            ne[q].entry_skip = t == -3 ? ne[q].isize : at; /* (no record starts in it: all of it is that tail) */
            break;
          }
        /* the tail of the stretch's last record, when a block boundary cuts it: the bytes of the following blocks up to their first record start */
        const uint64_t stop = i + 1 < n_rg ? rg[i + 1].lo : b->n_ent;
        for (uint64_t k = rg[i].hi + 1; k < stop && !rc; k++) {
          if (b->ent[k].isize == 0) continue;
          int64_t t;
          uint32_t at = 0;
          rc = first_tid_of(b, k, k_hdr, hdr_skip, raw, buf, &t, &at);
          if (rc) break;
          if (t != -3 && at == 0) break; /* starts at a record (htslib's writer: always) */
          if (o + 1 >= cap) {
            cap = cap * 2u + 16u;
            struct bs_ent *nn = realloc(ne, cap * sizeof *ne);
            if (!nn) {
              rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of memory");
              break;
            }
            ne = nn;
          }
          ne[o] = b->ent[k];
          if (t != -3) { /* up to its first record start, then the stretch is over */
            ne[o].full = ne[o].isize;
            ne[o].isize = at;
            o++;
            break;
          }
          o++; /* no record starts in it: all of it, and on */
        }
      }
      if (rc) free(ne);
      else {
        free(b->ent);
        b->ent = ne;
        b->n_ent = o;
        b->w_skip = 0; /* every stretch says where it starts */
      }
    }
  }
  free(raw);
  free(buf);
  free(rg);
  return rc;
}

/* every block of the file -> b->ent, then every block's place.  BSC_OK or an error code (message set) */
static int build_index(bsc_bamstream *b, int n_threads) {
  const size_t F = b->map_len;
  int T = n_threads < 1 ? 1 : (n_threads > 64 ? 64 : n_threads);
  while (T > 1 && F / (size_t)T < (4u << 20)) T--; /* stretches of 4 MiB at least */
  idx_job *job = calloc((size_t)T, sizeof *job);
  pthread_t *th = calloc((size_t)T, sizeof *th);
  if (!job || !th) {
    free(job);
    free(th);
    return bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of memory");
  }
  const size_t R = F / (size_t)T;
  for (int i = 0; i < T; i++) {
    job[i].b = b;
    job[i].from = (size_t)i * R;
    job[i].limit = i + 1 == T ? F : (size_t)(i + 1) * R;
    job[i].first = i == 0;
  }
  int started = 0;
  for (int i = 1; i < T; i++, started++)
    if (pthread_create(&th[i], NULL, idx_scan, &job[i])) break;
  idx_scan(&job[0]);
  for (int i = 1; i <= started; i++) pthread_join(th[i], NULL);
  int good = started == T - 1;
  for (int i = 0; i < T && good; i++) {
    if (job[i].err || !job[i].found) good = 0;
    else if (i + 1 < T && job[i + 1].found && job[i].end != job[i + 1].start) good = 0; /* the chains do not meet: a false start */
  }
  b->idx_threads = T;
  if (!good && T > 1) { /* one thread, from the file's first byte: the chain itself */
    for (int i = 1; i < T; i++) {
      free(job[i].ent);
      job[i].ent = NULL;
      job[i].n = 0;
    }
    job[0].limit = F;
    job[0].cap = job[0].n = 0;
    free(job[0].ent);
    job[0].ent = NULL;
    idx_scan(&job[0]);
    T = 1;
    b->idx_serial = 1;
  }
  int rc = BSC_OK;
  if (job[0].err && T == 1) rc = bsc_set_error(BSC_ERR_ARG, "%s", job[0].err);
  size_t total = 0;
  for (int i = 0; i < T; i++) total += job[i].n;
  if (!rc) {
    b->ent = malloc((total + 1) * sizeof *b->ent);
    if (!b->ent) rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of memory");
  }
  if (!rc) {
    size_t o = 0;
    for (int i = 0; i < T; i++) {
      if (job[i].n) memcpy(b->ent + o, job[i].ent, job[i].n * sizeof *b->ent);
      o += job[i].n;
    }
    b->n_ent = total;
  }
  for (int i = 0; i < T; i++) free(job[i].ent);
  free(job);
  free(th);
  if (rc) return rc;
  if (b->sel_n >= 0 && (rc = select_contigs(b))) return rc;
  /* every block's place: slab-loads filled in stream order, a block never cut */
  size_t cap_seq = 0;
  uint64_t stream = 0;
  uint32_t fill = 0, nblk = 0, sp = 0;
  int open_ = 0;
  for (uint64_t k = 0; k < b->n_ent; k++) {
    struct bs_ent *e = &b->ent[k];
    if (e->isize == 0) continue;
    if (open_ && ((size_t)fill + e->isize > b->slab_bytes || nblk >= b->blk_cap)) open_ = 0;
    if (!open_) {
      if (b->n_seq == cap_seq) {
        cap_seq = cap_seq * 2 + 256;
        struct bs_seq *ns = realloc(b->seq, cap_seq * sizeof *ns);
        if (!ns) return bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of memory");
        b->seq = ns;
      }
      struct bs_seq *q = &b->seq[b->n_seq++];
      q->stream_off = stream;
      q->n_bytes = q->n_ent = q->last = 0;
      fill = nblk = sp = 0;
      open_ = 1;
    }
    if (b->n_seq > 0xfffffff0ull) return bsc_set_error(BSC_ERR_ARG, "bsc_bamstream_open: file too large");
    struct bs_seq *q = &b->seq[b->n_seq - 1];
    e->seq = (uint32_t)(b->n_seq - 1);
    e->boff = fill;
    e->blk_ix = nblk++;
    e->sp_base = sp;
    sp += e->isize / 36u + 1u; /* a record is 36 bytes at least: no block has more starts than that */
    fill += e->isize;
    stream += e->isize;
    q->n_bytes = fill;
    q->n_ent = nblk;
  }
  if (b->n_seq) b->seq[b->n_seq - 1].last = 1;
  return BSC_OK;
}

/* one slab's buffers; 0 or -1 */
static int slab_alloc(bsc_bamstream *b, int i) {
  bs_slab *s = &b->slab[i];
  if (!b->unpinned) {
    s->bytes = bsc_alloc_host(b->slab_bytes);
    if (!s->bytes && i == 0) b->unpinned = 1; /* no device at all: the stream still works, from ordinary memory */
    else s->rec_off = bsc_alloc_host((size_t)b->rec_cap * 4u);
  }
  if (b->unpinned) {
    void *p1 = NULL, *p2 = NULL;
    if (posix_memalign(&p1, 4096, b->slab_bytes) || posix_memalign(&p2, 4096, (size_t)b->rec_cap * 4u)) {
      free(p1);
      return -1;
    }
    s->bytes = p1;
    s->rec_off = p2;
  }
  s->sparse = malloc((size_t)b->sparse_cap * 4u);
  s->blk = malloc((size_t)b->blk_cap * sizeof(struct bs_blk));
  if (!s->bytes || !s->rec_off || !s->sparse || !s->blk) return -1;
  __atomic_store_n(&s->allocated, 1, __ATOMIC_RELEASE);
  return 0;
}
static void *alloc_main(void *arg) {
  bsc_bamstream *b = (bsc_bamstream *)arg;
  if (b->alloc_dev >= 0) (void)hipSetDevice(b->alloc_dev);
  for (int i = 1; i < b->n_slabs; i++)
    if (slab_alloc(b, i)) {
      pthread_mutex_lock(&b->mu);
      set_err(b, "BAM: out of (page-locked) memory");
      pthread_mutex_unlock(&b->mu);
      break;
    }
  return NULL;
}

/* ---- the helpers ----------------------------------------------------------------------------------------------------------------- */
static void *helper(void *arg) {
  bsc_bamstream *b = (bsc_bamstream *)arg;
  uint8_t *raw = malloc(65536 + 64); /* one compressed block */
  if (!raw) {
    pthread_mutex_lock(&b->mu);
    set_err(b, "BAM: out of memory");
    pthread_mutex_unlock(&b->mu);
    return NULL;
  }
  for (;;) {
    const uint64_t k = __atomic_fetch_add(&b->next_ent, 1, __ATOMIC_RELAXED);
    if (k >= b->n_ent || __atomic_load_n(&b->closing, __ATOMIC_RELAXED) || __atomic_load_n(&b->has_err, __ATOMIC_RELAXED)) break;
    const struct bs_ent *e = &b->ent[k];
    if (e->seq == SEQ_NONE) continue;
    /* its slab: free once the consumer has handed back the load that used it before */
    for (unsigned spins = 0; (uint64_t)e->seq >= __atomic_load_n(&b->n_released, __ATOMIC_ACQUIRE) + (uint64_t)b->n_slabs; spins++) {
      if (__atomic_load_n(&b->closing, __ATOMIC_RELAXED)) {
        free(raw);
        return NULL;
      }
      if (spins < 64) __builtin_ia32_pause();
      else {
        struct timespec ts = {0, 50000}; /* the consumer is the slower side: nothing to hurry for */
        nanosleep(&ts, NULL);
      }
    }
    bs_slab *s = &b->slab[e->seq % (uint32_t)b->n_slabs];
    while (!__atomic_load_n(&s->allocated, __ATOMIC_ACQUIRE)) { /* (the first passes over the ring only) */
      if (__atomic_load_n(&b->closing, __ATOMIC_RELAXED) || __atomic_load_n(&b->has_err, __ATOMIC_RELAXED)) {
        free(raw);
        return NULL;
      }
      struct timespec ts = {0, 50000};
      nanosleep(&ts, NULL);
    }
    struct bs_blk *kb = &s->blk[e->blk_ix];
    kb->boff = e->boff;
    kb->isize = e->isize;
    kb->sp_base = e->sp_base;
    kb->sp_n = 0;
    kb->lead = 0;
    kb->valid = 0;
    kb->restart = (uint8_t)e->restart;
    kb->entry_skip = e->entry_skip;
    const char *er;
    if (pread(b->fd, raw, e->clen, (off_t)e->file_off) != (ssize_t)e->clen) er = "BAM: read error";
    else if (!e->full) er = bgzf_inflate_to(raw, e->clen, s->bytes + e->boff, e->isize, e->crc);
    else { /* the end of a selected stretch: the whole block is inflated (and checked), its first isize bytes are the stream's */
      uint8_t *tmp = malloc(65536);
      er = tmp ? bgzf_inflate_to(raw, e->clen, tmp, e->full, e->crc) : "BAM: out of memory";
      if (!er) memcpy(s->bytes + e->boff, tmp, e->isize);
      free(tmp);
    }
    if (er) {
      pthread_mutex_lock(&b->mu);
      set_err(b, er);
      pthread_mutex_unlock(&b->mu);
      break;
    }
    if (!b->dbg_nowalk) walk_speculative(b, kb, s->sparse, s->bytes + e->boff);
    if (__atomic_add_fetch(&s->done, 1, __ATOMIC_ACQ_REL) == b->seq[e->seq].n_ent) { /* the load is complete */
      pthread_mutex_lock(&b->mu);
      s->ready_for = (uint64_t)e->seq + 1u;
      pthread_cond_broadcast(&b->cv_ready);
      pthread_mutex_unlock(&b->mu);
    }
  }
  free(raw);
  return NULL;
}

/* ---- the header, read on the caller's thread --------------------------------------------------------------------------- */
typedef struct {
  const uint8_t *map;
  size_t len, pos;
  uint8_t buf[65536];
  uint32_t n, o;
  uint64_t taken;
} hdr_in;

static int hdr_read(hdr_in *h, void *dst, size_t n) { /* 1, 0 = clean end, -1 = error */
  uint8_t *d = (uint8_t *)dst;
  size_t done = 0;
  while (done < n) {
    if (h->o == h->n) {
      const uint8_t *payload;
      uint32_t clen, isize, crc, bsize;
      const char *e = NULL;
      const int r = bgzf_parse(h->map, h->len, h->pos, &payload, &clen, &isize, &crc, &bsize, &e);
      if (r < 0) return bsc_set_error(BSC_ERR_ARG, "%s", e), -1;
      if (r == 0) return done ? (bsc_set_error(BSC_ERR_ARG, "BAM: input truncated"), -1) : 0;
      h->pos += bsize;
      if ((e = bgzf_inflate_to(payload, clen, h->buf, isize, crc))) return bsc_set_error(BSC_ERR_ARG, "%s", e), -1;
      h->n = isize;
      h->o = 0;
      continue;
    }
    const size_t take = n - done < (size_t)(h->n - h->o) ? n - done : (size_t)(h->n - h->o);
    memcpy(d + done, h->buf + h->o, take);
    h->o += (uint32_t)take;
    done += take;
    h->taken += take;
  }
  return 1;
}

void bsc_bamstream_close(bsc_bamstream *b) {
  if (!b) return;
  if (b->n_threads) {
    pthread_mutex_lock(&b->mu);
    __atomic_store_n(&b->closing, 1, __ATOMIC_RELEASE);
    pthread_cond_broadcast(&b->cv_ready);
    pthread_mutex_unlock(&b->mu);
    for (int i = 0; i < b->n_threads; i++) pthread_join(b->th[i], NULL);
  }
  if (b->alloc_running) pthread_join(b->alloc_th, NULL);
  if (b->sync_made) {
    pthread_mutex_destroy(&b->mu);
    pthread_cond_destroy(&b->cv_ready);
  }
  free(b->th);
  free(b->ent);
  free(b->seq);
  free(b->sel_tid);
  if (b->slab)
    for (int i = 0; i < b->n_slabs; i++) {
      free(b->slab[i].sparse);
      free(b->slab[i].blk);
      if (b->unpinned) {
        free(b->slab[i].bytes);
        free(b->slab[i].rec_off);
      } else {
        bsc_free_host(b->slab[i].bytes);
        bsc_free_host(b->slab[i].rec_off);
      }
    }
  free(b->slab);
  if (b->map && b->map != MAP_FAILED) munmap((void *)b->map, b->map_len ? b->map_len : 1);
  if (b->fd >= 0) close(b->fd);
  free(b->text);
  if (b->ref_name)
    for (int32_t i = 0; i < b->n_ref; i++) free(b->ref_name[i]);
  free(b->ref_name);
  free(b->ref_len);
  free(b);
}

/* the CPUs this process may really use: its affinity mask, cut down to the container's CFS quota (cgroup v2 cpu.max, v1 cpu.cfs_quota_us) —
 * a GPU box shows every core of the host and gives a job a share of them; helpers beyond the share only get the caller's thread throttled
 * (measured: 16 helpers 5.5 GB/s of inflated bytes, 32 .. 128 the same, the block call behind them 2.6 x slower) */
static int cpu_share(void) {
  cpu_set_t set;
  int n = 0;
  if (sched_getaffinity(0, sizeof set, &set) == 0) n = CPU_COUNT(&set);
  if (n <= 0) n = (int)sysconf(_SC_NPROCESSORS_ONLN);
  long long quota = -1, period = -1;
  FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f) {
    char q[64];
    if (fscanf(f, "%63s %lld", q, &period) == 2 && strcmp(q, "max")) quota = atoll(q);
    fclose(f);
  } else {
    FILE *fq = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r"), *fp = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r");
    if (fq && fp && fscanf(fq, "%lld", &quota) == 1 && fscanf(fp, "%lld", &period) == 1) {
    } else
      quota = -1;
    if (fq) fclose(fq);
    if (fp) fclose(fp);
  }
  if (quota > 0 && period > 0) {
    const int share = (int)((quota + period - 1) / period);
    if (share >= 1 && share < n) n = share;
  }
  return n < 1 ? 1 : n;
}

int bsc_bamstream_default_threads(void) {
  int n = cpu_share(); /* all of it: the caller's thread sleeps while the helpers inflate, and drives the device when they are done */
  if (n > 64) n = 64;
  return n < 1 ? 1 : n;
}

int bsc_bamstream_open(const char *path, int n_threads, uint64_t slab_bytes, int n_slabs, bsc_bamstream **out) {
  return bsc_bamstream_open_contigs(path, n_threads, slab_bytes, n_slabs, NULL, -1, out);
}

int bsc_bamstream_open_contigs(const char *path, int n_threads, uint64_t slab_bytes, int n_slabs, const int32_t *tids, int n_tids, bsc_bamstream **out) {
  if (!path || !out || (n_tids > 0 && !tids)) return bsc_set_error(BSC_ERR_ARG, "bsc_bamstream_open: NULL argument");
  *out = NULL;
  if (n_threads <= 0) n_threads = bsc_bamstream_default_threads();
  if (n_threads > 128) n_threads = 128;
  if (slab_bytes == 0) slab_bytes = 16u << 20; /* (page-locking costs ~1 ms per MB: the ring is 6 x 16 MiB, not more) */
  if (slab_bytes < 65536u) slab_bytes = 65536u;
  if (slab_bytes > (1ull << 31)) slab_bytes = 1ull << 31;
  if (n_slabs <= 0) n_slabs = 6;
  if (n_slabs < 2) n_slabs = 2;
  bsc_bamstream *b = calloc(1, sizeof *b);
  if (!b) return bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of memory");
  b->sel_n = n_tids < 0 ? -1 : n_tids;
  if (n_tids > 0) {
    b->sel_tid = malloc((size_t)n_tids * sizeof *b->sel_tid);
    if (!b->sel_tid) {
      free(b);
      return bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of memory");
    }
    memcpy(b->sel_tid, tids, (size_t)n_tids * sizeof *tids);
    for (int i = 1; i < n_tids; i++) /* ascending, the unplaced reads (-1) last: the order of the file */
      for (int j = i; j > 0; j--) {
        const int64_t x = b->sel_tid[j - 1] < 0 ? (int64_t)1 << 40 : b->sel_tid[j - 1], y = b->sel_tid[j] < 0 ? (int64_t)1 << 40 : b->sel_tid[j];
        if (x <= y) break;
        const int32_t t = b->sel_tid[j - 1];
        b->sel_tid[j - 1] = b->sel_tid[j];
        b->sel_tid[j] = t;
      }
  }
  b->fd = open(path, O_RDONLY);
  if (b->fd < 0) {
    const int e = errno;
    bsc_bamstream_close(b);
    return bsc_set_error(BSC_ERR_ARG, "bsc_bamstream_open: cannot open '%s': %s", path, strerror(e));
  }
  struct stat st;
  if (fstat(b->fd, &st) || !S_ISREG(st.st_mode)) {
    bsc_bamstream_close(b);
    return bsc_set_error(BSC_ERR_ARG, "bsc_bamstream_open: '%s' is not a regular file", path);
  }
  b->map_len = (size_t)st.st_size;
  b->map = b->map_len ? mmap(NULL, b->map_len, PROT_READ, MAP_PRIVATE, b->fd, 0) : NULL;
  if (b->map == MAP_FAILED) {
    b->map = NULL;
    bsc_bamstream_close(b);
    return bsc_set_error(BSC_ERR_ARG, "bsc_bamstream_open: cannot map '%s'", path);
  }
  if (b->map_len) (void)madvise((void *)b->map, b->map_len, MADV_SEQUENTIAL);
  /* the header: magic, text, reference list.  Counts from the file are not believed before the bytes behind them have arrived. */
  hdr_in *h = calloc(1, sizeof *h);
  if (!h) {
    bsc_bamstream_close(b);
    return bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of memory");
  }
  h->map = b->map;
  h->len = b->map_len;
  uint8_t w[8];
  int rc = hdr_read(h, w, 8);
  if (rc <= 0 || memcmp(w, "BAM\1", 4)) {
    free(h);
    bsc_bamstream_close(b);
    return rc < 0 ? BSC_ERR_ARG : bsc_set_error(BSC_ERR_ARG, "bsc_bamstream_open: '%s' is not a BAM file (SAM text goes through bsc_bam_open)", path);
  }
  {
    const uint32_t l_text = le32(w + 4);
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
      if (hdr_read(h, b->text + b->l_text, step) <= 0) goto bad;
      b->l_text += step;
    }
    if (!b->text && !(b->text = malloc(1))) goto bad;
    b->text[b->l_text] = 0;
  }
  if (hdr_read(h, w, 4) <= 0) goto bad;
  {
    const int32_t n_ref = (int32_t)le32(w);
    if (n_ref < 0) goto bad;
    size_t ref_cap = (size_t)(n_ref < 1024 ? n_ref : 1024) + 1;
    b->ref_name = calloc(ref_cap, sizeof *b->ref_name);
    b->ref_len = calloc(ref_cap, sizeof *b->ref_len);
    if (!b->ref_name || !b->ref_len) goto bad;
    for (int32_t i = 0; i < n_ref; i++) {
      if ((size_t)i + 1 >= ref_cap) {
        ref_cap *= 2;
        char **nn = realloc(b->ref_name, ref_cap * sizeof *b->ref_name);
        if (nn) b->ref_name = nn;
        uint32_t *nl = realloc(b->ref_len, ref_cap * sizeof *b->ref_len);
        if (nl) b->ref_len = nl;
        if (!nn || !nl) goto bad;
      }
      if (hdr_read(h, w, 4) <= 0) goto bad;
      const uint32_t ln = le32(w);
      if (ln == 0 || ln > 65536) goto bad;
      char *nm = malloc(ln);
      if (!nm) goto bad;
      b->ref_name[i] = nm;
      b->n_ref = i + 1;
      if (hdr_read(h, nm, ln) <= 0 || hdr_read(h, w, 4) <= 0) goto bad;
      nm[ln - 1] = 0;
      b->ref_len[i] = le32(w);
    }
  }
  b->first_rec_off = h->taken;
  free(h);
  h = NULL;
  /* slabs and helpers */
  b->slab_bytes = (size_t)slab_bytes;
  b->n_slabs = n_slabs;
  b->rec_cap = (uint32_t)(slab_bytes / 36u + 2u);
  b->blk_cap = (uint32_t)(slab_bytes / 4096u + 64u);
  b->sparse_cap = b->rec_cap + b->blk_cap;
  b->slab = calloc((size_t)n_slabs, sizeof *b->slab);
  if (!b->slab) goto nomem;
  if (getenv("BSC_BAMSTREAM_UNPINNED")) b->unpinned = 1; /* measurement: ordinary memory */
  pthread_mutex_init(&b->mu, NULL);
  pthread_cond_init(&b->cv_ready, NULL);
  b->sync_made = 1;
  if (slab_alloc(b, 0)) goto nomem;
  b->alloc_dev = -1;
  (void)hipGetDevice(&b->alloc_dev);
  (void)hipGetLastError();
  b->alloc_running = pthread_create(&b->alloc_th, NULL, alloc_main, b) == 0; /* the others: while the index is built and the helpers start */
  if (!b->alloc_running)
    for (int i = 1; i < n_slabs; i++)
      if (slab_alloc(b, i)) goto nomem;
  b->w_skip = b->first_rec_off;
  b->dbg_nowalk = getenv("BSC_BAMSTREAM_NOWALK") != NULL;
  {
    const int rc = build_index(b, n_threads);
    if (rc) {
      bsc_bamstream_close(b);
      return rc;
    }
  }
  b->th = calloc((size_t)n_threads, sizeof *b->th);
  if (!b->th) goto nomem;
  for (int i = 0; i < n_threads; i++) {
    if (pthread_create(&b->th[b->n_threads], NULL, helper, b)) break;
    b->n_threads++;
  }
  if (b->n_threads == 0) {
    bsc_bamstream_close(b);
    return bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: no helper thread could be started");
  }
  *out = b;
  return BSC_OK;
nomem:
  bsc_bamstream_close(b);
  return bsc_set_error(BSC_ERR_NOMEM, "bsc_bamstream_open: out of (page-locked) memory");
bad:
  free(h);
  bsc_bamstream_close(b);
  return bsc_set_error(BSC_ERR_ARG, "bsc_bamstream_open: '%s': malformed or truncated BAM header", path);
}

int bsc_bamstream_n_refs(const bsc_bamstream *b) { return b ? b->n_ref : 0; }
const char *bsc_bamstream_ref_name(const bsc_bamstream *b, int i) { return (b && i >= 0 && i < b->n_ref) ? b->ref_name[i] : NULL; }
uint32_t bsc_bamstream_ref_len(const bsc_bamstream *b, int i) { return (b && i >= 0 && i < b->n_ref) ? b->ref_len[i] : 0; }
const char *bsc_bamstream_header_text(const bsc_bamstream *b) { return b ? b->text : NULL; }
int bsc_bamstream_threads(const bsc_bamstream *b) { return b ? b->n_threads : 0; }

/* 1 = *out filled (valid until bsc_bamstream_release of it), 0 = the stream has ended, < 0 = error */
int bsc_bamstream_next(bsc_bamstream *b, bsc_bam_slab *out) {
  if (!b || !out) return bsc_set_error(BSC_ERR_ARG, "bsc_bamstream_next: NULL argument");
  memset(out, 0, sizeof *out);
  if (b->cons_seq >= b->n_seq) {
    if (b->cons_seq == b->n_seq && !b->dbg_nowalk && (b->w_skip || b->w_hdr_n))
      return bsc_set_error(BSC_ERR_ARG, "BAM: input truncated (the last record is incomplete)");
    return 0;
  }
  const uint64_t q = b->cons_seq;
  bs_slab *s = &b->slab[q % (uint64_t)b->n_slabs];
  pthread_mutex_lock(&b->mu);
  while (s->ready_for != q + 1u && !b->err) pthread_cond_wait(&b->cv_ready, &b->mu);
  if (s->ready_for != q + 1u) {
    const char *e = b->err;
    pthread_mutex_unlock(&b->mu);
    return bsc_set_error(BSC_ERR_ARG, "%s", e);
  }
  s->out = 1;
  pthread_mutex_unlock(&b->mu);
  /* the chain, block by block: a block whose helper assumed the right entry state (nothing hanging over from its predecessor) keeps its
   * finds; any other is walked again from the true state */
  const struct bs_seq *sq = &b->seq[q];
  s->n_recs = 0;
  for (uint32_t i = 0; i < sq->n_ent && !b->dbg_nowalk; i++) {
    const struct bs_blk *k = &s->blk[i];
    if (k->restart) { /* a new stretch of the selection: its first block starts at a record (checked when the stretch was chosen) */
      if (b->w_skip || b->w_hdr_n) return bsc_set_error(BSC_ERR_ARG, "BAM: a record is cut where a selected stretch of the file ends");
      b->w_skip = k->entry_skip;
    }
    if (b->w_hdr_n == 0 && k->valid && b->w_skip == k->lead) { /* the predecessor's record ends where the helper's walk began */
      memcpy(s->rec_off + s->n_recs, s->sparse + k->sp_base, (size_t)k->sp_n * 4u);
      s->n_recs += k->sp_n;
      b->w_skip = k->exit_skip;
      b->w_hdr_n = k->exit_hdr_n;
      memcpy(b->w_hdr, k->exit_hdr, 4);
    } else {
      const char *e = walk_block(b, s, k->boff, s->bytes + k->boff, k->isize);
      if (e) return bsc_set_error(BSC_ERR_ARG, "%s", e);
      b->n_rewalked++;
    }
  }
  if (sq->last && (b->w_skip || b->w_hdr_n) && !b->dbg_nowalk) return bsc_set_error(BSC_ERR_ARG, "BAM: input truncated (the last record is incomplete)");
  out->bytes = s->bytes;
  out->n_bytes = sq->n_bytes;
  out->stream_off = sq->stream_off;
  out->rec_off = s->rec_off;
  out->n_recs = s->n_recs;
  out->last = (int32_t)sq->last;
  out->seq = q;
  b->cons_seq++;
  b->total_recs += s->n_recs;
  b->total_bytes += sq->n_bytes;
  return 1;
}

/* slabs are handed back in the order they were handed out */
int bsc_bamstream_release(bsc_bamstream *b, const bsc_bam_slab *sl) {
  if (!b || !sl) return bsc_set_error(BSC_ERR_ARG, "bsc_bamstream_release: NULL argument");
  bs_slab *s = &b->slab[sl->seq % (uint64_t)b->n_slabs];
  if (!s->out || s->bytes != sl->bytes || sl->seq != __atomic_load_n(&b->n_released, __ATOMIC_RELAXED))
    return bsc_set_error(BSC_ERR_ARG, "bsc_bamstream_release: not the oldest slab that is out");
  s->out = 0;
  __atomic_store_n(&s->done, 0, __ATOMIC_RELAXED);
  __atomic_store_n(&b->n_released, sl->seq + 1u, __ATOMIC_RELEASE);
  return BSC_OK;
}

uint64_t bsc_bamstream_first_record(const bsc_bamstream *b) { return b ? b->first_rec_off : 0; }
