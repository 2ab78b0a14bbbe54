/*
 * The BAM reader's helper threads (csrc/bamio.c: n_threads workers inflate BGZF blocks ahead of the parser, a ring of slots
 * handed over under one mutex and two condition variables) under ThreadSanitizer, CPU only: this program is compiled TOGETHER
 * with csrc/bamio.c and csrc/prep.c (-fsanitize=thread), reads a BAM file to its end with 0, 1, 3 and 8 helpers, early-closes
 * a reader in the middle of the file (the helpers are joined while blocks are still in flight) and checks that every run
 * delivers the same blocks (count, templates, a hash of every byte handed out).  tests/test_bam_tsan.py builds and runs it.
 * usage: bam_tsan <file.bam>
 */
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "bscall_amd.h"

/* the library's error sink (csrc/bscall_api.c), per thread as there */
static __thread char errbuf[512];
int bsc_set_error(int code, const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(errbuf, sizeof errbuf, fmt, ap);
  va_end(ap);
  return code;
}

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
  const unsigned char *c = (const unsigned char *)p;
  for (size_t i = 0; i < n; i++) h = (h ^ c[i]) * 0x100000001b3ull;
  return h;
}

static int drain(const char *path, int threads, long stop_after, uint64_t *hash, long *blocks, long *templates) {
  bsc_bam *b = NULL;
  int rc = bsc_bam_open_threads(path, threads, &b);
  if (rc) {
    fprintf(stderr, "open (%d threads): %s\n", threads, errbuf);
    return 1;
  }
  bsc_reader_params par;
  memset(&par, 0, sizeof par);
  par.mapq_thresh = 20;
  par.max_template_len = 1000;
  bsc_read_block blk;
  uint64_t h = 0xcbf29ce484222325ull;
  long nb = 0, nt = 0;
  while ((rc = bsc_bam_next_block(b, &par, &blk)) == 1) {
    h = fnv(h, &blk.tid, sizeof blk.tid);
    h = fnv(h, &blk.y, sizeof blk.y);
    h = fnv(h, blk.tpl, (size_t)blk.nr * sizeof *blk.tpl);
    h = fnv(h, blk.seq, (size_t)blk.seq_bytes);
    h = fnv(h, blk.misms, (size_t)blk.n_misms * sizeof *blk.misms);
    nb++;
    nt += blk.nr;
    if (stop_after && nb == stop_after) break; /* close with helpers busy and inflated blocks unconsumed */
  }
  bsc_bam_close(b);
  if (rc < 0) {
    fprintf(stderr, "next_block (%d threads): %s\n", threads, errbuf);
    return 1;
  }
  *hash = h;
  *blocks = nb;
  *templates = nt;
  return 0;
}

int main(int argc, char **argv) {
  if (argc < 2) return 2;
  static const int threads[] = {0, 1, 3, 8};
  uint64_t h0 = 0;
  long b0 = 0, t0 = 0;
  for (int i = 0; i < 4; i++) {
    uint64_t h;
    long nb, nt;
    if (drain(argv[1], threads[i], 0, &h, &nb, &nt)) return 1;
    if (i == 0) {
      h0 = h;
      b0 = nb;
      t0 = nt;
    } else if (h != h0 || nb != b0 || nt != t0) {
      fprintf(stderr, "%d helpers: %ld blocks, %ld templates, hash %016llx; without: %ld, %ld, %016llx\n", threads[i], nb, nt,
              (unsigned long long)h, b0, t0, (unsigned long long)h0);
      return 1;
    }
    for (long stop = 1; stop <= 3 && stop < b0; stop++) { /* early close */
      uint64_t hh;
      long bb, tt;
      if (drain(argv[1], threads[i], stop, &hh, &bb, &tt) || bb != stop) return 1;
    }
  }
  printf("bam_tsan: %ld blocks, %ld templates, hash %016llx with 0 / 1 / 3 / 8 helper threads; early closes ok\n", b0, t0,
         (unsigned long long)h0);
  return 0;
}
