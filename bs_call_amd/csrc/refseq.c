/*
 * refseq.c — the reference sequence of the process thread (host C + zlib):
 *   bsc_fasta_contig      one contig of a FASTA file as reference codes 0 = N, 1..4 = ACGT (either case) — what load_sequence
 *                         keeps, 3 bits per base, from the faidx'ed file (src/read_reference.c:44-131).  The file is read
 *                         sequentially through zlib (plain, gzip or bgzip FASTA); no .fai index is needed or used.
 *   bsc_block_reference   get_sequence_string (src/get_sequence.c:20-54): the codes of positions x .. x + sz - 1 of a contig
 *                         for one block (work->ref1: sz = y - x + 3).  Faithful to the reference's bounds: a position left
 *                         of the contig's first A/C/G/T base or AT OR BEYOND its last position reads 0 — the walk stops at
 *                         `x1 < contig->end_pos`, so the contig's very last base is an N to the caller (:40).
 */
#include <ctype.h>
#include <errno.h>
#include <stdio.h>
#include <string.h>
#include <zlib.h>

#include "../../include/bscall_amd.h"

int bsc_set_error(int code, const char *fmt, ...);

int bsc_fasta_contig(const char *path, const char *name, uint8_t *codes, uint64_t cap, uint64_t *len) {
  if (!path || !name || !len || (cap && !codes)) return bsc_set_error(BSC_ERR_ARG, "bsc_fasta_contig: NULL argument");
  *len = 0;
  gzFile f = gzopen(path, "rb");
  if (!f) return bsc_set_error(BSC_ERR_ARG, "bsc_fasta_contig: cannot open '%s': %s", path, strerror(errno));
  gzbuffer(f, 1 << 20);
  static const uint8_t code_of[256] = {['A'] = 1, ['C'] = 2, ['G'] = 3, ['T'] = 4, ['a'] = 1, ['c'] = 2, ['g'] = 3, ['t'] = 4};
  const size_t nl = strlen(name);
  char line[1 << 16];
  int in_contig = 0, found = 0, rc = BSC_OK;
  uint64_t n = 0;
  while (gzgets(f, line, sizeof line)) {
    if (line[0] == '>') {
      if (in_contig) break; /* the next record ends ours */
      const char *p = line + 1;
      /* the record's name is the header up to the first white space (faidx) */
      in_contig = !strncmp(p, name, nl) && (p[nl] == 0 || isspace((unsigned char)p[nl]));
      found |= in_contig;
      /* a header longer than the buffer: skip its tail */
      while (!strchr(line, '\n') && gzgets(f, line, sizeof line)) {}
      continue;
    }
    if (!in_contig) continue;
    for (const unsigned char *p = (const unsigned char *)line; *p; p++) {
      if (!isgraph(*p)) continue;
      if (n < cap) codes[n] = code_of[*p];
      n++;
    }
  }
  gzclose(f);
  if (!found) return bsc_set_error(BSC_ERR_ARG, "bsc_fasta_contig: no sequence '%s' in '%s'", name, path);
  *len = n;
  if (n > cap) rc = bsc_set_error(BSC_ERR_ARG, "bsc_fasta_contig: '%s' has %llu bases, the buffer holds %llu", name, (unsigned long long)n,
                                  (unsigned long long)cap);
  return rc;
}

int bsc_block_reference(const uint8_t *codes, uint64_t contig_len, uint32_t x, uint32_t sz, uint8_t *out) {
  if ((contig_len && !codes) || (sz && !out)) return bsc_set_error(BSC_ERR_ARG, "bsc_block_reference: NULL argument");
  if (x == 0) return bsc_set_error(BSC_ERR_ARG, "bsc_block_reference: positions are 1-based");
  for (uint32_t i = 0; i < sz; i++) {
    const uint64_t pos = (uint64_t)x + i; /* 1-based */
    out[i] = pos < contig_len ? codes[pos - 1] : 0; /* end_pos = the contig's length: that position itself is not read */
  }
  return BSC_OK;
}
