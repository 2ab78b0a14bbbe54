/*
 * synth_reads.c — host generator of synthetic read pairs ('L-reads') that exercise the
 * pile-up accumulate stage (reference HOT LOOP A, src/call_genotypes.c:180-226).  Bench/test support.
 *
 * Paired templates over positions [x, x + n_sites): read length 100, insert 300, forward starts evenly spaced
 * so that the mean depth is `coverage`; orientation and bisulfite strand Bernoulli(1/2); MAPQ 60 (or varied);
 * base = reference (synth.h syn_ref, with the every-1000th-site het SNP) with 0.5 % error; C->T on C2T reads /
 * G->A on G2A reads with p = 120/128 (20 % at CpG); base quality uniform 20..43.  To exercise the filters of
 * the loop: 3 % of bases get quality 1..19 (below min_qual), 1 % are N (byte 0), 12 % of reads carry a run of
 * trimmed bases (quality 63, src/read_utils.c:13-22) at one or both ends, 1 % of templates are single-end and
 * 0.5 % of reads are empty.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bscall_amd.h"
#include "synth.h"

/* Returns the number of templates written (<= max_templates) or -1 if a buffer is too small.
 * seq_used receives the number of bytes written to seq.  Templates come out in the order of the pairs' start positions; the 1 % that lost their forward read are
 * therefore out of leftmost-position order. */
int64_t bsc_synth_reads_host(uint64_t seed, uint32_t x, uint32_t n_sites, uint32_t coverage, uint32_t flags,
                             bsc_template *tpl, uint64_t max_templates, uint8_t *seq, uint64_t seq_cap,
                             uint64_t *seq_used) {
  const uint32_t RL = 100, INS = 300;
  if (n_sites == 0 || coverage == 0) {
    if (seq_used) *seq_used = 0;
    return 0;
  }
  const uint32_t y = x + n_sites - 1;
  /* templates start every 2*RL/coverage positions (fixed point, 16 fractional bits) */
  const uint64_t step_fp = ((uint64_t)2 * RL << 16) / coverage;
  uint64_t nt = 0, used = 0;
  uint64_t s = syn_mix(seed ^ 0x5bd1e9955bd1e995ull);
  if (s == 0) s = 1;
  for (uint64_t pfp = 0;; pfp += step_fp) {
    const uint32_t start = x + (uint32_t)(pfp >> 16);
    if (start > y) break;
    if (nt >= max_templates) return -1;
    uint64_t u = syn_next(&s);
    bsc_template t;
    memset(&t, 0, sizeof t);
    t.orientation = (uint8_t)(u & 1u);
    t.bs_strand = (uint8_t)(1u + ((u >> 1) & 1u)); /* 1 = C2T, 2 = G2A */
    t.mapq[0] = t.mapq[1] = 60;
    if (((u >> 2) & 15u) == 0) { /* some variety in MAPQ */
      t.mapq[0] = (uint8_t)(20u + ((u >> 8) % 41u));
      t.mapq[1] = (uint8_t)(20u + ((u >> 16) % 41u));
    }
    const int single = ((u >> 24) % 100u) == 0;
    for (int k = 0; k < 2; k++) {
      if (k == 1 && single) break;
      uint32_t rpos = start + (k ? INS - RL : 0);
      uint32_t rl = RL;
      uint64_t v = syn_next(&s);
      if ((v % 200u) == 0) rl = 0; /* empty read */
      if (rpos > y) rl = 0;
      if (rl == 0) continue;
      if (used + rl > seq_cap) return -1;
      uint8_t *sp = seq + used;
      /* trimmed runs */
      uint32_t ltrim = 0, rtrim = 0;
      if ((v >> 8) % 100u < 12u) {
        ltrim = (uint32_t)((v >> 16) % 12u);
        rtrim = (uint32_t)((v >> 24) % 12u);
        if (((v >> 32) & 63u) == 0) ltrim = rl; /* fully trimmed read */
      }
      for (uint32_t j = 0; j < rl; j++) {
        const uint64_t site = (uint64_t)rpos + j; /* genome position = synthetic site index */
        uint64_t w = syn_next(&s);
        uint32_t ref = syn_ref(seed, site, flags);
        uint32_t b;
        if (ref == 0) b = (uint32_t)(w & 3u);
        else {
          const int het = (site % 1000u) == 0;
          const uint32_t alt = (ref - 1u + 1u + (uint32_t)(syn_mix(seed + site) % 3u)) & 3u;
          b = (het && ((u >> (3 + k)) & 1u)) ? alt : ref - 1u;
        }
        if (((w >> 8) & 0xfffu) % 200u == 0) b = (b + 1u + (uint32_t)((w >> 20) % 3u)) & 3u;
        const uint32_t conv = (uint32_t)(w >> 32) & 127u;
        if (t.bs_strand == 1 && b == 1u) {
          const int cpg = syn_ref(seed, site + 1, flags) == 3u;
          if (cpg ? (conv % 5u == 0) : (conv < 120u)) b = 3u;
        } else if (t.bs_strand == 2 && b == 2u) {
          const int cpg = site > 0 && syn_ref(seed, site - 1, flags) == 2u;
          if (cpg ? (conv % 5u == 0) : (conv < 120u)) b = 0u;
        }
        uint32_t q = 20u + (uint32_t)((w >> 40) % 24u);
        const uint32_t r100 = (uint32_t)((w >> 48) % 100u);
        if (r100 < 3u) q = 1u + (uint32_t)((w >> 56) % 19u); /* below min_qual */
        uint8_t byte = (uint8_t)(b | (q << 2));
        if (r100 == 99u) byte = 0;                         /* N: base 0, quality 0 (src/input_sam.c:76-86) */
        if (j < ltrim || j + rtrim >= rl) byte = (uint8_t)(b | (63u << 2)); /* trimmed: q = FLT_QUAL */
        sp[j] = byte;
      }
      if (k == 0) t.flags = bsc_template_walk_flags(sp, rl); /* as a host that writes the bytes would (bsc_prepare_templates, the glue) */
      t.pos[k] = rpos;
      t.len[k] = rl;
      t.off[k] = used;
      used += rl;
    }
    if (t.len[0] == 0 && t.len[1] == 0) continue; /* nothing aligned */
    if (t.len[0] == 0) {
      t.pos[0] = 0;                                /* forward_position 0 = none (src/call_genotypes.c:183-185) */
      t.flags = BSC_TPL_WALK_KNOWN;                /* no read 0: nothing walked */
    }
    if (t.len[1] == 0) t.pos[1] = 0;
    tpl[nt++] = t;
  }
  if (seq_used) *seq_used = used;
  return (int64_t)nt;
}
