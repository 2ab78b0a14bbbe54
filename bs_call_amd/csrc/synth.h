/*
 * synth.h — deterministic synthetic 'L-pileup' generator, shared by the host
 * twin (bsc_synth_pileup_host) and the device kernel so both produce the same bits.
 *
 * Bench/test support, not part of the calling path.  Site i of the synthetic contig depends only on
 * (seed, i): PRNG = xorshift64 (x^=x<<13; x^=x>>7; x^=x<<17) seeded per site through a splitmix64
 * finaliser.  Model of a 30x-style WGBS pile-up:
 *   - reference base uniform over A,C,G,T (codes 1..4); with BSC_SYNTH_NRUNS, 1 % of 10-kb runs are N (code 0)
 *     and carry no reads
 *   - depth = coverage +- 25 % (uniform), 1/1024 of sites uncovered
 *   - every 1000th site is a het SNP (alt allele uniform over the other three bases)
 *   - each read: bisulfite strand C2T/G2A and orientation Bernoulli(1/2); 0.5 % uniform base error;
 *     C->T on C2T reads / G->A on G2A reads with p = 120/128 outside CpG, 20 % at CpG (about 80 % methylated);
 *     base quality uniform 20..43; MAPQ 60
 *   - classes from base_tab_st (reference src/call_genotypes.c:17-19)
 */
#ifndef BSCALL_AMD_SYNTH_H
#define BSCALL_AMD_SYNTH_H

#include <stdint.h>

#if defined(__HIPCC__)
#define SYN_FN __host__ __device__ static __forceinline__
#else
#define SYN_FN static inline __attribute__((always_inline))
#endif

#define BSC_SYNTH_NRUNS 1u

SYN_FN uint64_t syn_mix(uint64_t z) {
  z ^= z >> 30;
  z *= 0xbf58476d1ce4e5b9ull;
  z ^= z >> 27;
  z *= 0x94d049bb133111ebull;
  z ^= z >> 31;
  return z;
}

SYN_FN uint64_t syn_next(uint64_t *s) {
  uint64_t x = *s;
  x ^= x << 13;
  x ^= x >> 7;
  x ^= x << 17;
  *s = x;
  return x;
}

/* reference code 0..4 (N,A,C,G,T) of a site */
SYN_FN uint32_t syn_ref(uint64_t seed, uint64_t site, uint32_t flags) {
  if (flags & BSC_SYNTH_NRUNS) {
    uint64_t run = site / 10000u;
    if (syn_mix(seed ^ (run * 0xd1342543de82ef95ull + 0x632be59bd9b4e019ull)) % 100u == 0) return 0;
  }
  return 1u + (uint32_t)(syn_mix(seed + 0x9e3779b97f4a7c15ull * (site + 1)) >> 62);
}

/* Fills the 26 dwords of one `pileup` (counts[2][8], n, quality[8] as f32 bits, mapq2 as f32 bits). */
SYN_FN void syn_site(uint64_t seed, uint64_t site, uint32_t coverage, uint32_t flags, uint32_t *counts16,
                     uint32_t *n_out, float *quality8, float *mapq2_out, uint32_t *ref_out) {
  uint32_t ref = syn_ref(seed, site, flags);
  *ref_out = ref;
  uint32_t qs[8];
  for (int i = 0; i < 16; i++) counts16[i] = 0;
  for (int i = 0; i < 8; i++) qs[i] = 0;
  uint32_t n = 0;
  if (ref != 0) {
    uint64_t s = syn_mix(seed ^ (0xa0761d6478bd642full * (site + 0x1234567ull)));
    if (s == 0) s = 0x2545f4914f6cdd1dull;
    uint64_t u = syn_next(&s);
    uint32_t span = coverage / 2u + 1u; /* depth in [cov - cov/4, cov + cov/4] */
    uint32_t depth = coverage - coverage / 4u + (uint32_t)((u >> 8) % span);
    if ((u & 1023u) == 0) depth = 0;
    int het = (site % 1000u) == 0;
    uint32_t alt = (ref - 1u + 1u + (uint32_t)((u >> 40) % 3u)) & 3u; /* 0..3, != ref-1 */
    /* CpG context of this reference position */
    int cpg_c = (ref == 2u) && (syn_ref(seed, site + 1, flags) == 3u);
    int cpg_g = (ref == 3u) && (site > 0) && (syn_ref(seed, site - 1, flags) == 2u);
    for (uint32_t r = 0; r < depth; r++) {
      u = syn_next(&s);
      uint32_t strand = (u & 1u) ? 1u : 2u; /* 1 = C2T, 2 = G2A */
      uint32_t ori = (uint32_t)(u >> 1) & 1u;
      uint32_t b = (het && ((u >> 2) & 1u)) ? alt : (ref - 1u);
      if (((u >> 8) & 0xfffu) % 200u == 0) b = (b + 1u + (uint32_t)((u >> 20) % 3u)) & 3u;
      uint32_t conv = (uint32_t)(u >> 32) & 127u;
      if (strand == 1u && b == 1u) {
        int c = cpg_c ? (conv % 5u == 0) : (conv < 120u);
        if (c) b = 3u;
      } else if (strand == 2u && b == 2u) {
        int c = cpg_g ? (conv % 5u == 0) : (conv < 120u);
        if (c) b = 0u;
      }
      uint32_t q = 20u + (uint32_t)((u >> 40) % 24u);
      /* base_tab_st, 0-based: C2T -> A0 C5 G2 T7 ; G2A -> A4 C1 G6 T3 */
      uint32_t cls = (strand == 1u) ? ((b & 1u) ? b + 4u : b) : ((b & 1u) ? b : b + 4u);
      counts16[ori * 8u + cls]++;
      qs[cls] += q;
      n++;
    }
  }
  *n_out = n;
  for (int i = 0; i < 8; i++) quality8[i] = (float)qs[i]; /* exact: sums of small integers */
  *mapq2_out = (float)(3600u * n);                         /* MAPQ 60, exact below 2^24 */
}

#endif /* BSCALL_AMD_SYNTH_H */
