/*
 * accumulate.hip — gfx950 kernels of the pile-up accumulate stage: reads -> pileup[] for one block of
 * positions [x, y].  Replaces HOT LOOP A of call_genotypes_ML (reference src/call_genotypes.c:178-226),
 * which the reference runs serially on its process thread.
 *
 *   bsc_prep_reads_kernel   one thread per template: the leading/trailing scan that finds each read's first and
 *                           last countable base (:198-211), the orientation each read is counted with (:187,224,
 *                           including the reference's quirk that a skipped read 0 does not flip it), and a
 *                           compact per-read descriptor; also the template's leftmost position and the largest
 *                           template extent of the block.
 *   bsc_tile_lo_kernel      one thread per 64-position wave-tile: binary search for the first template that can
 *                           reach the tile (templates arrive sorted by leftmost position).
 *   bsc_accumulate_kernel   one wave per wave-tile, lane i OWNS position i of the tile: the wave walks the
 *                           candidate templates (64 descriptors per vector load, ballot-filtered to the reads
 *                           that overlap the tile, broadcast with v_readlane), every lane fetches "its" base of
 *                           the read (consecutive lanes = consecutive bytes: coalesced) and bumps its own
 *                           pile-up row in the wave's LDS slot.  A row has one writer, so there is no atomic
 *                           contention and the result does not depend on any ordering; the slot is the
 *                           reference's pileup[] layout and leaves with 16-byte-per-lane stores.
 *
 * Exactness: the reference sums base qualities and MAPQ^2 in float.  Sums of integers are exact in float
 * below 2^24, and then order-independent, so integer accumulation + one conversion gives the same bits.
 * quality[c]: q <= 43, exact up to 390 000 bases per class; mapq2: exact up to 258 bases at MAPQ 255, 4 660 at
 * MAPQ 60.  Beyond that the reference's own result depends on its summation order; positions that exceed the
 * bound are counted in counters[BSC_CNT_INEXACT] and reported by bsc_accumulate().
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "devtables.h"

#define IN_DW 26
#define SLOT_DW (64 * IN_DW)
#define ACC_WAVES 4 /* waves per workgroup */

/* compact read descriptor, 24 bytes; a > b: the read contributes nothing */
struct __attribute__((aligned(8))) bsc_read_desc {
  uint32_t a;    /* first countable position (absolute) */
  uint32_t b;    /* last countable position (absolute, already clipped to y) */
  int64_t base;  /* seq offset of position 0: byte of position p is seq[base + p] */
  uint32_t meta; /* bit0 orientation the read is counted with, bits 1-2 bs_strand, bits 8-23 mapq^2 */
  uint32_t _pad;
};

/* bsc_template (include/bscall_amd.h) as the kernels read it */
struct bsc_template_dev {
  uint32_t pos[2];
  uint32_t len[2];
  uint64_t off[2];
  uint8_t mapq[2];
  uint8_t orientation;
  uint8_t bs_strand;
  uint32_t _pad;
};

extern "C" __global__ __launch_bounds__(256) void bsc_prep_reads_kernel(const bsc_template_dev *__restrict__ tpl,
                                                                        uint32_t nr, const uint8_t *__restrict__ seq,
                                                                        uint32_t y, bsc_read_desc *__restrict__ rd,
                                                                        uint32_t *__restrict__ x1,
                                                                        unsigned long long *__restrict__ counters) {
  uint32_t span_max = 0;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nr; t += gridDim.x * blockDim.x) {
    const bsc_template_dev tp = tpl[t];
    uint32_t left = tp.pos[0]; /* src/call_genotypes.c:183-185 */
    if (left == 0) left = tp.pos[1];
    else if (tp.pos[1] > 0 && tp.pos[1] < left) left = tp.pos[1];
    x1[t] = left;
    uint32_t ori = tp.orientation & 1u;
    uint32_t reach = left;
    /* the end bytes of both reads, fetched together: almost every read starts and ends on a countable base, so
     * the scans below rarely need another load and the thread waits for memory once, not four times */
    uint32_t e_first[2], e_last[2];
#pragma unroll
    for (int k = 0; k < 2; k++) {
      const uint32_t rl = tp.len[k];
      e_first[k] = rl ? seq[tp.off[k]] : 0u;
      e_last[k] = rl ? seq[tp.off[k] + rl - 1] : 0u;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      bsc_read_desc d;
      d.a = 1;
      d.b = 0;
      d.base = 0;
      d.meta = 0;
      d._pad = 0;
      const uint32_t rl = tp.len[k];
      if (rl != 0) {
        const uint8_t *sp = seq + tp.off[k];
        uint32_t j = 0;
        for (; j < rl; j++) { /* :198-201 */
          const uint32_t q = (j == 0 ? e_first[k] : (uint32_t)sp[j]) >> 2;
          if (q > 0 && q != 63u) break;
        }
        if (j < rl) {
          const uint32_t first = j;
          for (j = rl; j > 0; j--) { /* :205-208 */
            const uint32_t q = (j == rl ? e_last[k] : (uint32_t)sp[j - 1]) >> 2;
            if (q > 0 && q != 63u) break;
          }
          const uint32_t last = j - 1;
          const uint64_t pa = (uint64_t)tp.pos[k] + first, pb = (uint64_t)tp.pos[k] + last;
          d.a = pa > 0xffffffffull ? 0xffffffffu : (uint32_t)pa;
          d.b = pb > y ? y : (uint32_t)pb; /* pos <= y, :214 */
          if (pa > y) { d.a = 1; d.b = 0; }
          d.base = (int64_t)tp.off[k] - (int64_t)tp.pos[k];
          d.meta = ori | ((uint32_t)tp.bs_strand << 1) | (((uint32_t)tp.mapq[k] * tp.mapq[k]) << 8);
          if (d.b >= d.a && d.b > reach) reach = d.b;
          ori ^= 1u; /* :224 — only a read that was walked flips the orientation */
        }
      }
      rd[2 * (uint64_t)t + k] = d;
    }
    const uint32_t span = reach - left;
    span_max = span > span_max ? span : span_max;
  }
  /* largest extent: wave max, one atomic per wave */
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t v = __shfl_xor(span_max, o);
    span_max = v > span_max ? v : span_max;
  }
  /* Same-address device-scope atomics serialise at the memory side (measured: ~6 ns each, 19 k waves), and most
   * waves see the same largest extent: look first (the counter only grows, so a stale smaller value costs an
   * atomic, never a wrong skip). */
  if ((threadIdx.x & 63u) == 0 && span_max &&
      __hip_atomic_load(&counters[BSC_CNT_SPAN], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < span_max)
    atomicMax(&counters[BSC_CNT_SPAN], (unsigned long long)span_max);
}

extern "C" __global__ __launch_bounds__(256) void bsc_tile_lo_kernel(const uint32_t *__restrict__ x1, uint32_t nr,
                                                                     uint32_t x, uint32_t n_wt,
                                                                     const unsigned long long *__restrict__ counters,
                                                                     uint32_t *__restrict__ tile_lo) {
  const uint32_t span = (uint32_t)counters[BSC_CNT_SPAN];
  for (uint32_t wt = blockIdx.x * blockDim.x + threadIdx.x; wt < n_wt; wt += gridDim.x * blockDim.x) {
    const uint64_t p0 = (uint64_t)x + (uint64_t)wt * 64u;
    const uint32_t key = p0 > span ? (uint32_t)(p0 - span) : 0u; /* first template with x1 >= key */
    uint32_t lo = 0, hi = nr;
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (x1[mid] < key) lo = mid + 1;
      else hi = mid;
    }
    tile_lo[wt] = lo;
  }
}

/* strand -> 4 * class, one byte per base code (reference base_tab_st, src/call_genotypes.c:17-19):
 * NON_CONVERTED 0 1 2 3 ; C2T 0 5 2 7 ; G2A 4 1 6 3.  Byte-indexed so that one v_perm_b32 turns a base code into
 * the byte offset of its class inside a pile-up row. */
#define LUT4(c0, c1, c2, c3) ((uint32_t)(4 * (c0)) | ((uint32_t)(4 * (c1)) << 8) | ((uint32_t)(4 * (c2)) << 16) | ((uint32_t)(4 * (c3)) << 24))

#ifndef ACC_DEPTH
#define ACC_DEPTH 4 /* reads whose byte loads are in flight together */
#endif

/*
 * The kernel is VALU-issue bound (a wave64 VALU instruction occupies its SIMD for 4 cycles; without the byte loads
 * or without the LDS updates it runs exactly as long), so everything that is the same for all lanes — the read's
 * extent, its sequence base, orientation, strand, MAPQ^2 — is kept in SGPRs (v_readlane results, SALU arithmetic)
 * and the per-lane work per read is: clamp, load, range test, quality test, class lookup, two LDS adds, one add.
 */
extern "C" __global__ __launch_bounds__(64 * ACC_WAVES) void bsc_accumulate_kernel(
    const bsc_read_desc *__restrict__ rd, const uint32_t *__restrict__ x1, uint32_t nr,
    const uint8_t *__restrict__ seq, uint32_t x, uint32_t y, uint32_t min_qual, const uint32_t *__restrict__ tile_lo,
    uint32_t *__restrict__ cts, unsigned long long *__restrict__ counters) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_slot[ACC_WAVES][SLOT_DW];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t *slot = lds_slot[wid];
  uint32_t *row = slot + lane * IN_DW;
  const uint32_t n_sites = y - x + 1;
  const uint32_t n_wt = (n_sites + 63u) / 64u;
  /* q counts iff min_qual <= q < 63 (:217)  <=>  (q - min_qual) <u q_span */
  const uint32_t q_span = min_qual < 63u ? 63u - min_qual : 0u;
  unsigned inexact = 0;
  for (uint32_t wt = blockIdx.x * ACC_WAVES + wid; wt < n_wt; wt += gridDim.x * ACC_WAVES) {
    /* x + 64 wt <= y: the tile's first position fits 32 bits; its last one is clipped to y */
    const uint32_t p0 = x + wt * 64u;
    const uint32_t p_last = y - p0 < 63u ? y : p0 + 63u;
    const bool valid = lane <= p_last - p0;
#pragma unroll
    for (int i = 0; i < IN_DW / 2; i++) reinterpret_cast<uint2 *>(row)[i] = make_uint2(0u, 0u);
    uint32_t m2sum = 0; /* mapq2 of this lane's position */

    uint32_t t0 = tile_lo[wt];
    bool more = true;
    while (more) {
      /* 64 candidate templates per pass: lane i holds the two read descriptors of template t0 + i */
      const uint32_t t = t0 + lane;
      const bool cand = t < nr && x1[t < nr ? t : 0] <= p_last;
      bsc_read_desc d0, d1;
      d0.a = d1.a = 1;
      d0.b = d1.b = 0;
      d0.base = d1.base = 0;
      d0.meta = d1.meta = 0;
      if (cand) {
        d0 = rd[2 * (uint64_t)t];
        d1 = rd[2 * (uint64_t)t + 1];
      }
      more = __all(cand); /* templates are sorted by x1: the candidates are a prefix of the batch */
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const bsc_read_desc &d = k ? d1 : d0;
        /* reads that overlap the tile at all */
        unsigned long long m = __ballot(d.b >= d.a && d.b >= p0 && d.a <= p_last);
        /* ACC_DEPTH reads per group: their byte loads are issued back to back and consumed afterwards, so the
         * wave waits for memory once per group; slots past the end of the list are skipped by scalar branches */
        while (m) {
          const int cnt = __popcll(m);
          uint32_t g_byte[ACC_DEPTH], g_lo[ACC_DEPTH], g_len[ACC_DEPTH], g_meta[ACC_DEPTH];
#pragma unroll
          for (int u = 0; u < ACC_DEPTH; u++) {
            if (u < cnt) {
              const int src = __builtin_ctzll(m);
              m &= m - 1;
              const uint32_t a = __builtin_amdgcn_readlane(d.a, src);
              const uint32_t b = __builtin_amdgcn_readlane(d.b, src);
              const uint32_t blo = __builtin_amdgcn_readlane((uint32_t)(uint64_t)d.base, src);
              const uint32_t bhi = __builtin_amdgcn_readlane((uint32_t)((uint64_t)d.base >> 32), src);
              g_meta[u] = __builtin_amdgcn_readlane(d.meta, src);
              /* the read's part of the tile, as lane numbers lo..hi (0 <= lo <= hi <= 63) */
              const uint32_t lo = (a > p0 ? a : p0) - p0;
              const uint32_t hi = (b < p_last ? b : p_last) - p0;
              g_lo[u] = lo;
              g_len[u] = hi - lo;
              /* byte of position p0 + i is sp[i]; lanes outside the read fetch its nearest byte (always a valid
               * address, same cache lines) and drop it */
              const uint8_t *sp = seq + ((int64_t)(((uint64_t)bhi << 32) | blo) + (int64_t)p0);
              const uint32_t pc = lane < lo ? lo : (lane > hi ? hi : lane);
              g_byte[u] = sp[pc];
            }
          }
#pragma unroll
          for (int u = 0; u < ACC_DEPTH; u++) {
            if (u < cnt) {
              const uint32_t byte = g_byte[u], meta = g_meta[u];
              const uint32_t q = byte >> 2;
              if (lane - g_lo[u] <= g_len[u] && q - min_qual < q_span) {
                const uint32_t strand = (meta >> 1) & 3u; /* uniform */
                const uint32_t lut = strand == 0 ? LUT4(0, 1, 2, 3) : (strand == 1 ? LUT4(0, 5, 2, 7) : LUT4(4, 1, 6, 3));
                /* selector bytes 1..3 = 0x0c: constant 0 */
                const uint32_t c4 = __builtin_amdgcn_perm(0u, lut, (byte & 3u) | 0x0c0c0c00u);
                uint32_t *rc = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(row) + c4);
                /* fire-and-forget ds_add_u32: the row has a single writer (this lane), no contention */
                if (meta & 1u) atomicAdd(rc + 8, 1u); /* counts[ori][c]++ */
                else atomicAdd(rc, 1u);
                atomicAdd(rc + 17, q); /* quality[c] += q (integer; converted below) */
                m2sum += meta >> 8;    /* mapq2 += mapq^2 */
              }
            }
          }
        }
      }
      t0 += 64u;
    }
    row[25] = m2sum;

    /* n = sum of counts; integer sums -> float */
    {
      uint32_t n = 0;
#pragma unroll
      for (int j = 0; j < 16; j++) n += row[j];
      row[16] = n;
#pragma unroll
      for (int j = 0; j < 8; j++) {
        const uint32_t qs = row[17 + j];
        inexact |= qs >= (1u << 24);
        row[17 + j] = __float_as_uint((float)qs);
      }
      const uint32_t m2 = row[25];
      inexact |= m2 >= (1u << 24);
      row[25] = __float_as_uint((float)m2);
    }
    /* the slot is the tile's pileup[] image: copy it out */
    const uint32_t nvalid = p_last - p0 + 1u;
    uint32_t *dst = cts + (uint64_t)wt * SLOT_DW;
    if (nvalid == 64u) {
      const uint4 *s4 = reinterpret_cast<const uint4 *>(slot);
      uint4 *d4 = reinterpret_cast<uint4 *>(dst);
#pragma unroll
      for (int v = 0; v < 6; v++) d4[v * 64 + lane] = s4[v * 64 + lane];
      if (lane < 32) d4[6 * 64 + lane] = s4[6 * 64 + lane];
    } else if (valid) {
#pragma unroll
      for (int i = 0; i < IN_DW; i++) dst[lane * IN_DW + i] = row[i];
    }
  }
  if (__any(inexact)) {
    const unsigned long long m = __ballot(inexact);
    if (lane == 0) atomicAdd(&counters[BSC_CNT_INEXACT], (unsigned long long)__popcll(m));
  }
}

/* ---- launcher ------------------------------------------------------------------------------------------- */
extern "C" int bsc_dev_launch_accumulate(const void *tpl, uint32_t nr, const void *seq, uint32_t x, uint32_t y,
                                         uint32_t min_qual, void *rd, void *x1, void *tile_lo, void *cts,
                                         void *counters, int num_cus, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n_sites = y - x + 1;
  const uint32_t n_wt = (n_sites + 63u) / 64u;
  if (nr) {
    unsigned g = (nr + 255u) / 256u;
    if (g > (unsigned)num_cus * 16u) g = (unsigned)num_cus * 16u;
    hipLaunchKernelGGL(bsc_prep_reads_kernel, dim3(g), dim3(256), 0, s, (const bsc_template_dev *)tpl, nr,
                       (const uint8_t *)seq, y, (bsc_read_desc *)rd, (uint32_t *)x1, (unsigned long long *)counters);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  {
    unsigned g = (n_wt + 255u) / 256u;
    if (g > (unsigned)num_cus * 16u) g = (unsigned)num_cus * 16u;
    hipLaunchKernelGGL(bsc_tile_lo_kernel, dim3(g), dim3(256), 0, s, (const uint32_t *)x1, nr, x, n_wt,
                       (const unsigned long long *)counters, (uint32_t *)tile_lo);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  {
    unsigned g = (n_wt + ACC_WAVES - 1) / ACC_WAVES;
    const unsigned cap = (unsigned)num_cus * 6u * 8u; /* 6 workgroups of 26 KB LDS fit a CU */
    if (g > cap) g = cap;
    hipLaunchKernelGGL(bsc_accumulate_kernel, dim3(g), dim3(64 * ACC_WAVES), 0, s, (const bsc_read_desc *)rd,
                       (const uint32_t *)x1, nr, (const uint8_t *)seq, x, y, min_qual, (const uint32_t *)tile_lo,
                       (uint32_t *)cts, (unsigned long long *)counters);
  }
  return (int)hipGetLastError();
}
