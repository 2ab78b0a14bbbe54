/*
 * accumulate.hip — gfx950 kernels of the pile-up accumulate stage: reads -> pileup[] for one block of
 * positions [x, y].  Replaces HOT LOOP A of call_genotypes_ML (reference src/call_genotypes.c:178-226),
 * which the reference runs serially on its process thread.
 *
 *   bsc_prep_reads_kernel   one thread per template: the reference's asserts on the template; per read the
 *                           orientation it is counted with (:187,224, including the reference's quirk that a read 0
 *                           without a countable base does not flip it), a compact descriptor and its sort key (the
 *                           position of its first base); also the longest read extent of the block.
 *   (sort.hip)              the block's reads ordered by that key on the device.
 *   bsc_tile_lo_kernel      one thread per 64-position wave-tile: binary search for the first read that can reach
 *                           the tile.
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

#include "accdev.h"
#include "devtables.h"

#define IN_DW 26
#define SLOT_DW (64 * IN_DW)
#define ACC_WAVES 4 /* waves per workgroup */

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

__device__ static __forceinline__ uint32_t leftmost(uint32_t p0, uint32_t p1) { /* src/call_genotypes.c:183-185 */
  return p0 == 0 ? p1 : (p1 > 0 && p1 < p0 ? p1 : p0);
}

/* The reference's asserts on a template (:186-188) and the bounds of the read buffer; 0 = fine, else BSC_TERR_*,
 * the first failing check in the reference's order. */
__device__ static __forceinline__ uint32_t template_error(const bsc_template_dev &tp, uint32_t left, uint32_t x,
                                                          uint64_t seq_bytes) {
  if (left < x) return BSC_TERR_LEFT;
  /* The reference checks only the leftmost non-zero position (:183-186) and then indexes counts + pos - x for BOTH reads
   * (:212): a read with bases whose own position is 0 or left of x is an out-of-bounds walk there.  Here it is the same
   * error as a template that starts left of the block (its sort key d.a - x would wrap and break the tiles' search). */
  if ((tp.len[0] && tp.pos[0] < x) || (tp.len[1] && tp.pos[1] < x)) return BSC_TERR_LEFT;
  if (tp.orientation > 1) return BSC_TERR_ORI;
  if (tp.bs_strand > 2) return BSC_TERR_STRAND;
  if (tp.len[0] && (tp.off[0] > seq_bytes || tp.len[0] > seq_bytes - tp.off[0])) return BSC_TERR_RANGE0;
  if (tp.len[1] && (tp.off[1] > seq_bytes || tp.len[1] > seq_bytes - tp.off[1])) return BSC_TERR_RANGE1;
  return 0;
}

/* One thread per template (in the caller's order): the reference's asserts (the lowest index of an invalid template with
 * its first failing check reaches the host through counters[BSC_CNT_ERR]; such a template contributes nothing), then per
 * read the leading/trailing scan, the orientation it is counted with and its descriptor rd[2t + k], and its sort key:
 * the first countable position relative to the block start, key_max for a read that contributes nothing.  READS, not
 * templates, are what the tiles search: a read's extent is bounded by its length, so mates that lie far apart (or a
 * pathological template) cannot widen every tile's candidate window. */
extern "C" __global__ __launch_bounds__(256) void bsc_prep_reads_kernel(const bsc_template_dev *__restrict__ tpl,
                                                                        uint32_t nr, const uint8_t *__restrict__ seq,
                                                                        uint64_t seq_bytes, uint32_t x, uint32_t y,
                                                                        uint32_t key_max, bsc_read_desc *__restrict__ rd,
                                                                        uint32_t *__restrict__ keys,
                                                                        unsigned long long *__restrict__ counters) {
  uint32_t span_max = 0;
  for (uint32_t t = blockIdx.x * blockDim.x + threadIdx.x; t < nr; t += gridDim.x * blockDim.x) {
    const bsc_template_dev tp = tpl[t];
    const uint32_t left = leftmost(tp.pos[0], tp.pos[1]);
    const uint32_t terr = template_error(tp, left, x, seq_bytes);
    if (terr) {
      atomicMin(&counters[BSC_CNT_ERR], ((unsigned long long)t << 8) | terr);
      bsc_read_desc d;
      acc_dead(d);
      rd[2 * (uint64_t)t] = d;
      rd[2 * (uint64_t)t + 1] = d;
      keys[2 * (uint64_t)t] = key_max;
      keys[2 * (uint64_t)t + 1] = key_max;
      continue;
    }
    uint32_t ori = tp.orientation & 1u;
    /*
     * The reference walks a read from its first to its last base with a quality other than 0 / 63 (:198-211) and tests every
     * base in between against min_qual (:217).  The bases it trims off the ends fail that test anyway (0 < min_qual, 63 is
     * excluded by name), so the device walks the whole read and this kernel does not look for the ends.  What the scan does
     * decide is whether a read counts as walked at all: only then is the orientation flipped for its mate (:224, and the
     * `continue`s of :203,:210 in front of it).  That needs the first countable base of read 0 — almost always its first
     * byte, so one byte per template is fetched here instead of four.
     */
    bool walked0 = false;
    {
      const uint32_t rl = tp.len[0];
      const uint8_t *sp = seq + tp.off[0];
      for (uint32_t j = 0; j < rl; j++) {
        const uint32_t q = (uint32_t)sp[j] >> 2;
        if (q > 0 && q != 63u) {
          walked0 = true;
          break;
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      bsc_read_desc d;
      acc_dead(d);
      const uint32_t rl = tp.len[k];
      if (rl != 0 && (k == 1 || walked0)) {
        const uint64_t pa = (uint64_t)tp.pos[k], pb = (uint64_t)tp.pos[k] + rl - 1u;
        if (pa <= y) {
          d.a = (uint32_t)pa;
          d.b = pb > y ? y : (uint32_t)pb; /* pos <= y, :214 */
          d.base = (int64_t)tp.off[k] - (int64_t)tp.pos[k];
          d.meta = ACC_META(ori, tp.mapq[k]);
          d.lut = tp.bs_strand == 0 ? LUT4(0, 1, 2, 3) : (tp.bs_strand == 1 ? LUT4(0, 5, 2, 7) : LUT4(4, 1, 6, 3));
        }
      }
      if (k == 0 && walked0) ori ^= 1u; /* :224 — only a read that was walked flips the orientation */
      rd[2 * (uint64_t)t + k] = d;
      const bool live = d.b >= d.a; /* then x <= a <= b <= y */
      keys[2 * (uint64_t)t + k] = live ? (d.a - x) >> ACC_BIN_SHIFT : key_max;
      if (live && d.b - d.a > span_max) span_max = d.b - d.a;
    }
  }
  /* longest read extent: wave max, one atomic per wave */
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

/* Tile wt starts at block-relative position base + wt * step (below 0: at the block start): step 64, base 0 for the
 * stand-alone accumulate kernel; step 60, base = window start - 2 for the reads-in chain (fused.hip). */
extern "C" __global__ __launch_bounds__(256) void bsc_tile_lo_kernel(const uint32_t *__restrict__ keys_sorted,
                                                                     uint32_t n_reads, uint32_t n_wt, int64_t base, uint32_t step,
                                                                     const unsigned long long *__restrict__ counters,
                                                                     uint32_t *__restrict__ tile_lo) {
  const int64_t span = (int64_t)(uint32_t)counters[BSC_CNT_SPAN]; /* longest read extent, b - a */
  for (uint32_t wt = blockIdx.x * blockDim.x + threadIdx.x; wt < n_wt; wt += gridDim.x * blockDim.x) {
    const int64_t r0 = base + (int64_t)wt * step;       /* the tile's first position, relative to the block start */
    const uint32_t key = r0 > span ? (uint32_t)(r0 - span) >> ACC_BIN_SHIFT : 0u; /* bin of the first read that can still reach it */
    uint32_t lo = 0, hi = n_reads;
    while (lo < hi) {
      const uint32_t mid = lo + ((hi - lo) >> 1);
      if (keys_sorted[mid] < key) lo = mid + 1;
      else hi = mid;
    }
    tile_lo[wt] = lo;
  }
}

/*
 * The kernel is instruction-issue bound (profiles/: VALU and scalar units each busy ~2/3 of the time, memory and LDS
 * far from their limits), so the work per (read, tile) pair is kept to a minimum: what depends on the read and the
 * tile but not on the lane — the read's lane range, the address of its first byte in the tile — is computed once per
 * descriptor lane, 64 reads at a time, and reaches the scalar registers with v_readlane; per lane that leaves
 * clamp, load, range test, quality test, class lookup (v_alignbyte on the strand's table), two LDS adds, one add.
 */
extern "C" __global__ __launch_bounds__(64 * ACC_WAVES) void bsc_accumulate_kernel(
    const bsc_read_desc *__restrict__ rd, const uint32_t *__restrict__ keys_sorted, const uint32_t *__restrict__ perm,
    uint32_t n_reads, const uint8_t *__restrict__ seq, uint32_t x, uint32_t y, uint32_t min_qual,
    const uint32_t *__restrict__ tile_lo, uint32_t *__restrict__ cts, unsigned long long *__restrict__ counters) {
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
  auto fetch = [&](uint32_t tb, uint32_t &kv, bsc_read_desc &e) { acc_fetch(rd, keys_sorted, perm, n_reads, tb, lane, kv, e); };
  /* Software pipeline over the wave's tiles: while tile i is processed, the first batch of tile i+1 and the start
   * index of tile i+2 are on their way, so the tile_lo -> key / index -> descriptor chain of dependent loads is off the
   * critical path. */
  const uint32_t wt_step = gridDim.x * ACC_WAVES;
  uint32_t wt = blockIdx.x * ACC_WAVES + wid;
  uint32_t t0 = 0, t0_next = 0, kv = 0xffffffffu;
  bsc_read_desc d;
  d.a = 1;
  d.b = 0;
  d.base = 0;
  d.meta = 0;
  d.lut = 0;
  if (wt < n_wt) {
    t0 = tile_lo[wt];
    if (wt + wt_step < n_wt && wt + wt_step > wt) t0_next = tile_lo[wt + wt_step];
    fetch(t0, kv, d);
  }
  for (; wt < n_wt; wt += wt_step) {
    /* x + 64 wt <= y: the tile's first position fits 32 bits; its last one is clipped to y */
    const uint32_t p0 = x + wt * 64u;
    const uint32_t p_last = y - p0 < 63u ? y : p0 + 63u;
    const uint32_t r_last = p_last - x; /* the tile's last position as a sort key */
    const bool valid = lane <= p_last - p0;
    const uint32_t wt1 = wt + wt_step, wt2 = wt1 + wt_step;
    const bool have1 = wt1 < n_wt && wt1 > wt, have2 = have1 && wt2 < n_wt && wt2 > wt1;
    uint32_t kv_n = 0xffffffffu, t0_nn = 0;
    bsc_read_desc dn;
    dn.a = 1;
    dn.b = 0;
    dn.base = 0;
    dn.meta = 0;
    dn.lut = 0;
    if (have1) fetch(t0_next, kv_n, dn);
    if (have2) t0_nn = tile_lo[wt2];
    {
      uint32_t w[IN_DW];
      inexact |= acc_tile(rd, keys_sorted, perm, n_reads, seq, lane, lane, row, p0, p_last, r_last, min_qual, q_span, t0, kv, d, w) ? 1u : 0u;
#pragma unroll
      for (int i = 0; i < IN_DW / 2; i++) reinterpret_cast<uint2 *>(row)[i] = make_uint2(w[2 * i], w[2 * i + 1]);
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
    t0 = t0_next;
    t0_next = t0_nn;
    kv = kv_n;
    d = dn;
  }
  if (__any(inexact)) {
    const unsigned long long m = __ballot(inexact);
    if (lane == 0) atomicAdd(&counters[BSC_CNT_INEXACT], (unsigned long long)__popcll(m));
  }
}

/* ---- launchers ------------------------------------------------------------------------------------------ */
extern "C" int bsc_dev_sort_templates(const void *keys, void *keys_sorted, void *perm, uint32_t nr, unsigned key_bits,
                                      void *tmp, size_t tmp_bytes, void *stream); /* sort.hip */

/* template checks, read descriptors and the ordering of the block's reads: rd[2 nr], keys_sorted[2 nr], perm[2 nr] */
extern "C" int bsc_dev_launch_prep_reads(const void *tpl, uint32_t nr, const void *seq, uint64_t seq_bytes, uint32_t x, uint32_t y,
                                         void *keys, void *keys_sorted, void *perm, void *sort_tmp, size_t sort_tmp_bytes,
                                         void *rd, void *counters, int num_cus, void *stream) {
  if (!nr) return 0;
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n_sites = y - x + 1;
  const uint32_t n_reads = 2u * nr; /* nr <= 2^31 - 1 is checked by the caller */
  unsigned g = (nr + 255u) / 256u;
  if (g > (unsigned)num_cus * 16u) g = (unsigned)num_cus * 16u;
  /* keys 0 .. key_max - 1: the 64-position bins of x .. y; key_max: a read that contributes nothing */
  const uint32_t key_max = ((n_sites - 1u) >> ACC_BIN_SHIFT) + 1u;
  unsigned key_bits = 1;
  while (key_bits < 32 && (key_max >> key_bits)) key_bits++;
  hipLaunchKernelGGL(bsc_prep_reads_kernel, dim3(g), dim3(256), 0, s, (const bsc_template_dev *)tpl, nr, (const uint8_t *)seq,
                     seq_bytes, x, y, key_max, (bsc_read_desc *)rd, (uint32_t *)keys, (unsigned long long *)counters);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  return bsc_dev_sort_templates(keys, keys_sorted, perm, n_reads, key_bits, sort_tmp, sort_tmp_bytes, stream);
}

/* first candidate read of every tile (see bsc_tile_lo_kernel) */
extern "C" int bsc_dev_launch_tile_lo(const void *keys_sorted, uint32_t n_reads, uint32_t n_tiles, int64_t base, uint32_t step,
                                      const void *counters, void *tile_lo, int num_cus, void *stream) {
  if (!n_tiles) return 0;
  unsigned g = (n_tiles + 255u) / 256u;
  if (g > (unsigned)num_cus * 16u) g = (unsigned)num_cus * 16u;
  hipLaunchKernelGGL(bsc_tile_lo_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const uint32_t *)keys_sorted, n_reads,
                     n_tiles, base, step, (const unsigned long long *)counters, (uint32_t *)tile_lo);
  return (int)hipGetLastError();
}

extern "C" int bsc_dev_launch_accumulate(const void *tpl, uint32_t nr, const void *seq, uint64_t seq_bytes, uint32_t x,
                                         uint32_t y, uint32_t min_qual, void *keys, void *keys_sorted, void *perm,
                                         void *sort_tmp, size_t sort_tmp_bytes, void *rd, void *tile_lo, void *cts,
                                         void *counters, int num_cus, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const uint32_t n_sites = y - x + 1;
  const uint32_t n_wt = (n_sites + 63u) / 64u;
  const uint32_t n_reads = 2u * nr;
  int rc = bsc_dev_launch_prep_reads(tpl, nr, seq, seq_bytes, x, y, keys, keys_sorted, perm, sort_tmp, sort_tmp_bytes, rd, counters,
                                     num_cus, stream);
  if (rc) return rc;
  if ((rc = bsc_dev_launch_tile_lo(keys_sorted, n_reads, n_wt, 0, 64u, counters, tile_lo, num_cus, stream))) return rc;
  {
    unsigned g = (n_wt + ACC_WAVES - 1) / ACC_WAVES;
    const unsigned cap = (unsigned)num_cus * 6u * 8u; /* 6 workgroups of 26 KB LDS fit a CU */
    if (g > cap) g = cap;
    hipLaunchKernelGGL(bsc_accumulate_kernel, dim3(g), dim3(64 * ACC_WAVES), 0, s, (const bsc_read_desc *)rd,
                       (const uint32_t *)keys_sorted, (const uint32_t *)perm, n_reads, (const uint8_t *)seq, x, y, min_qual,
                       (const uint32_t *)tile_lo, (uint32_t *)cts, (unsigned long long *)counters);
  }
  return (int)hipGetLastError();
}
