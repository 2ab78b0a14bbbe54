/*
 * accumulate.hip — gfx950 kernels of the pile-up accumulate stage: reads -> pileup[] for one block of
 * positions [x, y].  Replaces HOT LOOP A of call_genotypes_ML (reference src/call_genotypes.c:178-226),
 * which the reference runs serially on its process thread.
 *
 *   bsc_bin_count_kernel    BIN_TPT templates per thread (their loads issued back to back): the reference's asserts on the template; per read the
 *   bsc_bin_scatter_kernel  orientation it is counted with (:187,224, including the reference's quirk that a read 0
 *                           without a countable base does not flip it) and a compact descriptor; the reads grouped by the
 *                           64-position bin of their first base in two passes around a prefix sum (count in an LDS window
 *                           per workgroup, scatter by rank) — no sort: the sums do not depend on the order inside a bin.
 *   bsc_accumulate_kernel   one wave per 64-position wave-tile, lane i OWNS position i of the tile: the wave walks the
 *                           candidate reads (accdev.h: 64 descriptors per vector load, ballot-filtered to the reads that
 *                           overlap the tile, broadcast with v_readlane), every lane fetches "its" base of the read
 *                           (consecutive lanes = consecutive bytes: coalesced) and bumps its own pile-up row in the wave's
 *                           LDS slot.
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
#define SUM_DW 12   /*  dwords of a site summary (the kernel's summary form; read by fused.hip, which explains the size) */

/* bsc_template (include/bscall_amd.h) as the kernels read it */
struct bsc_template_dev {
  uint32_t pos[2];
  uint32_t len[2];
  uint64_t off[2];
  uint8_t mapq[2];
  uint8_t orientation;
  uint8_t bs_strand;
  uint32_t flags; /* BSC_TPL_* (include/bscall_amd.h) */
};
#define TPL_WALK_KNOWN 1u
#define TPL_WALKED0 2u

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
  if (tp.flags & ~(TPL_WALK_KNOWN | TPL_WALKED0)) return BSC_TERR_FLAGS; /* the device trusts the two bits: junk there is refused */
  return 0;
}

/*
 * What one template contributes: its two read descriptors (dead ones for a read that contributes nothing), and its error code.
 * `walked0`: read 0 has a base with a quality other than 0 / 63.  The reference walks a read from its first to its last such
 * base (:198-211) and tests every base in between against min_qual (:217); the bases it trims off the ends fail that test
 * anyway (0 < min_qual, 63 is excluded by name), so the device walks the whole read.  What the scan does decide is whether a
 * read counts as walked at all: only then is the orientation flipped for its mate (:224, and the `continue`s of :203,:210 in
 * front of it).
 */
__device__ static __forceinline__ uint32_t template_reads(const bsc_template_dev &tp, uint32_t x, uint32_t y, uint64_t seq_bytes,
                                                          bool walked0, bsc_read_desc d[2]) {
  acc_dead(d[0]);
  acc_dead(d[1]);
  const uint32_t terr = template_error(tp, leftmost(tp.pos[0], tp.pos[1]), x, seq_bytes);
  if (terr) return terr;
  uint32_t ori = tp.orientation & 1u;
#pragma unroll
  for (int k = 0; k < 2; k++) {
    const uint32_t rl = tp.len[k];
    if (rl != 0 && (k == 1 || walked0)) {
      const uint64_t pa = (uint64_t)tp.pos[k], pb = (uint64_t)tp.pos[k] + rl - 1u;
      if (pa <= y) {
        d[k].a = (uint32_t)pa;
        d[k].b = pb > y ? y : (uint32_t)pb; /* pos <= y, :214 */
        d[k].base = (int64_t)tp.off[k] - (int64_t)tp.pos[k];
        d[k].meta = ACC_META(ori, tp.mapq[k]);
        d[k].lut = tp.bs_strand == 0 ? LUT4(0, 1, 2, 3) : (tp.bs_strand == 1 ? LUT4(0, 5, 2, 7) : LUT4(4, 1, 6, 3));
      }
    }
    if (k == 0 && walked0) ori ^= 1u; /* :224 — only a read that was walked flips the orientation */
  }
  return 0;
}

/* Several blocks in one pass (bsc_blocks_records): template t belongs to the block whose tpl_end is the first one above t; its
 * positions and bins are that block's (devtables.h).  blk == NULL: one block, x .. y, bins from 0. */
__device__ static __forceinline__ void template_block(const bsc_chain_mblock *__restrict__ blk, uint32_t n_blk, uint32_t t, uint32_t &x,
                                                      uint32_t &y, uint32_t &bin0) {
  bin0 = 0;
  if (!blk) return;
  uint32_t lo = 0, hi = n_blk - 1u; /* t < blk[n_blk - 1].tpl_end */
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (t < blk[mid].tpl_end) hi = mid;
    else lo = mid + 1u;
  }
  x = blk[lo].x;
  y = blk[lo].x + (blk[lo].n - 1u);
  bin0 = blk[lo].bin0;
}

#ifndef BIN_WG
#define BIN_WG 256   /* threads per workgroup */
#endif
#ifndef BIN_TPT
#define BIN_TPT 4    /* templates per thread: a workgroup's lifetime is a chain of latencies (template load, LDS atomics behind a barrier, global
                        atomics), and with one template per thread the 29 k workgroups of a 50 Mb block at 30x were what the two passes waited
                        for — several templates per thread, their loads issued back to back, pay that chain once for all of them */
#endif
#define BIN_TPW (BIN_WG * BIN_TPT) /* templates per workgroup */
#ifndef BIN_WIN
#define BIN_WIN 1024 /* bins of the workgroup's LDS window: a coordinate-ordered align_list keeps a workgroup's 512 reads
                        within a few dozen bins; reads outside the window take a global atomic each */
#endif

/* the workgroup's window starts at the lowest bin among its live reads */
__device__ static __forceinline__ uint32_t wg_min_bin(uint32_t m, uint32_t *s_min) { /* m: the lowest bin among the thread's reads */
  if (threadIdx.x == 0) *s_min = 0xffffffffu;
  __syncthreads();
  for (int o = 32; o > 0; o >>= 1) {
    const uint32_t v = __shfl_xor(m, o);
    m = v < m ? v : m;
  }
  if ((threadIdx.x & 63u) == 0 && m != 0xffffffffu) atomicMin(s_min, m);
  __syncthreads();
  return *s_min;
}

/*
 * Grouping the block's reads by the 64-position bin of their first base, pass 1 of 2 — BIN_TPT templates per thread, in the
 * caller's order: the reference's asserts (the lowest index of an invalid template with its first failing check reaches the
 * host through counters[BSC_CNT_ERR]; such a template contributes nothing), whether read 0 was walked (as the host says,
 * bsc_template.flags, or else found out here: one byte of it, almost always; kept in tflag[] for pass 2), the bins of its two reads counted — in an LDS window of the workgroup, one
 * global atomic per non-empty bin and workgroup — and the longest read extent of the block.
 * READS, not templates, are what the tiles search: a read's extent is bounded by its length, so mates that lie far apart
 * (or a pathological template) cannot widen every tile's candidate window.
 */
extern "C" __global__ __launch_bounds__(BIN_WG) void bsc_bin_count_kernel(const bsc_template_dev *__restrict__ tpl, uint32_t nr,
                                                                          const uint8_t *__restrict__ seq, uint64_t seq_bytes,
                                                                          uint32_t x, uint32_t y, uint8_t *__restrict__ tflag,
                                                                          uint32_t *__restrict__ bin_cnt,
                                                                          unsigned long long *__restrict__ counters,
                                                                          const bsc_chain_mblock *__restrict__ blk, uint32_t n_blk) {
  __shared__ uint32_t s_cnt[BIN_WIN];
  __shared__ uint32_t s_min;
  for (unsigned i = threadIdx.x; i < BIN_WIN; i += BIN_WG) s_cnt[i] = 0;
  uint32_t bin[BIN_TPT][2], span_max = 0, bin_lo = 0xffffffffu;
  bsc_template_dev tps[BIN_TPT];
#pragma unroll
  for (int j = 0; j < BIN_TPT; j++) { /* the loads first, back to back */
    const uint32_t t = blockIdx.x * BIN_TPW + (uint32_t)j * BIN_WG + threadIdx.x;
    if (t < nr) tps[j] = tpl[t];
  }
#pragma unroll
  for (int j = 0; j < BIN_TPT; j++) {
    const uint32_t t = blockIdx.x * BIN_TPW + (uint32_t)j * BIN_WG + threadIdx.x;
    bin[j][0] = bin[j][1] = 0xffffffffu;
    if (t < nr) {
      const bsc_template_dev tp = tps[j];
      uint32_t bin0;
      template_block(blk, n_blk, t, x, y, bin0);
      bool walked0 = false;
      if (tp.flags & TPL_WALK_KNOWN) walked0 = (tp.flags & TPL_WALKED0) != 0; /* the host says (include/bscall_amd.h) */
      else if (template_error(tp, leftmost(tp.pos[0], tp.pos[1]), x, seq_bytes) == 0) { /* read 0's first countable base */
        const uint32_t rl = tp.len[0];
        const uint8_t *sp = seq + tp.off[0];
        for (uint32_t q_ = 0; q_ < rl; q_++) {
          const uint32_t q = (uint32_t)sp[q_] >> 2;
          if (q > 0 && q != 63u) {
            walked0 = true;
            break;
          }
        }
      }
      tflag[t] = walked0 ? 1 : 0;
      bsc_read_desc d[2];
      const uint32_t terr = template_reads(tp, x, y, seq_bytes, walked0, d);
      if (terr) atomicMin(&counters[BSC_CNT_ERR], ((unsigned long long)t << 8) | terr);
#pragma unroll
      for (int k = 0; k < 2; k++)
        if (d[k].b >= d[k].a) { /* live: x <= a <= b <= y */
          bin[j][k] = bin0 + ((d[k].a - x) >> ACC_BIN_SHIFT);
          if (bin[j][k] < bin_lo) bin_lo = bin[j][k];
          if (d[k].b - d[k].a > span_max) span_max = d[k].b - d[k].a;
        }
    }
  }
  const uint32_t base = wg_min_bin(bin_lo, &s_min); /* also orders the zeroing of s_cnt before its use */
#pragma unroll
  for (int j = 0; j < BIN_TPT; j++)
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (bin[j][k] != 0xffffffffu) {
        if (bin[j][k] - base < BIN_WIN) atomicAdd(&s_cnt[bin[j][k] - base], 1u);
        else atomicAdd(&bin_cnt[bin[j][k]], 1u);
      }
  __syncthreads();
  for (unsigned i = threadIdx.x; i < BIN_WIN; i += BIN_WG)
    if (s_cnt[i]) atomicAdd(&bin_cnt[base + i], s_cnt[i]);
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

/*
 * Pass 2: bin_cur[] starts as the exclusive prefix sum of the bin counts (= bin_off[]).  Every workgroup takes, per bin of
 * its window, as many consecutive slots of that bin as it has reads for it (one global atomic per non-empty bin) and its
 * reads take theirs by their rank inside the workgroup (LDS atomic); the descriptors land in rd[] bin after bin.  The order
 * inside a bin depends on the order the atomics were served in — the pile-up sums do not.
 */
extern "C" __global__ __launch_bounds__(BIN_WG) void bsc_bin_scatter_kernel(const bsc_template_dev *__restrict__ tpl, uint32_t nr,
                                                                            uint64_t seq_bytes, uint32_t x, uint32_t y,
                                                                            const uint8_t *__restrict__ tflag,
                                                                            uint32_t *__restrict__ bin_cur,
                                                                            bsc_read_desc *__restrict__ rd,
                                                                            const bsc_chain_mblock *__restrict__ blk, uint32_t n_blk) {
  __shared__ uint32_t s_cnt[BIN_WIN]; /* reads of the workgroup per bin of its window, then: the first slot they got */
  __shared__ uint32_t s_min;
  for (unsigned i = threadIdx.x; i < BIN_WIN; i += BIN_WG) s_cnt[i] = 0;
  uint32_t bin[BIN_TPT][2], rank[BIN_TPT][2], bin_lo = 0xffffffffu;
  bsc_read_desc d[BIN_TPT][2];
  bsc_template_dev tps[BIN_TPT];
  uint8_t tf[BIN_TPT];
#pragma unroll
  for (int j = 0; j < BIN_TPT; j++) { /* the loads first, back to back */
    const uint32_t t = blockIdx.x * BIN_TPW + (uint32_t)j * BIN_WG + threadIdx.x;
    tf[j] = 0;
    if (t < nr) {
      tps[j] = tpl[t];
      tf[j] = tflag[t];
    }
  }
#pragma unroll
  for (int j = 0; j < BIN_TPT; j++) {
    const uint32_t t = blockIdx.x * BIN_TPW + (uint32_t)j * BIN_WG + threadIdx.x;
    bin[j][0] = bin[j][1] = 0xffffffffu;
    rank[j][0] = rank[j][1] = 0;
    acc_dead(d[j][0]);
    acc_dead(d[j][1]);
    if (t < nr) {
      uint32_t bin0;
      template_block(blk, n_blk, t, x, y, bin0);
      (void)template_reads(tps[j], x, y, seq_bytes, tf[j] != 0, d[j]);
#pragma unroll
      for (int k = 0; k < 2; k++)
        if (d[j][k].b >= d[j][k].a) {
          bin[j][k] = bin0 + ((d[j][k].a - x) >> ACC_BIN_SHIFT);
          if (bin[j][k] < bin_lo) bin_lo = bin[j][k];
        }
    }
  }
  const uint32_t base = wg_min_bin(bin_lo, &s_min);
#pragma unroll
  for (int j = 0; j < BIN_TPT; j++)
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (bin[j][k] != 0xffffffffu && bin[j][k] - base < BIN_WIN) rank[j][k] = atomicAdd(&s_cnt[bin[j][k] - base], 1u);
  __syncthreads();
  for (unsigned i = threadIdx.x; i < BIN_WIN; i += BIN_WG)
    if (s_cnt[i]) s_cnt[i] = atomicAdd(&bin_cur[base + i], s_cnt[i]);
  __syncthreads();
#pragma unroll
  for (int j = 0; j < BIN_TPT; j++)
#pragma unroll
    for (int k = 0; k < 2; k++)
      if (bin[j][k] != 0xffffffffu) {
        const uint32_t slot = bin[j][k] - base < BIN_WIN ? s_cnt[bin[j][k] - base] + rank[j][k] : atomicAdd(&bin_cur[bin[j][k]], 1u);
        rd[slot] = d[j][k];
      }
}

/*
 * The stand-alone accumulate kernel: one wave per 64-position tile; lane i owns position i of the tile and bumps its own
 * pile-up row in the wave's LDS slot (accdev.h: the walk).  A row has one writer, so there is no atomic contention and the
 * result does not depend on any ordering; the slot becomes the reference's pileup[] layout and leaves with 16-byte-per-lane
 * stores.
 */
/* SUMM: instead of the 104-byte pile-up the tile leaves 48-byte SITE SUMMARIES — per class its count | its forward-strand part << 16, and the per-site summary of
 * call_thread (src/call_genotypes.c:44-59: rounded mean quality per class, mean quality, MQ; call_summary.inc, the statements
 * the calling kernels run) — which is what the chain kernel's summary-in form starts from: under half the bytes through HBM, and
 * the summary's arithmetic (a ninth of the chain kernel's vector instructions) moves to the kernel that has issue slots to
 * spare. */
template <bool SUMM>
__global__ __launch_bounds__(64 * ACC_WAVES) void bsc_accumulate_kernel_t(
    const bsc_read_desc *__restrict__ rd, const uint32_t *__restrict__ bin_off, uint32_t n_bins, const uint8_t *__restrict__ seq,
    uint32_t x, uint32_t y, uint32_t min_qual, uint32_t *__restrict__ cts, unsigned long long *__restrict__ counters,
    const bsc_chain_mblock *__restrict__ blk, uint32_t n_blk) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_slot[ACC_WAVES][SLOT_DW];
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t *slot = lds_slot[wid];
  uint32_t *row = slot + lane * IN_DW;
  /* blk != NULL (bsc_blocks_submit_to): several blocks, each from a multiple of 64 positions on in the pile-up array — tile wt
   * is bin wt of the grouped reads and belongs to the block whose [bin0, bin_end) holds it; n_bins = the tiles of all blocks */
  const uint32_t n_sites = y - x + 1;
  const uint32_t n_wt = blk ? n_bins : (n_sites + 63u) / 64u;
  /* q counts iff min_qual <= q < 63 (:217)  <=>  (q - min_qual) <u q_span */
  const uint32_t q_span = min_qual < 63u ? 63u - min_qual : 0u;
  unsigned inexact = 0;
  acc_reads R;
  R.rd = rd;
  R.bin_off = bin_off;
  R.seq = seq;
  R.n_bins = n_bins;
  R.x = x;
  uint32_t n_live = bin_off[n_bins];
  const uint32_t span = (uint32_t)counters[BSC_CNT_SPAN]; /* longest read extent, b - a */
  /* Software pipeline over the wave's tiles: while tile i is processed, the first batch of tile i+1 is on its way */
  const uint32_t wt_step = gridDim.x * ACC_WAVES;
  /* Workgroups are dealt round-robin over the 8 XCDs, each with an L2 of its own, and a read is wanted by every tile it
   * overlaps (2.6 of them at 100 bases): with consecutive workgroups on consecutive tile groups, every XCD fetches nearly
   * every read (FETCH_SIZE 3.8 GB per 50 M positions at 30x for 1.6 GB of reads).  So the workgroups that share an XCD
   * (blockIdx % 8: a label, not the XCD's id) take ACC_XCD_CHUNK CONSECUTIVE tile groups at a time — neighbours in the genome
   * are neighbours in one L2 — while the 8 XCDs work side by side in the same stretch of the genome (one contiguous stretch
   * per XCD and round halves the fetched bytes too, but was 3 % slower).  Whole chunks only: the ragged end of the grid stays
   * where it is; a wrong guess about the placement costs speed, nothing else.  ACC_NO_XCD_SWIZZLE: the A/B build. */
#ifndef ACC_XCD_CHUNK
#define ACC_XCD_CHUNK 16
#endif
#ifndef ACC_NO_XCD_SWIZZLE
  const uint32_t per8 = 8u * ACC_XCD_CHUNK, full8 = gridDim.x / per8 * per8;
  const uint32_t lb = blockIdx.x >> 3;
  const uint32_t vb = blockIdx.x < full8 ? (lb / ACC_XCD_CHUNK) * per8 + (blockIdx.x & 7u) * ACC_XCD_CHUNK + lb % ACC_XCD_CHUNK : blockIdx.x;
#else
  const uint32_t vb = blockIdx.x;
#endif
  uint32_t wt = vb * ACC_WAVES + wid;
  uint32_t t0 = 0, kv = 0xffffffffu;
  bsc_read_desc d;
  acc_dead(d);
  uint32_t wl = wt, wl_n = 0; /* the tile's index inside its block (blk: set by ACC_BLOCK_OF) */
  /* the block of tile w (wave-uniform: scalar loads): positions, bins, and where its reads end */
#define ACC_BLOCK_OF(w, wlocal)                                    \
  do {                                                             \
    uint32_t lo_ = 0, hi_ = n_blk - 1u;                            \
    while (lo_ < hi_) {                                            \
      const uint32_t mid_ = (lo_ + hi_) >> 1;                      \
      if ((w) < blk[mid_].bin_end) hi_ = mid_;                     \
      else lo_ = mid_ + 1u;                                        \
    }                                                              \
    x = blk[lo_].x;                                                \
    y = blk[lo_].x + (blk[lo_].n - 1u);                            \
    R.x = x;                                                       \
    R.bin_off = bin_off + blk[lo_].bin0;                           \
    n_live = bin_off[blk[lo_].bin_end];                            \
    (wlocal) = (w)-blk[lo_].bin0;                                  \
  } while (0)
  if (wt < n_wt) {
    if (blk) ACC_BLOCK_OF(wt, wl);
    t0 = acc_tile_start(R, (int64_t)wl * 64, span);
    acc_fetch(R, n_live, t0, lane, kv, d);
  }
  for (; wt < n_wt; wt += wt_step) {
    /* x + 64 wl <= y: the tile's first position fits 32 bits; its last one is clipped to y */
    const uint32_t p0 = x + wl * 64u;
    const uint32_t p_last = y - p0 < 63u ? y : p0 + 63u;
    const uint32_t r_last = p_last - x; /* the tile's last position relative to the block start */
    const bool valid = lane <= p_last - p0;
    const acc_reads Rc = R; /* this tile's block; the request below may move R on to the next tile's */
    const uint32_t n_live_c = n_live;
    const uint32_t wt1 = wt + wt_step;
    const bool have1 = wt1 < n_wt && wt1 > wt;
    uint32_t kv_n = 0xffffffffu, t0_n = 0;
    bsc_read_desc dn;
    acc_dead(dn);
    wl_n = wt1;
    if (have1) {
      if (blk) ACC_BLOCK_OF(wt1, wl_n);
      t0_n = acc_tile_start(R, (int64_t)wl_n * 64, span);
      acc_fetch(R, n_live, t0_n, lane, kv_n, dn);
    }
    {
      uint32_t w[IN_DW];
      bool packed;
      inexact |= acc_tile(Rc, n_live_c, lane, lane, row, p0, p_last, r_last, min_qual, q_span, t0, kv, d, w, &packed) ? 1u : 0u;
      if (SUMM) {
        if (__builtin_expect(!packed, 0)) { /* a tile more than 4 095 reads deep: do its counts still fit the summary's 16 bits? */
          uint32_t cmax = 0;
#pragma unroll
          for (int j = 0; j < 8; j++) cmax = max(cmax, w[j] + w[8 + j]);
          if (__any(cmax > 0xffffu) && lane == 0) atomicMax(&counters[BSC_CNT_DEEP], 1ull);
        }
        uint32_t o[SUM_DW];
        {
#include "call_summary.inc"
#pragma unroll
          for (int j = 0; j < 8; j++) o[j] = ((w[j] + w[8 + j]) & 0xffffu) | (w[j] << 16); /* class count | its forward-strand part */
          o[8] = qpack0;
          o[9] = qpack1;
          o[10] = ((uint32_t)aq & 0xffffu) | ((uint32_t)mq << 16);
          o[11] = covered ? n_reads : 0u;
        }
        /* the rows are re-laid at the summary's stride: every lane has read its own row (acc_tile) — but lane L's new row
         * overlaps old rows of lower lanes only (22 L < 26 L), all read by now: the wave's LDS operations execute in order */
        uint32_t *srow = slot + lane * SUM_DW;
#pragma unroll
        for (int i = 0; i < SUM_DW / 2; i++) reinterpret_cast<uint2 *>(srow)[i] = make_uint2(o[2 * i], o[2 * i + 1]);
      } else {
#pragma unroll
        for (int i = 0; i < IN_DW / 2; i++) reinterpret_cast<uint2 *>(row)[i] = make_uint2(w[2 * i], w[2 * i + 1]);
      }
    }
    /* the slot is the tile's pileup[] image: copy it out */
    const uint32_t nvalid = p_last - p0 + 1u;
    constexpr unsigned ROW = SUMM ? SUM_DW : IN_DW;
    uint32_t *dst = cts + (uint64_t)wt * (64u * ROW);
#ifdef ACC_EXPERIMENT_NOSTORE /* timing experiment: the walk without the tile image's way out */
    if (wt == 0xffffffffu)
#endif
    if (nvalid == 64u) {
      /* written once, read by another kernel much later: non-temporal, like the calling kernel's records (ACC_PLAIN_STORES:
       * the A/B build) */
      typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 *s4 = reinterpret_cast<const u32x4 *>(slot);
      u32x4 *d4 = reinterpret_cast<u32x4 *>(dst);
#ifdef ACC_PLAIN_STORES
#define ACC_ST(p, v) *(p) = (v)
#else
#define ACC_ST(p, v) __builtin_nontemporal_store((v), (p))
#endif
      constexpr int full = (int)(64u * ROW * 4u / 1024u); /* 6.5 KB of pile-ups, 5.5 KB of summaries */
#pragma unroll
      for (int v = 0; v < full; v++) ACC_ST(d4 + v * 64 + lane, s4[v * 64 + lane]);
      constexpr unsigned rem = (64u * ROW * 4u % 1024u) / 16u; /* lanes of the last, partial kilobyte */
      if (rem && lane < rem) ACC_ST(d4 + full * 64 + lane, s4[full * 64 + lane]);
#undef ACC_ST
    } else if (valid || blk) { /* blk: the positions between a block's end and the next multiple of 64 are called too: nothing piled up */
      const uint32_t *orow = slot + lane * ROW;
#pragma unroll
      for (int i = 0; i < (int)ROW; i++) dst[lane * ROW + i] = valid ? orow[i] : 0u;
    }
    t0 = t0_n;
    kv = kv_n;
    d = dn;
    wl = wl_n;
  }
  if (__any(inexact)) {
    const unsigned long long m = __ballot(inexact);
    if (lane == 0) atomicAdd(&counters[BSC_CNT_INEXACT], (unsigned long long)__popcll(m));
  }
}

/*
 * bsc_blocks_submit_to_inplace: the caller's reference codes — y - x + 3 per block, one block after another — into the layout the
 * calling kernel reads beside the pile-ups: a block's y - x + 1 codes from its multiple of 64 on, 0 up to the next block.
 */
extern "C" __global__ __launch_bounds__(256) void bsc_ref_pad_kernel(const uint8_t *__restrict__ packed, const bsc_chain_mblock *__restrict__ blk,
                                                                     uint32_t n_blk, uint8_t *__restrict__ padded, uint32_t n_pos) {
  for (uint32_t p = blockIdx.x * 256u + threadIdx.x; p < n_pos; p += gridDim.x * 256u) {
    uint32_t lo = 0, hi = n_blk; /* the last block whose pos_off is not behind p */
    while (hi - lo > 1u) {
      const uint32_t mid = (lo + hi) >> 1;
      if (blk[mid].pos_off <= p) lo = mid;
      else hi = mid;
    }
    const uint32_t i = p - blk[lo].pos_off;
    padded[p] = i < blk[lo].n ? packed[(uint64_t)blk[lo].ref_in + i] : (uint8_t)0;
  }
}

extern "C" int bsc_dev_launch_ref_pad(const void *packed, const void *d_blk, uint32_t n_blk, void *padded, uint32_t n_pos, int num_cus,
                                      void *stream) {
  if (!n_pos) return 0;
  unsigned g = (n_pos + 255u) / 256u;
  if (g > (unsigned)num_cus * 8u) g = (unsigned)num_cus * 8u;
  hipLaunchKernelGGL(bsc_ref_pad_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, (const uint8_t *)packed, (const bsc_chain_mblock *)d_blk,
                     n_blk, (uint8_t *)padded, n_pos);
  return (int)hipGetLastError();
}

/* ---- launchers ------------------------------------------------------------------------------------------ */
extern "C" int bsc_dev_scan_u32(const void *in, void *out, uint32_t n, void *tmp, size_t tmp_bytes, void *stream); /* sort.hip */

/* bins of a block of n_sites positions */
extern "C" uint32_t bsc_dev_n_bins(uint32_t n_sites) { return ((n_sites - 1u) >> ACC_BIN_SHIFT) + 1u; }

/* template checks, read descriptors and their grouping by bin: rd[<= 2 nr] (live reads, bin after bin), bin_off[n_bins + 1].
 * bin_cnt / bin_off / bin_cur: n_bins + 1 words each; tflag: nr bytes. */
static int launch_bin_reads(const void *tpl, uint32_t nr, const void *seq, uint64_t seq_bytes, uint32_t x, uint32_t y, uint32_t nb,
                            const bsc_chain_mblock *blk, uint32_t n_blk, void *tflag, void *bin_cnt, void *bin_off, void *bin_cur,
                            void *scan_tmp, size_t scan_tmp_bytes, void *rd, void *counters, void *stream) {
  hipStream_t s = (hipStream_t)stream;
  const size_t bytes = ((size_t)nb + 1u) * sizeof(uint32_t);
  if (!nr) return (int)hipMemsetAsync(bin_off, 0, bytes, s); /* no reads: every bin empty */
  hipError_t e = hipMemsetAsync(bin_cnt, 0, bytes, s);
  if (e != hipSuccess) return (int)e;
  const unsigned g = (nr + BIN_TPW - 1u) / BIN_TPW; /* nr <= 2^31 - 1 is checked by the caller */
  hipLaunchKernelGGL(bsc_bin_count_kernel, dim3(g), dim3(BIN_WG), 0, s, (const bsc_template_dev *)tpl, nr, (const uint8_t *)seq,
                     seq_bytes, x, y, (uint8_t *)tflag, (uint32_t *)bin_cnt, (unsigned long long *)counters, blk, n_blk);
  if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  int rc = bsc_dev_scan_u32(bin_cnt, bin_off, nb + 1u, scan_tmp, scan_tmp_bytes, stream);
  if (rc) return rc;
  if ((e = hipMemcpyAsync(bin_cur, bin_off, bytes, hipMemcpyDeviceToDevice, s)) != hipSuccess) return (int)e;
  hipLaunchKernelGGL(bsc_bin_scatter_kernel, dim3(g), dim3(BIN_WG), 0, s, (const bsc_template_dev *)tpl, nr, seq_bytes, x, y,
                     (const uint8_t *)tflag, (uint32_t *)bin_cur, (bsc_read_desc *)rd, blk, n_blk);
  return (int)hipGetLastError();
}

/* template checks, read descriptors and their grouping by bin: rd[<= 2 nr] (live reads, bin after bin), bin_off[n_bins + 1].
 * bin_cnt / bin_off / bin_cur: n_bins + 1 words each; tflag: nr bytes. */
extern "C" int bsc_dev_launch_bin_reads(const void *tpl, uint32_t nr, const void *seq, uint64_t seq_bytes, uint32_t x, uint32_t y,
                                        void *tflag, void *bin_cnt, void *bin_off, void *bin_cur, void *scan_tmp,
                                        size_t scan_tmp_bytes, void *rd, void *counters, void *stream) {
  return launch_bin_reads(tpl, nr, seq, seq_bytes, x, y, bsc_dev_n_bins(y - x + 1), NULL, 0, tflag, bin_cnt, bin_off, bin_cur, scan_tmp,
                          scan_tmp_bytes, rd, counters, stream);
}

/* the same over the templates of several blocks at once: d_blk[n_blk] (device) says which templates, positions and bins are
 * whose; n_bins = the bins of all blocks together */
extern "C" int bsc_dev_launch_bin_reads_multi(const void *tpl, uint32_t nr, const void *seq, uint64_t seq_bytes, const void *d_blk,
                                              uint32_t n_blk, uint32_t n_bins, void *tflag, void *bin_cnt, void *bin_off, void *bin_cur,
                                              void *scan_tmp, size_t scan_tmp_bytes, void *rd, void *counters, void *stream) {
  return launch_bin_reads(tpl, nr, seq, seq_bytes, 0, 0, n_bins, (const bsc_chain_mblock *)d_blk, n_blk, tflag, bin_cnt, bin_off, bin_cur,
                          scan_tmp, scan_tmp_bytes, rd, counters, stream);
}

static int launch_accumulate(const void *rd, const void *bin_off, const void *seq, uint32_t x, uint32_t y, uint32_t n_wt, uint32_t min_qual,
                             void *cts, void *counters, const bsc_chain_mblock *blk, uint32_t n_blk, int num_cus, void *stream,
                             bool summary = false) {
  unsigned g = (n_wt + ACC_WAVES - 1) / ACC_WAVES;
  const unsigned cap = (unsigned)num_cus * 6u * 8u; /* 6 workgroups of 26 KB LDS fit a CU */
  if (g > cap) g = cap;
  if (summary)
    hipLaunchKernelGGL(bsc_accumulate_kernel_t<true>, dim3(g), dim3(64 * ACC_WAVES), 0, (hipStream_t)stream, (const bsc_read_desc *)rd,
                       (const uint32_t *)bin_off, n_wt, (const uint8_t *)seq, x, y, min_qual, (uint32_t *)cts,
                       (unsigned long long *)counters, blk, n_blk);
  else
    hipLaunchKernelGGL(bsc_accumulate_kernel_t<false>, dim3(g), dim3(64 * ACC_WAVES), 0, (hipStream_t)stream, (const bsc_read_desc *)rd,
                       (const uint32_t *)bin_off, n_wt, (const uint8_t *)seq, x, y, min_qual, (uint32_t *)cts,
                       (unsigned long long *)counters, blk, n_blk);
  return (int)hipGetLastError();
}

/* the summary form (one block): cts receives (positions rounded up to 64) x 48 bytes */
extern "C" int bsc_dev_launch_accumulate_summary(const void *rd, const void *bin_off, const void *seq, uint32_t x, uint32_t y,
                                                 uint32_t min_qual, void *cts, void *counters, int num_cus, void *stream) {
  return launch_accumulate(rd, bin_off, seq, x, y, bsc_dev_n_bins(y - x + 1), min_qual, cts, counters, NULL, 0, num_cus, stream, true);
}
extern "C" size_t bsc_dev_summary_bytes(void) { return SUM_DW * 4u; }

extern "C" int bsc_dev_launch_accumulate(const void *rd, const void *bin_off, const void *seq, uint32_t x, uint32_t y, uint32_t min_qual,
                                         void *cts, void *counters, int num_cus, void *stream) {
  return launch_accumulate(rd, bin_off, seq, x, y, bsc_dev_n_bins(y - x + 1), min_qual, cts, counters, NULL, 0, num_cus, stream);
}

/* several blocks (d_blk[n_blk], device): n_bins = the 64-position tiles of all of them = the bins of their grouped reads */
extern "C" int bsc_dev_launch_accumulate_multi(const void *rd, const void *bin_off, const void *seq, const void *d_blk, uint32_t n_blk,
                                               uint32_t n_bins, uint32_t min_qual, void *cts, void *counters, int num_cus, void *stream) {
  return launch_accumulate(rd, bin_off, seq, 0, 0, n_bins, min_qual, cts, counters, (const bsc_chain_mblock *)d_blk, n_blk, num_cus, stream);
}
