/*
 * accdev.h — the device side of HOT LOOP A (reference src/call_genotypes.c:198-224) shared by the stand-alone accumulate
 * kernel (accumulate.hip) and the reads-in chain kernel (fused.hip): the read descriptor, the fetch of a batch of
 * candidate reads and the walk of a 64-position tile over its candidates.  One set of statements, hence the same sums.
 */
#ifndef BSCALL_AMD_ACCDEV_H
#define BSCALL_AMD_ACCDEV_H

#include <hip/hip_runtime.h>
#include <stdint.h>

/* Reads are grouped by the 64-position bin of their first base (accumulate.hip: count, scan, scatter — the sums do not
 * depend on the order inside a bin, so a full sort is not needed): rd[] holds the live reads bin after bin, bin_off[b] the
 * index of bin b's first read, bin_off[n_bins] their number.  A tile's candidates are the contiguous range from the bin of
 * (tile start - longest read extent) on. */
#define ACC_BIN_SHIFT 6

/* what the walk needs of the grouped reads */
struct acc_reads {
  const struct bsc_read_desc *rd; /* live reads, grouped by bin */
  const uint32_t *bin_off;        /* n_bins + 1 entries */
  const uint8_t *seq;
  uint32_t n_bins, x;             /* x: the block's first position (bin 0 starts there) */
};

/* timing experiments only (tools/build_variant_fused.sh): the walk without its loads / without its updates */
#if defined(ACC_EXPERIMENT_NOLOAD)
#define ACC_EXP_LOAD(u, rs, pks) g_byte[u] = (unsigned char)(((lane_p - ((pks)&63u)) <= (((pks) >> 6) & 63u)) ? (100u + ((lane_p + (pks)) & 3u)) : 0u);
#else
#define ACC_EXP_LOAD(u, rs, pks) g_byte[u] = __builtin_amdgcn_raw_buffer_load_b8(rs, (int)(lane_p - ((pks)&63u)), 0, 0);
#endif
#if defined(ACC_EXPERIMENT_NOUPDATE)
#define ACC_EXP_UPD(u) m2sum ^= (uint32_t)g_byte[u]; if (false)
#else
#define ACC_EXP_UPD(u)
#endif

#ifndef ACC_GROUP
#define ACC_GROUP 16 /* reads whose byte loads are issued back to back before the first is consumed (4, 8 or 16) */
#endif

/* compact read descriptor, 24 bytes; a > b: the read contributes nothing */
struct __attribute__((aligned(8))) bsc_read_desc {
  uint32_t a;    /* position of the read's first base (absolute) */
  uint32_t b;    /* position of its last base (absolute, already clipped to y) */
  int64_t base;  /* seq offset of position 0: byte of position p is seq[base + p] */
  uint32_t meta; /* ACC_META: bit 12 = the orientation the read is counted with, bits 16-31 = mapq^2 */
  uint32_t lut;  /* 4 * class of base codes 0..3 on the read's bisulfite strand, one byte each (LUT4 below) */
};
#define ACC_META(ori, mapq) ((((uint32_t)(ori)&1u) << 12) | (((uint32_t)(mapq) * (uint32_t)(mapq)) << 16))

/* strand -> 4 * class, one byte per base code (reference base_tab_st, src/call_genotypes.c:17-19):
 * NON_CONVERTED 0 1 2 3 ; C2T 0 5 2 7 ; G2A 4 1 6 3.  Byte-indexed so that one v_alignbyte_b32 turns a base code into
 * the byte offset of its class inside a pile-up row. */
#define LUT4(c0, c1, c2, c3) ((uint32_t)(4 * (c0)) | ((uint32_t)(4 * (c1)) << 8) | ((uint32_t)(4 * (c2)) << 16) | ((uint32_t)(4 * (c3)) << 24))

__device__ static __forceinline__ void acc_dead(bsc_read_desc &e) {
  e.a = 1;
  e.b = 0;
  e.base = 0;
  e.meta = 0;
  e.lut = 0;
}

/* 64 candidate reads per batch: lane i gets the descriptor of the read that comes (tb + i)-th in bin order and its bin */
__device__ static __forceinline__ void acc_fetch(const acc_reads &R, uint32_t n_live, uint32_t tb, unsigned lane, uint32_t &kv,
                                                 bsc_read_desc &e) {
  const uint32_t t = tb + lane;
  kv = 0xffffffffu;
  acc_dead(e);
  if (t < n_live) {
    e = R.rd[t];
    kv = (e.a - R.x) >> ACC_BIN_SHIFT;
  }
}

/* index of the first candidate read of a tile whose first position is r0 (relative to the block start; below 0: the block
 * start), given the longest read extent of the block */
__device__ static __forceinline__ uint32_t acc_tile_start(const acc_reads &R, int64_t r0, uint32_t span) {
  const int64_t k = r0 - (int64_t)span;
  return R.bin_off[k > 0 ? (uint32_t)k >> ACC_BIN_SHIFT : 0u];
}

/*
 * One tile: lane L owns genome position pa + (L - loff) for L - loff in 0 .. p_last - pa (lane_p = L - loff as an unsigned
 * number: the lanes in front of the tile's first position wrap to huge values and fall out of the range of every read) and
 * bumps its own pile-up row (`row`, 26 dwords in LDS, zeroed by the caller; counts and INTEGER quality sums, the caller
 * converts).  The wave walks the candidate reads from the t0-th in key order — the first batch (kv, d) fetched by the
 * caller — until a read starts right of the tile.  The lane's MAPQ^2 sum comes back in two parts (m2sum, m2cnt; see acc_unpack).
 *
 * The VALU is what this loop (and the kernels around it) run out of, so the work per (read, tile) pair is kept off it where
 * possible.  What depends on the read and the tile but not on the lane — the read's lane range, its orientation, its
 * MAPQ^2, the address of its first byte in the tile — is computed once per descriptor lane, 64 reads at a time, packed into
 * one dword and reaches the scalar registers with four v_readlane (packed word, class table, address); the scalar unit
 * unpacks it.  The bytes are fetched through a buffer descriptor built per read (base = the read's first byte in the tile,
 * num_records = its length there): a lane outside the read gets 0 from the range check of the load itself — quality 0,
 * which never counts (min_qual >= 1, src/parse_args.c:170-171) — so no lane clamps an index or tests a range.  Per lane
 * that leaves: offset, load, window test, class lookup (v_alignbyte on the strand's table), two address adds, the LDS
 * add(s), one add.
 *
 * The LDS is the other thing the loop runs out of (sixteen waves of a chain workgroup share one): PACKED = true keeps count
 * and quality of a (strand, class) cell in ONE dword — count in bits 0-11, the sum of the counted read BYTES (quality << 2 |
 * base) in bits 12-31 — so a counted base costs one ds_add_u32 instead of two; acc_unpack() takes the cell apart (a class's
 * base is class & 3, so its quality sum is (byte sum - base * count) >> 2).  A cell holds 4 095 bases (255 * 4 095 < 2^20):
 * the walk counts the reads it has applied and gives up (returns false, rows to be zeroed and walked again unpacked) before
 * their number could reach that.  PACKED = false is the plain layout: counts[2][8] at dwords 0-15, quality sums at 17-24.
 */
#define ACC_PACK_MAX 4095u
template <bool PACKED>
__device__ static __forceinline__ bool acc_walk(const acc_reads &R, uint32_t n_live, unsigned lane, uint32_t lane_p, uint32_t *row,
                                                uint32_t pa, uint32_t p_last, uint32_t r_last, uint32_t min_qual, uint32_t q_span,
                                                uint32_t t0, uint32_t kv, bsc_read_desc d, uint32_t m2_ref, uint32_t &m2sum,
                                                uint32_t &m2cnt) {
  const uint8_t *__restrict__ const seq = R.seq;
  m2sum = 0; /* MAPQ^2 of this lane's counted bases from reads whose MAPQ^2 is not m2_ref ... */
  m2cnt = 0; /* ... and their number (acc_unpack puts the lane's sum together) */
  uint32_t applied = 0; /* reads applied to the tile so far: no cell can hold more bases than that */
  const uint32_t b_lo = min_qual << 2, b_span = q_span << 2; /* the window test on the byte: quality in [min_qual, 63) */
  bool more = true;
  while (more) {
    const bool cand = kv <= (r_last >> ACC_BIN_SHIFT); /* reads are in key (bin) order: the candidates are a prefix of the batch */
    more = __all(cand);
    {
      /* reads that overlap the tile at all (a read past the candidates starts right of the tile: a > p_last) */
      unsigned long long m = __ballot(d.b >= d.a && d.b >= pa && d.a <= p_last);
      if (PACKED) {
        applied += (uint32_t)__builtin_popcountll(m);
        if (__builtin_expect(applied > ACC_PACK_MAX, 0)) return false;
      }
      /* per descriptor lane: the read's part of the tile as position offsets lo .. lo + len from pa (meaningless where the
       * read does not overlap, never used there) packed with its orientation and MAPQ^2, and the address of the byte at lo */
      const uint32_t lo = (d.a > pa ? d.a : pa) - pa;
      const uint32_t pk = lo | (((d.b < p_last ? d.b : p_last) - pa - lo) << 6) | d.meta;
      /* PACKED: a counted base touches one cell, counts[ori][c] — the strand's table with the orientation's 32 bytes folded
       * into every entry gives its offset in one lookup */
      const uint32_t lutv = PACKED ? d.lut + ((d.meta >> 12) & 1u) * 0x20202020u : d.lut;
      const uint64_t sp = (uint64_t)(uintptr_t)seq + (uint64_t)(d.base + (int64_t)pa + (int64_t)lo);
      /* Groups of ACC_GROUP, then smaller ones: the byte loads of a group are issued back to back and consumed
       * afterwards, so the wave waits for memory once per group.  Straight-line code per group size: slot
       * conditions inside a group would be evaluated on the VALU. */
      uint32_t cnt = (uint32_t)__builtin_popcountll(m);
      unsigned char g_byte[ACC_GROUP];
      uint32_t g_pk[ACC_GROUP], g_lut[ACC_GROUP];
#define ACC_LOAD(u)                                                                                              \
  {                                                                                                              \
    const int src = __builtin_ctzll(m);                                                                          \
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(src)); /* m &= m - 1 in one scalar instruction */                 \
    const uint32_t pks = (uint32_t)__builtin_amdgcn_readlane(pk, src);                                           \
    /* v_readlane returns int: without the casts the low word would be sign-extended into the high one */        \
    const uint32_t sp_lo = (uint32_t)__builtin_amdgcn_readlane((uint32_t)sp, src);                               \
    const uint32_t sp_hi = (uint32_t)__builtin_amdgcn_readlane((uint32_t)(sp >> 32), src);                       \
    g_lut[u] = (uint32_t)__builtin_amdgcn_readlane(lutv, src);                                                   \
    g_pk[u] = pks;                                                                                               \
    /* raw buffer over the read's bytes in the tile, lo .. lo + len: offsets beyond it (the lanes in front of the   \
     * read wrap to huge offsets) read as 0 */                                                                   \
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(                                         \
        (void *)(uintptr_t)(((uint64_t)sp_hi << 32) | sp_lo), 0, (int)(((pks >> 6) & 63u) + 1u), 0x00020000);    \
    ACC_EXP_LOAD(u, rs, pks)                                                                                     \
  }
#define ACC_UPDATE(u)                                                                                            \
  ACC_EXP_UPD(u) {                                                                                               \
    const uint32_t byte = g_byte[u], pks = g_pk[u];                                                              \
    if (byte - b_lo < b_span) { /* the base counts iff min_qual <= quality < 63 (quality = byte >> 2) */         \
      /* v_alignbyte_b32 shifts by 8 * (byte & 3): the class offset of this base arrives in the low byte */      \
      const uint32_t c4 = __builtin_amdgcn_alignbyte(0u, g_lut[u], byte) & 0xffu;                                \
      char *rc = reinterpret_cast<char *>(row) + c4 + (PACKED ? 0u : ((pks >> 7) & 32u)); /* the cell of counts[ori][c] */ \
      /* fire-and-forget ds_add_u32: the row has a single writer (this lane), no contention */                   \
      if (PACKED) atomicAdd(reinterpret_cast<uint32_t *>(rc), (byte << 12) | 1u); /* count++, byte sum += byte */ \
      else {                                                                                                     \
        atomicAdd(reinterpret_cast<uint32_t *>(rc), 1u);                                  /* counts[ori][c]++ */ \
        atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(row) + c4 + 68), byte >> 2); /* quality[c] += q */ \
      }                                                                                                          \
      /* mapq2 += mapq^2: a read with the tile's reference MAPQ^2 (nearly every read) costs nothing here — its bases \
       * are counted in the row anyway and acc_unpack multiplies; the others add theirs (v_add_u32 clamp: a sum   \
       * past 2^32 sticks there, so INEXACT cannot be missed) and are counted.  The test is on scalar registers;  \
       * the empty assembly keeps it a branch (if-converted it is three vector instructions for every read) */   \
      if ((pks >> 16) != m2_ref) {                                                                               \
        asm volatile("; a read off the reference MAPQ^2");                                                       \
        m2sum = __builtin_elementwise_add_sat(m2sum, pks >> 16);                                                 \
        m2cnt++;                                                                                                 \
      }                                                                                                          \
    }                                                                                                            \
  }
#if ACC_GROUP == 16
      for (; cnt >= 16u; cnt -= 16u) {
        ACC_LOAD(0) ACC_LOAD(1) ACC_LOAD(2) ACC_LOAD(3) ACC_LOAD(4) ACC_LOAD(5) ACC_LOAD(6) ACC_LOAD(7)
        ACC_LOAD(8) ACC_LOAD(9) ACC_LOAD(10) ACC_LOAD(11) ACC_LOAD(12) ACC_LOAD(13) ACC_LOAD(14) ACC_LOAD(15)
        ACC_UPDATE(0) ACC_UPDATE(1) ACC_UPDATE(2) ACC_UPDATE(3) ACC_UPDATE(4) ACC_UPDATE(5) ACC_UPDATE(6) ACC_UPDATE(7)
        ACC_UPDATE(8) ACC_UPDATE(9) ACC_UPDATE(10) ACC_UPDATE(11) ACC_UPDATE(12) ACC_UPDATE(13) ACC_UPDATE(14) ACC_UPDATE(15)
      }
      if (cnt & 8u) {
        ACC_LOAD(0) ACC_LOAD(1) ACC_LOAD(2) ACC_LOAD(3) ACC_LOAD(4) ACC_LOAD(5) ACC_LOAD(6) ACC_LOAD(7)
        ACC_UPDATE(0) ACC_UPDATE(1) ACC_UPDATE(2) ACC_UPDATE(3) ACC_UPDATE(4) ACC_UPDATE(5) ACC_UPDATE(6) ACC_UPDATE(7)
      }
      if (cnt & 4u) {
        ACC_LOAD(0) ACC_LOAD(1) ACC_LOAD(2) ACC_LOAD(3)
        ACC_UPDATE(0) ACC_UPDATE(1) ACC_UPDATE(2) ACC_UPDATE(3)
      }
#elif ACC_GROUP == 8
      for (; cnt >= 8u; cnt -= 8u) {
        ACC_LOAD(0) ACC_LOAD(1) ACC_LOAD(2) ACC_LOAD(3) ACC_LOAD(4) ACC_LOAD(5) ACC_LOAD(6) ACC_LOAD(7)
        ACC_UPDATE(0) ACC_UPDATE(1) ACC_UPDATE(2) ACC_UPDATE(3) ACC_UPDATE(4) ACC_UPDATE(5) ACC_UPDATE(6) ACC_UPDATE(7)
      }
      if (cnt & 4u) {
        ACC_LOAD(0) ACC_LOAD(1) ACC_LOAD(2) ACC_LOAD(3)
        ACC_UPDATE(0) ACC_UPDATE(1) ACC_UPDATE(2) ACC_UPDATE(3)
      }
#else
      for (; cnt >= 4u; cnt -= 4u) {
        ACC_LOAD(0) ACC_LOAD(1) ACC_LOAD(2) ACC_LOAD(3)
        ACC_UPDATE(0) ACC_UPDATE(1) ACC_UPDATE(2) ACC_UPDATE(3)
      }
#endif
      if (cnt & 2u) {
        ACC_LOAD(0) ACC_LOAD(1)
        ACC_UPDATE(0) ACC_UPDATE(1)
      }
      if (cnt & 1u) {
        ACC_LOAD(0)
        ACC_UPDATE(0)
      }
#undef ACC_LOAD
#undef ACC_UPDATE
    }
    t0 += 64u;
    if (more) acc_fetch(R, n_live, t0, lane, kv, d); /* further batches: deep data */
  }
  return true;
}

/* The lane's row after the walk -> the 26 dwords of its `pileup` record (include/bs_call.h:174-182) in w[]: counts[2][8], n,
 * the eight quality sums and the MAPQ^2 sum as the floats the reference accumulates (integer sums convert exactly below
 * 2^24: DESIGN.md section 2).  Returns whether a sum left that range (BSC_WARN_INEXACT). */
__device__ static __forceinline__ bool acc_unpack(const uint32_t *row, bool packed, uint32_t m2_ref, uint32_t m2sum, uint32_t m2cnt,
                                                  uint32_t w[26]) {
  uint32_t qs[8], n = 0;
  if (packed) {
#pragma unroll
    for (int j = 0; j < 8; j++) {
      const uint2 a = make_uint2(row[j], row[8 + j]);
      w[j] = a.x & 0xfffu;
      w[8 + j] = a.y & 0xfffu;
      qs[j] = ((a.x >> 12) + (a.y >> 12) - (uint32_t)(j & 3) * (w[j] + w[8 + j])) >> 2;
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; j++) w[j] = row[j];
#pragma unroll
    for (int j = 0; j < 8; j++) qs[j] = row[17 + j];
  }
#pragma unroll
  for (int j = 0; j < 16; j++) n += w[j];
  w[16] = n;
  /* the position's MAPQ^2 sum: its n - m2cnt bases from reads with the reference value, and the others' own sum; what the
   * reference's sequence of float additions gives is this integer as a float while it stays below 2^24, and a sum past 2^32
   * (where a chain of saturating additions would stick) is as inexact as one past 2^24 */
  const uint64_t m2w = (uint64_t)m2_ref * (uint64_t)(n - m2cnt) + (uint64_t)m2sum;
  const uint32_t m2 = (m2sum == 0xffffffffu || m2w > 0xffffffffull) ? 0xffffffffu : (uint32_t)m2w;
  bool inexact = m2 >= (1u << 24);
#pragma unroll
  for (int j = 0; j < 8; j++) {
    inexact |= qs[j] >= (1u << 24);
    w[17 + j] = __float_as_uint((float)qs[j]);
  }
  w[25] = __float_as_uint((float)m2);
  return inexact;
}

/* the whole tile: rows zeroed, walked packed, and — a tile more than ACC_PACK_MAX reads deep — once more unpacked */
__device__ static __forceinline__ bool acc_tile(const acc_reads &R, uint32_t n_live, unsigned lane, uint32_t lane_p, uint32_t *row,
                                                uint32_t pa, uint32_t p_last, uint32_t r_last, uint32_t min_qual, uint32_t q_span,
                                                uint32_t t0, uint32_t kv, const bsc_read_desc &d, uint32_t w[26], bool *was_packed = nullptr) {
  uint32_t m2sum, m2cnt;
  /* the tile's reference MAPQ^2: its first candidate's (any value gives the same sums; the common one gives them cheaply) */
  const uint32_t m2_ref = (uint32_t)__builtin_amdgcn_readfirstlane(d.meta) >> 16;
#pragma unroll
  for (int i = 0; i < 8; i++) reinterpret_cast<uint2 *>(row)[i] = make_uint2(0u, 0u);
  bool packed = acc_walk<true>(R, n_live, lane, lane_p, row, pa, p_last, r_last, min_qual, q_span, t0, kv, d, m2_ref, m2sum, m2cnt);
  if (__builtin_expect(!packed, 0)) {
#pragma unroll
    for (int i = 0; i < 13; i++) reinterpret_cast<uint2 *>(row)[i] = make_uint2(0u, 0u);
    (void)acc_walk<false>(R, n_live, lane, lane_p, row, pa, p_last, r_last, min_qual, q_span, t0, kv, d, m2_ref, m2sum, m2cnt);
  }
  if (was_packed) *was_packed = packed;
  return acc_unpack(row, packed, m2_ref, m2sum, m2cnt, w);
}

#endif /* BSCALL_AMD_ACCDEV_H */
