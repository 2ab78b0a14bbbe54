/*
 * bcfdev.hip — the tail of row f-1 on the device: a block's packed written records (bsc_vcf_rec, compact.hip) as the BCF2
 * records bcf_write() emits for them — the typed values _print_vcf_entry encodes with htslib's bcf_enc_* (src/print_vcf.c:160-222
 * the shared block, :267-378 the per-sample block) behind bcf_write's 32 bytes of fixed fields.  The host form is csrc/bcf.c
 * (bsc_bcf_record / bsc_bcf_block: the checker of this file, statement for statement the same rules — see its header for the
 * BCF2 typed-value rules and the dictionary indices); what crosses PCIe is the stream the output thread writes, ~113 bytes per
 * written record instead of 128, and no host core touches a record.
 *
 *   bsc_bcf_size_kernel    one wave per tile of 64 records: every lane the length of its record (the emitter below over a sink
 *                          that only counts), the tile's sum -> tile_bytes[tile] (u64); records beyond *n_recs count 0
 *   (exclusive scan of the tile sums, rocPRIM u64: sort.hip; one more entry behind the last tile = the stream's length)
 *   bsc_bcf_write_kernel_t one wave per tile: lane offsets from a wave prefix sum, every lane writes its record into the wave's LDS
 *                          image of the tile's span of the stream — field by field, a key and its value composed as one word in
 *                          registers (see "sinks" below); the image starts at the span's phase within 16 bytes — then the wave copies
 *                          the image out: whole 16-byte chunks as one dwordx4 store per lane, the ragged head and tail byte by byte
 *                          (the neighbouring tiles own the other bytes of those chunks).  A tile whose span does not fit the image
 *                          (8 KB: long IDs, wide dictionary indices, deep counts) goes out in 2, 4 or 8 parts.
 *
 * Two sources: packed records (bsc_vcf_compact_device's output; bsc_bcf_block_device), or the per-position arrays the reads-in chain leaves
 * (bsc_vcf_core + the 64-byte aux array that is the packed record's second half; bsc_bcf_sites_device, and what bsc_block_bcf runs): the
 * tiles are then tiles of 64 POSITIONS and the packing pass (128 bytes written and read again per record) does not run at all.  With the
 * chain's byte per position (bsc_bcf_sites_len_device; 0 = no record, 1 .. 254 = the record's BCF2 length, 255 = ask the record) neither
 * kernel touches a position that writes nothing, the write kernel fetches that byte a tile ahead, issues a record's eight loads together
 * and runs the emitter once (the length is the byte); without it a lane reads its position's first 16 bytes to learn whether there is a
 * record, and the emitter runs twice (over the counting sink for the lane offsets, then over the writing one).
 *
 * Names: the dbSNP name of a record whose rs_found flag is set comes from a table of the block's flagged positions (sorted
 * positions, offsets, bytes: bsc_dbsnp_names on the host, uploaded with the block) by binary search; a flagged record the table
 * does not list has no ID, as in bsc_bcf_block when bsc_dbsnp_name finds nothing.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "../../include/bscall_amd.h"

static_assert(sizeof(bsc_vcf_rec) == 128, "bsc_vcf_rec is 128 bytes");

enum { BT_INT8 = 1, BT_INT16 = 2, BT_INT32 = 3, BT_FLOAT = 5, BT_CHAR = 7 };
#define BCF_ID_MAX 63u        /* bsc_bcf_block's rs[64] */
#define BCF_REC_MAX 336u      /* 32 + shared (3 + 63 + 2 + 4 + 5 + 5 + 6) + per-sample (13 keys x 5 + 136): an upper bound of one record */
#define BCF_IMG_BYTES 10752u  /* the one-pass kernel's image per wave: 32 records of the longest kind (10 752 = 32 x 336), or 64 ordinary ones */
#define BCF_WAVES 4u
static_assert(BCF_IMG_BYTES >= 32u * BCF_REC_MAX, "half a tile of the longest records must fit the wave's image");
#define BCF_ONEPASS_WAVES_PER_EU 3
/* The write kernel's image per wave and the waves a SIMD is to hold, by form (tools/r06_ab_bcf_words.sh: the A/B of these).  A tile of the
 * per-position form carries ~32 records of ~113 bytes at WGBS densities (3.6 KB; every position written: 7.2 KB, two parts); a tile of
 * packed records 64 of them. */
#ifndef BCF_IMG_SITES
#define BCF_IMG_SITES 8192u
#endif
#ifndef BCF_WPE_SITES
#define BCF_WPE_SITES 4
#endif
#ifndef BCF_IMG_PACKED
#define BCF_IMG_PACKED 8192u
#endif
#ifndef BCF_WPE_PACKED
#define BCF_WPE_PACKED 4
#endif

struct bcf_args {
  const uint8_t *recs;                /* bsc_vcf_rec[] — or NULL: the records are taken where the chain left them, */
  const uint8_t *core, *aux;          /* bsc_vcf_core[] and the chain's aux array (64 B per position: the second half of a bsc_vcf_rec) */
  const unsigned long long *n_recs;   /* device: how many records / positions (NULL: max_recs of them) */
  uint64_t max_recs;                  /* never more than this (the arrays' size) */
  int32_t rid;
  bsc_bcf_ids ids;
  const uint32_t *name_pos;           /* n_names sorted 1-based positions, or NULL */
  const uint32_t *name_off;           /* n_names + 1 offsets into name_bytes */
  const uint8_t *name_bytes;
  uint32_t n_names;
  const uint8_t *gate;                /* per-position form: the chain's byte per position (0 = no record) or NULL — a position without a record
                                       * then costs that byte, not the 64-byte sector around its first 16 bytes; with one-byte dictionary
                                       * indices a byte of 1 .. 254 is the record's length (fused.hip: ebyte) */
};

/* ---- sinks: the emitter runs over one that counts and one that writes ----
 * Round 6 (second form): the writer composes FIELDS, not bytes — a key and its value are one 64-bit word in registers and two dword
 * stores.  The first form stored byte by byte (p[len++] = ...): 1 144 vector instructions and 99 LDS stores per tile of 64 positions, the
 * kernel at 0.70 of VALU issue (profiles/r06_bcf_write_sq_counters_before.txt; now 537 and 43, r06_bcf_write_sq_counters.txt: the LDS's
 * busy cycles are the nearest bound).  The LDS takes stores at any byte address (gfx950 under HSA: unaligned access mode; the compiler
 * emits ds_write_b16 / b32 for align-1 copies), and a store may write MORE bytes than the field has (put_n always writes 8): what follows
 * in the record overwrites the excess, and the excess behind a record's LAST field (<= 7 bytes) falls into the next lane's 32 fixed
 * bytes, which every lane writes AFTER all bodies are written (a wave barrier between; the image has 32 spare bytes behind the last
 * record). */
struct count_sink {
  unsigned len;
  __device__ __forceinline__ void u8(unsigned) { len++; }
  __device__ __forceinline__ void le(uint32_t, unsigned bytes) { len += bytes; }
  __device__ __forceinline__ void put_n(uint64_t, unsigned n) { len += n; }
  __device__ __forceinline__ void put_w(uint32_t) { len += 4u; }
  __device__ __forceinline__ void words6(const uint32_t *, unsigned n_used) { len += 4u * n_used; }
  __device__ __forceinline__ void cond8(unsigned, bool adv) { len += adv ? 1u : 0u; }
};
struct lds_sink {
  uint8_t *p;
  unsigned len;
  __device__ __forceinline__ void u8(unsigned v) { p[len++] = (uint8_t)v; }
  __device__ __forceinline__ void le(uint32_t v, unsigned bytes) {
    for (unsigned k = 0; k < bytes; k++) p[len++] = (uint8_t)(v >> (8u * k));
  }
  /* the low n bytes of bits (n <= 8); eight bytes are written, as two dwords (one store of eight bytes at an odd address measured
   * slower: 1.62 against 1.58 ms, and slower still for the fixed fields — profiles/r06_ab_bcf_words.txt), four when the compiler
   * knows that n <= 4 */
  __device__ __forceinline__ void put_n(uint64_t bits, unsigned n) {
    const uint32_t lo = (uint32_t)bits, hi = (uint32_t)(bits >> 32);
    __builtin_memcpy(p + len, &lo, 4);
    if (!(__builtin_constant_p(n) && n <= 4u)) __builtin_memcpy(p + len + 4u, &hi, 4);
    len += n;
  }
  __device__ __forceinline__ void put_w(uint32_t v) {
    __builtin_memcpy(p + len, &v, 4);
    len += 4u;
  }
  /* the first n_used of six dwords; all six are written */
  __device__ __forceinline__ void words6(const uint32_t *w, unsigned n_used) {
#pragma unroll
    for (int k = 0; k < 6; k++) __builtin_memcpy(p + len + 4u * (unsigned)k, &w[k], 4);
    len += 4u * n_used;
  }
  /* a byte that stays only if adv (the next one lands on it otherwise) */
  __device__ __forceinline__ void cond8(unsigned v, bool adv) {
    p[len] = (uint8_t)v;
    len += adv ? 1u : 0u;
  }
};

__device__ __forceinline__ int int_type(int32_t lo, int32_t hi) {
  if (hi <= 127 && lo >= -120) return BT_INT8;
  if (hi <= 32767 && lo >= -32760) return BT_INT16;
  return BT_INT32;
}
__device__ __forceinline__ unsigned type_bytes(int t) { return t == BT_INT8 ? 1u : (t == BT_INT16 ? 2u : 4u); }

/* a typed single integer as the low n bytes of a word: the descriptor, then the value in 1 / 2 / 4 bytes (nothing above them) */
__device__ __forceinline__ uint64_t enc_int(int32_t v, unsigned &n) {
  const bool i8 = v <= 127 && v >= -120, i16 = v <= 32767 && v >= -32760;
  const uint32_t t = i8 ? (uint32_t)BT_INT8 : (i16 ? (uint32_t)BT_INT16 : (uint32_t)BT_INT32);
  const uint32_t val = i8 ? ((uint32_t)v & 0xffu) : (i16 ? ((uint32_t)v & 0xffffu) : (uint32_t)v);
  n = i8 ? 2u : (i16 ? 3u : 5u);
  return (uint64_t)(0x10u | t) | (uint64_t)val << 8;
}
template <class S>
__device__ __forceinline__ void put_int(S &s, int32_t v) { /* a typed single integer */
  unsigned n;
  const uint64_t b = enc_int(v, n);
  s.put_n(b, n);
}
/* a dictionary index as a typed integer; SHORT: every index of the block is known to be 0 .. 127 (the launcher looked), two bytes */
template <bool SHORT>
__device__ __forceinline__ uint64_t enc_key(int32_t key, unsigned &n) {
  if (SHORT) {
    n = 2u;
    return (uint64_t)((1u << 4 | (uint32_t)BT_INT8) | (uint32_t)key << 8);
  }
  return enc_int(key, n);
}
/* two typed integers (a key and its value), one store when they fit a word */
template <bool SHORT, class S>
__device__ __forceinline__ void put_int2(S &s, int32_t key, int32_t v) {
  unsigned n1, n2;
  const uint64_t b1 = enc_key<SHORT>(key, n1), b2 = enc_int(v, n2);
  if (n1 + n2 <= 8u)
    s.put_n(b1 | b2 << (8u * n1), n1 + n2);
  else {
    s.put_n(b1, n1);
    s.put_n(b2, n2);
  }
}
/* a typed integer key, then n (<= 6) bytes that are already encoded */
template <bool SHORT, class S>
__device__ __forceinline__ void put_key_then(S &s, int32_t key, uint64_t bits, unsigned n) {
  unsigned n1;
  const uint64_t b1 = enc_key<SHORT>(key, n1);
  if (n1 + n <= 8u)
    s.put_n(b1 | bits << (8u * n1), n1 + n);
  else {
    s.put_n(b1, n1);
    s.put_n(bits, n);
  }
}
template <class S>
__device__ __forceinline__ void put_descriptor(S &s, uint32_t n, int type) {
  if (n >= 15u) {
    s.u8(15u << 4 | (unsigned)type);
    put_int(s, (int32_t)n);
  } else
    s.u8(n << 4 | (unsigned)type);
}

/* the record's fields as the emitter reads them: one 128-byte record in eight 16-byte loads */
struct rec_regs {
  uint32_t w[32];
  __device__ __forceinline__ uint8_t byte(unsigned o) const { return (uint8_t)(w[o >> 2] >> (8u * (o & 3u))); }
};

/* Everything behind the 32 fixed bytes, in the reference's order; returns l_shared (the per-sample block follows it in the sink).
 * id / id_len: the record's name (global memory).  bad: set for a record bsc_bcf_record refuses (gt > 9, n_gl > 6). */
template <bool SHORT, class S>
__device__ __forceinline__ unsigned bcf_emit_body(S &s, const rec_regs &r, const bcf_args &a, const uint8_t *id, unsigned id_len, bool &bad) {
  const unsigned gt_raw = r.byte(5), n_gl_raw = r.byte(10);
  bad = gt_raw > 9u || n_gl_raw > 6u;
  const unsigned gt = gt_raw > 9u ? 9u : gt_raw, n_gl = n_gl_raw > 6u ? 6u : n_gl_raw;
  const unsigned flt = r.byte(8), phred = r.byte(9), gt_enc = r.byte(7);
  const unsigned alt0 = r.byte(12), alt1 = r.byte(13);
  const bool het = (0x16Eu >> gt) & 1u; /* gt_het {0,1,1,1,0,1,1,0,1,0} (src/init_param.c:16) */
  const unsigned s0 = s.len;
  /* ---- shared: ID, REF, ALT, FILTER, INFO CX (:165-221) ---- */
  {
    uint64_t bits;
    unsigned n;
    if (id_len) { /* a dbSNP name: byte by byte from the table */
      put_descriptor(s, id_len, BT_CHAR);
      for (unsigned k = 0; k < id_len; k++) s.u8(id[k]);
      bits = 0ull;
      n = 0u;
    } else { /* no ID: an empty string */
      bits = (uint64_t)(0u << 4 | BT_CHAR);
      n = 1u;
    }
    bits |= (uint64_t)((1u << 4 | BT_CHAR) | (uint32_t)r.byte(16) << 8) << (8u * n); /* REF = cx_ref[2] */
    n += 2u;
    if (alt0) {
      bits |= (uint64_t)((1u << 4 | BT_CHAR) | alt0 << 8) << (8u * n);
      n += 2u;
      if (alt1) {
        bits |= (uint64_t)((1u << 4 | BT_CHAR) | alt1 << 8) << (8u * n);
        n += 2u;
      }
    }
    s.put_n(bits, n); /* <= 7 bytes */
  }
  {
    unsigned n1, n2;
    const uint64_t b1 = enc_key<SHORT>(flt == 0u ? a.ids.pass : ((flt & 128u) ? a.ids.mac1 : a.ids.fail), n1), b2 = enc_key<SHORT>(a.ids.info_cx, n2);
    if (n1 + n2 <= 8u)
      s.put_n(b1 | b2 << (8u * n1), n1 + n2);
    else {
      s.put_n(b1, n1);
      s.put_n(b2, n2);
    }
  }
  /* five characters: bytes 14 .. 18 of the record */
  s.put_n((uint64_t)(5u << 4 | BT_CHAR) | (uint64_t)(r.w[3] >> 16) << 8 | (uint64_t)(r.w[4] & 0xffffffu) << 24, 6u);
  const unsigned l_shared = s.len - s0;
  /* ---- per sample: GT FT DP MQ GQ QD GL MC8 [AMQ] CS CG CX [FS] (:267-378) ---- */
  put_key_then<SHORT>(s, a.ids.fmt_gt, (uint64_t)(2u << 4 | BT_INT8) | (uint64_t)(gt_enc >> 4) << 8 | (uint64_t)(gt_enc & 15u) << 16, 3u); /* two allele codes below 16 */
  if (flt & 15u) { /* each name WITH its terminator, ';' between (:283-296): "q20\0;qd2\0" */
    const unsigned nf = (unsigned)__popc(flt & 15u);
    const unsigned ft_len = ((flt & 1u) ? 4u : 0u) + ((flt & 2u) ? 4u : 0u) + ((flt & 4u) ? 5u : 0u) + ((flt & 8u) ? 5u : 0u) + nf - 1u;
    {
      unsigned n1;
      const uint64_t b1 = enc_key<SHORT>(a.ids.fmt_ft, n1);
      s.put_n(b1, n1);
    }
    put_descriptor(s, ft_len, BT_CHAR);
    bool first = true;
#define FT_NAME(bit, c0, c1, c2, c3, n)                                                                                                   \
  if (flt & (bit)) {                                                                                                                      \
    const uint64_t nm = (uint64_t)(c0) | (uint64_t)(c1) << 8 | (uint64_t)(c2) << 16 | ((n) == 4 ? (uint64_t)(c3) << 24 : 0ull); /* + '\0' */ \
    s.put_n(first ? nm : ((uint64_t)';' | nm << 8), (n) + (first ? 1u : 2u));                                                              \
    first = false;                                                                                                                        \
  }
    FT_NAME(1u, 'q', '2', '0', 0, 3)
    FT_NAME(2u, 'q', 'd', '2', 0, 3)
    FT_NAME(4u, 'f', 's', '6', '0', 4)
    FT_NAME(8u, 'm', 'q', '4', '0', 4)
#undef FT_NAME
  } else
    put_key_then<SHORT>(s, a.ids.fmt_ft, (uint64_t)(4u << 4 | BT_CHAR) | (uint64_t)'P' << 8 | (uint64_t)'A' << 16 | (uint64_t)'S' << 24 | (uint64_t)'S' << 32, 5u);
  put_int2<SHORT>(s, a.ids.fmt_dp, (int32_t)r.w[8]);  /* dp */
  put_int2<SHORT>(s, a.ids.fmt_mq, (int32_t)r.w[26]); /* mq */
  put_int2<SHORT>(s, a.ids.fmt_gq, (int32_t)phred);
  put_int2<SHORT>(s, a.ids.fmt_qd, (int32_t)r.w[7]);  /* qd */
  put_key_then<SHORT>(s, a.ids.fmt_gl, (uint64_t)(n_gl << 4 | BT_FLOAT), 1u);
  s.words6(&r.w[9], n_gl); /* (the dwords behind the n_gl-th are overwritten: at least 27 bytes of this record follow) */
  {
    int32_t lo = (int32_t)r.w[16], hi = lo;
#pragma unroll
    for (int k = 1; k < 8; k++) {
      const int32_t v = (int32_t)r.w[16 + k];
      lo = v < lo ? v : lo;
      hi = v > hi ? v : hi;
    }
    const int t = int_type(lo, hi);
    put_key_then<SHORT>(s, a.ids.fmt_mc8, (uint64_t)(8u << 4 | (unsigned)t), 1u);
    if (t == BT_INT8) { /* eight counts below 128: one word */
      const uint32_t b0 = (r.w[16] & 0xffu) | (r.w[17] & 0xffu) << 8 | (r.w[18] & 0xffu) << 16 | r.w[19] << 24;
      const uint32_t b1 = (r.w[20] & 0xffu) | (r.w[21] & 0xffu) << 8 | (r.w[22] & 0xffu) << 16 | r.w[23] << 24;
      s.put_n((uint64_t)b0 | (uint64_t)b1 << 32, 8u);
    } else if (t == BT_INT16) {
#pragma unroll
      for (int k = 0; k < 8; k += 2) s.put_w((r.w[16 + k] & 0xffffu) | r.w[17 + k] << 16);
    } else {
#pragma unroll
      for (int k = 0; k < 8; k++) s.put_w(r.w[16 + k]);
    }
  }
  {
    unsigned n_amq = 0;
    int32_t lo = 255, hi = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
      if (r.w[16 + k] > 0u) {
        const int32_t q = (int32_t)r.byte(96u + (unsigned)k);
        n_amq++;
        lo = q < lo ? q : lo;
        hi = q > hi ? q : hi;
      }
    if (n_amq) {
      const int t = int_type(lo, hi);
      put_key_then<SHORT>(s, a.ids.fmt_amq, (uint64_t)(n_amq << 4 | (unsigned)t), 1u); /* one element: put_int's descriptor is the same byte */
      if (t == BT_INT8) { /* every byte is stored where the stream stands; it stays if its count is not 0 (a CS key follows: >= 4 bytes) */
#pragma unroll
        for (int k = 0; k < 8; k++) s.cond8(r.byte(96u + (unsigned)k), r.w[16 + k] > 0u);
      } else {
#pragma unroll
        for (int k = 0; k < 8; k++)
          if (r.w[16 + k] > 0u) s.le(r.byte(96u + (unsigned)k), type_bytes(t));
      }
    }
  }
  { /* cs_str {"NA","+","-","NA","+","+-","+","-","-","NA"} (src/print_vcf.c:58-59) */
    const bool na = (0x209u >> gt) & 1u, plus = (0x72u >> gt) & 1u, minus = (0x1A4u >> gt) & 1u;
    const uint64_t str = na ? ((uint64_t)(2u << 4 | BT_CHAR) | (uint64_t)'N' << 8 | (uint64_t)'A' << 16)
                            : ((uint64_t)(((plus ? 1u : 0u) + (minus ? 1u : 0u)) << 4 | BT_CHAR) | (uint64_t)(plus ? '+' : '-') << 8 | (uint64_t)'-' << 16);
    put_key_then<SHORT>(s, a.ids.fmt_cs, str, na ? 3u : 1u + (plus ? 1u : 0u) + (minus ? 1u : 0u));
  }
  put_key_then<SHORT>(s, a.ids.fmt_cg, (uint64_t)(1u << 4 | BT_CHAR) | (uint64_t)r.byte(11) << 8, 2u);
  /* five characters: bytes 19 .. 23 of the record */
  put_key_then<SHORT>(s, a.ids.fmt_cx, (uint64_t)(5u << 4 | BT_CHAR) | (uint64_t)(r.w[4] >> 24) << 8 | (uint64_t)r.w[5] << 16, 6u);
  if (het) put_int2<SHORT>(s, a.ids.fmt_fs, (int32_t)r.w[6]);
  return l_shared;
}

/* the 32 fixed bytes bcf_write puts in front */
__device__ __forceinline__ void bcf_emit_fixed(uint8_t *p, const rec_regs &r, const bcf_args &a, unsigned l_shared, unsigned l_indiv) {
  const unsigned gt = r.byte(5) > 9u ? 9u : r.byte(5);
  const unsigned alt0 = r.byte(12), alt1 = r.byte(13);
  const uint32_t n_allele = 1u + (alt0 ? 1u : 0u) + (alt0 && alt1 ? 1u : 0u);
  bool amq = false;
#pragma unroll
  for (int k = 0; k < 8; k++) amq |= r.w[16 + k] > 0u;
  const uint32_t n_fmt = 11u + (amq ? 1u : 0u) + (((0x16Eu >> gt) & 1u) ? 1u : 0u);
  const float qual = (float)r.byte(9);
  lds_sink f = {p, 0u};
  f.put_w(l_shared + 24u);
  f.put_w(l_indiv);
  f.put_w((uint32_t)a.rid);
  f.put_w(r.w[0] - 1u);
  f.put_w(1u); /* rlen */
  f.put_w(__float_as_uint(qual));
  f.put_w(n_allele << 16 | 1u); /* one INFO field */
  f.put_w(n_fmt << 24 | 1u);    /* one sample */
}

/* record / position i into registers; false: nothing is written for it (emit == 0).  Without the chain's byte (gate < 0) the flag is in
 * the record's first 16 bytes: a position that writes no record costs those (their 64-byte sector), and a record's other loads wait for
 * them.  With it (gate: the byte) a position without a record costs nothing more, and a record's eight loads leave together. */
__device__ __forceinline__ bool load_rec(rec_regs &r, const bcf_args &a, uint64_t i, int gate = -1) {
  if (gate == 0) return false;
  const uint4 *lo = reinterpret_cast<const uint4 *>(a.recs ? a.recs + i * 128u : a.core + i * 64u);
  const uint4 v0 = lo[0];
  r.w[0] = v0.x; r.w[1] = v0.y; r.w[2] = v0.z; r.w[3] = v0.w;
  if (gate < 0 && !(v0.y & 0xffu)) return false; /* bsc_vcf_core.emit */
  const uint4 *hi = a.recs ? lo + 4 : reinterpret_cast<const uint4 *>(a.aux + i * 64u);
#pragma unroll
  for (int k = 1; k < 4; k++) {
    const uint4 v = lo[k];
    r.w[4 * k] = v.x; r.w[4 * k + 1] = v.y; r.w[4 * k + 2] = v.z; r.w[4 * k + 3] = v.w;
  }
#pragma unroll
  for (int k = 0; k < 4; k++) {
    const uint4 v = hi[k];
    r.w[16 + 4 * k] = v.x; r.w[17 + 4 * k] = v.y; r.w[18 + 4 * k] = v.z; r.w[19 + 4 * k] = v.w;
  }
  return true;
}

/* the name of a flagged record: binary search of its position in the block's table */
__device__ __forceinline__ unsigned find_name(const bcf_args &a, const rec_regs &r, const uint8_t *&id) {
  id = nullptr;
  if (!a.n_names || !r.byte(113)) return 0u;
  const uint32_t pos = r.w[0];
  uint32_t lo = 0, hi = a.n_names;
  while (lo < hi) {
    const uint32_t mid = (lo + hi) >> 1;
    if (a.name_pos[mid] < pos) lo = mid + 1u; else hi = mid;
  }
  if (lo >= a.n_names || a.name_pos[lo] != pos) return 0u;
  const uint32_t o0 = a.name_off[lo], o1 = a.name_off[lo + 1u];
  id = a.name_bytes + o0;
  const uint32_t l = o1 - o0;
  return l > BCF_ID_MAX ? BCF_ID_MAX : l;
}

__device__ __forceinline__ uint64_t clamp_n(const bcf_args &a) {
  if (!a.n_recs) return a.max_recs;
  const unsigned long long n = *a.n_recs;
  return n < a.max_recs ? n : a.max_recs;
}

/* length of record i (0 beyond n, 0 for a record that is not written).  gate: the chain's byte of the position, or -1; len_known: a byte
 * of 1 .. 254 IS the length (SHORT indices, and the chain gives 255 to every record with a name or one the encoder refuses) */
template <bool SHORT>
__device__ __forceinline__ unsigned rec_len(const bcf_args &a, uint64_t i, uint64_t n, rec_regs &r, const uint8_t *&id, unsigned &id_len, bool &bad,
                                            int gate = -1, bool len_known = false) {
  bad = false;
  id_len = 0;
  id = nullptr;
  if (i >= n) return 0u;
  if (!load_rec(r, a, i, gate)) return 0u;
  if (len_known && gate > 0 && gate != 255) return (unsigned)gate;
  id_len = find_name(a, r, id);
  count_sink c = {0u};
  (void)bcf_emit_body<SHORT>(c, r, a, id, id_len, bad);
  return 32u + c.len;
}

/* n_tiles = tiles of max_recs; tile_bytes[n_tiles] = 0 (so that the scan's last output is the stream's length); err[0] += records
 * refused, err[1] += records written */
extern "C" __global__ __launch_bounds__(256) void bsc_bcf_size_kernel(bcf_args a, uint32_t n_tiles, unsigned long long *__restrict__ tile_bytes,
                                                                      unsigned long long *__restrict__ err) {
  const unsigned lane = threadIdx.x & 63u;
  const uint64_t n = clamp_n(a);
  if (blockIdx.x == 0 && threadIdx.x == 0) tile_bytes[n_tiles] = 0ull;
  unsigned n_written = 0; /* wave-uniform */
  for (uint32_t tile = blockIdx.x * BCF_WAVES + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * BCF_WAVES) {
    rec_regs r;
    const uint8_t *id;
    unsigned id_len;
    bool bad;
    unsigned len = rec_len<false>(a, (uint64_t)tile * 64u + lane, n, r, id, id_len, bad);
    if (bad) atomicAdd(err, 1ull);
    n_written += (unsigned)__popcll(__ballot(len != 0u));
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) len += __shfl_xor(len, d);
    if (lane == 0) tile_bytes[tile] = len;
  }
  /* totals[2]: once per workgroup (n_written is wave-uniform) — atomics on one word are served one after the other, 23 ns each */
  __shared__ unsigned s_written;
  if (threadIdx.x == 0) s_written = 0;
  __syncthreads();
  if (lane == 0 && n_written) atomicAdd(&s_written, n_written);
  __syncthreads();
  if (threadIdx.x == 0 && s_written) atomicAdd(err + 1, (unsigned long long)s_written);
}

/* The same sums from the chain kernel's byte per position (csrc/fused.hip, emit_out): 0 = no record, 1 .. 254 = the record's length with
 * one-byte dictionary indices and no ID — what the launcher has checked the indices to be —, 255 = look at the record (dbSNP-flagged
 * positions, records longer than 254 bytes).  64 bytes a tile instead of 64 sectors and more. */
extern "C" __global__ __launch_bounds__(256) void bsc_bcf_size_bytes_kernel(bcf_args a, const uint8_t *__restrict__ emit, uint32_t n_tiles,
                                                                            unsigned long long *__restrict__ tile_bytes, unsigned long long *__restrict__ err) {
  /* a THREAD per tile (round 6, second form; the first was a wave per tile, a byte per lane and six shuffles): the tile's 64 bytes are four
   * 16-byte loads of a lane, summed four at a time (v_sad_u8), 4 KB a wave instruction: 30 us per 50 M positions */
  const uint64_t n = clamp_n(a);
  if (blockIdx.x == 0 && threadIdx.x == 0) tile_bytes[n_tiles] = 0ull;
  unsigned n_written = 0;
  for (uint64_t tile = (uint64_t)blockIdx.x * 256u + threadIdx.x; tile < n_tiles; tile += (uint64_t)gridDim.x * 256u) {
    const uint64_t i0 = tile * 64u;
    uint32_t w[16];
    if (i0 + 64u <= n) {
#pragma unroll
      for (int q = 0; q < 4; q++) __builtin_memcpy(&w[4 * q], emit + i0 + 16u * q, 16);
    } else { /* the block's last, ragged tile */
#pragma unroll
      for (int q = 0; q < 16; q++) {
        w[q] = 0u;
        for (int t = 0; t < 4; t++)
          if (i0 + 4u * q + t < n) w[q] |= (uint32_t)emit[i0 + 4u * q + t] << (8 * t);
      }
    }
    uint32_t len = 0, any255 = 0;
#pragma unroll
    for (int q = 0; q < 16; q++) {
      const uint32_t x = w[q];
      len = __builtin_amdgcn_sad_u8(x, 0u, len);
      n_written += (uint32_t)__popc((x | ((x & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u); /* bytes that are not 0 */
      any255 |= x & (x << 1) & (x << 2) & (x << 3) & (x << 4) & (x << 5) & (x << 6) & (x << 7) & 0x80808080u; /* bit 7 of a byte that is all ones */
    }
    if (any255) { /* look at those records: dbSNP-flagged positions, records longer than 254 bytes (the heterozygous calls' bytes are final too since
                     the chain kernel writes them again behind Fisher's test: with every het call a 255, a tile in five had one and this
                     branch was the kernel's whole 0.2 ms) */
      for (int q = 0; q < 16; q++)
        for (int t = 0; t < 4; t++)
          if (((w[q] >> (8 * t)) & 0xffu) == 0xffu) {
            rec_regs r;
            const uint8_t *id;
            unsigned id_len;
            bool bad;
            const unsigned l = rec_len<false>(a, i0 + 4u * q + t, n, r, id, id_len, bad);
            if (bad) atomicAdd(err, 1ull);
            len += l - 255u;
            if (l == 0u) n_written--; /* (a 255 whose record is not written after all: cannot happen, kept consistent) */
          }
    }
    tile_bytes[tile] = len;
  }
  /* totals[2]: once per WORKGROUP, and few workgroups — atomics on one word are served one after the other, 23 ns each: one per wave of
   * 12 288 waves was the whole 0.25 ms of the first form of this kernel */
  __shared__ unsigned s_written;
  if (threadIdx.x == 0) s_written = 0;
  __syncthreads();
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) n_written += __shfl_xor(n_written, d);
  if ((threadIdx.x & 63u) == 0 && n_written) atomicAdd(&s_written, n_written);
  __syncthreads();
  if (threadIdx.x == 0 && s_written) atomicAdd(err + 1, (unsigned long long)s_written);
}

/* SHORT: every dictionary index is 0 .. 127 (bcf_emit_body); IMG: the wave's image of its part of the stream, bytes.  A tile whose span
 * does not fit the image goes out in 2, 4 or 8 parts of 32, 16 or 8 lanes (8 records of the longest kind always fit: IMG >= 8 x 336). */
template <bool SHORT, unsigned IMG, unsigned WPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void bsc_bcf_write_kernel_t(
    bcf_args a, uint32_t n_tiles, const unsigned long long *__restrict__ tile_off, uint8_t *__restrict__ out, uint64_t out_cap,
    unsigned long long *__restrict__ total) {
  static_assert(IMG >= 8u * BCF_REC_MAX && IMG % 16u == 0u, "an eighth of a tile of the longest records must fit the wave's image");
  __shared__ __attribute__((aligned(16))) uint8_t s_img[BCF_WAVES][IMG + 32u]; /* 15 bytes of phase in front, 7 of a last field's excess behind */
  const unsigned lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  uint8_t *const img = s_img[wid];
  const uint64_t n = clamp_n(a);
  if (blockIdx.x == 0 && threadIdx.x == 0) *total = tile_off[n_tiles];
  /* the chain's byte of a position (0: no record; with SHORT indices 1 .. 254: the record's length) is fetched a tile ahead: the record
   * loads then wait for nothing but themselves, and all eight leave together */
  const uint32_t tile0 = blockIdx.x * BCF_WAVES + wid;
  int gate_next = -1;
  if (a.gate && tile0 < n_tiles && (uint64_t)tile0 * 64u + lane < n) gate_next = a.gate[(uint64_t)tile0 * 64u + lane];
  for (uint32_t tile = tile0; tile < n_tiles; tile += gridDim.x * BCF_WAVES) {
    if ((uint64_t)tile * 64u >= n) break; /* wave-uniform; later tiles of this wave lie further out still */
    const int gate = gate_next;
    {
      const uint64_t nt = (uint64_t)tile + (uint64_t)gridDim.x * BCF_WAVES;
      gate_next = -1;
      if (a.gate && nt < n_tiles && nt * 64u + lane < n) gate_next = a.gate[nt * 64u + lane];
    }
    rec_regs r;
    const uint8_t *id;
    unsigned id_len;
    bool bad;
    const unsigned len = rec_len<SHORT>(a, (uint64_t)tile * 64u + lane, n, r, id, id_len, bad, gate, SHORT);
    /* exclusive prefix of the lengths over the wave */
    unsigned inc = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned v = __shfl_up(inc, d);
      if (lane >= (unsigned)d) inc += v;
    }
    const unsigned excl = inc - len;
    const unsigned t_all = (unsigned)__builtin_amdgcn_readlane((int)inc, 63);
    const uint64_t g_tile = tile_off[tile];
    if (g_tile + t_all > out_cap) continue; /* the host reports the overflow from *total */
    /* one part when the tile's span fits the image (the usual case), else the fewest parts of equal lane counts that do */
    unsigned parts = 1u;
    if (t_all > IMG) {
      for (parts = 2u; parts < 8u; parts <<= 1) {
        const unsigned step = 64u / parts;
        bool fits = true;
        unsigned prev = 0u;
        for (unsigned q = 0; q < parts; q++) {
          const unsigned e = (unsigned)__builtin_amdgcn_readlane((int)inc, (int)(step * (q + 1u) - 1u));
          fits = fits && e - prev <= IMG;
          prev = e;
        }
        if (fits) break;
      }
    }
    const unsigned step = 64u / parts;
    unsigned b0 = 0u; /* the part's first byte within the tile */
    for (unsigned ps = 0; ps < parts; ps++) {
      const unsigned b1 = (unsigned)__builtin_amdgcn_readlane((int)inc, (int)(step * (ps + 1u) - 1u)); /* one past its last */
      const bool mine = len && excl >= b0 && excl < b1;
      const uint64_t g0 = g_tile + b0;
      const unsigned ph = (unsigned)(g0 & 15u);
      uint8_t *const p = img + ph + (excl - b0);
      lds_sink w = {p + 32u, 0u};
      unsigned l_shared = 0u;
      if (mine) {
        bool bad2;
        l_shared = bcf_emit_body<SHORT>(w, r, a, id, id_len, bad2);
        /* the length the lane offsets were made of (the chain's byte, or the counting run) is the length written: anything else would be a
         * stream with a hole or an overlap — counted with the refused records, so the block fails instead (never seen; the chain's formula
         * and this emitter are two statements of one rule) */
        if (32u + w.len != len) atomicAdd(total + 1, 1ull);
      }
      /* every body before any of the fixed fields: a body's last store may reach into the next record's first bytes */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (mine) bcf_emit_fixed(p, r, a, l_shared, w.len - l_shared);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      /* the image [ph, ph + t) -> out[g0, g0 + t) */
      const unsigned t = b1 - b0, end = ph + t;
      uint8_t *const dst = out + (g0 - ph);
      const unsigned head_end = ph ? (end < 16u ? end : 16u) : 0u; /* bytes [ph, head_end) singly */
      if (lane >= ph && lane < head_end) dst[lane] = img[lane];
      const unsigned body0 = ph ? 16u : 0u, body1 = end & ~15u;
      for (unsigned o = body0 + 16u * lane; o < body1; o += 1024u) *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(img + o);
      const unsigned tail0 = body1 > head_end ? body1 : head_end;
      if (tail0 + lane < end) dst[tail0 + lane] = img[tail0 + lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      b0 = b1;
    }
  }
}

/*
 * Round 6: the two kernels above read every record twice (sizes, then bytes: 2 x (64 B per position + 64 B per written record) for the
 * per-position form).  ONE kernel with a decoupled look-back (Merrill & Garland's single-pass scan): a wave takes the next tile from a
 * counter (tiles are therefore started in order: a wave never waits for a tile nobody runs), sums its lanes' record lengths, publishes the
 * sum, and finds its place in the stream by walking back over its predecessors' published words — a tile's own sum (flag 1) is added and
 * the walk goes on, a tile's inclusive prefix (flag 2) ends it — then publishes its own inclusive prefix and writes its records as the
 * write kernel does.  A word = flag << 62 | bytes: one 8-byte store publishes both.  The records are read once.
 * The words are published and polled with RELAXED atomic stores and loads at agent scope (they bypass the XCD's L2, nothing else): a word
 * carries everything it has to say — nothing else is ordered by it.  (First form: release stores and acquire loads at agent scope —
 * 31 ms against the two kernels' 0.87 per 10 M records: every publish wrote the XCD's L2 back, every poll invalidated it.)
 */
#define BCF_FLAG_SUM (1ull << 62)
#define BCF_FLAG_PREFIX (2ull << 62)
#define BCF_VAL_MASK ((1ull << 62) - 1ull)
extern "C" __global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(BCF_ONEPASS_WAVES_PER_EU, BCF_ONEPASS_WAVES_PER_EU))) void bsc_bcf_onepass_kernel(
    bcf_args a, uint32_t n_tiles, unsigned long long *__restrict__ state /* [n_tiles], zeroed */, unsigned int *__restrict__ next_tile /* zeroed */,
    uint8_t *__restrict__ out, uint64_t out_cap, unsigned long long *__restrict__ totals /* [0] length, [1] += refused, [2] += written */) {
  __shared__ __attribute__((aligned(16))) uint8_t s_img[BCF_WAVES][BCF_IMG_BYTES + 32u]; /* 15 bytes of phase in front, 7 of a last field's excess behind */
  const unsigned lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  uint8_t *const img = s_img[wid];
  const uint64_t n = clamp_n(a);
  unsigned n_written = 0, n_bad = 0;
  /* Tiles are dealt round-robin over the launch's waves, every wave taking its tiles in rising order: the launch is sized so that all its
   * waves are resident at once, so the tile a wave waits for is in the hands of a running wave.  (Tried: one atomic counter handing out
   * single tiles in order — 157 k claims on one word took 23 ns each, the kernel's whole time; the counter handing out chunks of 32 tiles
   * to a workgroup — a chunk's first tile then waits for ALL of the chunk before it: 318 ms.)  A wave that polls a word far longer than
   * any launch lasts gives up and says so (next_tile[0] = 1: the host runs the two kernels instead) — a launch that was NOT all resident
   * (the device shared with another process) must still drain. */
  bool gave_up = false;
  for (uint32_t tile = blockIdx.x * BCF_WAVES + wid; tile < n_tiles && !gave_up; tile += gridDim.x * BCF_WAVES) {
    rec_regs r;
    const uint8_t *id;
    unsigned id_len;
    bool bad;
    const unsigned len = rec_len<false>(a, (uint64_t)tile * 64u + lane, n, r, id, id_len, bad);
    n_bad += (unsigned)__popcll(__ballot(bad));
    n_written += (unsigned)__popcll(__ballot(len != 0u));
    unsigned inc = len;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const unsigned v = __shfl_up(inc, d);
      if (lane >= (unsigned)d) inc += v;
    }
    const unsigned excl = inc - len;
    const unsigned t_all = __shfl(inc, 63);
    /* the tile's place: its predecessors' bytes */
    unsigned long long before = 0ull;
    if (tile == 0u) {
      if (lane == 0) __hip_atomic_store(&state[0], BCF_FLAG_PREFIX | (unsigned long long)t_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      if (lane == 0) __hip_atomic_store(&state[tile], BCF_FLAG_SUM | (unsigned long long)t_all, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      /* 64 predecessors at a time, lane l looking at tile - 1 - l (- 64 per round): the nearest inclusive prefix ends the walk */
      int64_t base = (int64_t)tile - 1;
      for (;;) {
        const int64_t j = base - (int64_t)lane;
        unsigned long long w = BCF_FLAG_PREFIX; /* before tile 0: a prefix of nothing */
        if (j >= 0) {
          for (unsigned polls = 0;; polls++) {
            w = __hip_atomic_load(&state[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((w >> 62) != 0ull) break;
            if (polls > (1u << 22)) { /* ~ a second */
              w = BCF_FLAG_PREFIX;
              gave_up = true;
              break;
            }
            __builtin_amdgcn_s_sleep(2);
          }
        }
        if (__ballot(gave_up)) {
          gave_up = true;
          if (lane == 0) {
            atomicExch(next_tile, 1u);
            atomicMax(totals, ~0ull); /* a length no buffer holds: the host's check of totals[0] fails the call */
          }
          break;
        }
        const unsigned long long is_prefix = __ballot((w >> 62) == 2ull);
        const unsigned first = (unsigned)__builtin_ctzll(is_prefix ? is_prefix : 1ull); /* the nearest prefix among these 64 */
        unsigned long long v = (is_prefix == 0ull || lane <= first) ? (w & BCF_VAL_MASK) : 0ull;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d);
        before += v;
        if (is_prefix) break;
        base -= 64;
      }
      if (lane == 0) __hip_atomic_store(&state[tile], BCF_FLAG_PREFIX | (before + (unsigned long long)t_all), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (gave_up) {
      if (lane == 0) __hip_atomic_store(&state[tile], BCF_FLAG_PREFIX, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); /* nobody waits for this one in vain */
      break;
    }
    if (tile == n_tiles - 1u && lane == 0) atomicMax(totals, before + (unsigned long long)t_all);
    const uint64_t g_tile = before;
    if ((uint64_t)tile * 64u >= n || g_tile + t_all > out_cap) continue; /* nothing to write / the host reports the overflow from totals[0] */
    const unsigned t_half = __shfl(inc, 31);
    const unsigned passes = t_all <= BCF_IMG_BYTES ? 1u : 2u;
    for (unsigned ps = 0; ps < passes; ps++) {
      const unsigned b0 = ps ? t_half : 0u;
      const unsigned b1 = passes == 1u ? t_all : (ps ? t_all : t_half);
      const bool mine = len && excl >= b0 && excl < b1;
      const uint64_t g0 = g_tile + b0;
      const unsigned ph = (unsigned)(g0 & 15u);
      uint8_t *const p = img + ph + (excl - b0);
      lds_sink w = {p + 32u, 0u};
      unsigned l_shared = 0u;
      if (mine) {
        bool bad2;
        l_shared = bcf_emit_body<false>(w, r, a, id, id_len, bad2);
      }
      /* every body before any of the fixed fields: a body's last store may reach into the next record's first bytes */
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      if (mine) bcf_emit_fixed(p, r, a, l_shared, w.len - l_shared);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const unsigned t = b1 - b0, end = ph + t;
      uint8_t *const dst = out + (g0 - ph);
      const unsigned head_end = ph ? (end < 16u ? end : 16u) : 0u;
      if (lane >= ph && lane < head_end) dst[lane] = img[lane];
      const unsigned body0 = ph ? 16u : 0u, body1 = end & ~15u;
      for (unsigned o = body0 + 16u * lane; o < body1; o += 1024u) *reinterpret_cast<uint4 *>(dst + o) = *reinterpret_cast<const uint4 *>(img + o);
      const unsigned tail0 = body1 > head_end ? body1 : head_end;
      if (tail0 + lane < end) dst[tail0 + lane] = img[tail0 + lane];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
  }
  if (lane == 0) {
    if (n_bad) atomicAdd(totals + 1, (unsigned long long)n_bad);
    if (n_written) atomicAdd(totals + 2, (unsigned long long)n_written);
  }
}

extern "C" int bsc_dev_scan_u64(const void *in, void *out, uint32_t n, void *tmp, size_t tmp_bytes, void *stream); /* sort.hip */

/*
 * recs[<= max_recs] packed records, *n_recs of them (device u64) — or, recs == NULL, core[max_recs] / aux[max_recs] as the reads-in chain
 * leaves them (n_recs may then be NULL: every position) — -> out[<= out_cap] BCF bytes; totals[0] = the stream's length (also when it
 * exceeds out_cap: then only the tiles that fit whole are written), totals[1] += records bsc_bcf_record refuses (and records whose written length is not the length they were placed by: never seen), totals[2] += records
 * written.  tile_bytes / tile_off: max_recs / 64 (rounded up) + 1 u64 each; scan_tmp: bsc_dev_scan_tmp_bytes_u64 of that many.
 */
extern "C" int bsc_dev_launch_bcf(const void *recs, const void *core, const void *aux, const void *n_recs, uint64_t max_recs, int32_t rid,
                                  const bsc_bcf_ids *ids, const void *name_pos, const void *name_off, const void *name_bytes, uint32_t n_names,
                                  void *tile_bytes, void *tile_off, void *scan_tmp, size_t scan_tmp_bytes, void *out, uint64_t out_cap, void *totals,
                                  int num_cus, void *stream, const void *emit_len) {
  hipStream_t s = (hipStream_t)stream;
  bcf_args a;
  a.recs = (const uint8_t *)recs;
  a.core = (const uint8_t *)core;
  a.aux = (const uint8_t *)aux;
  a.n_recs = (const unsigned long long *)n_recs;
  a.max_recs = max_recs;
  a.rid = rid;
  a.ids = *ids;
  a.name_pos = (const uint32_t *)name_pos;
  a.name_off = (const uint32_t *)name_off;
  a.name_bytes = (const uint8_t *)name_bytes;
  a.n_names = name_pos ? n_names : 0u;
  a.gate = nullptr;
  const uint64_t nt64 = (max_recs + 63u) / 64u;
  if (nt64 > 0x7fffffffull) return (int)hipErrorInvalidValue;
  const uint32_t n_tiles = (uint32_t)nt64;
  unsigned grid = (n_tiles + BCF_WAVES - 1u) / BCF_WAVES;
  if (grid > (unsigned)num_cus * 12u) grid = (unsigned)num_cus * 12u;
  if (grid == 0) grid = 1;
  static int one_pass = -1; /* BSC_BCF_ONE_PASS in the environment: the look-back kernel (the A/B of tools/bench_bcf.py; slower, see there) */
  if (one_pass < 0) one_pass = getenv("BSC_BCF_ONE_PASS") != nullptr;
  if (one_pass) {
    if (n_tiles == 0) return (int)hipMemsetAsync(totals, 0, sizeof(unsigned long long), s);
    hipError_t e1 = hipMemsetAsync(tile_off, 0, (size_t)n_tiles * 8u, s);
    if (e1 == hipSuccess) e1 = hipMemsetAsync(tile_bytes, 0, 8, s);
    if (e1 != hipSuccess) return (int)e1;
    /* as many workgroups as are resident at once (the look-back's guarantee of progress) */
    static int per_cu = 0;
    if (!per_cu) {
      int nb = 0;
      const hipError_t eo = hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bsc_bcf_onepass_kernel, 256, 0);
      per_cu = (eo == hipSuccess && nb > 0) ? nb : 1;
      (void)hipGetLastError();
    }
    unsigned g1 = (n_tiles + BCF_WAVES - 1u) / BCF_WAVES;
    const unsigned cap1 = (unsigned)num_cus * (unsigned)per_cu;
    if (g1 > cap1) g1 = cap1;
    hipLaunchKernelGGL(bsc_bcf_onepass_kernel, dim3(g1), dim3(256), 0, s, a, n_tiles, (unsigned long long *)tile_off, (unsigned int *)tile_bytes, (uint8_t *)out, out_cap,
                       (unsigned long long *)totals);
    return (int)hipGetLastError();
  }
  /* the sizes from the chain's length bytes when they can be trusted: per-position form, no names, every dictionary index in one byte */
  bool short_ids = true;
  {
    const int32_t *iv = &ids->pass;
    for (size_t k = 0; k < sizeof(bsc_bcf_ids) / sizeof(int32_t); k++) short_ids = short_ids && iv[k] >= 0 && iv[k] <= 127;
  }
  static int no_len = -1; /* BSC_BCF_NO_LEN_BYTES: the size pass over the records, as in round 5 (A/B) */
  if (no_len < 0) no_len = getenv("BSC_BCF_NO_LEN_BYTES") != nullptr;
  if (emit_len && !recs && short_ids && !a.n_names && !no_len)
  {
    unsigned gb = (n_tiles + 255u) / 256u; /* a thread per tile */
    if (gb > (unsigned)num_cus * 2u) gb = (unsigned)num_cus * 2u; /* (few: each ends in an atomic on one word) */
    if (gb == 0) gb = 1;
    hipLaunchKernelGGL(bsc_bcf_size_bytes_kernel, dim3(gb), dim3(256), 0, s, a, (const uint8_t *)emit_len, n_tiles, (unsigned long long *)tile_bytes,
                       (unsigned long long *)totals + 1);
  }
  else
    hipLaunchKernelGGL(bsc_bcf_size_kernel, dim3(grid), dim3(256), 0, s, a, n_tiles, (unsigned long long *)tile_bytes, (unsigned long long *)totals + 1);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  const int rc = bsc_dev_scan_u64(tile_bytes, tile_off, n_tiles + 1u, scan_tmp, scan_tmp_bytes, stream);
  if (rc) return rc;
  static int no_gate = -1; /* BSC_BCF_NO_GATE: the write kernel finds a position's flag in its record, as before the gate (A/B) */
  if (no_gate < 0) no_gate = getenv("BSC_BCF_NO_GATE") != nullptr;
  if (emit_len && !recs && !no_len && !no_gate) a.gate = (const uint8_t *)emit_len; /* not 0 <=> bsc_vcf_core.emit (fused.hip: ebyte) */
#define BCF_LAUNCH_WRITE(SH, IMG, WPE)                                                                                                      \
  hipLaunchKernelGGL((bsc_bcf_write_kernel_t<SH, IMG, WPE>), dim3(grid), dim3(256), 0, s, a, n_tiles, (const unsigned long long *)tile_off, \
                     (uint8_t *)out, out_cap, (unsigned long long *)totals)
  if (a.gate) {
    grid = (n_tiles + BCF_WAVES - 1u) / BCF_WAVES;
    if (grid > (unsigned)num_cus * 4u * BCF_WPE_SITES) grid = (unsigned)num_cus * 4u * BCF_WPE_SITES;
    if (grid == 0) grid = 1;
    if (short_ids)
      BCF_LAUNCH_WRITE(true, BCF_IMG_SITES, BCF_WPE_SITES);
    else
      BCF_LAUNCH_WRITE(false, BCF_IMG_SITES, 3); /* (wide dictionary indices: the general emitter needs 168 registers to stay out of scratch) */
  } else if (short_ids)
    BCF_LAUNCH_WRITE(true, BCF_IMG_PACKED, BCF_WPE_PACKED);
  else
    BCF_LAUNCH_WRITE(false, BCF_IMG_PACKED, 3);
#undef BCF_LAUNCH_WRITE
  return (int)hipGetLastError();
}
