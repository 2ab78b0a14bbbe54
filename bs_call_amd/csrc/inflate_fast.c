/*
 * inflate_fast.c — raw DEFLATE (RFC 1951) decoding and CRC-32 for the BGZF blocks of the device reader's host half (csrc/bamstream.c).
 * The host's share of a file-to-file run IS the inflation (the container's CPU share, 16 cores, inflates 5.5 GB/s through zlib: 0.5 s of
 * the 0.98 s a 50 Mb contig at 30x takes), so the routine is written for that one job: a whole block in, a whole block out, both in memory.
 *
 *   bsc_inflate_raw   one call per BGZF block: `in` = the deflate stream, `out` = exactly the block's ISIZE bytes.  A 64-bit bit buffer
 *                     refilled with one unaligned 8-byte load; literal / length symbols through an 11-bit table (a literal, or a length
 *                     with its extra bits' count, in one entry; longer codes through second-level tables behind it), distances through an
 *                     8-bit table; up to three literals per refill; matches copied 8 bytes at a time (short distances: the pattern
 *                     widened first).  The fast loop runs while 8 input bytes and 258 + 8 output bytes of slack remain, a careful loop
 *                     finishes.  Returns 0, or -1 for a stream that is not valid DEFLATE or does not produce exactly out_len bytes.
 *   bsc_crc32         zlib's CRC-32 (RFC 1952 section 8): carry-less multiplication where the CPU has it, slicing-by-8 tables otherwise.
 *
 * Both are checked against zlib on random and adversarial streams (tests/test_inflate_fast.py: every zlib level and strategy, stored and
 * fixed blocks, truncated and damaged streams); csrc/bamio.c — the checker of the device reader — keeps zlib.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/bscall_amd.h"

/* ---- CRC-32 ------------------------------------------------------------------------------------------------------------------ */
static uint32_t crc_tab[8][256];
static volatile int crc_ready;
static void crc_init(void) {
  uint32_t t[8][256];
  for (uint32_t i = 0; i < 256; i++) {
    uint32_t c = i;
    for (int k = 0; k < 8; k++) c = (c >> 1) ^ ((c & 1u) ? 0xedb88320u : 0u);
    t[0][i] = c;
  }
  for (uint32_t i = 0; i < 256; i++)
    for (int s = 1; s < 8; s++) t[s][i] = (t[s - 1][i] >> 8) ^ t[0][t[s - 1][i] & 0xffu];
  memcpy(crc_tab, t, sizeof t);
  __atomic_store_n(&crc_ready, 1, __ATOMIC_RELEASE);
}
static uint32_t crc_slice(uint32_t c, const uint8_t *p, size_t n) { /* the register's update over n bytes (no inversions) */
  while (n && ((uintptr_t)p & 7u)) {
    c = (c >> 8) ^ crc_tab[0][(c ^ *p++) & 0xffu];
    n--;
  }
  while (n >= 8) {
    uint64_t v;
    memcpy(&v, p, 8);
    v ^= c;
    c = crc_tab[7][v & 0xffu] ^ crc_tab[6][(v >> 8) & 0xffu] ^ crc_tab[5][(v >> 16) & 0xffu] ^ crc_tab[4][(v >> 24) & 0xffu] ^ crc_tab[3][(v >> 32) & 0xffu] ^
        crc_tab[2][(v >> 40) & 0xffu] ^ crc_tab[1][(v >> 48) & 0xffu] ^ crc_tab[0][v >> 56];
    p += 8;
    n -= 8;
  }
  while (n--) c = (c >> 8) ^ crc_tab[0][(c ^ *p++) & 0xffu];
  return c;
}

#if defined(__x86_64__)
#include <immintrin.h>
/* Carry-less multiplication (Gopal et al., "Fast CRC computation for generic polynomials using PCLMULQDQ"): four 16-byte lanes folded
 * 64 bytes ahead with x^(512 +- 32) mod P, then into one with x^(128 +- 32) mod P (the constants of the reflected CRC-32, as Linux's
 * crc32-pclmul uses them); the last 128 bits are a 16-byte message with the same remainder — finished by the table, like the tail.
 * 20 GB/s a core where the tables do 2.8: a BGZF block's checksum stops being a fifth of its inflation. */
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc_clmul(uint32_t c, const uint8_t *p, size_t n) {
  __m128i x1 = _mm_loadu_si128((const __m128i *)p), x2 = _mm_loadu_si128((const __m128i *)(p + 16)), x3 = _mm_loadu_si128((const __m128i *)(p + 32)),
          x4 = _mm_loadu_si128((const __m128i *)(p + 48));
  x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)c));
  p += 64;
  n -= 64;
  const __m128i k512 = _mm_set_epi64x(0x1c6e41596ll, 0x154442bd4ll), k128 = _mm_set_epi64x(0x0ccaa009ell, 0x1751997d0ll);
#define CRC_FOLD(x, k, d)                                  \
  do {                                                     \
    const __m128i t_ = _mm_clmulepi64_si128((x), (k), 0x00); \
    (x) = _mm_clmulepi64_si128((x), (k), 0x11);            \
    (x) = _mm_xor_si128(_mm_xor_si128((x), t_), (d));      \
  } while (0)
  while (n >= 64) {
    CRC_FOLD(x1, k512, _mm_loadu_si128((const __m128i *)p));
    CRC_FOLD(x2, k512, _mm_loadu_si128((const __m128i *)(p + 16)));
    CRC_FOLD(x3, k512, _mm_loadu_si128((const __m128i *)(p + 32)));
    CRC_FOLD(x4, k512, _mm_loadu_si128((const __m128i *)(p + 48)));
    p += 64;
    n -= 64;
  }
  CRC_FOLD(x1, k128, x2);
  CRC_FOLD(x1, k128, x3);
  CRC_FOLD(x1, k128, x4);
  while (n >= 16) {
    CRC_FOLD(x1, k128, _mm_loadu_si128((const __m128i *)p));
    p += 16;
    n -= 16;
  }
#undef CRC_FOLD
  uint8_t acc[16];
  _mm_storeu_si128((__m128i *)acc, x1);
  return crc_slice(crc_slice(0u, acc, 16), p, n);
}
#endif

uint32_t bsc_crc32(const uint8_t *p, size_t n) {
  if (!__atomic_load_n(&crc_ready, __ATOMIC_ACQUIRE)) crc_init(); /* (idempotent: two threads at once write the same values) */
#if defined(__x86_64__)
  static int have_clmul = -1;
  if (have_clmul < 0) have_clmul = __builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") ? 1 : 0;
  if (have_clmul && n >= 64) return ~crc_clmul(0xffffffffu, p, n);
#endif
  return ~crc_slice(0xffffffffu, p, n);
}

/* ---- decode tables ------------------------------------------------------------------------------------------------------------- */
#define LL_BITS 11u
#define D_BITS 8u
#define LL_SIZE (2048u + 2048u) /* primary + room for the second-level tables of a complete code of <= 288 symbols of <= 15 bits */
#define D_SIZE (256u + 1024u)
/* an entry, 32 bits (the literal / length table is 16 KB: half the L1): bits 0-3 the code's length (the bits it consumes at this level),
 * bits 4-6 kind: 0 literal, 1 length / distance (extra-bit count in bits 8-11, base in bits 16..), 2 end of block, 3 link to a second-level
 * table (its index in bits 16.., its width in 8-11), 4 invalid */
typedef uint32_t ent_t;
#define E_LEN(e) ((unsigned)((e) & 15u))
#define E_KIND(e) ((unsigned)(((e) >> 4) & 7u))
#define E_XBITS(e) ((unsigned)(((e) >> 8) & 15u))
#define E_VAL(e) ((uint32_t)((e) >> 16))
#define E_MAKE(val, xb, kind, len) ((ent_t)(val) << 16 | (ent_t)(xb) << 8 | (ent_t)(kind) << 4 | (ent_t)(len))
enum { K_LIT = 0, K_BASE = 1, K_EOB = 2, K_LINK = 3, K_BAD = 4 };

static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t len_xbits[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t dist_xbits[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

static inline ent_t sym_entry(int is_dist, unsigned sym, unsigned len) { /* is_dist 2: the code-length code's plain symbols */
  if (is_dist == 2) return E_MAKE(sym, 0, K_LIT, len);
  if (is_dist) {
    if (sym >= 30) return E_MAKE(0, 0, K_BAD, len);
    return E_MAKE(dist_base[sym], dist_xbits[sym], K_BASE, len);
  }
  if (sym < 256) return E_MAKE(sym, 0, K_LIT, len);
  if (sym == 256) return E_MAKE(0, 0, K_EOB, len);
  if (sym >= 286) return E_MAKE(0, 0, K_BAD, len);
  return E_MAKE(len_base[sym - 257], len_xbits[sym - 257], K_BASE, len);
}

static inline unsigned rev_bits(unsigned v, unsigned n) { /* the low n bits of v, reversed */
  unsigned r = 0;
  for (unsigned i = 0; i < n; i++) r |= ((v >> i) & 1u) << (n - 1u - i);
  return r;
}

/* canonical Huffman code of lens[0 .. n) (0 = unused) -> a table of `bits` primary bits (+ second levels); 0, or -1 for a code zlib's
 * inflate_table refuses: over-subscribed, or incomplete — unless it has no code at all, or (not the code-length code) one code of one bit. */
static int build_table(const uint8_t *lens, unsigned n, int is_dist, unsigned bits, ent_t *tab, unsigned tab_cap) {
  unsigned count[16] = {0}, next[16];
  for (unsigned i = 0; i < n; i++) count[lens[i]]++;
  count[0] = 0;
  unsigned code = 0, left = 1, max = 0;
  for (unsigned l = 1; l <= 15; l++) {
    left <<= 1;
    if (count[l] > left) return -1;
    left -= count[l];
    code = (code + count[l - 1]) << 1;
    next[l] = code;
    if (count[l]) max = l;
  }
  if (max && left > 0 && (is_dist == 2 || max != 1)) return -1;
  const unsigned psize = 1u << bits;
  for (unsigned i = 0; i < psize; i++) tab[i] = E_MAKE(0, 0, K_BAD, 1);
  unsigned used = psize;
  /* short codes: every primary slot whose low `l` bits are the (bit-reversed) code */
  for (unsigned s = 0; s < n; s++) {
    const unsigned l = lens[s];
    if (!l || l > bits) continue;
    const unsigned r = rev_bits(next[l]++, l);
    const ent_t e = sym_entry(is_dist, s, l);
    for (unsigned i = r; i < psize; i += 1u << l) tab[i] = e;
  }
  /* long codes, by their first `bits` bits: one second-level table per prefix, as wide as the prefix's longest code needs */
  {
    /* widths first */
    unsigned char width[1u << 11];
    unsigned have_long = 0;
    memset(width, 0, psize);
    unsigned nx[16];
    { /* recompute the first codes: next[] was advanced for the short ones only, which is what the long ones continue from */
      unsigned c = 0;
      for (unsigned l = 1; l <= 15; l++) {
        c = (c + count[l - 1]) << 1;
        nx[l] = c;
      }
    }
    unsigned cur[16];
    memcpy(cur, nx, sizeof cur);
    for (unsigned s = 0; s < n; s++) {
      const unsigned l = lens[s];
      if (!l) continue;
      const unsigned c = cur[l]++;
      if (l <= bits) continue;
      const unsigned r = rev_bits(c, l), pre = r & (psize - 1u);
      if (l - bits > width[pre]) width[pre] = (unsigned char)(l - bits);
      have_long = 1;
    }
    if (have_long) {
      unsigned base[1u << 11];
      for (unsigned p = 0; p < psize; p++)
        if (width[p]) {
          if (used + (1u << width[p]) > tab_cap) return -1;
          base[p] = used;
          for (unsigned i = 0; i < (1u << width[p]); i++) tab[used + i] = E_MAKE(0, 0, K_BAD, 1);
          tab[p] = E_MAKE(used, width[p], K_LINK, bits);
          used += 1u << width[p];
        }
      memcpy(cur, nx, sizeof cur);
      for (unsigned s = 0; s < n; s++) {
        const unsigned l = lens[s];
        if (!l) continue;
        const unsigned c = cur[l]++;
        if (l <= bits) continue;
        const unsigned r = rev_bits(c, l), pre = r & (psize - 1u), sub = r >> bits, w = width[pre];
        const ent_t e = sym_entry(is_dist, s, l - bits);
        for (unsigned i = sub; i < (1u << w); i += 1u << (l - bits)) tab[base[pre] + i] = e;
      }
    }
  }
  return 0;
}

typedef struct {
  ent_t ll[LL_SIZE], d[D_SIZE];
} tables;

static tables fixed_tab;
static volatile int fixed_ready;
static void fixed_init(void) {
  static tables t;
  uint8_t l[288];
  for (int i = 0; i < 144; i++) l[i] = 8;
  for (int i = 144; i < 256; i++) l[i] = 9;
  for (int i = 256; i < 280; i++) l[i] = 7;
  for (int i = 280; i < 288; i++) l[i] = 8;
  build_table(l, 288, 0, LL_BITS, t.ll, LL_SIZE);
  for (int i = 0; i < 32; i++) l[i] = 5; /* thirty distance codes and two that are never valid: a complete 5-bit code */
  build_table(l, 32, 1, D_BITS, t.d, D_SIZE);
  memcpy(&fixed_tab, &t, sizeof t);
  __atomic_store_n(&fixed_ready, 1, __ATOMIC_RELEASE);
}

/* ---- the decoder ----------------------------------------------------------------------------------------------------------------- */
#define REFILL()                                         \
  do {                                                   \
    if (ip + 8 <= iend) {                                \
      uint64_t v_;                                       \
      memcpy(&v_, ip, 8);                                \
      bb |= v_ << bn;                                    \
      ip += (63u - bn) >> 3;                             \
      bn |= 56u;                                         \
    } else {                                             \
      while (bn <= 56u && ip < iend) {                   \
        bb |= (uint64_t)*ip++ << bn;                     \
        bn += 8u;                                        \
      }                                                  \
    }                                                    \
  } while (0)
#define TAKE(n_) (bb >>= (n_), bn -= (n_))

int bsc_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len) {
  const uint8_t *ip = in, *const iend = in + in_len;
  uint8_t *op = out, *const oend = out + out_len;
  uint64_t bb = 0;
  unsigned bn = 0;
  static __thread tables dyn_tls; /* a dynamic block's tables (30 KB): per thread, not per call */
  tables *const dyn = &dyn_tls;
  int rc = -1, last;
  do {
    REFILL();
    if (bn < 3) goto done;
    last = (int)(bb & 1u);
    const unsigned type = (unsigned)(bb >> 1) & 3u;
    TAKE(3);
    if (type == 0) { /* stored: to the byte boundary, LEN, NLEN, the bytes */
      TAKE(bn & 7u);
      /* give the whole bytes in the bit buffer back to the input */
      ip -= bn >> 3;
      bb = 0;
      bn = 0;
      if (iend - ip < 4) goto done;
      const unsigned len = ip[0] | (unsigned)ip[1] << 8, nlen = ip[2] | (unsigned)ip[3] << 8;
      ip += 4;
      if ((len ^ 0xffffu) != nlen || (size_t)(iend - ip) < len || (size_t)(oend - op) < len) goto done;
      memcpy(op, ip, len);
      op += len;
      ip += len;
      continue;
    }
    const tables *T;
    if (type == 1) {
      if (!__atomic_load_n(&fixed_ready, __ATOMIC_ACQUIRE)) fixed_init();
      T = &fixed_tab;
    } else if (type == 2) {
      REFILL();
      if (bn < 14) goto done;
      const unsigned hlit = (unsigned)(bb & 31u) + 257u, hdist = (unsigned)((bb >> 5) & 31u) + 1u, hclen = (unsigned)((bb >> 10) & 15u) + 4u;
      TAKE(14);
      if (hlit > 286 || hdist > 30) goto done;
      static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
      uint8_t cl[19] = {0}, lens[320];
      for (unsigned i = 0; i < hclen; i++) {
        REFILL();
        if (bn < 3) goto done;
        cl[order[i]] = (uint8_t)(bb & 7u);
        TAKE(3);
      }
      ent_t ct[128 + 64];
      if (build_table(cl, 19, 2, 7, ct, 128)) goto done; /* (kind 2: plain symbols — see below) */
      unsigned n = 0;
      while (n < hlit + hdist) {
        REFILL();
        const ent_t e = ct[bb & 127u];
        if (E_KIND(e) == K_BAD || E_LEN(e) > bn) goto done;
        TAKE(E_LEN(e));
        const unsigned sym = E_VAL(e);
        if (sym < 16) lens[n++] = (uint8_t)sym;
        else {
          unsigned rep, val = 0;
          if (sym == 16) {
            if (!n || bn < 2) goto done;
            val = lens[n - 1];
            rep = 3u + (unsigned)(bb & 3u);
            TAKE(2);
          } else if (sym == 17) {
            if (bn < 3) goto done;
            rep = 3u + (unsigned)(bb & 7u);
            TAKE(3);
          } else {
            if (bn < 7) goto done;
            rep = 11u + (unsigned)(bb & 127u);
            TAKE(7);
          }
          if (n + rep > hlit + hdist) goto done;
          memset(lens + n, (int)val, rep);
          n += rep;
        }
      }
      if (lens[256] == 0) goto done; /* no end-of-block code */
      if (build_table(lens, hlit, 0, LL_BITS, dyn->ll, LL_SIZE) || build_table(lens + hlit, hdist, 1, D_BITS, dyn->d, D_SIZE)) goto done;
      T = dyn;
    } else
      goto done;
    /* ---- the block's symbols ---- */
    for (;;) {
      /* fast: 8 bytes of input to load from, room for a longest match plus a wide copy's overshoot */
      while (iend - ip >= 16 && oend - op >= 258 + 16) {
        REFILL(); /* >= 56 bits */
        ent_t e = T->ll[bb & ((1u << LL_BITS) - 1u)];
        if (E_KIND(e) == K_LIT) { /* up to three literals on one refill: 3 x 15 < 56 */
          TAKE(E_LEN(e));
          *op++ = (uint8_t)E_VAL(e);
          e = T->ll[bb & ((1u << LL_BITS) - 1u)];
          if (E_KIND(e) == K_LIT) {
            TAKE(E_LEN(e));
            *op++ = (uint8_t)E_VAL(e);
            e = T->ll[bb & ((1u << LL_BITS) - 1u)];
            if (E_KIND(e) == K_LIT) {
              TAKE(E_LEN(e));
              *op++ = (uint8_t)E_VAL(e);
            }
          }
          continue; /* whatever follows the literals is looked at behind the next refill */
        }
        if (E_KIND(e) == K_LINK) {
          const unsigned w = E_XBITS(e);
          TAKE(E_LEN(e));
          e = T->ll[E_VAL(e) + (unsigned)(bb & ((1u << w) - 1u))];
          if (E_KIND(e) == K_LIT) {
            TAKE(E_LEN(e));
            *op++ = (uint8_t)E_VAL(e);
            continue;
          }
        }
        if (E_KIND(e) == K_EOB) {
          TAKE(E_LEN(e));
          goto block_done;
        }
        if (E_KIND(e) != K_BASE) goto done;
        TAKE(E_LEN(e));
        unsigned len = E_VAL(e) + (unsigned)(bb & ((1u << E_XBITS(e)) - 1u));
        TAKE(E_XBITS(e));
        if (bn < 32u) REFILL(); /* a distance needs up to 15 + 13 bits */
        ent_t de = T->d[bb & ((1u << D_BITS) - 1u)];
        if (E_KIND(de) == K_LINK) {
          const unsigned w = E_XBITS(de);
          TAKE(E_LEN(de));
          de = T->d[E_VAL(de) + (unsigned)(bb & ((1u << w) - 1u))];
        }
        if (E_KIND(de) != K_BASE) goto done;
        TAKE(E_LEN(de));
        const unsigned dist = E_VAL(de) + (unsigned)(bb & ((1u << E_XBITS(de)) - 1u));
        TAKE(E_XBITS(de));
        if (dist > (size_t)(op - out)) goto done;
        const uint8_t *src = op - dist;
        uint8_t *const end = op + len;
        if (dist >= 8) {
          do {
            uint64_t v;
            memcpy(&v, src, 8);
            memcpy(op, &v, 8);
            src += 8;
            op += 8;
          } while (op < end);
        } else if (dist == 1) {
          memset(op, *src, len);
        } else { /* 2 .. 7: byte by byte for the first 8, then the widened pattern repeats at a distance that is a multiple of dist >= 8 */
          do *op++ = *src++;
          while (op < end);
        }
        op = end;
      }
      /* careful: a symbol at a time, every bound checked */
      REFILL();
      ent_t e = T->ll[bb & ((1u << LL_BITS) - 1u)];
      if (E_KIND(e) == K_LINK) {
        if (E_LEN(e) > bn) goto done;
        const unsigned w = E_XBITS(e);
        TAKE(E_LEN(e));
        e = T->ll[E_VAL(e) + (unsigned)(bb & ((1u << w) - 1u))];
      }
      if (E_KIND(e) == K_BAD || E_LEN(e) > bn) goto done;
      TAKE(E_LEN(e));
      if (E_KIND(e) == K_LIT) {
        if (op >= oend) goto done;
        *op++ = (uint8_t)E_VAL(e);
        continue;
      }
      if (E_KIND(e) == K_EOB) goto block_done;
      if (E_XBITS(e) > bn) goto done;
      unsigned len = E_VAL(e) + (unsigned)(bb & ((1u << E_XBITS(e)) - 1u));
      TAKE(E_XBITS(e));
      REFILL();
      ent_t de = T->d[bb & ((1u << D_BITS) - 1u)];
      if (E_KIND(de) == K_LINK) {
        if (E_LEN(de) > bn) goto done;
        const unsigned w = E_XBITS(de);
        TAKE(E_LEN(de));
        de = T->d[E_VAL(de) + (unsigned)(bb & ((1u << w) - 1u))];
      }
      if (E_KIND(de) != K_BASE || E_LEN(de) + E_XBITS(de) > bn) goto done;
      TAKE(E_LEN(de));
      const unsigned dist = E_VAL(de) + (unsigned)(bb & ((1u << E_XBITS(de)) - 1u));
      TAKE(E_XBITS(de));
      if (dist > (size_t)(op - out) || len > (size_t)(oend - op)) goto done;
      for (const uint8_t *src = op - dist; len; len--) *op++ = *src++;
    }
  block_done:;
  } while (!last);
  rc = op == oend ? 0 : -1;
done:
  return rc;
}
