/*
 * prepdev.hip — read pre-processing ON THE DEVICE: from the templates the reader hands over (reads as
 * base | qual << 2 bytes with their CIGAR-derived mismatch lists) to the templates the pile-up accumulate stage consumes — what the
 * reference's process thread does to every template of a block before it calls call_genotypes_ML
 * (src/process_template.c:36-111): trim_read (src/read_utils.c:13-26), trim_soft_clips (src/al_utils.c:122-162), handle_overlap
 * (src/al_utils.c:164-318), the indel normalisation (src/process_template.c:62-108).  The host form is csrc/prep.c; the bar is
 * byte equality with it (and with the tests' pure-Python restatement), quirks included (tests/test_gpu_prep.py).
 *
 * The host form edits every read in place (memmove).  Here nothing moves until the end:
 *   bsc_prep_plan_kernel   one thread per template.  The reads' bytes are not touched except for the mean qualities that break a
 *                          tie between mates of equal span.  What the trims, clips and the overlap do to a read is a WINDOW of it
 *                          (first byte, length) and an edited copy of its mismatch list; what the normalisation does is, per list
 *                          entry, where it cuts or pads (ix1, src/process_template.c:93).  Out: the plan of both reads, the output
 *                          lengths (for the prefix sum that places the reads in the output buffer), the template's positions.
 *                          Per read a 16-byte descriptor (where its bytes start, length, left mark, flags); the full 56-byte plan
 *                          only for the reads that need it (the list cuts or pads, a right trim, very short / very long).
 *   bsc_prep_copy_kernel   one wave per 64 reads at a time: lane r fetches read r's descriptor and place and does the per-read
 *                          bookkeeping; then the reads whose output is their window as it stands (16 .. 128 bytes: nearly all) move EIGHT
 *                          A ROUND, sixteen bytes a lane — one byte-unaligned 16-byte load and store per lane, what a round needs of a
 *                          read fetched from the read's own lane by ds_bpermute, two rounds' loads in flight ahead of the rounds being
 *                          stored (round 6; round 5: one read a round, four bytes a lane); the left trim's mark (quality 63) and the
 *                          base counters of the statistics (:50-59) are byte-parallel arithmetic on the dwords, bsc_template.flags (was
 *                          read 0 walked, src/call_genotypes.c:198-211) a bit per round OR-ed over a read's lanes.  The other reads
 *                          follow one at a time: longer / shorter / marked ones four bytes a lane, edited ones byte by byte — output
 *                          byte j from the inverse of the list's edits (walked backwards, a padded deletion gives 0, everything else
 *                          an index into the window) through the fixed trims' marks (the right trim takes the BASE of the byte
 *                          mirrored at the read's other end, as the reference's loop does).
 *   bsc_prep_refmask_kernel / bsc_prep_profile_kernel (round 6)   the non-CpG read profile as a pass of its own over the prepared bytes.
 * Measured (tools/bench_prep.py, 50 Mb at 30x, 7.5 M templates, 1.48 G bases, every 50th read with an indel pair): plan 0.22 ms,
 * prefix sum 0.11 ms, copy 1.11 ms (2 x 1.5 GB of read bytes + 0.5 GB of descriptors and places at 3.8 TB/s), with the read profile
 * + 0.04 + 0.74 ms (profiles/r06_prep_profile_kernels_timed_3.txt).
 * The round-5 form it replaces — one wave per template, its plan through the scalar cache, a byte per lane — took 0.9 + 6.4 ms:
 * profiles/r05_prep.json.
 * Where the reference aborts (an illegal soft clip ...) the lowest offending template and its first failing check come back
 * through one atomicMin; the host entry runs csrc/prep.c on that one template for the message.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bscall_amd.h"

#define FLT_QUAL 63u

/* what the plan kernel leaves per read (56 bytes) */
struct bsc_prep_plan {
  uint64_t src;     /* offset of the ORIGINAL read's first byte in seq */
  uint64_t ms;      /* index of its edited mismatch list in ms_work */
  uint32_t rl0;     /* the original read's length */
  uint32_t w0, wl;  /* the window of it that is left: first byte, length */
  uint32_t nm;      /* entries of the edited list; their `position` is ix1 (src/process_template.c:93) */
  uint32_t mark_l, mark_r; /* fixed trims: this many bytes from either end of the original read carry quality 63 */
  uint32_t out_len;
  uint32_t present; /* the reader delivered a vector for this read (t->len[k] != 0): it counts in filter_cts (:57-58) */
  uint32_t trim_l, trim_r; /* bases cut from either end by the soft clips and the overlap (src/al_utils.c:309-313): the read profile's
                              positions in the ORIGINAL read (src/process_template.c:76-87) */
};

/* what the copy kernel needs of a read whose bytes are its window as it stands (16 bytes): almost every read.  The full plan
 * is written — and read — only for the others (PD_SLOW). */
struct bsc_prep_desc {
  uint64_t srcw; /* offset in seq of output byte 0: src + w0 (a read that goes the long way: unused) */
  uint32_t pk;   /* out_len (20 bits; a longer read: 0xfffff and PD_SLOW) | left mark << 20 (8 bits: output bytes below it carry the
                    left trim's quality 63) | PD_* << 28 */
  int32_t pc;    /* the read profile: window byte s lies at pc + s (read 0), pc - s (read 1, counted from its far end:
                    src/process_template.c:76-87) of the ORIGINAL read */
};
#define PD_PRESENT 1u
#define PD_EDITED 2u /* the edited list holds an INS or a DEL: the inverse mapping runs */
#define PD_SLOW 8u   /* byte by byte from the full plan in plan[]: an edited read, one with a right trim (mirrored bases), one
                        shorter than 4 or longer than 2^20 - 2 bytes, one marked further than 255 bytes in */

#define PREP_CNT_WORDS 8u  /* cnt[]: [0] lowest error, [1] base_clip, [2] base_overlap (both: see the slots), [3..7] the copy kernel's sums */
#define PREP_CNT_SLOTS 64u /* ... followed by PREP_CNT_SLOTS x 8 words: per slot {base_clip, base_overlap} of the plan kernel's waves */
/* error codes, in the order csrc/prep.c makes its checks; the low byte of the word the kernels atomicMin */
#define PE_ORI 1u
#define PE_READ 2u  /* + k */
#define PE_LIST 4u  /* + k */
#define PE_SOFT_POS 6u
#define PE_SOFT_ILL 7u
#define PE_INDEL 8u
#define PE_CAP 9u   /* the output buffer is too small */
#define PE_PROF_CAP 10u   /* a read position beyond the profile (the host form checks this first) */
#define PE_PROF_RANGE 11u /* a read outside the profile's reference */

struct prep_rd {
  uint32_t w0, wl;
};
__device__ static __forceinline__ void d_left_trim(prep_rd &r, uint32_t l) { /* src/al_utils.c:103-113 */
  if (l > 0) {
    if (l >= r.wl) r.wl = 0;
    else {
      r.w0 += l;
      r.wl -= l;
    }
  }
}
__device__ static __forceinline__ void d_right_trim(prep_rd &r, uint32_t l) { /* src/al_utils.c:115-120 */
  if (l > 0) {
    if (l >= r.wl) r.wl = 0;
    else r.wl -= l;
  }
}

/* quality of byte i of the original read after the fixed trims */
__device__ static __forceinline__ uint32_t d_qual(const uint8_t *sp, uint32_t i, uint32_t rl0, uint32_t mark_l, uint32_t mark_r) {
  return (i < mark_l || i >= rl0 - mark_r) ? FLT_QUAL : (uint32_t)sp[i] >> 2;
}

__device__ static __forceinline__ uint32_t d_ld32(const uint8_t *p);
/* src/al_utils.c:191-203 over the read's window: the mean of the qualities that are not 63.  The bytes the fixed trims mark carry 63, so
 * what is summed lies in [max(w0, mark_l), min(w0 + wl, rl0 - mark_r)); four bytes at a time — one unaligned dword load, the qualities
 * side by side in the dword (6 bits each: bit 7 of a byte is free for "is not 63"), sixteen a load — since overlapping mates of equal span, the common
 * pair of a short-insert library, come through here twice per template: byte by byte the plan kernel took 17 ms per 7.5 M such
 * templates (profiles/r05_prep_overlap.txt). */
__device__ static uint32_t d_mean_qual(const uint8_t *sp, const prep_rd &r, uint32_t rl0, uint32_t mark_l, uint32_t mark_r) {
  const uint32_t lo = r.w0 > mark_l ? r.w0 : mark_l;
  const uint32_t e0 = r.w0 + r.wl, e1 = rl0 - mark_r; /* (unsigned, as d_qual's own test) */
  const uint32_t hi = e0 < e1 ? e0 : e1;
  uint32_t tot = 0, n = 0, i = lo;
#define MQ_DWORD(w_)                                                                                                                   \
  {                                                                                                                                    \
    const uint32_t q = ((w_) >> 2) & 0x3f3f3f3fu;                                                                                      \
    const uint32_t nz = ((q ^ 0x3f3f3f3fu) + 0x7f7f7f7fu) & 0x80808080u; /* bit 7 of a byte: its quality is not 63 (no carry leaves a byte) */ \
    n += (uint32_t)__popc(nz);                                                                                                         \
    tot = __builtin_amdgcn_sad_u8(q & ((nz >> 7) * 0xffu), 0u, tot); /* + the bytes that count */                                      \
  }
  for (; i + 16u <= hi; i += 16u) { /* one (unaligned) 16-byte load: a lane's requests are what the pass is short of, each lane reads its own read */
    uint4 v;
    __builtin_memcpy(&v, sp + i, 16);
    MQ_DWORD(v.x) MQ_DWORD(v.y) MQ_DWORD(v.z) MQ_DWORD(v.w)
  }
  for (; i + 4u <= hi; i += 4u) MQ_DWORD(d_ld32(sp + i))
#undef MQ_DWORD
  for (; i < hi; i++) {
    const uint32_t q = (uint32_t)sp[i] >> 2;
    if (q != FLT_QUAL) {
      tot += q;
      n++;
    }
  }
  return n > 0 ? tot / n : 0;
}

__global__ __launch_bounds__(256) void bsc_prep_plan_kernel(const bsc_raw_template *__restrict__ raw, uint32_t nr, const uint8_t *__restrict__ seq,
                                                            uint64_t seq_bytes, const bsc_misms *__restrict__ misms_in, uint64_t n_misms_in,
                                                            bsc_prep_params par, bsc_misms *__restrict__ ms_work,
                                                            bsc_prep_plan *__restrict__ plan, bsc_prep_desc *__restrict__ desc,
                                                            unsigned long long *__restrict__ out_len,
                                                            bsc_template *__restrict__ tpl_out, unsigned long long *__restrict__ cnt,
                                                            uint32_t *__restrict__ max_pos1) {
  const uint32_t ti = blockIdx.x * 256u + threadIdx.x;
  unsigned long long n_clip = 0, n_overlap = 0;
  uint32_t err = 0;
  if (ti < nr) {
    const bsc_raw_template t = raw[ti];
    bsc_prep_plan P[2];
    prep_rd rd[2];
    uint32_t nm[2];
    bsc_misms *ms[2];
    bool edited[2] = {false, false};
    uint32_t pos[2] = {t.pos[0], t.pos[1]};
    if (t.orientation > 1) err = PE_ORI;
#pragma unroll
    for (int k = 0; k < 2 && !err; k++) {
      if (t.len[k] && (t.off[k] > seq_bytes || t.len[k] > seq_bytes - t.off[k])) err = PE_READ + (uint32_t)k;
      else if (t.n_misms[k] && (t.misms_off[k] > n_misms_in || t.n_misms[k] > n_misms_in - t.misms_off[k])) err = PE_LIST + (uint32_t)k;
    }
    if (!err) {
      /* 1. fixed trims (src/process_template.c:39-41): read[0] is R1 on a FORWARD template, R2 on a REVERSE one — as marks */
      const int msk = t.orientation == 0 ? 0 : 1;
#pragma unroll
      for (int k = 0; k < 2; k++) {
        const int side = k ^ msk; /* rd[k] is trimmed with left_trim[side], right_trim[side] */
        const uint32_t rl = t.len[k];
        const int32_t lt = side ? par.left_trim[1] : par.left_trim[0], rt = side ? par.right_trim[1] : par.right_trim[0];
        P[k].src = t.off[k];
        P[k].ms = t.misms_off[k];
        P[k].rl0 = rl;
        P[k].mark_l = lt > 0 ? ((uint32_t)lt < rl ? (uint32_t)lt : rl) : 0u;
        P[k].mark_r = rt > 0 ? ((uint32_t)rt < rl ? (uint32_t)rt : rl) : 0u;
        P[k].present = rl != 0;
        P[k].trim_l = P[k].trim_r = 0;
        rd[k].w0 = 0;
        rd[k].wl = rl;
        nm[k] = t.n_misms[k];
        ms[k] = ms_work + t.misms_off[k];
        for (uint32_t z = 0; z < nm[k]; z++) ms[k][z] = misms_in[t.misms_off[k] + z];
      }
      /* 2. soft clips (src/al_utils.c:122-162) */
#pragma unroll
      for (int k = 0; k < 2 && !err; k++) {
        const uint32_t rl = rd[k].wl;
        if (rl == 0) continue;
        int nclip = 0;
        uint32_t adj = 0;
        const uint32_t n0 = nm[k];
        for (uint32_t z = 0; z < n0; z++) {
          bsc_misms *m = ms[k] + z;
          if (m->type == BSC_MISMS_SOFT) {
            if (z && z != n0 - 1) { err = PE_SOFT_POS; break; }
            nclip++;
            if (!m->position) {
              if (m->size >= rl) { err = PE_SOFT_ILL; break; }
              adj = m->size;
              n_clip += adj;
              P[k].trim_l = adj;
              d_left_trim(rd[k], adj);
            } else {
              if (m->position + m->size != rl) { err = PE_SOFT_ILL; break; }
              d_right_trim(rd[k], m->size);
              P[k].trim_r = m->size;
              n_clip += m->size;
            }
          } else if (nclip) {
            m->position -= adj;
            ms[k][z - (uint32_t)nclip] = *m;
          }
        }
        if (nclip) nm[k] -= (uint32_t)nclip;
      }
    }
    if (!err) {
      /* 3. mate overlap (src/al_utils.c:164-318) */
      const uint32_t rdl[2] = {rd[0].wl, rd[1].wl};
      if (rdl[0] > 0 && rdl[1] > 0) {
        int rev;
        int32_t overlap;
        if (pos[0] <= pos[1]) {
          overlap = (int32_t)(t.reference_span[0] - pos[1] + pos[0]);
          rev = 0;
        } else {
          overlap = (int32_t)(t.reference_span[1] + pos[1] - pos[0]);
          rev = 1;
        }
        if (pos[0] + t.reference_span[0] >= pos[1]) {
          int tr; /* the read that is trimmed */
          if (t.reference_span[0] > t.reference_span[1]) tr = 1;
          else if (t.reference_span[0] < t.reference_span[1]) tr = 0;
          else
            tr = d_mean_qual(seq + P[0].src, rd[0], P[0].rl0, P[0].mark_l, P[0].mark_r) <=
                         d_mean_qual(seq + P[1].src, rd[1], P[1].rl0, P[1].mark_l, P[1].mark_r)
                     ? 0
                     : 1;
          const int right = (rev && tr) || !(rev || tr); /* trim the right end of read tr; else its left end */
          if (!right) { /* a left trim moves the read's start */
            if (tr) pos[1] += (uint32_t)overlap;
            else pos[0] += (uint32_t)overlap;
          }
          /* (the trimmed read's state in scalars: arrays indexed by `tr` would live in scratch memory) */
          bsc_misms *mm = tr ? ms[1] : ms[0];
          prep_rd rdt = tr ? rd[1] : rd[0];
          const uint32_t rdl_t = tr ? rdl[1] : rdl[0], span_t = tr ? t.reference_span[1] : t.reference_span[0];
          uint32_t num = tr ? nm[1] : nm[0];
          if (!num) {
            if (right) d_right_trim(rdt, (uint32_t)overlap);
            else d_left_trim(rdt, (uint32_t)overlap);
          } else {
            int trimmed = 0;
            if (right) {
              const uint32_t xx = span_t - (uint32_t)overlap;
              int64_t adj = 0;
              for (uint32_t z = 0; z < num; z++) {
                bsc_misms *m = mm + z;
                if ((int64_t)m->position + adj >= (int64_t)xx) {
                  const int64_t trim = (int64_t)rdl_t - xx + adj;
                  d_right_trim(rdt, (uint32_t)trim);
                  num = z;
                  trimmed = 1;
                  break;
                }
                if (m->type == BSC_MISMS_INS) {
                  if ((int64_t)m->position + adj + m->size >= (int64_t)xx) {
                    const int64_t trim = (int64_t)rdl_t - m->position;
                    m->size = (uint32_t)((int64_t)xx - ((int64_t)m->position + adj));
                    d_right_trim(rdt, (uint32_t)trim);
                    num = z + 1;
                    trimmed = 1; /* (no break in the reference: the walk goes on over the shortened list) */
                  }
                  adj += m->size;
                } else if (m->type == BSC_MISMS_DEL) adj -= m->size;
              }
              if (!trimmed) d_right_trim(rdt, (uint32_t)overlap);
            } else {
              const uint32_t xx = (uint32_t)overlap;
              int64_t adj = 0;
              uint32_t z;
              for (z = 0; z < num; z++) {
                bsc_misms *m = mm + z;
                if ((int64_t)m->position + adj >= (int64_t)xx) {
                  const uint32_t trim = (uint32_t)((int64_t)overlap - adj);
                  d_left_trim(rdt, trim);
                  trimmed = 1;
                  if (z) {
                    for (uint32_t z1 = z; z1 < num; z1++) {
                      mm[z1].position -= trim;
                      const bsc_misms a = mm[z1], b = mm[z1 - z];
                      mm[z1 - z] = a;
                      mm[z1] = b;
                    }
                    num -= z;
                  } else {
                    for (uint32_t z1 = 0; z1 < num; z1++) mm[z1].position -= trim;
                  }
                  break;
                }
                if (m->type == BSC_MISMS_INS) {
                  if ((int64_t)m->position + adj + m->size >= (int64_t)xx) {
                    m->size = (uint32_t)((int64_t)m->position + m->size + adj - xx);
                    const uint32_t trim = m->position;
                    d_left_trim(rdt, trim);
                    trimmed = 1;
                    const uint32_t z2 = m->size ? z : z + 1;
                    for (uint32_t z1 = z2; z1 < num; z1++) {
                      mm[z1].position -= trim;
                      if (z2) {
                        const bsc_misms a = mm[z1], b = mm[z1 - z2];
                        mm[z1 - z2] = a;
                        mm[z1] = b;
                      }
                    }
                    num -= z2;
                    break;
                  }
                  adj += m->size;
                } else if (m->type == BSC_MISMS_DEL) adj -= m->size;
              }
              if (!trimmed) {
                d_left_trim(rdt, (uint32_t)((int64_t)overlap - adj));
                num = 0;
              }
            }
          }
          if (tr) {
            nm[1] = num;
            rd[1] = rdt;
          } else {
            nm[0] = num;
            rd[0] = rdt;
          }
          n_overlap += (rdl[0] - rd[0].wl) + (rdl[1] - rd[1].wl);
          const uint32_t cut = tr ? rdl[1] - rd[1].wl : rdl[0] - rd[0].wl; /* src/al_utils.c:309-313 */
          if (right) {
            if (tr) P[1].trim_r += cut;
            else P[0].trim_r += cut;
          } else {
            if (tr) P[1].trim_l += cut;
            else P[0].trim_l += cut;
          }
        }
      }
      /* 4. indel normalisation (src/process_template.c:62-108): where every entry cuts or pads — ix1 — and the length that results */
#pragma unroll
      for (int k = 0; k < 2 && !err; k++) {
        const uint32_t rl = rd[k].wl;
        uint32_t adj = 0;
        for (uint32_t z = 0; z < nm[k]; z++) {
          bsc_misms *m = ms[k] + z;
          const uint32_t ix1 = m->position + adj;
          edited[k] |= m->type == BSC_MISMS_INS || m->type == BSC_MISMS_DEL;
          if (m->type == BSC_MISMS_INS) {
            if (ix1 > rl + adj) { err = PE_INDEL; break; }
            adj += m->size;
          } else if (m->type == BSC_MISMS_DEL) {
            if ((uint64_t)ix1 + m->size > (uint64_t)rl + adj) { err = PE_INDEL; break; }
            adj -= m->size;
          }
          m->position = ix1;
        }
        P[k].w0 = rd[k].w0;
        P[k].wl = rl;
        P[k].nm = nm[k];
        P[k].out_len = rl + adj;
      }
    }
    if (err) {
#pragma unroll
      for (int k = 0; k < 2; k++) {
        P[k].src = P[k].ms = 0;
        P[k].rl0 = P[k].w0 = P[k].wl = P[k].nm = P[k].mark_l = P[k].mark_r = P[k].out_len = P[k].present = P[k].trim_l = P[k].trim_r = 0;
      }
      atomicMin(&cnt[0], ((unsigned long long)ti << 8) | err);
      n_clip = n_overlap = 0;
    }
#pragma unroll
    for (int k = 0; k < 2; k++) {
      bsc_prep_desc d;
      const uint32_t ol = P[k].out_len;
      d.srcw = P[k].src + P[k].w0;
      const uint32_t ml = P[k].mark_l > P[k].w0 ? (P[k].mark_l - P[k].w0 < ol ? P[k].mark_l - P[k].w0 : ol) : 0u;
      const uint32_t keep = P[k].rl0 - P[k].mark_r; /* original bytes from here on are the right trim's */
      const uint32_t hr = keep > P[k].w0 ? (keep - P[k].w0 < ol ? keep - P[k].w0 : ol) : 0u;
      const bool ed = edited[k] && !err;
      const bool slow = ol != 0 && (ed || hr < ol || ol < 4u || ol >= 0xfffffu || ml >= 256u);
      d.pk = (ol < 0xfffffu ? ol : 0xfffffu) | ((ml & 0xffu) << 20) |
             (((P[k].present ? PD_PRESENT : 0u) | (ed ? PD_EDITED : 0u) | (slow ? PD_SLOW : 0u)) << 28);
      d.pc = k ? (int32_t)(P[k].wl + P[k].trim_r) - 1 : (int32_t)P[k].trim_l;
      desc[2u * ti + (uint32_t)k] = d;
      if (slow) plan[2u * ti + (uint32_t)k] = P[k];
    }
    if (max_pos1) { /* the read profile's vector must reach the template's last read position (src/process_template.c:76-89) */
      int32_t max_pos = 0;
#pragma unroll
      for (int k = 0; k < 2; k++)
        if (P[k].present) {
          const int32_t mpos = k ? (int32_t)(P[k].wl + P[k].trim_r) - 1 : (int32_t)(P[k].trim_l + P[k].wl);
          if (mpos > max_pos) max_pos = mpos;
        }
      max_pos1[ti] = err ? 0u : (uint32_t)max_pos + 1u;
    }
    out_len[2u * ti] = P[0].out_len;
    out_len[2u * ti + 1u] = P[1].out_len;
    bsc_template o;
    o.pos[0] = pos[0];
    o.pos[1] = pos[1];
    o.len[0] = P[0].out_len;
    o.len[1] = P[1].out_len;
    o.off[0] = o.off[1] = 0; /* the copy kernel knows where the reads land */
    o.mapq[0] = t.mapq[0];
    o.mapq[1] = t.mapq[1];
    o.orientation = t.orientation;
    o.bs_strand = t.bs_strand;
    o.flags = 0;
    tpl_out[ti] = o;
  }
  /* base_clip, base_overlap: wave sums, one atomic each */
  for (int o = 32; o > 0; o >>= 1) {
    n_clip += __shfl_xor(n_clip, o);
    n_overlap += __shfl_xor(n_overlap, o);
  }
  if ((threadIdx.x & 63u) == 0) {
    /* 64 slots, a 64-byte line each, behind the eight words the kernels share (the host adds them up): with overlapping mates every
     * wave has a sum to add, and 117 k atomics on ONE word are 0.7 ms at the memory side (profiles/r05_prep_overlap.txt) */
    unsigned long long *slot = cnt + PREP_CNT_WORDS + (blockIdx.x & (PREP_CNT_SLOTS - 1u)) * 8u;
    if (n_clip) atomicAdd(&slot[0], n_clip);
    if (n_overlap) atomicAdd(&slot[1], n_overlap);
  }
}

/* byte i (0 <= i < rl0) of the original read after the fixed trims (src/read_utils.c:13-26): the left trim keeps the base; the
 * right trim's loop writes sp[rl - 1 - k1] = base of sp[k1] — the mirrored byte, unless that one was itself overwritten earlier
 * in the loop (the trim reaches past the middle), in which case the byte gets its own base back */
__device__ static __forceinline__ uint32_t d_marked(const uint8_t *sp, uint32_t i, uint32_t rl0, uint32_t mark_l, uint32_t mark_r) {
  const bool in_r = i >= rl0 - mark_r;
  const uint32_t b = sp[(in_r && 2u * i > rl0 - 1u) ? rl0 - 1u - i : i];
  return (i < mark_l || in_r) ? (b & 3u) | (FLT_QUAL << 2) : (uint32_t)sp[i];
}

/*
 * The non-CpG read profile (meth_profile, src/meth_profile.c:48-77; host form: csrc/prep.c profile_read): for every prepared base,
 * by its position in the ORIGINAL read, whether it is a C / G of the reference outside a CpG and what the read shows there.  The
 * reference walks a read with a state of two reference codes; per base that is a function of the three codes around it —
 * mask(before) from (ref[v - 1], ref[v]), mask(after) from (ref[v], ref[v + 1]) — with v = the base's index in the block's codes,
 * except for a read that starts at the block's first position, whose walk starts one code late (v one lower, codes in front of
 * the block reading 0).  The vector of counts grows to the largest read position seen so far, and growing clears everything behind
 * its old end: a count is kept iff its index is inside the vector as the template's own growth left it (used_scan: the running
 * maximum over the templates in order).
 */
struct bsc_prep_prof {
  const uint8_t *ref;            /* codes of x .. x + n_ref - 1 (device) */
  uint32_t x, n_ref, cap, used0; /* the profile's capacity and its length before this call */
  const uint32_t *used_scan;     /* per template: max over the templates up to it of (last read position + 1) */
  unsigned long long *table;     /* [cap][4], zeroed by the caller: this call's counts */
};
__device__ static const uint8_t PROF_REF[64] = { /* src/meth_profile.c:14-23: 4 = C not followed by G, 8 = G not preceded by C */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 8, 0, 0, 0, 0, 0, 4, 4, 0, 4, 0, 0, 0, 0, 0, 0, 8, 0, 0, 0, 0,
    0, 0, 0, 8, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
__device__ static __forceinline__ uint32_t d_profile_base(uint32_t bs_strand, uint32_t c) { /* src/init_param.c:57-70 */
  const uint32_t q = c >> 2;
  if (q < 20u /* MIN_QUAL */ || q >= FLT_QUAL || bs_strand > 2u) return 0u;
  /* {11, 6, 10, 7}, {11, 4, 10, 5}, {9, 6, 8, 7}: a nibble each */
  return ((bs_strand == 0 ? 0x7A6Bu : (bs_strand == 1 ? 0x5A4Bu : 0x7869u)) >> (4u * (c & 3u))) & 15u;
}

#define PREP_WAVES 4
#define PREP_BATCH 4u  /* templates a wave places at a time: the loads of their eight reads are in flight together */
#define PROF_LDS 512u  /* read positions whose counts a workgroup keeps in LDS */

/* one prepared base for the read profile; s = its index in the window, v = its index in F.ref (see above) */
__device__ static __forceinline__ void d_prof_base(const bsc_prep_prof &F, uint32_t *s_prof, uint32_t strand, uint32_t used_t, int64_t v,
                                                   int32_t orig, uint32_t byte) {
  const uint32_t xx = d_profile_base(strand, byte);
  if (xx && v + 1 < (int64_t)F.n_ref) {
    const uint32_t ra = v >= 1 ? F.ref[v - 1] : 0u, rb = v >= 0 ? F.ref[v] : 0u, rc = F.ref[v + 1];
    const uint32_t m_before = PROF_REF[((ra << 3) | rb) & 63u], m_after = PROF_REF[((rb << 3) | rc) & 63u];
    if ((((xx & m_after) | ((xx & m_before) >> 1)) >> 2) & 1u) {
      const uint32_t ix = (uint32_t)(orig + 1);
      if (ix < used_t && ix < F.cap) {
        if (ix < PROF_LDS) atomicAdd(&s_prof[(xx & 3u) * PROF_LDS + ix], 1u); /* [column][position]: neighbours in a read, neighbours in the banks */
        else atomicAdd(&F.table[(uint64_t)ix * 4u + (xx & 3u)], 1ull);
      }
    }
  }
}

/*
 * One wave per 64 reads (32 templates) at a time.  Lane r fetches read r's descriptor and place (coalesced; streaming them through
 * the scalar cache instead holds the whole kernel to 0.5 TB/s) and does the per-read bookkeeping.  The reads then move eight a round,
 * sixteen bytes a lane (the loop's own comment in bsc_prep_copy_kernel); the base counters are byte-parallel arithmetic on the dwords
 * (qualities are 6 bits: bit 6 / bit 7 of every byte are free for the carries) and per-lane population counts.  Reads longer than 128
 * or shorter than 16 bytes and reads a left trim marks move one at a time, four bytes a lane, their descriptors broadcast by
 * v_readlane; reads the list cuts or pads, reads with a right trim (mirrored bases), reads shorter than four bytes and reads that do
 * not fit the output go the long way, byte by byte (prep_slow_read).
 */
__device__ static __forceinline__ uint32_t d_bcast(uint32_t v, uint32_t l) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)l); }
__device__ static __forceinline__ unsigned long long d_bcast64(unsigned long long v, uint32_t l) {
  return (unsigned long long)d_bcast((uint32_t)v, l) | ((unsigned long long)d_bcast((uint32_t)(v >> 32), l) << 32);
}
__device__ static __forceinline__ uint32_t d_ld32(const uint8_t *p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);
  return v;
}
__device__ static __forceinline__ void d_st32(uint8_t *p, uint32_t v) { __builtin_memcpy(p, &v, 4); }

struct prep_slow_ret {
  uint32_t c63, cge; /* this lane's bytes of quality 63 / of quality >= mq among those that count */
  uint32_t walked;
};

/* output byte j of a read that goes the long way -> its index in the read's window, or `pad` (a padded deletion: no byte of the read):
 * the list's edits undone, last first — INS (a deletion from the reference) padded `size` zeros in at ix1, DEL (an insertion) cut `size`
 * bytes out at ix1 */
__device__ static __forceinline__ uint32_t prep_slow_src(const bsc_prep_plan &P, const bsc_misms *__restrict__ ms, bool edited, uint32_t j, bool &pad) {
  uint32_t s = j;
  pad = false;
  if (edited) {
    for (uint32_t z = P.nm; z-- > 0;) {
      const bsc_misms m = ms[z];
      if (m.type == BSC_MISMS_INS) {
        if (s >= m.position) {
          if (s - m.position < m.size) {
            pad = true;
            break;
          }
          s -= m.size;
        }
      } else if (m.type == BSC_MISMS_DEL) {
        if (s >= m.position) s += m.size;
      }
    }
  }
  return s;
}

__device__ __noinline__ static prep_slow_ret prep_slow_read(const bsc_prep_plan *__restrict__ plan_r, uint32_t flags, uint8_t *__restrict__ dp,
                                                            const uint8_t *__restrict__ seq, const bsc_misms *__restrict__ ms_work, uint32_t mq) {
  const unsigned lane = threadIdx.x & 63u;
  prep_slow_ret r = {0u, 0u, 0u};
  const bsc_prep_plan P = *plan_r;
  const uint8_t *const sp = seq + P.src;
  const bool edited = (flags & PD_EDITED) != 0;
  if (edited) { /* the base counters run over the window, before the normalisation (src/process_template.c:50-59) */
    for (uint32_t s0 = 0; s0 < P.wl; s0 += 64u) {
      const uint32_t s = s0 + lane;
      if (s < P.wl) {
        const uint32_t q = d_marked(sp, P.w0 + s, P.rl0, P.mark_l, P.mark_r) >> 2;
        r.c63 += q == FLT_QUAL;
        r.cge += q >= mq;
      }
    }
  }
  const bsc_misms *const ms = ms_work + P.ms;
  bool walked = false;
  for (uint32_t j0 = 0; j0 < P.out_len; j0 += 64u) {
    const uint32_t j = j0 + lane;
    if (j < P.out_len) {
      bool pad;
      const uint32_t s = prep_slow_src(P, ms, edited, j, pad);
      const uint32_t byte = pad ? 0u : d_marked(sp, P.w0 + s, P.rl0, P.mark_l, P.mark_r);
      const uint32_t q = byte >> 2;
      if (!edited) { /* nothing cut or padded: the output IS the window, counted as it passes */
        r.c63 += q == FLT_QUAL;
        r.cge += q >= mq;
      }
      walked |= q != 0 && q != FLT_QUAL;
      dp[j] = (uint8_t)byte; /* (a read that does not fit the output never gets here) */
    }
  }
  r.walked = __any(walked) ? 1u : 0u;
  return r;
}

__global__ __launch_bounds__(64 * PREP_WAVES) void bsc_prep_copy_kernel(const bsc_prep_plan *__restrict__ plan,
                                                                        const bsc_prep_desc *__restrict__ desc, uint32_t nr,
                                                                        const uint8_t *__restrict__ seq, const bsc_misms *__restrict__ ms_work,
                                                                        const unsigned long long *__restrict__ out_off, int32_t min_qual,
                                                                        bsc_template *__restrict__ tpl_out, uint8_t *__restrict__ seq_out,
                                                                        uint64_t seq_out_cap, unsigned long long *__restrict__ cnt) {
  const unsigned lane = threadIdx.x & 63u, lane4 = lane * 4u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * PREP_WAVES + (threadIdx.x >> 6)));
  const uint32_t n_waves = gridDim.x * PREP_WAVES;
  const uint32_t mq = min_qual < 0 ? 0u : (min_qual > 64 ? 64u : (uint32_t)min_qual), mq4 = mq * 0x01010101u;
  /* per lane: bytes that count, those of quality 63, those of quality >= mq; reads, their bases */
  unsigned long long l_total = 0, l_bases = 0;
  uint32_t c63 = 0, cge = 0, l_reads = 0;
  const uint32_t n_reads2 = 2u * nr, n_groups = (n_reads2 + 63u) / 64u;
  for (uint32_t g = wave; g < n_groups; g += n_waves) {
    const uint32_t g0 = g * 64u, n = n_reads2 - g0 < 64u ? n_reads2 - g0 : 64u;
    /* ---- lane r: read g0 + r ---- */
    const uint32_t ri = g0 + (lane < n ? lane : 0u), my_ti = ri >> 1, my_k = ri & 1u;
    bsc_prep_desc L = desc[ri];
    const unsigned long long l_off = out_off[ri];
    if (lane >= n) L.pk = 0;
    if (lane < n) {
      const uint32_t fl = L.pk >> 28;
      uint32_t ol = L.pk & 0xfffffu, wl = ol; /* the read's length as it goes out, its window's */
      if (fl & PD_SLOW) {
        ol = plan[ri].out_len;
        wl = plan[ri].wl;
      }
      tpl_out[my_ti].off[my_k] = l_off;
      if (l_off > seq_out_cap || seq_out_cap - l_off < ol) { /* does not fit: refused, and not written at all */
        atomicMin(&cnt[0], ((unsigned long long)my_ti << 8) | PE_CAP);
        L.pk &= 0xf0000000u;
      }
      l_total += (fl & PD_EDITED) ? wl : ol;
      if (fl & PD_PRESENT) {
        l_reads++;
        l_bases += wl;
      }
    }
    /* ---- what the walk needs of a read, made here once per lane so that the walk itself is short: every instruction of the
     * walk — vector or scalar — is an issue slot of the wave, and the issue slots are what this kernel runs out of ----
     * l_ctl: length (20 bits) | left mark << 20 (8 bits) | kind << 28 | edited << 30;  kind 0: nothing to do,
     * KIND_LEAN: the window as it stands, at most 256 bytes, no mark, KIND_FAST: the window, any length, marked below ml,
     * KIND_SLOW: byte by byte from the full plan */
    enum : uint32_t { KIND_LEAN = 1u, KIND_FAST = 2u, KIND_SLOW = 3u };
    const uint32_t my_len = L.pk & 0xfffffu, my_ml = (L.pk >> 20) & 0xffu;
    const uint32_t my_kind = my_len == 0 ? 0u : ((L.pk >> 28) & PD_SLOW ? KIND_SLOW : ((my_len <= 256u && my_ml == 0) ? KIND_LEAN : KIND_FAST));
    const uint32_t l_ctl = (L.pk & 0x0fffffffu) | (my_kind << 28) | (((L.pk >> 28) & PD_EDITED) ? 1u << 30 : 0u);
    const bool my_fetch = my_kind == KIND_LEAN || my_kind == KIND_FAST; /* (then my_len >= 4) */
    /* where the read's bytes start, as an offset from desc[] (a pointer the compiler knows to be global memory; a broadcast
     * pointer would be a generic one); a read that is not fetched: desc[] itself, four bytes that exist */
    const unsigned long long l_src = my_fetch ? (unsigned long long)(seq + L.srcw) - (unsigned long long)desc : 0ull;
    unsigned long long walked_mask = 0; /* bit r: read r (an even one: a template's read 0) was walked */
    /* EIGHT READS A ROUND, sixteen bytes a lane (round 6, second form of this loop; before: one read a round, four bytes a lane — 25 of 64
     * lanes carrying a 100-base read's bytes, 52 instructions a read, and the issue slots were what the kernel ran out of): lanes 8 s .. 8 s + 7
     * take read 8 s + i in round i, lane h of them bytes 16 h .. 16 h + 15 with ONE byte-unaligned 16-byte load and store — the last lane of
     * a length that is no multiple of sixteen moves the read's last sixteen bytes, overlapping its neighbour (same values twice; the bytes
     * it shares are masked out of the counters).  For the reads that move as they stand and are 16 .. 128 bytes long; the others (longer,
     * shorter, marked by a left trim, or byte by byte from their plan) follow one at a time, the whole wave on each.  What a round needs of a
     * read comes from the read's own lane by ds_bpermute: length | its place from the group's lowest << 8, and where its bytes start from
     * the group's lowest source.  R rounds' loads are asked for a batch ahead of the batch being counted and stored (two register buffers in
     * turn, no branch around a load). */
    const unsigned long long obase = d_bcast64(l_off, 0);                       /* (the places ascend with the reads) */
    const unsigned long long m_w0 = __ballot(my_kind == KIND_LEAN && my_len >= 16u && my_len <= 128u);
    const unsigned long long sbase = m_w0 ? d_bcast64(L.srcw, (uint32_t)__builtin_ctzll(m_w0)) : 0ull;
    const unsigned long long orel = l_off - obase, srel = L.srcw - sbase;
    const bool my_wide = my_kind == KIND_LEAN && my_len >= 16u && my_len <= 128u && orel < (1ull << 24) && srel < (1ull << 31);
    const uint32_t w_a = my_wide ? my_len | ((uint32_t)orel << 8) : 0u, w_s = my_wide ? (uint32_t)srel : 0u;
    const uint32_t l16 = (lane & 7u) * 16u, sub = lane & 56u;
    const uint8_t *const sp_w = seq + sbase;
    uint8_t *const dp_w = seq_out + obase;
    uint32_t walk_bits = 0; /* bit i: this lane saw a byte of its round-i read with a quality that is neither 0 nor 63 */
    constexpr uint32_t RW = 2u; /* rounds in a batch */
    auto fetch8 = [&](uint32_t i0, uint4(&vv)[RW]) {
#pragma unroll
      for (uint32_t i = 0; i < RW; i++) {
        const int src = (int)(sub | ((i0 + i) & 7u));
        const uint32_t len = (uint32_t)__shfl((int)w_a, src) & 0xffu, so = (uint32_t)__shfl((int)w_s, src);
        const uint32_t last = len ? len - 16u : 0u;
        __builtin_memcpy(&vv[i], sp_w + (so + (l16 < last ? l16 : last)), 16); /* (idle lanes fetch the read's last sixteen bytes; no read: the group's first) */
      }
    };
    auto process8 = [&](const uint32_t i0, const uint4(&v)[RW]) {
#pragma unroll
      for (uint32_t i = 0; i < RW; i++) {
        const uint32_t l = i0 + i;
        const int src = (int)(sub | l);
        const uint32_t a_ = (uint32_t)__shfl((int)w_a, src), len = a_ & 0xffu;
        const bool act = l16 < len;
        const uint32_t o = l16 < len - 16u ? l16 : len - 16u, dup = act ? l16 - o : 16u; /* bytes this lane shares with its neighbour: not counted twice */
        const uint32_t w4[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        uint32_t any = 0;
#pragma unroll
        for (uint32_t d = 0; d < 4u; d++) {
          /* bytes whose index in the sixteen is below dup: out (an idle lane: all of them) — what is left in their place is quality 0,
           * base 0: neither 63 nor walked nor (mq > 0) >= mq */
          const uint32_t ge = (((0x83828180u + 0x04040404u * d) - dup * 0x01010101u) >> 7) & 0x01010101u; /* 1: index >= dup */
          const uint32_t wc = w4[d] & (ge * 0xffu);
          const uint32_t qh = ((wc >> 2) & 0x3f3f3f3fu) | 0x80808080u; /* 0x80 + quality, byte by byte */
          const uint32_t q1 = qh + 0x01010101u;                        /* bit 6: quality 63; bits 1..5 clear: quality 0 or 63 */
          c63 += (uint32_t)__popc(q1 & 0x40404040u);
          cge += (uint32_t)__popc((qh - mq4) & 0x80808080u);           /* bit 7: quality >= mq */
          any |= q1 & 0x3e3e3e3eu;
        }
        if (any) walk_bits |= 1u << l;
        if (act) __builtin_memcpy(dp_w + ((a_ >> 8) + o), &v[i], 16);
      }
    };
    /* one read, the whole wave, four bytes a lane: longer than 128 bytes or shorter than 16, marked by a left trim, or byte by byte from its plan */
    auto one = [&](const uint32_t l) {
      const uint32_t k = l & 1u;
      const uint32_t ctl = d_bcast(l_ctl, l), kind = (ctl >> 28) & 3u, len = ctl & 0xfffffu;
      uint8_t *const dp = seq_out + d_bcast64(l_off, l);
      if (kind == KIND_SLOW) {
        const prep_slow_ret r = prep_slow_read(plan + g0 + l, (ctl >> 30) & 1u ? PD_EDITED : 0u, dp, seq, ms_work, mq);
        c63 += r.c63;
        cge += r.cge;
        if (k == 0 && r.walked) walked_mask |= 1ull << l;
        return;
      }
      const uint32_t ml = (ctl >> 20) & 0xffu;
      bool walked = false;
      for (uint32_t base = 0; base < len; base += 256u) {
        const uint32_t nominal = base + lane4;
        const bool act = nominal < len;
        const uint32_t o = nominal < len - 4u ? nominal : len - 4u;
        uint32_t w = act ? d_ld32((const uint8_t *)desc + d_bcast64(l_src, l) + o) : 0u;
        if (ml + 3u > base) { /* the left trim's mark: quality 63 on bytes below ml (a last, overlapping dword reaches up to
                                 three bytes back into the round before: they must be stored as that round marked them) */
          const uint32_t nmk = act && ml > o ? (ml - o < 4u ? ml - o : 4u) : 0u;
          const uint32_t mm = nmk >= 4u ? 0xffffffffu : (1u << (8u * nmk)) - 1u;
          w = (w & ~mm) | (((w & 0x03030303u) | 0xfcfcfcfcu) & mm);
        }
        /* bytes this lane shares with its neighbour (the low ones of a last, overlapping dword) do not count twice: shifted out; what
         * comes in is quality 0, like an idle lane's zero */
        const uint32_t wc = act ? w >> (8u * (nominal - o)) : 0u;
        const uint32_t qh = ((wc >> 2) & 0x3f3f3f3fu) | 0x80808080u;
        const uint32_t q1 = qh + 0x01010101u;
        c63 += (uint32_t)__popc(q1 & 0x40404040u);
        cge += (uint32_t)__popc((qh - mq4) & 0x80808080u);
        if (k == 0) walked |= __any((q1 & 0x3e3e3e3eu) != 0u) != 0; /* was read 0 walked (src/call_genotypes.c:198-211) */
        if (act) d_st32(dp + o, w);
      }
      if (k == 0 && walked) walked_mask |= 1ull << l;
    };
    if (__ballot(my_wide)) {
      /* two buffers in turn: no register is copied, so nothing waits for a round's loads before that round's turn */
      uint4 va[RW], vb[RW];
      fetch8(0, va);
#pragma unroll
      for (uint32_t i0 = 0; i0 < 8u; i0 += 2u * RW) {
        fetch8(i0 + RW, vb);
        process8(i0, va);
        fetch8(i0 + 2u * RW, va);
        process8(i0 + RW, vb);
      }
      /* was read r walked: some lane of its eight saw such a byte in round r & 7 */
      walk_bits |= (uint32_t)__shfl_xor((int)walk_bits, 1);
      walk_bits |= (uint32_t)__shfl_xor((int)walk_bits, 2);
      walk_bits |= (uint32_t)__shfl_xor((int)walk_bits, 4);
      walked_mask |= __ballot((walk_bits >> (lane & 7u)) & 1u);
    }
    for (unsigned long long rest = __ballot(my_kind != 0u && !my_wide); rest; rest &= rest - 1ull) one((uint32_t)__builtin_ctzll(rest));
    if (lane < n && my_k == 0) tpl_out[my_ti].flags = BSC_TPL_WALK_KNOWN | (((walked_mask >> lane) & 1ull) ? BSC_TPL_WALKED0 : 0u);
  }
  unsigned long long w[5] = {l_total, c63, cge, l_reads, l_bases};
  for (int i = 0; i < 5; i++)
    for (int o = 32; o > 0; o >>= 1) w[i] += __shfl_xor(w[i], o);
  if (mq == 0) w[2] = w[0]; /* (idle lanes and shifted-in bytes read quality 0: with mq = 0 they were counted; every byte is >= 0) */
  if (lane == 0) {
    /* base_trim = quality 63; base_lowqual = below min_qual and not 63; base_none = the rest */
    const unsigned long long low = w[0] - w[2] - (mq > FLT_QUAL ? w[1] : 0ull), none = w[0] - w[1] - low;
    if (none) atomicAdd(&cnt[3], none);
    if (w[1]) atomicAdd(&cnt[4], w[1]);
    if (low) atomicAdd(&cnt[5], low);
    if (w[3]) atomicAdd(&cnt[6], w[3]);
    if (w[4]) atomicAdd(&cnt[7], w[4]);
  }
}

/*
 * The read profile, a pass of its own behind the copy (round 6; inside the copy kernel — an LDS atomic per counted base, and the walk held
 * to the slower of its two forms — it made that kernel 4.7 ms against 1.2).  What counts is a property of (reference position, base,
 * quality); where it counts is the base's position in the ORIGINAL read.  So:
 *   bsc_prep_refmask_kernel  once per call, a byte per reference position: 4 = a C followed by A / C / T, 8 = a G preceded by A / G / T
 *                            (PROF_REF over the two pairs of codes around it, 0 where the walk has no code behind it), 8 bytes of
 *                            zeros in front and behind, so that a dword of it can be fetched anywhere near the block.
 *   bsc_prep_profile_kernel  lanes own READ POSITIONS, not output bytes: lane h holds positions 8h .. 8h + 7 of every read it sees and
 *                            keeps their counts in registers — four bytes side by side in a dword, two dwords per base code, flushed to
 *                            the workgroup's table in LDS every 255 reads — so a counted base costs a byte-parallel add and no atomic.
 *                            The prepared bytes of those positions are one unaligned 8-byte load (byte-swapped for read 1, whose
 *                            positions run against its bytes), the mask bytes another.  Reads of up to 128 positions (pc + length) go
 *                            four to a wave, sixteen lanes each; up to 256, two to a wave; the rest — and the reads the list cut or
 *                            padded, whose bytes are not at a fixed distance from their positions, and strand 0 — byte by byte with
 *                            atomics as before.  (First form: four bytes a lane, two / one reads to a wave: 0.87 ms against 0.76.)
 * bs_strand 2 counts a base in the column bs_strand 1 counts its complement in (src/init_param.c:57-70: {11,4,10,5} / {9,6,8,7}), so the
 * registers are kept by base code (complemented for strand 2) and the column is looked up at the flush.
 */
__global__ __launch_bounds__(256) void bsc_prep_refmask_kernel(const uint8_t *__restrict__ ref, uint32_t n_ref, uint8_t *__restrict__ mask) {
  const uint64_t i = ((uint64_t)blockIdx.x * 256u + threadIdx.x) * 4u; /* mask[i .. i + 3]: positions v = i - 8 .. (the arrays' starts are 4-byte aligned) */
  if (i >= (uint64_t)n_ref + 24u) return;
  uint32_t m = 0;
  if (i >= 12u && i - 8u + 7u <= n_ref) { /* codes v - 1 .. v + 6 exist: four positions at once, byte-parallel */
    const uint64_t v = i - 8u;
    const uint32_t lo = d_ld32(ref + v - 1u) & 0x07070707u, hi = d_ld32(ref + v + 3u) & 0x07070707u;
    const uint32_t r_at = __builtin_amdgcn_alignbyte(hi, lo, 1u), r_next = __builtin_amdgcn_alignbyte(hi, lo, 2u);
#define PROF_NE(x, c) ((((x) ^ ((c) * 0x01010101u)) + 0x7f7f7f7fu) & 0x80808080u) /* bit 7 of every byte: the code (< 8) is not c */
    const uint32_t over_c = ~PROF_NE(r_at, 2u) & PROF_NE(r_next, 0u) & PROF_NE(r_next, 3u) & 0x80808080u; /* C, then A / C / T */
    const uint32_t over_g = ~PROF_NE(r_at, 3u) & PROF_NE(lo, 0u) & PROF_NE(lo, 2u) & 0x80808080u;         /* G, after A / G / T */
#undef PROF_NE
    m = (over_c >> 5) | (over_g >> 4);
  } else {
    for (uint32_t t = 0; t < 4u; t++) {
      const uint64_t it = i + t;
      if (it >= 8u && it - 8u + 1u < n_ref) {
        const uint64_t v = it - 8u;
        const uint32_t ra = v >= 1 ? ref[v - 1] : 0u, rb = ref[v], rc = ref[v + 1];
        m |= ((PROF_REF[((rb << 3) | rc) & 63u] & 4u) | (PROF_REF[((ra << 3) | rb) & 63u] & 8u)) << (8u * t);
      }
    }
  }
  __builtin_memcpy(mask + i, &m, 4); /* (the workspace is n_ref + 24 bytes rounded up to 4: 8 bytes of zeros in front, 16 behind) */
}

__global__ __launch_bounds__(64 * PREP_WAVES) void bsc_prep_profile_kernel(const bsc_prep_plan *__restrict__ plan, const bsc_prep_desc *__restrict__ desc,
                                                                           uint32_t nr, const bsc_misms *__restrict__ ms_work,
                                                                           const unsigned long long *__restrict__ out_off,
                                                                           const bsc_template *__restrict__ tpl_out, const uint8_t *__restrict__ seq_out,
                                                                           uint64_t seq_out_cap, unsigned long long *__restrict__ cnt,
                                                                           const uint8_t *__restrict__ mask, const bsc_prep_prof F) {
  __shared__ uint32_t s_prof[PROF_LDS * 4u];
  for (unsigned i = threadIdx.x; i < PROF_LDS * 4u; i += 64 * PREP_WAVES) s_prof[i] = 0;
  __syncthreads();
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * PREP_WAVES + (threadIdx.x >> 6)));
  const uint32_t n_waves = gridDim.x * PREP_WAVES;
  const uint32_t n_reads2 = 2u * nr, n_groups = (n_reads2 + 63u) / 64u;
  uint32_t acc[4][2] = {{0u, 0u}, {0u, 0u}, {0u, 0u}, {0u, 0u}}; /* by base code: counts of this lane's eight positions, a byte each */
  uint32_t acc_n = 0;                 /* reads since the last flush (wave-uniform) */
  bool acc_quarter = true;            /* whose positions the registers hold: 8 (lane & 15) .. or 8 (lane & 31) .. */
  auto flush = [&]() {
    const uint32_t p0 = 8u * (acc_quarter ? (lane & 15u) : (lane & 31u));
#pragma unroll
    for (uint32_t b = 0; b < 4u; b++) {
      const uint32_t col = (0x63u >> (2u * b)) & 3u; /* strand 1's column of base code b */
#pragma unroll
      for (uint32_t t = 0; t < 8u; t++) {
        const uint32_t c = (acc[b][t >> 2] >> (8u * (t & 3u))) & 0xffu;
        if (c) atomicAdd(&s_prof[col * PROF_LDS + p0 + t + 1u], c); /* (p0 + t + 1 <= 256 < PROF_LDS; kept positions only: see lim) */
      }
      acc[b][0] = acc[b][1] = 0u;
    }
    acc_n = 0;
  };
  for (uint32_t g = wave; g < n_groups; g += n_waves) {
    const uint32_t g0 = g * 64u, n = n_reads2 - g0 < 64u ? n_reads2 - g0 : 64u;
    /* ---- lane r: read g0 + r ---- */
    const uint32_t ri = g0 + (lane < n ? lane : 0u), my_ti = ri >> 1, my_k = ri & 1u;
    const bsc_prep_desc L = desc[ri];
    const unsigned long long l_off = out_off[ri];
    /* (asked for whether the read turns out to be profiled or not: one wait for the six loads of a lane instead of two in a row) */
    const uint32_t us_prev = my_ti ? F.used_scan[my_ti - 1u] : 0u, us_mine = F.used_scan[my_ti];
    const uint32_t t_strand = tpl_out[my_ti].bs_strand, t_pos = tpl_out[my_ti].pos[my_k];
    /* c1: index of output byte 0 in the codes + 1 (0: a read at the block's first position, whose walk starts a code late);
     * kind: 0 nothing, 1 in registers, 2 the long way */
    uint32_t c1 = 0, kind = 0, l_used = 0, l_strand = 0, l_ol = 0, top = 0;
    if (lane < n) {
      const uint32_t fl = L.pk >> 28;
      uint32_t ol = L.pk & 0xfffffu;
      if (fl & PD_SLOW) ol = plan[ri].out_len;
      const bool fits = !(l_off > seq_out_cap || seq_out_cap - l_off < ol); /* (else the copy kernel refused it: the call fails) */
      if ((fl & PD_PRESENT) && ol != 0 && fits) { /* this read's place in the block's codes, its template's share of the vector */
        const uint32_t before = my_ti ? (us_prev > F.used0 ? us_prev : F.used0) : F.used0;
        l_used = us_mine > before ? us_mine : before;
        /* growing the vector past its capacity is where the host form gives up (csrc/prep.c) */
        if (l_used > before && (unsigned long long)l_used + 1ull > F.cap) atomicMin(&cnt[0], ((unsigned long long)my_ti << 8) | PE_PROF_CAP);
        l_strand = t_strand;
        const uint32_t pos = t_pos;
        l_ol = ol;
        const bool in_ref = pos >= F.x && (uint64_t)pos - F.x + ol + 1u <= F.n_ref;
        if (!in_ref) atomicMin(&cnt[0], ((unsigned long long)my_ti << 8) | PE_PROF_RANGE);
        /* "this read is profiled": inside the block's codes (else the call fails anyway), on a strand that has a profile
         * (src/init_param.c:57-70: bs_strand 0 .. 2) */
        if (in_ref && l_strand <= 2u) {
          const uint32_t lim = l_used < F.cap ? l_used : F.cap; /* positions (+ 1) that are kept */
          const int32_t pc = L.pc;
          top = my_k ? (uint32_t)(pc < 0 ? 0 : pc) : (uint32_t)pc + ol - 1u; /* the read's highest position */
          /* in registers: bytes at a fixed distance from their positions, a strand whose column follows from the base, every position of
           * the read inside the window of 256 and kept (top + 1 < lim: the vector only cuts reads while it still grows), and eight bytes
           * seven to either side of the read inside the buffer */
          const bool regs = !(fl & PD_SLOW) && l_strand != 0u && ol <= 256u && pc >= 0 && pc < 512 && top < 256u && top + 1u < lim && l_off >= 8u &&
                            seq_out_cap - l_off - ol >= 8u;
          kind = regs ? 1u : 2u;
          c1 = pos > F.x ? pos - F.x + 1u : 0u;
        }
      }
    }
    /* ---- the reads whose counts stay in registers ---- */
    unsigned long long m_regs = __ballot(kind == 1u);
    if (m_regs) {
      /* what a round needs of a read, two words a lane fetches from the read's own lane: where its first code's byte lies in mask[]
       * (29 bits) | strand 2 << 29 | 1 << 31;  length - 1 (8 bits) | pc << 8 (9 bits) | its place, from the group's lowest << 17 */
      const uint32_t first = (uint32_t)__builtin_ctzll(m_regs);
      const unsigned long long obase = d_bcast64(l_off, first); /* (the places ascend with the reads) */
      const unsigned long long orel = l_off - obase;
      if (kind == 1u && orel >= (1ull << 15)) kind = 2u; /* (far from the group's first — long reads in between: the long way) */
      const uint32_t A1 = kind == 1u ? (c1 + 7u) | (l_strand == 2u ? 1u << 29 : 0u) | (1u << 31) : 0u; /* code index v = c1 - 1 lies at mask[v + 8] */
      const uint32_t A2 = kind == 1u ? (l_ol - 1u) | ((uint32_t)L.pc << 8) | ((uint32_t)orel << 17) : 0u;
      const bool quarter = __ballot(kind == 1u && top >= 128u) == 0ull;
      if (quarter != acc_quarter) {
        flush();
        acc_quarter = quarter;
      }
      const uint8_t *const sb = seq_out + obase - 8u; /* obase >= 8 */
      const uint32_t h = quarter ? (lane & 15u) : (lane & 31u);
      const int32_t p0 = (int32_t)(8u * h);
      const uint32_t hsel = quarter ? (lane & 48u) : (lane & 32u), rounds = quarter ? 16u : 32u;
      constexpr uint32_t B = 2u; /* rounds in flight: their loads are asked for a batch ahead */
      struct rnd {
        uint32_t a1;
        unsigned long long w, mw;
        int32_t jc, len8; /* len8 = 8 - the read's length */
      };
      /* reads alternate read 0 / read 1 of their templates, in both forms of the rounds: REV is a round's parity */
      auto ask = [&](uint32_t i, rnd &r) {
        const int src = (int)(hsel | (i & (rounds - 1u)));
        r.a1 = (uint32_t)__shfl((int)A1, src);
        const uint32_t a2 = (uint32_t)__shfl((int)A2, src);
        const int32_t len = (int32_t)(a2 & 0xffu) + 1;
        r.len8 = 8 - len;
        const int32_t pc = (int32_t)((a2 >> 8) & 0x1ffu);
        /* output byte of this lane's lowest-addressed position: read 0: j = p - pc, ascending; read 1: j = pc - p, descending */
        const int32_t j0 = (i & 1u) ? pc - p0 - 7 : p0 - pc;
        const bool some = (int32_t)r.a1 < 0 && (uint32_t)(j0 + 7) < (uint32_t)(len + 7);
        r.jc = some ? j0 : 0;
        __builtin_memcpy(&r.w, sb + ((a2 >> 17) + (uint32_t)(r.jc + 8)), 8);
        /* a round that is not this lane's (no read there, or none of its bytes at this lane's positions): mask bytes that are 0 — the
         * pad in front of the codes — and whatever the other load brings counts nothing */
        __builtin_memcpy(&r.mw, mask + (some ? (r.a1 & 0x1fffffffu) + (uint32_t)j0 : 0u), 8);
      };
      auto count = [&](uint32_t i, const rnd &r) {
        const int32_t lo = r.jc < 0 ? -r.jc : 0, hi = r.jc + r.len8 > 0 ? r.jc + r.len8 : 0; /* bytes of the eight in front of / behind the read: <= 7 */
        unsigned long long w8 = r.w & (~0ull << (8 * lo)) & (~0ull >> (8 * hi)); /* not the read's: quality 0, never counted */
        unsigned long long m8 = r.mw;
        if (i & 1u) { /* positions ascend as the bytes descend */
          w8 = __builtin_bswap64(w8);
          m8 = __builtin_bswap64(m8);
        }
        const uint32_t sx = (r.a1 >> 29) & 1u ? 0xffffffffu : 0u; /* strand 2: by the complement */
#pragma unroll
        for (int d = 0; d < 2; d++) {
          const uint32_t w = (uint32_t)(w8 >> (32 * d)), mw = (uint32_t)(m8 >> (32 * d));
          /* a base counts iff its quality is in [20, 63) and either it reads C / T (base code odd) over mask 4, or A / G over mask 8 */
          const uint32_t odd = w << 7;
          const uint32_t qf = ((w >> 2) & 0x3f3f3f3fu) | 0x80808080u;
          const uint32_t q_ok = (qf - 20u * 0x01010101u) & ~((qf + 0x01010101u) << 1) & 0x80808080u;
          const uint32_t hb = (((odd & (mw << 5)) | (~odd & (mw << 4))) & q_ok) >> 7;
          const uint32_t wb = w ^ sx;
          const uint32_t t1 = hb & (wb >> 1), t0 = hb ^ t1, t3 = t1 & wb, t2 = t0 & wb;
          acc[3][d] += t3;
          acc[2][d] += t1 ^ t3;
          acc[1][d] += t2;
          acc[0][d] += t0 ^ t2;
        }
      };
      rnd ra[B], rb[B];
#pragma unroll
      for (uint32_t u = 0; u < B; u++) ask(u, ra[u]);
      for (uint32_t i0 = 0; i0 < rounds; i0 += 2u * B) { /* (a batch past the last wraps round to the first reads: loads of bytes that exist, unused) */
        if (acc_n + 2u * B > 255u) flush();
#pragma unroll
        for (uint32_t u = 0; u < B; u++) ask(i0 + B + u, rb[u]);
#pragma unroll
        for (uint32_t u = 0; u < B; u++) count(u, ra[u]);
#pragma unroll
        for (uint32_t u = 0; u < B; u++) ask(i0 + 2u * B + u, ra[u]);
#pragma unroll
        for (uint32_t u = 0; u < B; u++) count(B + u, rb[u]);
        acc_n += 2u * B;
      }
    }
    /* ---- the others, one at a time, a byte per lane ---- */
    unsigned long long m_long = __ballot(kind == 2u);
    while (m_long) {
      const uint32_t l = (uint32_t)__builtin_ctzll(m_long);
      m_long &= m_long - 1ull;
      const uint32_t ol = d_bcast(l_ol, l), k = l & 1u, strand = d_bcast(l_strand, l), used_t = d_bcast(l_used, l);
      const int32_t pc = (int32_t)d_bcast((uint32_t)L.pc, l);
      const int64_t v0 = (int64_t)d_bcast(c1, l) - 1;
      const uint8_t *const dp = seq_out + d_bcast64(l_off, l);
      const bool slow = (d_bcast(L.pk, l) >> 28) & PD_SLOW;
      bsc_prep_plan P;
      bool edited = false;
      if (slow) {
        P = plan[g0 + l];
        edited = (d_bcast(L.pk, l) >> 28) & PD_EDITED;
      }
      for (uint32_t jb = 0; jb < ol; jb += 64u) {
        const uint32_t j = jb + lane;
        if (j < ol) {
          bool pad = false;
          const uint32_t s = slow ? prep_slow_src(P, ms_work + P.ms, edited, j, pad) : j;
          if (!pad) d_prof_base(F, s_prof, strand, used_t, v0 + j, k ? pc - (int32_t)s : pc + (int32_t)s, dp[j]);
        }
      }
    }
  }
  flush();
  __syncthreads();
  for (unsigned i = threadIdx.x; i < PROF_LDS * 4u; i += 64 * PREP_WAVES)
    if (s_prof[i] && i % PROF_LDS < F.cap) atomicAdd(&F.table[(uint64_t)(i % PROF_LDS) * 4u + i / PROF_LDS], (unsigned long long)s_prof[i]);
}

/* ---- launcher ------------------------------------------------------------------------------------------------------------------ */
extern "C" int bsc_dev_scan_u64(const void *in, void *out, uint32_t n, void *tmp, size_t tmp_bytes, void *stream); /* sort.hip */

extern "C" size_t bsc_dev_prep_plan_bytes(void) { return sizeof(bsc_prep_plan) + sizeof(bsc_prep_desc); } /* per read */

/*
 * cnt[8] (device, unsigned long long): [0] the error word (all ones = none: set by the caller), [1] base_clip, [2] base_overlap,
 * [3] base_none, [4] base_trim, [5] base_lowqual, [6] reads, [7] read_bases — zeroed by the caller.  out_len / out_off: 2 nr + 1
 * words each; ms_work: n_misms entries; plan: 2 nr entries.  After the launch out_off[2 nr] = the bytes written.
 */
extern "C" int bsc_dev_scan_max_u32(const void *in, void *out, uint32_t n, void *tmp, size_t tmp_bytes, void *stream); /* sort.hip */

/* prof_ref != NULL: the read profile too — prof_ref = the codes of prof_x .. prof_x + prof_n_ref - 1 (device), prof_table = u64 [prof_cap][4]
 * zeroed by the caller (this call's counts), prof_used0 = the vector's length before the call, max_pos1 / used_scan: nr words each */
extern "C" int bsc_dev_launch_prep(const void *raw, uint32_t nr, const void *seq, uint64_t seq_bytes, const void *misms, uint64_t n_misms,
                                   const bsc_prep_params *par, void *ms_work, void *plan, void *out_len, void *out_off, void *scan_tmp,
                                   size_t scan_tmp_bytes, void *tpl_out, void *seq_out, uint64_t seq_out_cap, void *cnt, int num_cus,
                                   void *stream, const void *prof_ref, uint32_t prof_x, uint32_t prof_n_ref, uint32_t prof_cap,
                                   uint32_t prof_used0, void *prof_table, void *max_pos1, void *used_scan, void *prof_mask) {
  hipStream_t s = (hipStream_t)stream;
  bsc_prep_desc *const desc = (bsc_prep_desc *)((bsc_prep_plan *)plan + 2ull * nr); /* the descriptors lie behind the plans */
  if (!nr) return (int)hipMemsetAsync(out_off, 0, sizeof(unsigned long long), s);
  hipError_t e = hipMemsetAsync((unsigned long long *)out_len + 2ull * nr, 0, sizeof(unsigned long long), s);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(bsc_prep_plan_kernel, dim3((nr + 255u) / 256u), dim3(256), 0, s, (const bsc_raw_template *)raw, nr, (const uint8_t *)seq,
                     seq_bytes, (const bsc_misms *)misms, n_misms, *par, (bsc_misms *)ms_work, (bsc_prep_plan *)plan, desc,
                     (unsigned long long *)out_len, (bsc_template *)tpl_out, (unsigned long long *)cnt, prof_ref ? (uint32_t *)max_pos1 : nullptr);
  if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  int rc = bsc_dev_scan_u64(out_len, out_off, 2u * nr + 1u, scan_tmp, scan_tmp_bytes, stream);
  if (rc) return rc;
  bsc_prep_prof F = {nullptr, 0u, 0u, 0u, 0u, nullptr, nullptr};
  if (prof_ref) {
    if ((rc = bsc_dev_scan_max_u32(max_pos1, used_scan, nr, scan_tmp, scan_tmp_bytes, stream))) return rc;
    F.ref = (const uint8_t *)prof_ref;
    F.x = prof_x;
    F.n_ref = prof_n_ref;
    F.cap = prof_cap;
    F.used0 = prof_used0;
    F.used_scan = (const uint32_t *)used_scan;
    F.table = (unsigned long long *)prof_table;
  }
  unsigned g = ((nr + 31u) / 32u + PREP_WAVES - 1u) / PREP_WAVES;
  /* as many workgroups as are resident at once: every wave strides over the groups of reads, one that starts late would do its
   * whole share after the others have finished */
  static int per_cu[2] = {0, 0};
  for (int pv = 0; pv < (prof_ref ? 2 : 1); pv++)
    if (!per_cu[pv]) {
      int nb = 0;
      const hipError_t eo = pv ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bsc_prep_profile_kernel, 64 * PREP_WAVES, 0)
                               : hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bsc_prep_copy_kernel, 64 * PREP_WAVES, 0);
      per_cu[pv] = (eo == hipSuccess && nb > 0) ? nb : 4;
      (void)hipGetLastError();
    }
  unsigned gc = g, gp = g;
  if (gc > (unsigned)num_cus * (unsigned)per_cu[0]) gc = (unsigned)num_cus * (unsigned)per_cu[0];
  hipLaunchKernelGGL(bsc_prep_copy_kernel, dim3(gc), dim3(64 * PREP_WAVES), 0, s, (const bsc_prep_plan *)plan, (const bsc_prep_desc *)desc, nr,
                     (const uint8_t *)seq, (const bsc_misms *)ms_work, (const unsigned long long *)out_off, par->min_qual, (bsc_template *)tpl_out,
                     (uint8_t *)seq_out, seq_out_cap, (unsigned long long *)cnt);
  if ((e = hipGetLastError()) != hipSuccess) return (int)e;
  if (prof_ref) { /* the read profile: a pass of its own over the prepared bytes */
    hipLaunchKernelGGL(bsc_prep_refmask_kernel, dim3((unsigned)(((uint64_t)prof_n_ref + 24u + 1023u) / 1024u)), dim3(256), 0, s, (const uint8_t *)prof_ref,
                       prof_n_ref, (uint8_t *)prof_mask);
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    if (gp > (unsigned)num_cus * (unsigned)per_cu[1]) gp = (unsigned)num_cus * (unsigned)per_cu[1];
    hipLaunchKernelGGL(bsc_prep_profile_kernel, dim3(gp), dim3(64 * PREP_WAVES), 0, s, (const bsc_prep_plan *)plan, (const bsc_prep_desc *)desc, nr,
                       (const bsc_misms *)ms_work, (const unsigned long long *)out_off, (const bsc_template *)tpl_out, (const uint8_t *)seq_out,
                       seq_out_cap, (unsigned long long *)cnt, (const uint8_t *)prof_mask, F);
  }
  return (int)hipGetLastError();
}
