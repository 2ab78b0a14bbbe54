/*
 * bamdev_core.h — the statements of the device BAM reader (round 6), written once: included by csrc/bamdev.hip (compiled by hipcc
 * for gfx950: the kernels call these per lane) and by tests/emul/bamdev_emul.cpp (compiled by g++: the SAME statements run by
 * loops on the CPU, so that their logic can be checked against csrc/bamio.c and the test suite's independent Python restatement in a container without a GPU —
 * test infrastructure, never loaded by the product).
 *
 * What is restated, with the reference line each part follows (through csrc/bamio.c, which cites them statement by statement):
 *   bd_parse            one alignment record -> a 64-byte descriptor: get_next_align_details (src/input_sam.c:222-312) — the flag /
 *                       MAPQ / insert-size / orientation filters and their reasons, forward / reverse position, orientation —,
 *                       the CIGAR's span and length in the read (get_bam_misms, :90-136), get_bs_strand (:144-220)
 *   bd_replay           read_input (src/get_template_vector.c:49-389) over the descriptors of the records that passed, statement for
 *                       statement, by ONE lane: block segmentation (:141-207), the pair table (:223-276), duplicate resolution with
 *                       the reference's accidents (:281-326,345-372).  Exact for any input; the slow path.
 *   bd_f_*              the same decisions as independent pieces — a prefix maximum for the blocks, one lane per start position for
 *                       the duplicates, one lane per backwards-facing mate for the joins — valid for input whose records are sorted
 *                       and whose read names pair up the usual way; bd_f_* report anything else (`irregular`) and the replay decides.
 *   bd_decode_read, bd_misms, bd_template     get_seq_and_qual (src/input_sam.c:61-88), get_bam_misms, align_details -> the
 *                       bsc_raw_template[] + read bytes + lists that bsc_prepare_templates_device takes
 */
#ifndef BSC_BAMDEV_CORE_H
#define BSC_BAMDEV_CORE_H

#include <stdint.h>
#include <string.h>

#if defined(__HIPCC__)
#define BD_FN __host__ __device__ static inline
#else
#define BD_FN static inline
#endif

#define BD_NONE 0xffffffffu

enum { BD_F_PAIRED = 1, BD_F_PROPER = 2, BD_F_UNMAP = 4, BD_F_MUNMAP = 8, BD_F_REVERSE = 16, BD_F_READ2 = 128, BD_F_SECONDARY = 256,
       BD_F_QCFAIL = 512, BD_F_DUP = 1024, BD_F_SUPP = 2048 };
enum { BD_FLT_NONE, BD_FLT_UNMAPPED, BD_FLT_QC, BD_FLT_SECONDARY, BD_FLT_MATE_UNMAPPED, BD_FLT_DUPLICATE, BD_FLT_NOPOS, BD_FLT_NOMATEPOS,
       BD_FLT_MISMATCH_CHR, BD_FLT_ORIENTATION, BD_FLT_INSERT_SIZE, BD_FLT_NOSEQ, BD_FLT_MAPQ, BD_FLT_NOT_ALIGNED, BD_FLT_PAIR_NOT_FOUND };
/* bd_desc.status */
enum { BD_ST_USE = 0, BD_ST_FILTERED = 1, BD_ST_ABSENT = 2, BD_ST_MALFORMED_CIGAR = 3, BD_ST_ERR_SIZE = 4, BD_ST_ERR_TRUNC = 5, BD_ST_ERR_RECORD = 6 };
/* error codes of the replay (bd_ws.err[0] = code, err[1] = used index) */
enum { BD_E_OK = 0, BD_E_TID = 1, BD_E_MATES_DISAGREE = 2, BD_E_DUP_NAME = 3, BD_E_MATE_OPENS_BLOCK = 4, BD_E_BLOCK_START = 5, BD_E_NOSEQ_DUP = 6,
       BD_E_RECORD = 7 };

typedef struct {
  uint32_t mapq_thresh;
  uint32_t keep_unmatched, ignore_duplicates, keep_duplicates;
  uint64_t max_template_len;
  int32_t region_tid;
  uint32_t region_start, region_stop;
  int32_t n_ref;
  uint32_t keep_unplaced;  /* with tid_keep: the records without a contig (and with an invalid one) are this reader's too */
  const uint8_t *tid_keep; /* a selection of contigs (one byte per contig of the header, 1 = this reader's), NULL = all: a sharded run's ranks
                            * each read the stretches of the file that hold their contigs; a stretch's other records are another rank's */
} bd_params;

typedef struct {
  uint64_t off;  /* stream offset of the record's block_size field */
  uint64_t hash; /* of the read name, its terminator included (the reference keys its pair table on l_qname bytes) */
  int32_t tid;
  uint32_t fwd, rev;                       /* forward_position, reverse_position */
  uint32_t span, aln_len, l_seq, del_len;  /* reference_span, align_length, l_seq, the bases its deletions pad (BSC_MISMS_INS sizes) */
  uint32_t bs;                             /* block_size */
  uint16_t aflag, n_cigar, n_ms;           /* alignment_flag, CIGAR operations, entries of its mismatch list */
  uint8_t status, flt, mapq, rev_ori;      /* BD_ST_*, the filter reason, MAPQ, bit 0 reverse strand | bit 1 orientation */
  uint8_t bs_strand, l_name, q0, q1;       /* gt_bs_strand, l_read_name, the qualities of decoded bytes 0 and 1 (get_al_qual's sq[k]) */
  uint16_t pad_;
} bd_desc;

BD_FN uint32_t bd_le32(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8 | (uint32_t)p[2] << 16 | (uint32_t)p[3] << 24; }
BD_FN uint32_t bd_le16(const uint8_t *p) { return p[0] | (uint32_t)p[1] << 8; }
BD_FN uint32_t bd_st(const bd_desc &d) { return (d.rev_ori & 1u) ? d.rev : d.fwd; }

BD_FN uint64_t bd_hash_name(const uint8_t *s, uint32_t n) {
  uint64_t h = 0xcbf29ce484222325ull;
  for (uint32_t i = 0; i < n; i++) h = (h ^ s[i]) * 0x100000001b3ull;
  h ^= h >> 33; /* a finaliser: FNV's low bits are weak, and the table and the sort both use them */
  h *= 0xff51afd7ed558ccdull;
  h ^= h >> 33;
  h *= 0xc4ceb9fe1a85ec53ull;
  h ^= h >> 33;
  return h;
}

/* get_bs_strand, src/input_sam.c:144-220 (csrc/bamio.c bs_strand_of) */
BD_FN uint8_t bd_bs_strand(const uint8_t *s, const uint8_t *end) {
  uint8_t strand = 0;
  int ok = 1;
  while (ok && s + 4 <= end) {
    /* 0 unknown, 1 GEM, 2 BOWTIE, 3 NOVALIGN, 4 BSMAP, 5 BWAMETH */
    int al = 0;
    if (s[0] == 'Z') al = s[1] == 'B' ? 3 : (s[1] == 'S' ? 4 : 0);
    else if (s[0] == 'X') al = s[1] == 'G' ? 2 : (s[1] == 'B' ? 1 : 0);
    else if (s[0] == 'Y' && s[1] == 'D') al = 5;
    s += 2;
    const uint8_t type = *s++;
    switch (type) {
      case 'A':
        if (al == 1) strand = *s == 'C' ? 1 : (*s == 'G' ? 2 : strand);
        s++;
        break;
      case 'C': case 'c': s++; break;
      case 'S': case 's':
        if (s + 2 <= end) s += 2; else ok = 0;
        break;
      case 'I': case 'i': case 'f':
        if (s + 4 <= end) s += 4; else ok = 0;
        break;
      case 'd':
        if (s + 8 <= end) s += 8; else ok = 0;
        break;
      case 'Z':
        if (al == 2 || al == 3) strand = *s == 'C' ? 1 : (*s == 'G' ? 2 : strand);
        else if (al == 4) strand = *s == '+' ? 1 : (*s == '-' ? 2 : strand);
        else if (al == 5) strand = *s == 'f' ? 1 : (*s == 'r' ? 2 : strand);
        /* fall through */
      case 'H':
        while (s < end && *s) s++;
        if (s < end) s++; else ok = 0;
        break;
      case 'B': {
        const uint8_t st = *s++;
        unsigned sz = 0;
        if (st == 'c' || st == 'C' || st == 'A') sz = 1;
        else if (st == 's' || st == 'S') sz = 2;
        else if (st == 'i' || st == 'I' || st == 'f') sz = 4;
        else if (st == 'd') sz = 8;
        else if (st == 'Z' || st == 'H' || st == 'B') sz = st; /* the reference's table holds the letter itself for these */
        if (s + 4 <= end && sz != 0) {
          const uint32_t n = bd_le32(s);
          s += 4;
          if ((uint64_t)n * sz <= (uint64_t)(end - s)) s += (size_t)n * sz; else ok = 0;
        } else ok = 0;
      } break;
      default: break;
    }
  }
  return strand;
}

/* the decoded byte of base i: base | min(q, 43) << 2, anything but A C G T = 0 (get_seq_and_qual, src/input_sam.c:76-86) */
BD_FN uint8_t bd_base_byte(const uint8_t *seq4, const uint8_t *qual, uint32_t i) {
  const unsigned c4 = (seq4[i >> 1] >> ((~i & 1u) << 2)) & 15u;
  unsigned q = qual[i];
  if (q > 43) q = 43;
  const unsigned base = c4 == 1 ? 1 : (c4 == 2 ? 2 : (c4 == 4 ? 3 : (c4 == 8 ? 4 : 0)));
  return base ? (uint8_t)((base - 1) | (q << 2)) : 0;
}

/* One record at rec[0 .. avail): its descriptor.  Follows csrc/bamio.c next_record (= get_next_align_details) in its order. */
BD_FN void bd_parse(const uint8_t *rec, uint64_t avail, uint64_t off, const bd_params &par, bd_desc &d) {
  memset(&d, 0, sizeof d);
  d.off = off;
  if (avail < 4) {
    d.status = BD_ST_ERR_TRUNC;
    return;
  }
  const uint32_t bs = bd_le32(rec);
  d.bs = bs;
  if (bs < 32 || bs > (1u << 29)) {
    d.status = BD_ST_ERR_SIZE;
    return;
  }
  if ((uint64_t)bs + 4u > avail) {
    d.status = BD_ST_ERR_TRUNC;
    return;
  }
  const uint8_t *p = rec + 4;
  const int32_t tid = (int32_t)bd_le32(p), pos = (int32_t)bd_le32(p + 4), mtid = (int32_t)bd_le32(p + 20), mpos = (int32_t)bd_le32(p + 24);
  const int32_t isize = (int32_t)bd_le32(p + 28);
  const uint32_t l_name = p[8], mapq = p[9], n_cigar = bd_le16(p + 12), flag = bd_le16(p + 14), l_seq = bd_le32(p + 16);
  const uint64_t need = 32ull + l_name + 4ull * n_cigar + ((uint64_t)l_seq + 1) / 2 + l_seq;
  if (need > bs || l_name == 0) {
    d.status = BD_ST_ERR_RECORD;
    return;
  }
  d.tid = tid;
  if (par.tid_keep && !((tid >= 0 && tid < par.n_ref) ? par.tid_keep[tid] != 0 : par.keep_unplaced != 0)) {
    d.status = BD_ST_ABSENT; /* another reader's */
    return;
  }
  d.l_name = (uint8_t)l_name;
  d.n_cigar = (uint16_t)n_cigar;
  d.l_seq = l_seq;
  d.mapq = (uint8_t)mapq;
  const uint8_t *cig = p + 32 + l_name;
  const uint8_t *seq4 = cig + 4 * n_cigar;
  const uint8_t *qual = seq4 + (l_seq + 1) / 2;
  const uint8_t *aux = qual + l_seq;
  const uint8_t *end = p + bs;
  if (l_seq && n_cigar) { /* a CIGAR whose query length is not l_seq: the record does not exist for the reader (counted apart) */
    uint64_t qlen = 0;
    for (uint32_t i = 0; i < n_cigar; i++) {
      const uint32_t c = bd_le32(cig + 4 * i), op = c & 15u;
      if (op == 0 || op == 1 || op == 4 || op == 7 || op == 8) qlen += c >> 4;
    }
    if (qlen != l_seq) {
      d.status = BD_ST_MALFORMED_CIGAR;
      return;
    }
  }
  if (par.region_stop) {
    uint32_t reflen = 0;
    for (uint32_t i = 0; i < n_cigar; i++) {
      const uint32_t c = bd_le32(cig + 4 * i), op = c & 15u;
      if (op == 0 || op == 2 || op == 3 || op == 7 || op == 8) reflen += c >> 4;
    }
    const int64_t beg = (int64_t)par.region_start - 1, stop = par.region_stop;
    const int64_t rend = (int64_t)pos + (reflen ? reflen : 1);
    if (tid != par.region_tid || pos >= stop || rend <= beg) {
      d.status = BD_ST_ABSENT;
      return;
    }
  }
  int flt = BD_FLT_NONE;
  if ((flag & BD_F_PAIRED) && !par.keep_unmatched) {
    if ((flag & (BD_F_PROPER | BD_F_UNMAP | BD_F_MUNMAP | BD_F_QCFAIL | BD_F_SECONDARY | BD_F_SUPP | BD_F_DUP)) != BD_F_PROPER) {
      if (flag & (BD_F_SECONDARY | BD_F_SUPP)) flt = BD_FLT_SECONDARY;
      else if (flag & BD_F_UNMAP) flt = BD_FLT_UNMAPPED;
      else if (flag & BD_F_MUNMAP) flt = BD_FLT_MATE_UNMAPPED;
      else if (flag & BD_F_QCFAIL) flt = BD_FLT_QC;
      else if (flag & BD_F_DUP) {
        if (!par.ignore_duplicates) flt = BD_FLT_DUPLICATE;
      } else flt = BD_FLT_NOT_ALIGNED;
    }
  } else if (flag & (BD_F_UNMAP | BD_F_QCFAIL | BD_F_SECONDARY | BD_F_SUPP | BD_F_DUP)) {
    if (flag & (BD_F_SECONDARY | BD_F_SUPP)) flt = BD_FLT_SECONDARY;
    else if (flag & BD_F_UNMAP) flt = BD_FLT_UNMAPPED;
    else if (flag & BD_F_QCFAIL) flt = BD_FLT_QC;
    else if (flag & BD_F_DUP) flt = BD_FLT_DUPLICATE;
  }
  int mis_matched = (flag & (BD_F_MUNMAP | BD_F_PROPER)) != BD_F_PROPER;
  const int reverse = (flag & BD_F_REVERSE) != 0, second = (flag & BD_F_READ2) != 0;
  const unsigned orientation = ((second && reverse) || !(second || reverse)) ? 0u : 1u;
  d.rev_ori = (uint8_t)((reverse ? 1u : 0u) | orientation << 1);
  const int mult_seg = (flag & (BD_F_PAIRED | BD_F_MUNMAP)) == BD_F_PAIRED;
  uint32_t fwd, rev;
  if (reverse) {
    fwd = (uint32_t)mpos + 1u;
    rev = (uint32_t)pos + 1u;
  } else {
    fwd = (uint32_t)pos + 1u;
    rev = (uint32_t)mpos + 1u;
  }
  if (mapq < par.mapq_thresh && !flt) flt = BD_FLT_MAPQ;
  uint32_t aflag = flag;
  if (mult_seg) {
    if (tid != mtid) {
      if (!flt) flt = BD_FLT_MISMATCH_CHR;
      if (par.keep_unmatched) mis_matched = 1;
    }
    if (!flt && (uint64_t)(isize < 0 ? -(int64_t)isize : (int64_t)isize) > par.max_template_len) {
      flt = BD_FLT_INSERT_SIZE;
      if (par.keep_unmatched) mis_matched = 1;
    }
    if (reverse) {
      if (pos < mpos) {
        if (!flt) flt = BD_FLT_ORIENTATION;
        if (par.keep_unmatched) mis_matched = 1;
      }
      if (mis_matched) fwd = 0;
    } else {
      if (pos > mpos) {
        if (!flt) flt = BD_FLT_ORIENTATION;
        if (par.keep_unmatched) mis_matched = 1;
      }
      if (mis_matched) rev = 0;
    }
  }
  if (!mult_seg || mis_matched) aflag &= ~(uint32_t)BD_F_PAIRED;
  d.aflag = (uint16_t)aflag;
  d.fwd = fwd;
  d.rev = rev;
  d.flt = (uint8_t)flt;
  if (flt && !(par.keep_unmatched && (flt == BD_FLT_INSERT_SIZE || flt == BD_FLT_MISMATCH_CHR || flt == BD_FLT_ORIENTATION))) {
    d.status = BD_ST_FILTERED;
    return;
  }
  uint32_t span = 0, position = 0, nm = 0, del = 0;
  for (uint32_t i = 0; i < n_cigar; i++) {
    const uint32_t c = bd_le32(cig + 4 * i), len = c >> 4;
    switch (c & 15u) {
      case 0: case 7: case 8: position += len; span += len; break;
      case 6: case 4: case 1: position += len; nm++; break;
      case 2: span += len; del += len; nm++; break;
      default: break;
    }
  }
  d.span = span;
  d.aln_len = position;
  d.n_ms = (uint16_t)nm;
  d.del_len = del;
  d.bs_strand = bd_bs_strand(aux, end);
  d.hash = bd_hash_name(p + 32, l_name);
  d.q0 = l_seq > 0 ? (uint8_t)(bd_base_byte(seq4, qual, 0) >> 2) : 0;
  d.q1 = l_seq > 1 ? (uint8_t)(bd_base_byte(seq4, qual, 1) >> 2) : 0;
  d.status = BD_ST_USE;
}

/* ---- the plan: what becomes of every record that passed ----------------------------------------------------------------------
 * Arrays over the USED index u (the records with status BD_ST_USE, in file order; U[u] = record index).  A template SLOT is named
 * after the used record that created it (k->n++ in csrc/bamio.c); a pair-table ENTRY after the used record whose name it holds. */
typedef struct {
  const uint8_t *arena; /* inflated bytes; arena[0] is stream offset arena_base */
  uint64_t arena_base;
  const bd_desc *D;
  const uint32_t *U;
  uint32_t n_used;
  uint32_t *occ;       /* slot -> the used record whose positions / orientation / strand the template carries */
  uint32_t *side0, *side1; /* slot -> the used record whose read it holds on that side, or BD_NONE */
  uint32_t *waiting;   /* slot -> the entry al_hash_list keeps against it (its owner), or BD_NONE */
  uint32_t *ent_slot;  /* entry owner -> slot */
  uint8_t *ent_alive;  /* entry owner -> the entry is in the table */
  uint8_t *slot_made;  /* u created a slot */
  uint8_t *blk_open;   /* u opens a block */
  uint32_t *max_at;    /* max_pos after u (the block's y is its last record's) */
  uint32_t *slot_list; /* scratch: the slots created since the block / group began, in order */
  unsigned long long *cts, *bases; /* [15] each: filter_cts / filter_bases additions of the segmentation and pairing */
  unsigned long long *err;         /* [2]: BD_E_* and the used index it names */
  uint32_t *tab;       /* the replay's pair table: open addressing over entry owners + 1 (0 = empty) */
  uint32_t *tab_blk;   /* the block an entry belongs to: entries of finished blocks count as empty (name_clear) */
  uint32_t tab_mask;
} bd_ws;

#define BD_D(ws, u) ((ws).D[(ws).U[u]])

BD_FN const uint8_t *bd_name_ptr(const bd_ws &ws, uint32_t u) { return ws.arena + (BD_D(ws, u).off - ws.arena_base) + 36; }
BD_FN int bd_same_name(const bd_ws &ws, uint32_t a, uint32_t b) {
  const bd_desc &x = BD_D(ws, a), &y = BD_D(ws, b);
  if (x.hash != y.hash || x.l_name != y.l_name) return 0;
  const uint8_t *p = bd_name_ptr(ws, a), *q = bd_name_ptr(ws, b);
  for (uint32_t i = 0; i < x.l_name; i++)
    if (p[i] != q[i]) return 0;
  return 1;
}
BD_FN uint32_t bd_len_of(const bd_ws &ws, uint32_t u) { return u == BD_NONE ? 0u : BD_D(ws, u).l_seq; }

/* get_al_qual with the reference's sq[k] indexing (bsc_template_qual, csrc/prep.c): read k counts len[k] times the quality of ITS
 * byte k — byte 0 of read 0, byte 1 of read 1 */
BD_FN uint32_t bd_tplq(const bd_ws &ws, uint32_t s0, uint32_t s1) {
  uint32_t qual = 0, n = 0;
  if (s0 != BD_NONE && BD_D(ws, s0).l_seq) {
    qual += BD_D(ws, s0).l_seq * BD_D(ws, s0).q0;
    n += BD_D(ws, s0).l_seq;
  }
  if (s1 != BD_NONE && BD_D(ws, s1).l_seq) {
    qual += BD_D(ws, s1).l_seq * BD_D(ws, s1).q1;
    n += BD_D(ws, s1).l_seq;
  }
  return n > 0 ? qual / n : 0;
}

/* the pair table of the replay */
BD_FN uint32_t bd_tab_find(const bd_ws &ws, uint32_t u, uint32_t blk) {
  const uint64_t h = BD_D(ws, u).hash;
  for (uint32_t i = (uint32_t)h & ws.tab_mask;; i = (i + 1) & ws.tab_mask) {
    const uint32_t e = ws.tab[i];
    if (e == 0 || ws.tab_blk[i] != blk) return BD_NONE; /* empty, or left over from a finished block */
    const uint32_t o = e - 1;
    if (ws.ent_alive[o] && bd_same_name(ws, o, u)) return o;
  }
}
BD_FN void bd_tab_add(const bd_ws &ws, uint32_t u, uint32_t blk) {
  const uint64_t h = BD_D(ws, u).hash;
  for (uint32_t i = (uint32_t)h & ws.tab_mask;; i = (i + 1) & ws.tab_mask) {
    const uint32_t e = ws.tab[i];
    if (e == 0 || ws.tab_blk[i] != blk || !ws.ent_alive[e - 1]) { /* free, stale, or a removed entry's place */
      /* a removed entry's place may only be reused if the probe chain behind it is not cut: it is not — the place stays occupied */
      ws.tab[i] = u + 1;
      ws.tab_blk[i] = blk;
      return;
    }
  }
}

/* The table through which one position group looks (the parallel path): entries of the group's own records.  Names that could meet
 * an entry from elsewhere are `irregular` and never come here. */
BD_FN uint32_t bd_grp_find(const bd_ws &ws, uint32_t u, uint32_t g_first) {
  for (uint32_t v = g_first; v < u; v++)
    if (ws.ent_alive[v] && BD_D(ws, v).hash == BD_D(ws, u).hash && bd_same_name(ws, v, u)) return v;
  return BD_NONE;
}

typedef struct {
  uint32_t curr_pos, start_idx, n_slots; /* read_input's curr_pos / start_idx and k->n, counted over slot_list[list0 ..) */
  uint32_t list0;
  uint32_t start_pos;
  uint32_t blk;      /* the replay's block number (table epochs) */
  uint32_t g_first;  /* the group's first used index (parallel path) */
  int local;         /* 1: the parallel path's group table */
  unsigned long long *cts, *bases; /* where this run's filter_cts / filter_bases additions go ([15] each) */
} bd_run;

BD_FN int bd_kind(const bd_desc &d);
/* (the group table is asked only by records whose mate shares their position: every other name is alone in its group, or `irregular`) */
BD_FN uint32_t bd_find(const bd_ws &ws, const bd_run &r, uint32_t u) {
  if (!r.local) return bd_tab_find(ws, u, r.blk);
  return bd_kind(BD_D(ws, u)) == 'E' ? bd_grp_find(ws, u, r.g_first) : BD_NONE;
}
BD_FN void bd_add(const bd_ws &ws, const bd_run &r, uint32_t u, uint32_t slot) {
  ws.ent_alive[u] = 1;
  ws.ent_slot[u] = slot;
  if (!r.local) bd_tab_add(ws, u, r.blk);
}
BD_FN void bd_new_slot(const bd_ws &ws, bd_run &r, uint32_t u, uint32_t ix, uint32_t entry) {
  ws.occ[u] = u;
  ws.side0[u] = ix ? BD_NONE : u;
  ws.side1[u] = ix ? u : BD_NONE;
  ws.waiting[u] = entry;
  ws.slot_made[u] = 1;
  ws.slot_list[r.list0 + r.n_slots++] = u;
}

/* additions to the counters: the parallel path's lanes share them (atomics), the replay's one lane adds to its own block's (plain: they
 * live in its private memory, which flat atomics must not touch) */
#if defined(__HIP_DEVICE_COMPILE__)
#define BD_ADD64(p, v)                                      \
  do {                                                      \
    if (r.local) atomicAdd((p), (unsigned long long)(v));   \
    else *(p) += (unsigned long long)(v);                   \
  } while (0)
#define BD_ADD64_SHARED(p, v) atomicAdd((p), (unsigned long long)(v))
#else
#define BD_ADD64(p, v) (*(p) += (unsigned long long)(v))
#define BD_ADD64_SHARED(p, v) (*(p) += (unsigned long long)(v))
#endif

/*
 * What read_input does with used record u once its block is settled (csrc/bamio.c bsc_bam_next_block from "const int ix" on;
 * src/get_template_vector.c:223-372).  insert: the record is stored as a new template (a forward-facing read, or a lone one);
 * else it is the backwards-facing mate of a stored one.  *x_out: the position its alignment reaches when it is kept without a mate
 * (0 = no such contribution).  Returns BD_E_*.
 */
BD_FN int bd_step(const bd_ws &ws, const bd_params &par, bd_run &r, uint32_t u, int insert, uint32_t *x_out) {
  const bd_desc &d = BD_D(ws, u);
  const uint32_t ix = d.rev_ori & 1u;
  *x_out = 0;
  int append = 0;
  if (d.aflag & BD_F_PAIRED) {
    if (!insert) {
      const uint32_t q = bd_find(ws, r, u);
      if (q != BD_NONE) {
        const uint32_t s = ws.ent_slot[q];
        const bd_desc &o = BD_D(ws, ws.occ[s]);
        if (o.fwd != d.fwd || o.rev != d.rev) return BD_E_MATES_DISAGREE;
        if (ix) ws.side1[s] = u; else ws.side0[s] = u;
        ws.waiting[s] = BD_NONE;
        ws.ent_alive[q] = 0;
      } else {
        BD_ADD64(&r.cts[BD_FLT_PAIR_NOT_FOUND], 1);
        BD_ADD64(&r.bases[BD_FLT_PAIR_NOT_FOUND], d.l_seq);
        int skip = 0;
        if (!par.keep_duplicates && bd_st(d) >= r.start_pos) skip = 1;
        if (!skip && par.keep_unmatched) {
          *x_out = (d.fwd > 0 ? d.fwd : d.rev) + d.aln_len;
          append = 1;
        }
      }
    } else {
      int skip = 0;
      if (!par.keep_duplicates) {
        const uint32_t pos = d.fwd > 0 ? d.fwd : d.rev;
        if (pos == r.curr_pos) {
          for (uint32_t j = r.start_idx; j < r.n_slots; j++) {
            const uint32_t s = ws.slot_list[r.list0 + j];
            const bd_desc &o = BD_D(ws, ws.occ[s]);
            if (d.fwd != o.fwd || d.rev != o.rev || d.bs_strand != o.bs_strand) continue;
            int maxq1 = 0, kn1 = 0;
            const uint32_t h0 = ws.side0[s], h1 = ws.side1[s];
            if (bd_len_of(ws, h0) > 0) {
              maxq1 += BD_D(ws, h0).mapq;
              kn1++;
            }
            if (bd_len_of(ws, h1) > 0) {
              maxq1 += BD_D(ws, h1).mapq;
              kn1++;
            }
            if (kn1 == 0) return BD_E_NOSEQ_DUP; /* the reference divides by zero here */
            const int maxq = d.mapq;
            maxq1 /= kn1;
            const uint32_t l1 = bd_len_of(ws, h0), l2 = bd_len_of(ws, h1); /* of the template in the slot */
            uint32_t dl1 = ix ? 0u : d.l_seq, dl2 = ix ? d.l_seq : 0u;    /* of the one that goes: the newcomer unless it wins */
            if (maxq1 < maxq || (maxq == maxq1 && bd_tplq(ws, h0, h1) < bd_tplq(ws, ix ? BD_NONE : u, ix ? u : BD_NONE))) {
              uint32_t q = bd_find(ws, r, u);
              if (q != BD_NONE && ws.waiting[s] != BD_NONE) return BD_E_DUP_NAME;
              if (q == BD_NONE) q = ws.waiting[s];
              if (q != BD_NONE) ws.ent_alive[q] = 0;
              const int had_entry = ws.waiting[s] != BD_NONE;
              bd_add(ws, r, u, s);
              dl1 = l1;
              dl2 = l2;
              ws.occ[s] = u;
              ws.side0[s] = ix ? BD_NONE : u;
              ws.side1[s] = ix ? u : BD_NONE;
              ws.waiting[s] = had_entry ? u : BD_NONE; /* a fresh entry is not recorded against the template (:300-309) */
            }
            BD_ADD64(&r.cts[BD_FLT_DUPLICATE], (dl1 && dl2) ? 2 : 1);
            BD_ADD64(&r.bases[BD_FLT_DUPLICATE], (uint64_t)dl1 + dl2);
            skip = 1;
          }
        } else {
          r.curr_pos = pos;
          r.start_idx = r.n_slots;
        }
      }
      if (!skip) {
        if (bd_find(ws, r, u) != BD_NONE) return BD_E_DUP_NAME;
        bd_add(ws, r, u, u);
        bd_new_slot(ws, r, u, ix, u);
      }
    }
  } else {
    int skip = 0;
    if (!par.keep_duplicates) {
      const uint32_t pos = d.fwd > 0 ? d.fwd : d.rev;
      if (pos == r.curr_pos) {
        for (uint32_t j = r.start_idx; j < r.n_slots; j++) {
          const uint32_t s = ws.slot_list[r.list0 + j];
          const bd_desc &o = BD_D(ws, ws.occ[s]);
          const uint32_t w = ws.waiting[s];
          if (d.fwd != o.fwd || d.rev != o.rev || d.bs_strand != o.bs_strand) continue;
          if (!(w == BD_NONE || (BD_D(ws, w).aflag & 9u) == 9u || (BD_D(ws, w).aflag & 9u) == 0u)) continue;
          const uint32_t h0 = ws.side0[s], h1 = ws.side1[s];
          /* the reference compares mapq[0] of both, whichever read they hold (:357) */
          const uint32_t m1 = h0 != BD_NONE ? BD_D(ws, h0).mapq : 0u, mc = ix ? 0u : d.mapq;
          uint32_t dropped_len = d.l_seq; /* dropped.len[ix] */
          if (m1 < mc || (m1 == mc && bd_tplq(ws, h0, h1) < bd_tplq(ws, ix ? BD_NONE : u, ix ? u : BD_NONE))) {
            dropped_len = bd_len_of(ws, ix ? h1 : h0);
            ws.occ[s] = u;
            ws.side0[s] = ix ? BD_NONE : u;
            ws.side1[s] = ix ? u : BD_NONE;
          }
          BD_ADD64(&r.cts[BD_FLT_DUPLICATE], 1);
          BD_ADD64(&r.bases[BD_FLT_NONE], dropped_len); /* its bases are added to the PASSED column (:361-364) */
          skip = 1;
        }
      } else {
        r.curr_pos = pos;
        r.start_idx = r.n_slots;
      }
    }
    append = !skip;
  }
  if (append) bd_new_slot(ws, r, u, ix, BD_NONE);
  return BD_E_OK;
}

/* does a record that would be inserted open a new block, given the rightmost covered position so far (:139-147) */
BD_FN int bd_gap(const bd_desc &d, uint32_t max_pos) {
  if (d.fwd > 0) return d.fwd > max_pos && (d.rev > max_pos || d.rev == 0) && d.fwd - max_pos > 1;
  return d.rev > max_pos && d.rev - max_pos > 1;
}

/* kinds of a used record for the pair table: 0 = not paired, 'I' stored by its positions, 'N' the backwards-facing mate, 'E' both
 * mates at one position (decided by the table) */
BD_FN int bd_kind(const bd_desc &d) {
  if (!(d.aflag & BD_F_PAIRED)) return 0;
  if (d.fwd > 0 && d.rev > 0) {
    if (d.fwd == d.rev) return 'E';
    if (d.rev_ori & 1u) return d.fwd > d.rev ? 'I' : 'N';
    return d.fwd < d.rev ? 'I' : 'N';
  }
  return 'I';
}

/*
 * read_input over the used records, by one lane (csrc/bamio.c bsc_bam_next_block, statement for statement).  u0 .. n_used: the
 * records not yet handed out; `final`: the input has ended, so the block in hand is complete.  Fills the plan; *n_done = the used
 * records of complete blocks (the next call starts there).  Returns BD_E_* (err[1] names the record).
 */
BD_FN int bd_replay(const bd_ws &ws, const bd_params &par, uint32_t u0, int final, uint32_t *n_done) {
  int32_t curr_tid = -1;
  uint32_t max_pos = 0;
  uint32_t blk_first = u0;
  bd_run r;
  memset(&r, 0, sizeof r);
  r.blk = 1;
  r.list0 = u0;
  unsigned long long lc[15], lb[15]; /* the block in hand: committed when it is complete */
  for (int i = 0; i < 15; i++) lc[i] = lb[i] = 0;
  r.cts = lc;
  r.bases = lb;
  *n_done = u0;
  for (uint32_t u = u0; u < ws.n_used; u++) {
    const bd_desc &d = BD_D(ws, u);
    ws.slot_made[u] = 0;
    ws.ent_alive[u] = 0;
    ws.blk_open[u] = 0;
    int new_block = 0, new_contig = 0;
    if (curr_tid < 0 || curr_tid != d.tid) {
      new_contig = new_block = 1;
      curr_tid = d.tid;
      if (d.tid < 0 || d.tid >= par.n_ref) {
        ws.err[1] = u;
        return BD_E_TID;
      }
    }
    int insert = 1;
    if (!new_contig) {
      if ((d.aflag & BD_F_PAIRED) && d.fwd > 0 && d.rev > 0) {
        if (d.fwd == d.rev) insert = bd_tab_find(ws, u, r.blk) == BD_NONE;
        else if (d.rev_ori & 1u) insert = d.fwd > d.rev;
        else insert = d.fwd < d.rev;
      }
      if (insert && r.start_pos > 0 && bd_gap(d, max_pos)) new_block = 1;
    }
    if (new_block) {
      if (!insert) {
        ws.err[1] = u;
        return BD_E_MATE_OPENS_BLOCK;
      }
      for (int i = 0; i < 15; i++) {
        ws.cts[i] += lc[i];
        ws.bases[i] += lb[i];
        lc[i] = lb[i] = 0;
      }
      r.blk++; /* name_clear */
      r.curr_pos = 0;
      r.start_idx = 0;
      r.n_slots = 0;
      r.list0 = u;
      ws.blk_open[u] = 1;
      blk_first = u;
      *n_done = u; /* everything before u belongs to complete blocks */
      max_pos = r.start_pos = 0;
    }
    {
      const uint32_t st = bd_st(d), ml = st + d.span;
      if (ml > max_pos) max_pos = ml;
      if (r.start_pos == 0 || r.start_pos > st) r.start_pos = st;
    }
    uint32_t x = 0;
    const int e = bd_step(ws, par, r, u, insert, &x);
    if (e) {
      ws.err[1] = u;
      return e;
    }
    if (x > max_pos) max_pos = x;
    ws.max_at[u] = max_pos;
  }
  (void)blk_first;
  if (final) {
    *n_done = ws.n_used;
    for (int i = 0; i < 15; i++) {
      ws.cts[i] += lc[i];
      ws.bases[i] += lb[i];
    }
  }
  return BD_E_OK;
}

/* ---- the parallel path -------------------------------------------------------------------------------------------------------
 * Over used records u (lane per record unless said).  `irregular` (one word, any lane sets it): the input is not of the kind these
 * pieces are exact for, and bd_replay decides.
 *
 * 1. bd_f_key        run | ml packed for ONE inclusive max-scan: the contig run in the high half (runs only grow in file order), the
 *                    rightmost position the record covers in the low — the scan's value at u is max_pos after u of u's contig run.
 *                    Blocks of one contig never need a reset: a block's first record starts right of everything before it.
 * 2. bd_f_open       u opens a block: its contig run differs from its predecessor's, or it would be inserted and bd_gap holds against
 *                    the scan's value at u - 1.  Also the group starts: a block start, or a start position other than u - 1's.
 * 3. bd_f_group      one lane per group: bd_step over the group's records in order with the group's own table.
 * 4. bd_f_join       one lane per backwards-facing mate: its partner (the other record of its name, from the sort by name) is looked
 *                    up; found alive in its block it gives its read to the partner's slot.
 */
BD_FN uint64_t bd_f_key(uint32_t run, const bd_desc &d) { return (uint64_t)run << 32 | (uint64_t)(uint32_t)(bd_st(d) + d.span); }

/* is u the first record of a contig run (u > 0) */
BD_FN int bd_f_run_start(const bd_ws &ws, uint32_t u) { return u == 0 || BD_D(ws, u).tid != BD_D(ws, u - 1).tid; }

/* prev_max: the scan's low half at u - 1 (meaningful when u continues its contig run).  Returns flags: 1 block start, 2 group start */
BD_FN uint32_t bd_f_open(const bd_ws &ws, const bd_params &par, uint32_t u, uint32_t prev_max, uint32_t *irregular) {
  const bd_desc &d = BD_D(ws, u);
  const uint32_t st = bd_st(d);
  const int kind = bd_kind(d);
  if (d.tid < 0 || d.tid >= par.n_ref || st == 0 || d.l_seq == 0) *irregular = 1;
  if (kind != 'N' && (d.fwd > 0 ? d.fwd : d.rev) != st) *irregular = 1; /* the duplicate runs would not be the position groups */
  if (bd_f_run_start(ws, u)) return 3u;
  const bd_desc &p = BD_D(ws, u - 1);
  if (bd_st(p) > st) *irregular = 1; /* not sorted */
  const int insert = kind != 'N';
  if (insert && bd_gap(d, prev_max)) return 3u;
  return bd_st(p) != st ? 2u : 0u;
}

/* one group: used records g_first .. g_end - 1 (one block, one start position) */
BD_FN void bd_f_group(const bd_ws &ws, const bd_params &par, uint32_t g_first, uint32_t g_end, uint32_t blk_start_pos, uint32_t *irregular) {
  bd_run r;
  memset(&r, 0, sizeof r);
  r.local = 1;
  r.g_first = g_first;
  r.list0 = g_first;
  r.start_pos = blk_start_pos;
  r.cts = ws.cts;
  r.bases = ws.bases;
  for (uint32_t u = g_first; u < g_end; u++) {
    ws.slot_made[u] = 0;
    ws.ent_alive[u] = 0;
  }
  for (uint32_t u = g_first; u < g_end; u++) {
    const bd_desc &d = BD_D(ws, u);
    const int kind = bd_kind(d);
    const int first = bd_f_run_start(ws, u); /* a contig's first record is inserted whatever its positions say (:176-183) */
    if (kind == 'N' && !first) continue; /* bd_f_join's */
    int insert = 1;
    if (kind == 'E' && !first) insert = bd_grp_find(ws, u, g_first) == BD_NONE;
    uint32_t x = 0;
    if (bd_step(ws, par, r, u, insert, &x)) {
      *irregular = 1;
      return;
    }
  }
}

/* partner[u]: the other used record of u's name in u's contig run when there are exactly two, else BD_NONE; n_same[u]: how many there
 * are (1, 2, 3 = more).  blk_of[u]: u's block number.  The join of one backwards-facing mate. */
BD_FN void bd_f_join(const bd_ws &ws, const bd_params &par, uint32_t u, const uint32_t *partner, const uint32_t *blk_of, const uint64_t *scan,
                     uint32_t *win0, uint32_t *win1, uint32_t *irregular) {
  const bd_desc &d = BD_D(ws, u);
  if (bd_kind(d) != 'N' || bd_f_run_start(ws, u)) return;
  const uint32_t ix = d.rev_ori & 1u;
  const uint32_t p = partner[u];
  if (p != BD_NONE && p < u && blk_of[p] == blk_of[u] && ws.ent_alive[p]) {
    const uint32_t s = ws.ent_slot[p];
    const bd_desc &o = BD_D(ws, ws.occ[s]);
    if (o.fwd != d.fwd || o.rev != d.rev) {
      *irregular = 1; /* the reference stops here ("mates disagree"): the replay says so */
      return;
    }
    /* store_read; the last mate to arrive keeps the side (several entries may point at one slot, csrc/bamio.c "accidents") */
#if defined(__HIP_DEVICE_COMPILE__)
    atomicMax(ix ? &win1[s] : &win0[s], u + 1);
#else
    uint32_t *w = ix ? &win1[s] : &win0[s];
    if (*w < u + 1) *w = u + 1;
#endif
    return;
  }
  BD_ADD64_SHARED(&ws.cts[BD_FLT_PAIR_NOT_FOUND], 1);
  BD_ADD64_SHARED(&ws.bases[BD_FLT_PAIR_NOT_FOUND], d.l_seq);
  if (par.keep_duplicates && par.keep_unmatched) { /* kept as a template of its own (sorted input: `skip` otherwise) */
    const uint32_t x = (d.fwd > 0 ? d.fwd : d.rev) + d.aln_len;
    if (x > (uint32_t)scan[u]) *irregular = 1; /* it would move max_pos: the segmentation is not the scan's */
    ws.occ[u] = u;
    ws.side0[u] = ix ? BD_NONE : u;
    ws.side1[u] = ix ? u : BD_NONE;
    ws.waiting[u] = BD_NONE;
    ws.slot_made[u] = 1;
  }
}

/* the names of a contig run, sorted by (run, hash) with file order kept inside: chain = the used records sorted[c0 .. c1).  Writes
 * partner[] for chains of two, flags what the parallel path is not exact for. */
BD_FN void bd_f_chain(const bd_ws &ws, const uint32_t *sorted, uint32_t c0, uint32_t c1, const uint32_t *blk_of, uint32_t *partner,
                      uint32_t *irregular) {
  const uint32_t n = c1 - c0;
  if (n == 1) {
    partner[sorted[c0]] = BD_NONE;
    return;
  }
  if (n > 2) {
    for (uint32_t i = c0; i < c1; i++) partner[sorted[i]] = BD_NONE;
    *irregular = 1;
    return;
  }
  const uint32_t a = sorted[c0], b = sorted[c0 + 1]; /* a < b: the sort is stable */
  partner[a] = b;
  partner[b] = a;
  const int ka = bd_kind(BD_D(ws, a)), kb = bd_kind(BD_D(ws, b));
  if (!bd_same_name(ws, a, b)) { /* two names, one hash */
    *irregular = 1;
    return;
  }
  if (blk_of[a] != blk_of[b]) {
    if (ka == 'E' || kb == 'E') *irregular = 1;
    return;
  }
  if (ka == 'I' && kb == 'N') return;
  if (ka == 'E' && kb == 'E' && bd_st(BD_D(ws, a)) == bd_st(BD_D(ws, b))) return;
  *irregular = 1;
}

/* ---- assembly ---------------------------------------------------------------------------------------------------------------- */
typedef struct { /* bsc_raw_template, include/bscall_amd.h */
  uint32_t pos[2];
  uint32_t reference_span[2];
  uint32_t len[2];
  uint32_t n_misms[2];
  uint64_t off[2];
  uint64_t misms_off[2];
  uint8_t mapq[2];
  uint8_t orientation;
  uint8_t bs_strand;
  uint32_t pad_;
} bd_raw_template;
typedef struct {
  uint32_t type, position, size;
} bd_misms;

/* seq_off[u] / ms_off[u]: where used record u's read bytes / list entries go (prefix sums over the used records);
 * seq_base / ms_base: the block's first (subtracted, so that a block's templates index the block's own stretch) */
BD_FN void bd_template(const bd_ws &ws, uint32_t s, const uint64_t *seq_off, const uint64_t *ms_off, uint64_t seq_base, uint64_t ms_base,
                       bd_raw_template *t) {
  memset(t, 0, sizeof *t);
  const bd_desc &o = BD_D(ws, ws.occ[s]);
  t->pos[0] = o.fwd;
  t->pos[1] = o.rev;
  t->orientation = (uint8_t)((o.rev_ori >> 1) & 1u);
  t->bs_strand = o.bs_strand;
  for (int k = 0; k < 2; k++) {
    const uint32_t h = k ? ws.side1[s] : ws.side0[s];
    if (h == BD_NONE) continue;
    const bd_desc &d = BD_D(ws, h);
    t->off[k] = seq_off[h] - seq_base;
    t->len[k] = d.l_seq;
    t->misms_off[k] = ms_off[h] - ms_base;
    t->n_misms[k] = d.n_ms;
    t->mapq[k] = d.mapq;
    t->reference_span[k] = d.span;
  }
}

BD_FN void bd_misms_of(const uint8_t *rec, const bd_desc &d, bd_misms *out) {
  const uint8_t *cig = rec + 4 + 32 + d.l_name;
  uint32_t position = 0, nm = 0;
  for (uint32_t i = 0; i < d.n_cigar; i++) {
    const uint32_t c = bd_le32(cig + 4 * i), len = c >> 4;
    bd_misms m = {0, position, len};
    switch (c & 15u) {
      case 0: case 7: case 8: position += len; continue;
      case 6: case 4: m.type = 3; position += len; break; /* BSC_MISMS_SOFT */
      case 1: m.type = 2; position += len; break;         /* BSC_MISMS_DEL: inserted in the read */
      case 2: m.type = 1; break;                          /* BSC_MISMS_INS: deleted from the read */
      default: continue;
    }
    out[nm++] = m;
  }
}

#endif /* BSC_BAMDEV_CORE_H */
