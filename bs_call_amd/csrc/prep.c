/*
 * prep.c — read pre-processing on the host: from the templates the reader hands over (reads as base|qual<<2 bytes with
 * their CIGAR-derived mismatch lists) to the templates the pile-up accumulate stage consumes (bsc_template: byte j of
 * read k sits at genome position pos[k] + j).  This is what the reference's process thread does to every template of a
 * block before it calls call_genotypes_ML (src/process_template.c:36-111), in the same order:
 *   1. fixed trims      trim_read          src/read_utils.c:13-26     marks bases with q = FLT_QUAL (63); the right trim
 *                                                                     copies the BASE from the left end (sp[k1]), as the
 *                                                                     reference does
 *   2. soft clips       trim_soft_clips    src/al_utils.c:122-162     removes clipped bases, shifts the mismatch list
 *   3. mate overlap     handle_overlap     src/al_utils.c:164-318     keeps the mate with the longer reference span (mean
 *                                                                     quality breaks ties), trims the other to the
 *                                                                     non-overlapping part, walking its indels
 *   4. indel normalisation                 src/process_template.c:62-108   a deletion from the reference (INS in the
 *                                                                     reference's naming: CIGAR D) becomes bytes 0, an
 *                                                                     insertion (DEL: CIGAR I) is removed
 * plus get_al_qual (src/al_utils.c:19-35, the duplicate-resolution score, with its sq[k] indexing).
 * Host C, as in the reference; nothing here touches the GPU.  Where the reference aborts (gt_fatal_error_msg on an
 * illegal soft clip) the call returns BSC_ERR_ARG naming the template.
 */
#include <stdlib.h>
#include <string.h>

#include "../../include/bscall_amd.h"

int bsc_set_error(int code, const char *fmt, ...);

#define FLT_QUAL 63u
#define GET_QUAL(x) ((x) >> 2)

/* a read being edited: bytes in a private buffer with head-room for the padding of deletions */
typedef struct {
  uint8_t *p;
  uint32_t len;
} prep_read;

static void left_trim(prep_read *r, uint32_t l) { /* src/al_utils.c:103-113 */
  if (l > 0) {
    if (l >= r->len) r->len = 0;
    else {
      memmove(r->p, r->p + l, r->len - l);
      r->len -= l;
    }
  }
}

static void right_trim(prep_read *r, uint32_t l) { /* src/al_utils.c:115-120 */
  if (l > 0) {
    if (l >= r->len) r->len = 0;
    else r->len -= l;
  }
}

/* src/read_utils.c:13-26 */
static void trim_read(prep_read *r, int left, int right) {
  const uint32_t rl = r->len;
  if (rl > 0) {
    uint8_t *const sp = r->p;
    for (int k1 = 0; k1 < left && (uint32_t)k1 < rl; k1++) sp[k1] = (uint8_t)((sp[k1] & 3u) | (FLT_QUAL << 2));
    for (int k1 = 0; k1 < right && (uint32_t)k1 < rl; k1++) sp[rl - k1 - 1] = (uint8_t)((sp[k1] & 3u) | (FLT_QUAL << 2)); /* sp[k1]: sic */
  }
}

/* src/al_utils.c:19-35 */
uint32_t bsc_template_qual(const bsc_raw_template *t, const uint8_t *seq) {
  uint32_t qual = 0, n = 0;
  for (int k = 0; k < 2; k++) {
    if (t->len[k]) {
      const uint8_t *sq = seq + t->off[k];
      for (uint32_t j = 0; j < t->len[k]; j++) {
        const uint8_t q = (uint8_t)GET_QUAL(sq[k]); /* sq[k], not sq[j]: the reference's indexing */
        if (q != FLT_QUAL) {
          qual += q;
          n++;
        }
      }
    }
  }
  return n > 0 ? qual / n : 0;
}

static uint32_t mean_qual(const prep_read *r) { /* src/al_utils.c:191-203 */
  uint32_t tot = 0;
  int n = 0;
  for (uint32_t i = 0; i < r->len; i++) {
    const uint8_t q = (uint8_t)GET_QUAL(r->p[i]);
    if (q != FLT_QUAL) {
      tot += q;
      n++;
    }
  }
  return n > 0 ? tot / (uint32_t)n : 0;
}

/*
 * The non-CpG read profile (src/meth_profile.c:48-77): for every base of a prepared read, by its position in the
 * ORIGINAL read, whether it is a C / G of the reference outside a CpG and what the read shows there — the four counts
 * per read position from which the conversion rate along the read is estimated.  The reference's finite-state walk:
 * `state` = previous and current reference code, 3 bits each.
 */
static const uint8_t PROFILE_REF[64] = { /* src/meth_profile.c:14-23: 4 = C not followed by G, 8 = G not preceded by C */
    0, 0, 0, 0, 0, 0, 0, 0, /* N x */
    0, 0, 0, 8, 0, 0, 0, 0, /* A x */
    0, 4, 4, 0, 4, 0, 0, 0, /* C x */
    0, 0, 0, 8, 0, 0, 0, 0, /* G x */
    0, 0, 0, 8, 0, 0, 0, 0, /* T x */
    0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};

/* src/init_param.c:57-70: base | quality << 2 -> count index (bits 0-1), "a C / T observation" (4), "a G / A observation" (8);
 * zero for qualities outside MIN_QUAL .. FLT_QUAL - 1 */
static uint8_t profile_base(uint32_t bs_strand, uint8_t c) {
  static const uint8_t tab[3][4] = {{11, 6, 10, 7}, {11, 4, 10, 5}, {9, 6, 8, 7}};
  const unsigned q = GET_QUAL(c);
  if (q < 20u /* MIN_QUAL */ || q >= FLT_QUAL || bs_strand > 2) return 0;
  return tab[bs_strand][c & 3u];
}

static int profile_read(bsc_read_profile *pf, const uint8_t *sp, const int32_t *orig, uint32_t rl, uint32_t pos, uint32_t bs_strand,
                        uint32_t ti) {
  const uint32_t x = pf->x;
  if (pos < x || (uint64_t)pos - x + rl + 1 > pf->n_ref)
    return bsc_set_error(BSC_ERR_ARG, "bsc_prepare_templates: template %u lies outside the profile's reference (%u .. %u + %u)", ti, pos, x,
                         pf->n_ref);
  const uint8_t *rf = pf->ref + (pos - x);
  uint8_t state = pos > x ? (uint8_t)((pf->ref[pos - x - 1] << 3) | (*rf++)) : 0;
  uint8_t mask = PROFILE_REF[state & 63u];
  for (uint32_t j = 0; j < rl; j++) {
    const uint8_t xx = profile_base(bs_strand, sp[j]);
    uint64_t *cts = pf->counts + ((size_t)(orig[j] + 1)) * 4u; /* element 0 collects the padded deletions (orig = -1) */
    const uint8_t mask1 = (uint8_t)((xx & mask) >> 1);
    state = (uint8_t)(((state << 3) | (*rf++)) & 63u); /* pos >= x always holds here */
    mask = PROFILE_REF[state];
    cts[xx & 3u] += (uint64_t)((((xx & mask) | mask1) >> 2) & 1u);
  }
  return BSC_OK;
}

int bsc_prepare_templates(const bsc_raw_template *raw, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes, const bsc_misms *misms_in,
                          uint64_t n_misms_in, const bsc_prep_params *par, bsc_template *tpl_out, uint8_t *seq_out,
                          uint64_t seq_out_cap, uint64_t *seq_out_used, bsc_prep_stats *stats) {
  return bsc_prepare_templates_profile(raw, nr, seq, seq_bytes, misms_in, n_misms_in, par, tpl_out, seq_out, seq_out_cap, seq_out_used,
                                       stats, NULL);
}

int bsc_prepare_templates_profile(const bsc_raw_template *raw, uint32_t nr, const uint8_t *seq, uint64_t seq_bytes,
                                  const bsc_misms *misms_in, uint64_t n_misms_in, const bsc_prep_params *par, bsc_template *tpl_out,
                                  uint8_t *seq_out, uint64_t seq_out_cap, uint64_t *seq_out_used, bsc_prep_stats *stats,
                                  bsc_read_profile *pf) {
  if ((nr && (!raw || !tpl_out)) || !seq_out_used || !par) return bsc_set_error(BSC_ERR_ARG, "bsc_prepare_templates: NULL argument");
  if (pf && (!pf->ref || !pf->counts || pf->used > pf->cap)) return bsc_set_error(BSC_ERR_ARG, "bsc_prepare_templates: bad read profile");
  int32_t *origs[2] = {NULL, NULL}; /* per read: position of every prepared base in the original read */
  size_t cap_orig[2] = {0, 0};
  uint64_t used = 0;
  *seq_out_used = 0;
  bsc_prep_stats st;
  memset(&st, 0, sizeof st);
  /* scratch: one template's two reads and mismatch lists */
  size_t cap_rd[2] = {0, 0}, cap_ms[2] = {0, 0};
  uint8_t *buf[2] = {NULL, NULL};
  bsc_misms *ms[2] = {NULL, NULL};
  int rc = BSC_OK;
#define FAIL(...)                                   \
  do {                                              \
    rc = bsc_set_error(BSC_ERR_ARG, __VA_ARGS__);   \
    goto done;                                      \
  } while (0)
  for (uint32_t ti = 0; ti < nr; ti++) {
    const bsc_raw_template *t = raw + ti;
    if (t->orientation > 1) FAIL("bsc_prepare_templates: template %u has orientation %u", ti, t->orientation);
    prep_read rd[2];
    uint32_t nm[2];
    uint32_t pos[2] = {t->pos[0], t->pos[1]};
    for (int k = 0; k < 2; k++) {
      if (t->len[k] && (!seq || t->off[k] > seq_bytes || t->len[k] > seq_bytes - t->off[k]))
        FAIL("bsc_prepare_templates: read %d of template %u lies outside the read buffer", k, ti);
      if (t->n_misms[k] && (!misms_in || t->misms_off[k] > n_misms_in || t->n_misms[k] > n_misms_in - t->misms_off[k]))
        FAIL("bsc_prepare_templates: mismatch list %d of template %u lies outside the list buffer", k, ti);
      /* head-room: every deletion from the reference is padded in place */
      uint64_t pad = 0;
      for (uint32_t z = 0; z < t->n_misms[k]; z++)
        if (misms_in[t->misms_off[k] + z].type == BSC_MISMS_INS) pad += misms_in[t->misms_off[k] + z].size;
      const size_t need = (size_t)t->len[k] + (size_t)pad + 1;
      if (need > cap_rd[k]) {
        uint8_t *nb = realloc(buf[k], need * 2);
        if (!nb) {
          rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_prepare_templates: out of memory");
          goto done;
        }
        buf[k] = nb;
        cap_rd[k] = need * 2;
      }
      if ((size_t)t->n_misms[k] + 1 > cap_ms[k]) {
        bsc_misms *nb = realloc(ms[k], ((size_t)t->n_misms[k] + 1) * 2 * sizeof(bsc_misms));
        if (!nb) {
          rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_prepare_templates: out of memory");
          goto done;
        }
        ms[k] = nb;
        cap_ms[k] = ((size_t)t->n_misms[k] + 1) * 2;
      }
      if (t->len[k]) memcpy(buf[k], seq + t->off[k], t->len[k]);
      if (t->n_misms[k]) memcpy(ms[k], misms_in + t->misms_off[k], (size_t)t->n_misms[k] * sizeof(bsc_misms));
      rd[k].p = buf[k];
      rd[k].len = t->len[k];
      nm[k] = t->n_misms[k];
    }
    /* 1. fixed trims (src/process_template.c:39-41): read[0] is R1 on a FORWARD template, R2 on a REVERSE one */
    const int msk = t->orientation == 0 ? 0 : 1;
    if (par->left_trim[0] || par->right_trim[0]) trim_read(&rd[0 ^ msk], par->left_trim[0], par->right_trim[0]);
    if (par->left_trim[1] || par->right_trim[1]) trim_read(&rd[1 ^ msk], par->left_trim[1], par->right_trim[1]);
    /* 2. soft clips (src/al_utils.c:122-162) */
    uint32_t trim_l[2] = {0, 0}, trim_r[2] = {0, 0}; /* bases cut from either end (for the read profile's positions) */
    for (int k = 0; k < 2; k++) {
      const uint32_t rl = rd[k].len;
      if (rl == 0) continue;
      int nclip = 0;
      uint32_t adj = 0;
      const uint32_t n0 = nm[k];
      for (uint32_t z = 0; z < n0; z++) {
        bsc_misms *m = ms[k] + z;
        if (m->type == BSC_MISMS_SOFT) {
          if (z && z != n0 - 1) FAIL("bsc_prepare_templates: template %u read %d: soft clip not at an extremity of the read", ti, k);
          nclip++;
          if (!m->position) {
            if (m->size >= rl) FAIL("bsc_prepare_templates: template %u read %d: illegal soft clip (%u %u %u %u)", ti, k, z, m->position, m->size, rl);
            adj = m->size;
            st.base_clip += adj;
            trim_l[k] = adj;
            left_trim(&rd[k], adj);
          } else {
            if (m->position + m->size != rl) FAIL("bsc_prepare_templates: template %u read %d: illegal soft clip (%u %u %u %u)", ti, k, z, m->position, m->size, rl);
            right_trim(&rd[k], m->size);
            trim_r[k] = m->size;
            st.base_clip += m->size;
          }
        } else if (nclip) {
          m->position -= adj;
          ms[k][z - (uint32_t)nclip] = *m;
        }
      }
      if (nclip) nm[k] -= (uint32_t)nclip;
    }
    /* 3. mate overlap (src/al_utils.c:164-318) */
    uint32_t rdl[2] = {rd[0].len, rd[1].len};
    if (rdl[0] > 0 && rdl[1] > 0) {
      int rev;
      int32_t overlap;
      if (pos[0] <= pos[1]) {
        overlap = (int32_t)(t->reference_span[0] - pos[1] + pos[0]);
        rev = 0;
      } else {
        overlap = (int32_t)(t->reference_span[1] + pos[1] - pos[0]);
        rev = 1;
      }
      if (pos[0] + t->reference_span[0] >= pos[1]) {
        const uint32_t *rspan = t->reference_span;
        int tr; /* the read that is trimmed */
        if (rspan[0] > rspan[1]) tr = 1;
        else if (rspan[0] < rspan[1]) tr = 0;
        else tr = mean_qual(&rd[0]) <= mean_qual(&rd[1]) ? 0 : 1;
        const int right = (rev && tr) || !(rev || tr); /* trim the right end of read tr; else its left end */
        if (!right) { /* a left trim moves the read's start */
          if (tr) pos[1] += (uint32_t)overlap;
          else pos[0] += (uint32_t)overlap;
        }
        bsc_misms *mm = ms[tr];
        uint32_t num = nm[tr];
        if (!num) {
          if (right) right_trim(&rd[tr], (uint32_t)overlap);
          else left_trim(&rd[tr], (uint32_t)overlap);
        } else {
          int trimmed = 0;
          if (right) {
            const uint32_t xx = t->reference_span[tr] - (uint32_t)overlap;
            int64_t adj = 0;
            for (uint32_t z = 0; z < num; z++) {
              bsc_misms *m = mm + z;
              if ((int64_t)m->position + adj >= (int64_t)xx) {
                const int64_t trim = (int64_t)rdl[tr] - xx + adj;
                right_trim(&rd[tr], (uint32_t)trim);
                num = z;
                trimmed = 1;
                break;
              }
              if (m->type == BSC_MISMS_INS) {
                if ((int64_t)m->position + adj + m->size >= (int64_t)xx) {
                  const int64_t trim = (int64_t)rdl[tr] - m->position;
                  m->size = (uint32_t)((int64_t)xx - ((int64_t)m->position + adj));
                  right_trim(&rd[tr], (uint32_t)trim);
                  num = z + 1;
                  trimmed = 1; /* (no break in the reference: the walk goes on over the shortened list) */
                }
                adj += m->size;
              } else if (m->type == BSC_MISMS_DEL) adj -= m->size;
            }
            if (!trimmed) right_trim(&rd[tr], (uint32_t)overlap);
          } else {
            const uint32_t xx = (uint32_t)overlap;
            int64_t adj = 0;
            uint32_t z;
            for (z = 0; z < num; z++) {
              bsc_misms *m = mm + z;
              if ((int64_t)m->position + adj >= (int64_t)xx) {
                const uint32_t trim = (uint32_t)((int64_t)overlap - adj);
                left_trim(&rd[tr], trim);
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
                  left_trim(&rd[tr], trim);
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
              left_trim(&rd[tr], (uint32_t)((int64_t)overlap - adj));
              num = 0;
            }
          }
        }
        nm[tr] = num;
        st.base_overlap += (rdl[0] - rd[0].len) + (rdl[1] - rd[1].len);
        if (right) trim_r[tr] += rdl[tr] - rd[tr].len; /* src/al_utils.c:309-313 */
        else trim_l[tr] += rdl[tr] - rd[tr].len;
      }
    }
    /* 4. indel normalisation (src/process_template.c:62-108) and hand-over */
    bsc_template *o = tpl_out + ti;
    memset(o, 0, sizeof *o);
    o->pos[0] = pos[0];
    o->pos[1] = pos[1];
    o->mapq[0] = t->mapq[0];
    o->mapq[1] = t->mapq[1];
    o->orientation = t->orientation;
    o->bs_strand = t->bs_strand;
    int32_t max_pos = 0;
    uint32_t out_len2[2] = {0, 0};
    for (int k = 0; k < 2; k++) {
      const uint32_t rl = rd[k].len;
      uint8_t *sp = rd[k].p;
      if (pf && t->len[k]) { /* positions in the original read, read 1 counted from its far end (:76-87) */
        uint64_t pad = 0;
        for (uint32_t z = 0; z < nm[k]; z++)
          if (ms[k][z].type == BSC_MISMS_INS) pad += ms[k][z].size;
        const size_t want = (size_t)rl + (size_t)pad + 1;
        if (want > cap_orig[k]) {
          int32_t *nb = realloc(origs[k], want * 2 * sizeof(int32_t));
          if (!nb) {
            rc = bsc_set_error(BSC_ERR_NOMEM, "bsc_prepare_templates: out of memory");
            goto done;
          }
          origs[k] = nb;
          cap_orig[k] = want * 2;
        }
        int32_t *orig = origs[k];
        int32_t mpos;
        if (k) {
          const int32_t posx = (int32_t)(rl + trim_r[k]) - 1;
          for (uint32_t k1 = 0; k1 < rl; k1++) orig[k1] = posx - (int32_t)k1;
          mpos = posx;
        } else {
          const int32_t posx = (int32_t)trim_l[k];
          for (uint32_t k1 = 0; k1 < rl; k1++) orig[k1] = posx + (int32_t)k1;
          mpos = posx + (int32_t)rl;
        }
        if (mpos > max_pos) max_pos = mpos;
      }
      for (uint32_t k1 = 0; k1 < rl; k1++) { /* the base counters of the statistics (:50-59) */
        const uint8_t q = (uint8_t)GET_QUAL(sp[k1]);
        if (q == FLT_QUAL) st.base_trim++;
        else if ((int)q < par->min_qual) st.base_lowqual++;
        else st.base_none++;
      }
      if (t->len[k]) { /* the reference counts every read it was given a vector for (:57-58) */
        st.reads++;
        st.read_bases += rl;
      }
      uint32_t adj = 0;
      for (uint32_t z = 0; z < nm[k]; z++) {
        const bsc_misms *m = ms[k] + z;
        const uint32_t ix1 = m->position + adj;
        if (m->type == BSC_MISMS_INS) {
          if (ix1 > rl + adj) FAIL("bsc_prepare_templates: template %u read %d: indel beyond the read", ti, k);
          memmove(sp + ix1 + m->size, sp + ix1, rl + adj - ix1);
          memset(sp + ix1, 0, m->size);
          if (pf) {
            int32_t *orig = origs[k];
            memmove(orig + ix1 + m->size, orig + ix1, sizeof(int32_t) * (rl + adj - ix1));
            for (uint32_t k1 = 0; k1 < m->size; k1++) orig[ix1 + k1] = -1;
          }
          adj += m->size;
        } else if (m->type == BSC_MISMS_DEL) {
          if ((uint64_t)ix1 + m->size > (uint64_t)rl + adj) FAIL("bsc_prepare_templates: template %u read %d: indel beyond the read", ti, k);
          memmove(sp + ix1, sp + ix1 + m->size, rl + adj - ix1 - m->size);
          if (pf) memmove(origs[k] + ix1, origs[k] + ix1 + m->size, sizeof(int32_t) * (rl + adj - ix1 - m->size));
          adj -= m->size;
        }
      }
      const uint32_t out_len = rl + adj;
      out_len2[k] = out_len;
      if (used + out_len > seq_out_cap || (out_len && !seq_out)) {
        rc = bsc_set_error(BSC_ERR_ARG, "bsc_prepare_templates: seq_out too small (template %u)", ti);
        goto done;
      }
      if (out_len) memcpy(seq_out + used, sp, out_len);
      if (k == 0) o->flags = bsc_template_walk_flags(sp, out_len); /* every base was touched above: HOT LOOP A's scan comes for free */
      o->len[k] = out_len;
      o->off[k] = used;
      used += out_len;
    }
    if (pf) { /* meth_profile() (src/meth_profile.c:48-77), once both reads of the template are prepared */
      if ((uint32_t)max_pos + 1u > pf->used) {
        if ((uint32_t)max_pos + 2u > pf->cap) FAIL("bsc_prepare_templates: template %u: read position %d beyond the profile (%u)", ti, max_pos, pf->cap);
        /* gt_vector_reserve(.., true) clears everything behind the old end: a count left one past it by a reverse read is lost */
        memset(pf->counts + (size_t)pf->used * 4u, 0, (size_t)(pf->cap - pf->used) * 4u * sizeof(uint64_t));
        pf->used = (uint32_t)max_pos + 1u;
      }
      for (int k = 0; k < 2; k++) {
        if (!t->len[k] || !out_len2[k]) continue;
        if ((rc = profile_read(pf, rd[k].p, origs[k], out_len2[k], k ? pos[1] : pos[0], t->bs_strand, ti))) goto done;
      }
    }
  }
  *seq_out_used = used;
  if (stats) *stats = st;
done:
  free(buf[0]);
  free(buf[1]);
  free(ms[0]);
  free(ms[1]);
  free(origs[0]);
  free(origs[1]);
  return rc;
#undef FAIL
}

/* bsc_template.flags of a template whose read 0 is read0[0 .. len0): has it a base the scan of src/call_genotypes.c:198-211
 * stops at (quality neither 0 nor 63)? */
uint32_t bsc_template_walk_flags(const uint8_t *read0, uint32_t len0) {
  for (uint32_t j = 0; j < len0; j++) {
    const uint32_t q = (uint32_t)read0[j] >> 2;
    if (q != 0 && q != 63u) return BSC_TPL_WALK_KNOWN | BSC_TPL_WALKED0;
  }
  return BSC_TPL_WALK_KNOWN;
}

/* The block a list of prepared templates spans, as process_template_vector derives it before it calls
 * call_genotypes_ML (src/process_template.c:22-28): x = the first template's start - 2 (at least 1). */
uint32_t bsc_block_start(const bsc_raw_template *first) {
  uint32_t x = first->pos[0];
  if (x == 0) x = first->pos[1];
  return x > 2 ? x - 2 : 1;
}
