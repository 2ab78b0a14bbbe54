/*
 * orc_vcf.c — CPU restatement of the VCF record formation of bs_call's print thread.
 *
 * *** TEST INFRASTRUCTURE, NOT PRODUCT *** (same rules as bsc_oracle.c).
 *
 * Follows, statement by statement and with the same static sliding-window state:
 *   orc_vcf_push   <- print_vcf_entry      src/print_vcf.c:548-594   (5-site window, reference context)
 *   orc_vcf_flush  <- flush_vcf_entries    src/print_vcf.c:536-546
 *   orc_vcf_entry  <- _print_vcf_entry     src/print_vcf.c:32-381    (everything up to the htslib encoding)
 * and orc_vcf_block drives them like print_thread does for one block (src/process.c:87-104): every position of the
 * block in order, then a flush.  Instead of encoding a BCF record through htslib (absent here), the fields that would
 * be encoded are stored in an orc_vcf_core record.  The statistics block of _print_vcf_entry (:382-526) is restated in
 * orc_vcf_stats_update; dbSNP names, the JSON rendering and the header are not restated.
 *
 * The per-genotype lookup tables of the reference (ref_alt, all_idx, gt_int, gt_flag, cs_str, cflag, gflag) are
 * regenerated from their rule (alleles of the genotype vs. the reference base) in orc_vcf_tables_init and
 * spot-checked in tests/test_vcf_core.py.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#define ORC_LN10 (2.30258509299404568402)

/* include/bs_call.h:152-160 */
typedef struct {
  uint64_t counts[8];
  int32_t qual[8];
  double gt_prob[10];
  double fisher_strand;
  int32_t mq;
  int32_t aq;
  uint8_t max_gt;
} orc_gt_meth;

/* same layout as bsc_vcf_core (include/bscall_amd.h), 64 bytes */
typedef struct {
  uint32_t pos;     /* 1-based position */
  uint8_t emit;     /* 1: a record is written (the reference's !skip) */
  uint8_t gt;       /* called genotype 0..9 (argmax of gt_prob, first maximum) */
  uint8_t ref_code; /* reference base code 0..4 (N,A,C,G,T) as seen through the context window */
  uint8_t gt_enc;   /* the two GT allele codes, (allele+1)<<1 each, high and low nibble (reference gt_int) */
  uint8_t flt;      /* 1 q20, 2 qd2, 4 fs60, 8 mq40, 128 mac1 */
  uint8_t phred;    /* QUAL and GQ */
  uint8_t n_gl;     /* number of GL values */
  char cg;          /* FORMAT CG: one of "CG" -> 'C', 'H', 'N', '?', '.' (the reference writes the first char) */
  char alt[2];      /* ALT alleles (0-terminated/padded) */
  char cx_ref[5];   /* INFO CX: reference context */
  char cx_gt[5];    /* FORMAT CX: IUPAC context of the called genotypes */
  int32_t fs;       /* FORMAT FS (written only for heterozygous genotypes) */
  uint32_t qd;      /* FORMAT QD */
  uint32_t dp;      /* FORMAT DP = non-informative depth */
  float gl[6];      /* FORMAT GL */
  uint32_t _pad;
} orc_vcf_core;

int orc_sizeof_vcf_core(void) { return (int)sizeof(orc_vcf_core); }

/* same layout as bsc_site_stats (include/bscall_amd.h): the sum fields of bs_stats (include/bs_call.h:124-146) with its
 * gt_vectors and its coverage hash made dense */
#define ORC_COV_CAP 4096
typedef struct {
  uint64_t snps[2], indels[2], multi[2], dbSNP_sites[2], dbSNP_var[2], CpG_ref[2], CpG_nonref[2];
  uint64_t mut_counts[12][2], dbSNP_mut_counts[12][2];
  uint64_t qual[4][256];
  uint64_t filter_counts[2][32];
  uint64_t qd_stats[256][2], fs_stats[256][2], mq_stats[256][2];
  uint64_t cov[ORC_COV_CAP][6]; /* all, var, CpG[2], CpG_inf[2] */
  double CpG_ref_meth[2][101], CpG_nonref_meth[2][101];
} orc_site_stats;

int orc_sizeof_site_stats(void) { return (int)sizeof(orc_site_stats); }

double lgamma(double);
static double orc_vcf_lfact(uint32_t x, const double *lfact_store) { /* lfact2, include/bs_call.h:335 */
  return x < 256 ? lfact_store[x] : lgamma((double)x + 1.0);
}

/* genotype order AA AC AG AT CC CG CT GG GT TT; alleles as base codes 1..4 */
static const uint8_t G_A[10] = {1, 1, 1, 1, 2, 2, 2, 3, 3, 4};
static const uint8_t G_B[10] = {1, 2, 3, 4, 2, 3, 4, 3, 4, 4};

static char t_ref_alt[10][5][3];
static int t_all_idx[10][5][2];
static uint8_t t_gt_int[10][5];
static uint8_t t_gt_flag[10][5];
static uint8_t t_cflag[10], t_gflag[10], t_het[10];
static int t_init;

static void orc_vcf_tables_init(void) {
  static const char base[] = "NACGT";
  for (int g = 0; g < 10; g++) {
    const int a = G_A[g], b = G_B[g];
    t_het[g] = a != b;
    t_cflag[g] = a == 2 || b == 2;
    t_gflag[g] = a == 3 || b == 3;
    for (int r = 0; r < 5; r++) {
      int n = 0;
      memset(t_ref_alt[g][r], 0, 3);
      t_all_idx[g][r][0] = t_all_idx[g][r][1] = 0;
      if (a != r) { t_ref_alt[g][r][n] = base[a]; t_all_idx[g][r][n++] = a; }
      if (b != a && b != r) { t_ref_alt[g][r][n] = base[b]; t_all_idx[g][r][n++] = b; }
      if (a == b) t_gt_int[g][r] = a == r ? 0x22 : 0x44;
      else t_gt_int[g][r] = (a == r || b == r) ? 0x24 : 0x48;
      t_gt_flag[g][r] = (g == 0 && r == 1) || (g == 9 && r == 4);
    }
  }
  t_init = 1;
}

void orc_vcf_tables(char *ref_alt_10x5x3, int *all_idx_10x5x2, uint8_t *gt_int_10x5, uint8_t *gt_flag_10x5) {
  if (!t_init) orc_vcf_tables_init();
  memcpy(ref_alt_10x5x3, t_ref_alt, sizeof t_ref_alt);
  memcpy(all_idx_10x5x2, t_all_idx, sizeof t_all_idx);
  memcpy(gt_int_10x5, t_gt_int, sizeof t_gt_int);
  memcpy(gt_flag_10x5, t_gt_flag, sizeof t_gt_flag);
}

typedef struct {
  int all_positions;   /* sr_param.all_positions (-A) */
  uint32_t reg_start;  /* region clip: emit only reg_start <= x <= reg_stop (ctg->curr_reg or 1..ctg->end_pos) */
  uint32_t reg_stop;
} orc_vcf_params;

/* the printer's static state (src/print_vcf.c:529-533) */
typedef struct {
  char gt_store[5];
  uint32_t store_x;
  orc_gt_meth gtm_store[5];
  char rf_ctxt[8];
  uint8_t dbsnp_store[5]; /* rs_found of the stored positions (our addition: the reference looks it up at print time) */
  orc_vcf_core *out;
  uint32_t x0;
  const orc_vcf_params *par;
  /* statistics (NULL = par->work.stats == NULL) and the printer's static CpG state (src/print_vcf.c:105-106) */
  orc_site_stats *stats;
  const double *lfact_store, *logp;
  uint32_t prev_cpg_x;
  int prev_cpg_flt;
} orc_vcf_state;

/* stats_mut (include/bs_call.h:46) of reference base X -> allele Y, base codes 1..4 */
static int orc_mut_xy(int x, int y) { return 3 * (x - 1) + (y < x ? y - 1 : y - 2); }
/* mut_type[gt][rfix] (src/print_vcf.c:46-57), regenerated from its rule; 12 = mut_no */
static int orc_mut_type(int gt, int rfix) {
  const int a = G_A[gt], b = G_B[gt];
  if (rfix == 0) return 12;
  if (a == b) return a == rfix ? 12 : orc_mut_xy(rfix, a);
  if (a == rfix) return orc_mut_xy(rfix, b);
  if (b == rfix) return orc_mut_xy(rfix, a);
  return 12;
}

/* src/print_vcf.c:382-526.  `skip`, `flt`, `phred`, `qd`, `fs` as _print_vcf_entry has them at that point; by then `alt`
 * points at the terminator of the ALT string for every written record (the while loop at :177-181 walked it), so the
 * reference's `alt[0] != '.'` is always true and `alt[1] == ','` reads the byte behind the literal — taken as "not a
 * comma" here.  The coverage hash is the dense table `cov`, the per-contig copies and the GC histogram are left out. */
static void orc_vcf_stats_update(orc_vcf_state *st, const orc_gt_meth *gtm, uint32_t x, int gt, int rfix, int skip,
                                 uint32_t flt, int phred, uint32_t qd, int fs, const char *cpg, const char *prf_ctxt,
                                 uint32_t dp, uint32_t d_inf, uint8_t rs_found) {
  orc_site_stats *stats = st->stats;
  const uint64_t *counts = gtm->counts;
  uint64_t *gcov = stats->cov[dp < ORC_COV_CAP ? dp : ORC_COV_CAP - 1];
  gcov[0]++; /* all */
  if (skip) return;
  const int het = t_het[gt];
  const int snp = 1; /* see above */
  stats->snps[0]++;
  if (!flt) stats->snps[1]++;
  stats->qual[1][phred]++; /* variant_sites */
  gcov[1]++;               /* var */
  stats->qd_stats[qd > 255 ? 255 : qd][het]++;
  if (fs >= 0) stats->fs_stats[fs > 255 ? 255 : fs][het]++; /* a negative index is undefined behaviour in the reference */
  stats->mq_stats[gtm->mq < 0 ? 0 : (gtm->mq > 255 ? 255 : gtm->mq)][het]++;
  stats->filter_counts[het ? 1 : 0][flt & 31]++;
  stats->qual[0][phred]++; /* all_sites */
  if (rs_found) {
    stats->dbSNP_sites[0]++;
    if (snp) stats->dbSNP_var[0]++;
    if (!flt) {
      stats->dbSNP_sites[1]++;
      if (snp) stats->dbSNP_var[1]++;
    }
  }
  if (!strcmp(cpg, "CG")) {
    static const char *cs_str[10] = {"NA", "+", "-", "NA", "+", "+-", "+", "-", "-", "NA"};
    int ref_cpg = 0, cpg_ok = 0;
    uint32_t a = 0, b = 0;
    if (!strcmp(cs_str[gt], "+")) {
      st->prev_cpg_x = x;
      st->prev_cpg_flt = (flt != 0);
      if (!strncmp(prf_ctxt + 2, "CG", 2)) ref_cpg = 1;
      a = (uint32_t)counts[5];
      b = (uint32_t)counts[7];
      cpg_ok = 1;
    } else if (!strcmp(cs_str[gt], "-")) {
      if (!strncmp(prf_ctxt + 1, "CG", 2)) ref_cpg = 1;
      if (x - st->prev_cpg_x == 1) {
        if (ref_cpg) {
          stats->CpG_ref[0]++;
          if (!(st->prev_cpg_flt || flt)) stats->CpG_ref[1]++;
        } else {
          stats->CpG_nonref[0]++;
          if (!(st->prev_cpg_flt || flt)) stats->CpG_nonref[1]++;
        }
      }
      a = (uint32_t)counts[6];
      b = (uint32_t)counts[4];
      cpg_ok = 1;
    }
    if (cpg_ok) {
      stats->qual[ref_cpg ? 2 : 3][phred]++;
      gcov[ref_cpg ? 2 : 3]++;
      stats->cov[d_inf < ORC_COV_CAP ? d_inf : ORC_COV_CAP - 1][ref_cpg ? 4 : 5]++;
      if (a + b) {
        double meth[101];
        double konst = orc_vcf_lfact(a + b + 1, st->lfact_store) - orc_vcf_lfact(a, st->lfact_store) -
                       orc_vcf_lfact(b, st->lfact_store);
        double sum = 0.0;
        if (a) meth[0] = 0.0;
        else sum = meth[0] = exp(konst);
        if (b) meth[100] = 0.0;
        else sum = (meth[100] = exp(konst));
        double da = (double)a, db = (double)b;
        for (int i = 1; i < 100; i++) sum += (meth[i] = exp(konst + st->logp[i - 1] * da + st->logp[99 - i] * db));
        for (int i = 0; i < 101; i++) {
          double z = meth[i] / sum;
          if (ref_cpg) {
            stats->CpG_ref_meth[0][i] += z;
            if (!flt) stats->CpG_ref_meth[1][i] += z;
          } else {
            stats->CpG_nonref_meth[0][i] += z;
            if (!flt) stats->CpG_nonref_meth[1][i] += z;
          }
        }
      }
    }
  }
  const int mut = orc_mut_type(gt, rfix);
  if (mut != 12) {
    stats->mut_counts[mut][0]++;
    if (!flt) stats->mut_counts[mut][1]++;
    if (rs_found) {
      stats->dbSNP_mut_counts[mut][0]++;
      if (!flt) stats->dbSNP_mut_counts[mut][1]++;
    }
  }
}

/* src/print_vcf.c:32-381 without the htslib calls */
static void orc_vcf_entry(orc_vcf_state *st, const orc_gt_meth *gtm, const char *rf_ctxt, uint32_t x, const char *gt_store,
                          uint8_t rs_found) {
  static const char pbase[] = "NACGT";
  static const char iupac[] = "NAMRWCSYGKT";
  if (x == 0) return;
  const uint64_t *counts = gtm->counts;
  uint32_t dp = 0, d_inf = 0, dp1 = 0;
  for (int i = 0; i < 4; i++) dp1 += counts[i];
  for (int i = 4; i < 8; i++) d_inf += counts[i];
  dp = dp1 + d_inf;
  if (!dp) return;
  orc_vcf_core *o = st->out + (x - st->x0);
  memset(o, 0, sizeof *o);
  char prf_ctxt[5];
  for (int i = 0; i < 5; i++) prf_ctxt[i] = pbase[(int)rf_ctxt[i]];
  int rfix = (int)rf_ctxt[2];
  int gt = gt_store[2] - 1;
  int skip = (!st->par->all_positions && !(rs_found & 2) && t_gt_flag[gt][rfix]);
  double z = gtm->gt_prob[gt];
  int phred;
  double z1 = exp(z * ORC_LN10);
  if (z1 >= 1.0) phred = 255;
  else {
    phred = (int)(-10.0 * log(1.0 - z1) / ORC_LN10);
    if (phred > 255) phred = 255;
  }
  const char *alt = t_ref_alt[gt][rfix];
  const int fs = (int)(-gtm->fisher_strand * 10.0 + 0.5);
  const uint32_t qd = dp1 > 0 ? (uint32_t)phred / dp1 : (uint32_t)phred; /* int / uint32 is unsigned in the reference too */
  uint32_t flt = 0;
  if (!skip) skip = (x < st->par->reg_start || x > st->par->reg_stop);
  if (!skip) {
    if (phred < 20) flt |= 1;
    if (qd < 2) flt |= 2;
    if (fs > 60) flt |= 4;
    if (gtm->mq < 40) flt |= 8;
    if (!flt) {
      int mac1 = 0;
      switch (gt) {
        case 1: mac1 = (counts[1] + counts[5] + counts[7] <= 1 || counts[0] + counts[4] <= 1); break;
        case 2: mac1 = (counts[2] + counts[6] <= 1 || counts[0] <= 1); break;
        case 3: mac1 = (counts[3] + counts[7] <= 1 || counts[0] + counts[4] <= 1); break;
        case 5: mac1 = (counts[2] + counts[6] + counts[4] <= 1 || counts[1] + counts[5] + counts[7] <= 1); break;
        case 6: mac1 = (counts[3] <= 1 || counts[1] + counts[5] <= 1); break;
        case 8: mac1 = (counts[3] + counts[7] <= 1 || counts[2] + counts[6] + counts[4] <= 1); break;
      }
      if (mac1) flt |= 128;
    }
  }
  /* genotype context and CpG status (:227-266), computed whether or not the record is written */
  char ctxt[5];
  for (int i = 0; i < 5; i++) ctxt[i] = iupac[(int)gt_store[i]];
  const char *cpg = ".";
  if ((gt_store[2] == 5 && gt_store[3] == 8) || (gt_store[2] == 8 && gt_store[1] == 5)) cpg = "CG";
  else if (gt_store[2] == 5) {
    if (gt_store[3]) cpg = t_gflag[(int)gt_store[3] - 1] ? "H" : "N";
    else cpg = "?";
  } else if (gt_store[2] == 8) {
    if (gt_store[1]) cpg = t_cflag[(int)gt_store[1] - 1] ? "H" : "N";
    else cpg = "?";
  } else if (t_cflag[(int)gt_store[2] - 1]) {
    if (gt_store[3]) cpg = t_gflag[(int)gt_store[3] - 1] ? "H" : "N";
    else cpg = "?";
  } else if (t_gflag[(int)gt_store[2] - 1]) {
    if (gt_store[1]) cpg = t_cflag[(int)gt_store[1] - 1] ? "H" : "N";
    else cpg = ".";
  }
  o->pos = x;
  o->gt = (uint8_t)gt;
  o->ref_code = (uint8_t)rfix;
  /* the record of a position that is not written keeps no QUAL / QD (include/bscall_amd.h: bsc_vcf_core): the reference computes
   * them for every position (:140-152) but reads them behind `skip` only (:185-217,382-398) */
  o->phred = skip ? 0 : (uint8_t)phred;
  o->fs = fs;
  o->qd = skip ? 0 : qd;
  o->dp = dp1;
  o->cg = *cpg;
  memcpy(o->cx_ref, prf_ctxt, 5);
  memcpy(o->cx_gt, ctxt, 5);
  if (!skip) {
    o->emit = 1;
    o->flt = (uint8_t)flt;
    o->alt[0] = alt[0];
    o->alt[1] = alt[0] ? alt[1] : 0;
    o->gt_enc = t_gt_int[gt][rfix];
    /* GL (:319-347) */
    const int *aix = t_all_idx[gt][rfix];
    float gtl[6];
    if (rfix) {
      int j = rfix * (9 - rfix) / 2 + rfix - 5;
      z = gtm->gt_prob[j];
      if (z < -99.999) z = -99.999;
    } else z = -99.999;
    gtl[0] = z;
    int n_gt = 1;
    for (int i = 0; i < 2 && aix[i] > 0; i++) {
      int j;
      if (rfix) {
        if (rfix < aix[i]) j = rfix * (9 - rfix) / 2 + aix[i] - 5;
        else j = aix[i] * (9 - aix[i]) / 2 + rfix - 5;
        z = gtm->gt_prob[j];
        if (z < -99.999) z = -99.999;
        gtl[n_gt++] = z;
      }
      /* (the reference's inner loop over k < i computes a value it never stores) */
      j = aix[i] * (9 - aix[i]) / 2 + aix[i] - 5;
      z = gtm->gt_prob[j];
      if (z < -99.999) z = -99.999;
      gtl[n_gt++] = z;
    }
    o->n_gl = (uint8_t)n_gt;
    for (int i = 0; i < n_gt; i++) o->gl[i] = gtl[i];
  }
  if (st->stats) orc_vcf_stats_update(st, gtm, x, gt, rfix, skip, flt, phred, qd, fs, cpg, prf_ctxt, dp, d_inf, rs_found);
}

/* src/print_vcf.c:548-594 */
static void orc_vcf_push(orc_vcf_state *st, const orc_gt_meth *gtm, const char *rf, uint32_t x, uint32_t xstart, int skip,
                         uint8_t rs_found) {
  uint32_t l = x - st->store_x;
  if (l < 5) {
    memmove(st->gt_store, st->gt_store + l, 5 - l);
    memmove(st->gtm_store, st->gtm_store + l, (5 - l) * sizeof(orc_gt_meth));
    memmove(st->dbsnp_store, st->dbsnp_store + l, 5 - l);
    for (uint32_t i = 4; i >= 5 - l; i--) st->gt_store[i] = 0;
  } else memset(st->gt_store, 0, 5);
  st->store_x = x;
  memcpy(st->gtm_store + 4, gtm, sizeof(orc_gt_meth));
  st->dbsnp_store[4] = rs_found;
  if (x - xstart >= 4) strncpy(st->rf_ctxt, rf + x - xstart - 4, 7); /* NB: strncpy stops at a 0 (= N) byte */
  else {
    uint32_t l2 = x - xstart;
    for (uint32_t i = 0; i < 4 - l2; i++) st->rf_ctxt[i] = 0;
    strncpy(st->rf_ctxt + 4 - l2, rf, 3 + l2);
  }
  if (skip) st->gt_store[4] = 0;
  else {
    double z = gtm->gt_prob[0];
    int gt = 0;
    for (int i = 1; i < 10; i++)
      if (gtm->gt_prob[i] > z) {
        z = gtm->gt_prob[i];
        gt = i;
      }
    st->gt_store[4] = gt + 1;
  }
  if (st->gt_store[2]) orc_vcf_entry(st, st->gtm_store + 2, st->rf_ctxt, x - 2, st->gt_store, st->dbsnp_store[2]);
}

/* src/print_vcf.c:536-546 */
static void orc_vcf_flush(orc_vcf_state *st) {
  if (st->store_x) {
    for (int i = 0; i < 2; i++) {
      memmove(st->gt_store, st->gt_store + 1, 4);
      memmove(st->gtm_store, st->gtm_store + 1, 4 * sizeof(orc_gt_meth));
      memmove(st->dbsnp_store, st->dbsnp_store + 1, 4);
      memmove(st->rf_ctxt, st->rf_ctxt + 1, 6);
      if (st->gt_store[2]) orc_vcf_entry(st, st->gtm_store + 2, st->rf_ctxt, st->store_x - 1 + i, st->gt_store, st->dbsnp_store[2]);
    }
    st->store_x = 0;
  }
}

/*
 * One block as the print thread handles it (src/process.c:87-104): gtm[i], skip[i] for positions x .. x+n-1, ref = the
 * reference codes of x .. x+n+1 (work->ref holds x .. y+2) as a C string is used by the reference (strncpy!), dbsnp[i] the
 * rs_found value (0/1/3) of each position or NULL.  out[n] is zeroed; positions that produce no call stay zero
 * (emit = 0, pos = 0).
 */
void orc_vcf_block(const orc_gt_meth *gtm, const uint8_t *skip, const char *ref, uint32_t n, uint32_t x,
                   const orc_vcf_params *par, const uint8_t *dbsnp, orc_vcf_core *out) {
  if (!t_init) orc_vcf_tables_init();
  orc_vcf_state st;
  memset(&st, 0, sizeof st);
  st.out = out;
  st.x0 = x;
  st.par = par;
  memset(out, 0, (size_t)n * sizeof *out);
  for (uint32_t i = 0; i < n; i++) orc_vcf_push(&st, gtm + i, ref, x + i, x, skip[i], dbsnp ? dbsnp[i] : 0);
  orc_vcf_flush(&st);
}

/*
 * The same with par->work.stats != NULL: the block's contributions are added to *stats.  lfact_store[256] and
 * logp[100] are the reference's tables (src/stats_utils.c:14-21, src/init_param.c:56); carry[2] = {prev_cpg_x,
 * prev_cpg_flt}, the printer's static CpG state, read before and written after the block.
 */
void orc_vcf_block_stats(const orc_gt_meth *gtm, const uint8_t *skip, const char *ref, uint32_t n, uint32_t x,
                         const orc_vcf_params *par, const uint8_t *dbsnp, orc_vcf_core *out, orc_site_stats *stats,
                         const double *lfact_store, const double *logp, uint32_t *carry) {
  if (!t_init) orc_vcf_tables_init();
  orc_vcf_state st;
  memset(&st, 0, sizeof st);
  st.out = out;
  st.x0 = x;
  st.par = par;
  st.stats = stats;
  st.lfact_store = lfact_store;
  st.logp = logp;
  st.prev_cpg_x = carry[0];
  st.prev_cpg_flt = (int)carry[1];
  memset(out, 0, (size_t)n * sizeof *out);
  for (uint32_t i = 0; i < n; i++) orc_vcf_push(&st, gtm + i, ref, x + i, x, skip[i], dbsnp ? dbsnp[i] : 0);
  orc_vcf_flush(&st);
  carry[0] = st.prev_cpg_x;
  carry[1] = (uint32_t)st.prev_cpg_flt;
}
