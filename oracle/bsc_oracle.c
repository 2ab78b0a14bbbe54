/*
 * bsc_oracle.c — CPU restatement of bs_call's per-site calling path.
 *
 * *** TEST INFRASTRUCTURE, NOT PRODUCT. ***  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The product (bs_call_amd/) never links,
 * imports or calls anything in oracle/.
 *
 * What it restates (reference heathsc/bs_call v2.1.7, paths relative to the reference root):
 *   orc_tables_init      <- fill_base_prob_table   src/genotype_model.c:8-21
 *                           lfact_store_init       src/stats_utils.c:14-21
 *                           model defaults         src/init_param.c:16,26-31 ; include/bs_call.h:14-42
 *   orc_accumulate       <- HOT LOOP A             src/call_genotypes.c:17-19,180-226
 *   orc_site_*           <- HOT LOOP B body        src/call_genotypes.c:44-113
 *   orc_calc_gt_prob_*   <- calc_gt_prob / get_Z   src/genotype_model.c:23-246
 *   orc_fisher_*         <- fisher / lfact2        src/stats_utils.c:25-91 ; include/bs_call.h:335
 *   orc_call_sites       <- the calc-thread pool with the reference's interleaved striding
 *                                                  src/call_genotypes.c:36-43,124-138,260-272
 *
 * Two flavours of the transcendental functions, same statements otherwise:
 *   flavour 0 "libm": log/exp/lgamma from libm, exactly what the reference links.  This is the
 *                     restatement of the reference and the CPU baseline that bench.py times.
 *   flavour 1 "bsm" : the explicit-FMA replicas of glibc's log/exp/lgamma in bs_call_amd/csrc/bsmath.h.  The gfx950
 *                     kernels use the same header, so GPU output must equal this flavour bit for bit — and on a host
 *                     whose libm is glibc's FMA variant the two flavours are themselves identical (tests/test_bsmath.py).
 *
 * Pinning status: the reference's own sources for this path all include include/bs_call.h, which
 * includes <htslib/sam.h>, <htslib/vcf.h>, <htslib/faidx.h>; htslib is not in this image and writing
 * stand-in headers is not allowed, so the reference is NOT built here (no oracle/_ref).  The libm
 * flavour is pinned bit-for-bit by the ten known-answer vectors that SURVEY.md section 8c records from
 * the reference itself (tests/golden/kav_survey8c.json, tests/test_oracle_kav.py) and cross-checked
 * against an independently written pure-Python restatement (oracle/py_model.py).
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../bs_call_amd/csrc/bsmath.h"

#define ORC_MAX_QUAL 43          /* include/bs_call.h:27 */
#define ORC_FLT_QUAL 63          /* include/bs_call.h:30 */
#define ORC_LFACT_STORE_SIZE 256 /* include/bs_call.h:39 */
#define ORC_LN10 (2.30258509299404568402) /* LOG10, include/bs_call.h:36 */

/* include/bs_call.h:174-182, 104 bytes */
typedef struct {
  uint32_t counts[2][8];
  uint32_t n;
  float quality[8];
  float mapq2;
} orc_pileup;

/* include/bs_call.h:152-160, 200 bytes */
typedef struct {
  uint64_t counts[8];
  int32_t qual[8];
  double gt_prob[10];
  double fisher_strand;
  int32_t mq;
  int32_t aq;
  uint8_t max_gt;
} orc_gt_meth;

/* include/bs_call.h:148-150 */
typedef struct {
  double e, k, ln_k, ln_k_half, ln_k_one;
} orc_qual_prob;

typedef struct {
  orc_qual_prob q_prob[ORC_MAX_QUAL + 1];
  double lfact_store[ORC_LFACT_STORE_SIZE];
  double under_conv, over_conv, ref_bias;
  double lrb, lrb1; /* log(ref_bias), log(0.5*(1+ref_bias)): src/genotype_model.c:88-89 */
  int32_t min_qual;
  int32_t _pad;
} orc_tables;

/* One template (read pair) flattened out of align_details (include/bs_call.h:64-73): read k holds
 * len[k] bytes base|qual<<2 at seq + off[k], aligned so that byte j sits at pos[k] + j. */
typedef struct {
  uint32_t pos[2]; /* forward_position, reverse_position; 0 = none */
  uint32_t len[2]; /* 0 = read absent */
  uint64_t off[2];
  uint8_t mapq[2];
  uint8_t orientation; /* 0 FORWARD, 1 REVERSE */
  uint8_t bs_strand;   /* 0 NON_CONVERTED, 1 STRAND_C2T, 2 STRAND_G2A */
  uint32_t _pad;
} orc_template;

/* src/init_param.c:16 */
static const uint8_t orc_gt_het[10] = {0, 1, 1, 1, 0, 1, 1, 0, 1, 0};
/* src/call_genotypes.c:17-19 */
static const int8_t orc_base_tab_st[3][4] = {{1, 2, 3, 4}, {1, 6, 3, 8}, {5, 2, 7, 4}};

int orc_sizeof_pileup(void) { return (int)sizeof(orc_pileup); }
int orc_sizeof_gt_meth(void) { return (int)sizeof(orc_gt_meth); }
int orc_sizeof_tables(void) { return (int)sizeof(orc_tables); }
int orc_sizeof_template(void) { return (int)sizeof(orc_template); }

void orc_tables_init(orc_tables *tb, double under_conv, double over_conv, double ref_bias, int min_qual) {
  memset(tb, 0, sizeof *tb);
  for (int q = 0; q <= ORC_MAX_QUAL; q++) { /* src/genotype_model.c:10-21 */
    double e = exp(-.1 * (double)q * ORC_LN10);
    if (e > .5) e = .5;
    double k = e / (3.0 - 4.0 * e);
    tb->q_prob[q].e = e;
    tb->q_prob[q].k = k;
    tb->q_prob[q].ln_k = log(k);
    tb->q_prob[q].ln_k_half = log(0.5 + k);
    tb->q_prob[q].ln_k_one = log(1.0 + k);
  }
  tb->lfact_store[0] = tb->lfact_store[1] = 0.0; /* src/stats_utils.c:14-21 */
  double l = 0.0;
  for (int i = 2; i < ORC_LFACT_STORE_SIZE; i++) {
    l += log((double)i);
    tb->lfact_store[i] = l;
  }
  tb->under_conv = under_conv;
  tb->over_conv = over_conv;
  tb->ref_bias = ref_bias;
  tb->lrb = log(ref_bias);
  tb->lrb1 = log(0.5 * (1.0 + ref_bias));
  if (min_qual < 1) min_qual = 1; /* src/parse_args.c:170-171 */
  if (min_qual > ORC_MAX_QUAL) min_qual = ORC_MAX_QUAL;
  tb->min_qual = min_qual;
}

/* ---- flavour 0: libm ---------------------------------------------------------------------- */
#define ORC_SUF _libm
#define ORC_LOG(x) log(x)
#define ORC_EXP(x) exp(x)
#define ORC_LFACT_BIG(x) lgamma((double)((x) + 1))
#include "orc_model.inc"
#undef ORC_SUF
#undef ORC_LOG
#undef ORC_EXP
#undef ORC_LFACT_BIG

/* ---- flavour 1: bsmath (the kernels' twin) ------------------------------------------------- */
#define ORC_SUF _bsm
#define ORC_LOG(x) bsm_log(x)
#define ORC_EXP(x) bsm_exp(x)
#define ORC_LFACT_BIG(x) bsm_lfact_big(x)
#include "orc_model.inc"
#undef ORC_SUF
#undef ORC_LOG
#undef ORC_EXP
#undef ORC_LFACT_BIG

/* scalar entry points for the known-answer tests */
double orc_log(double x, int flavour) { return flavour ? bsm_log(x) : log(x); }
double orc_exp(double x, int flavour) { return flavour ? bsm_exp(x) : exp(x); }
double orc_lfact(int x, const orc_tables *tb, int flavour) {
  return flavour ? orc_lfact_bsm(x, tb->lfact_store) : orc_lfact_libm(x, tb->lfact_store);
}
void orc_log_array(const double *x, double *y, uint64_t n, int flavour) {
  for (uint64_t i = 0; i < n; i++) y[i] = flavour ? bsm_log(x[i]) : log(x[i]);
}
void orc_exp_array(const double *x, double *y, uint64_t n, int flavour) {
  for (uint64_t i = 0; i < n; i++) y[i] = flavour ? bsm_exp(x[i]) : exp(x[i]);
}
void orc_calc_gt_prob(orc_gt_meth *gt, const orc_tables *tb, char rf, int flavour) {
  if (flavour) orc_calc_gt_prob_bsm(gt, tb, rf);
  else orc_calc_gt_prob_libm(gt, tb, rf);
}
/* calc_gt_prob() over an array of prepared gt_meth records (counts, qual filled): the "model only" CPU timing of
 * SURVEY.md section 8d (i).  Single thread. */
void orc_calc_gt_prob_array(orc_gt_meth *gt, const char *ref, uint64_t n, const orc_tables *tb, int flavour) {
  if (flavour)
    for (uint64_t i = 0; i < n; i++) orc_calc_gt_prob_bsm(gt + i, tb, ref[i]);
  else
    for (uint64_t i = 0; i < n; i++) orc_calc_gt_prob_libm(gt + i, tb, ref[i]);
}

/* c[4] is modified, as in the reference */
double orc_fisher(int *c, const orc_tables *tb, int flavour) {
  return flavour ? orc_fisher_bsm(c, tb->lfact_store) : orc_fisher_libm(c, tb->lfact_store);
}

/* ---- HOT LOOP A: src/call_genotypes.c:178-226 ------------------------------------------------ */
/* counts[] has y-x+1 entries and is zeroed here (:178).  Returns 0, or -1 where the reference asserts
 * (:158 y>=x, :186 x1>=x, :188 ori<2). */
int orc_accumulate(const orc_template *tpl, uint32_t nr, const uint8_t *seq, uint32_t x, uint32_t y, int min_qual,
                   orc_pileup *counts) {
  if (y < x) return -1;
  uint32_t sz = y - x + 1;
  memset(counts, 0, sizeof(orc_pileup) * (size_t)sz);
  for (uint32_t ix = 0; ix < nr; ix++) {
    const orc_template *al = tpl + ix;
    uint32_t x1 = al->pos[0];
    if (x1 == 0) x1 = al->pos[1];
    else if (al->pos[1] > 0 && al->pos[1] < x1) x1 = al->pos[1];
    if (x1 < x) return -1;
    int ori = al->orientation;
    if (ori >= 2) return -1;
    int st = al->bs_strand;
    for (int k = 0; k < 2; k++) {
      uint32_t rl = al->len[k];
      if (rl == 0) continue;
      float mapq2 = al->mapq[k] * al->mapq[k];
      const uint8_t *sp = seq + al->off[k];
      uint32_t pos = al->pos[k];
      uint32_t j;
      for (j = 0; j < rl; j++) {
        uint8_t q = sp[j] >> 2;
        if (q > 0 && q != ORC_FLT_QUAL) break;
      }
      uint32_t read_start;
      if (j < rl) read_start = j;
      else continue; /* note: `continue` skips the ori flip, as in the reference (:204) */
      for (j = rl; j > 0; j--) {
        uint8_t q = sp[j - 1] >> 2;
        if (q > 0 && q != ORC_FLT_QUAL) break;
      }
      uint32_t read_end;
      if (j > 0) read_end = j - 1;
      else continue;
      pos += read_start;
      orc_pileup *curr_loc = counts + (pos - x);
      for (j = read_start; j <= read_end && pos <= y; j++, pos++, curr_loc++) {
        int c = orc_base_tab_st[st][sp[j] & 3] - 1;
        uint8_t q = sp[j] >> 2;
        if (q >= min_qual && q != ORC_FLT_QUAL) {
          curr_loc->n++;
          curr_loc->quality[c] += (float)q;
          curr_loc->mapq2 += mapq2;
          curr_loc->counts[ori][c]++;
        }
      }
      ori ^= 1;
    }
  }
  return 0;
}

/* ---- HOT LOOP B over a block, with the reference's worker striding ---------------------------- */
typedef struct {
  const orc_pileup *cts;
  const char *ref;
  orc_gt_meth *out;
  uint8_t *skip;
  const orc_tables *tb;
  uint64_t n;     /* end of this worker's range (exclusive) */
  uint64_t first; /* first site of this worker */
  int step, flavour;
} orc_job;

static void orc_run_libm(const orc_job *jb) {
  for (uint64_t i = jb->first; i < jb->n; i += jb->step) {
    memset(jb->out + i, 0, sizeof(orc_gt_meth));
    jb->skip[i] = (uint8_t)orc_site_libm(jb->cts + i, jb->out + i, jb->ref[i], jb->tb);
  }
}

__attribute__((target_clones("fma", "default"))) static void orc_run_bsm(const orc_job *jb) {
  for (uint64_t i = jb->first; i < jb->n; i += jb->step) {
    memset(jb->out + i, 0, sizeof(orc_gt_meth));
    jb->skip[i] = (uint8_t)orc_site_bsm(jb->cts + i, jb->out + i, jb->ref[i], jb->tb);
  }
}

static void *orc_worker(void *arg) {
  const orc_job *jb = arg;
  if (jb->flavour) orc_run_bsm(jb);
  else orc_run_libm(jb);
  return NULL;
}

/* out[i] is zeroed then filled; skip[i] = 1 where n == 0 (src/call_genotypes.c:44,112).  ref[i] is the
 * reference code 0..4 of site i (N,A,C,G,T: src/get_sequence.c:20-54).
 * nthreads > 0: the reference's scheme, worker t handles sites t, t+T, ... (src/call_genotypes.c:264-270);
 * nthreads < 0: |nthreads| workers on contiguous ranges (no false sharing of the 104/200-byte records) —
 *               the friendlier CPU baseline; results are identical either way. */
int orc_call_sites(const orc_pileup *cts, const char *ref, uint64_t n, const orc_tables *tb, orc_gt_meth *out,
                   uint8_t *skip, int flavour, int nthreads) {
  int blocked = nthreads < 0;
  if (blocked) nthreads = -nthreads;
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  orc_job jobs[256];
  pthread_t thr[256];
  for (int t = 0; t < nthreads; t++) {
    orc_job jb = {cts, ref, out, skip, tb, n, (uint64_t)t, nthreads, flavour};
    if (blocked) {
      jb.first = n * (uint64_t)t / (uint64_t)nthreads;
      jb.n = n * (uint64_t)(t + 1) / (uint64_t)nthreads;
      jb.step = 1;
    }
    jobs[t] = jb;
  }
  if (nthreads == 1) {
    orc_worker(&jobs[0]);
    return 0;
  }
  for (int t = 0; t < nthreads; t++)
    if (pthread_create(&thr[t], NULL, orc_worker, &jobs[t])) return -1;
  for (int t = 0; t < nthreads; t++) pthread_join(thr[t], NULL);
  return 0;
}

/* ---- timing helpers for bench.py's cpu_baseline leg (the three CPU timings) ------------------ */
typedef struct {
  orc_gt_meth *gt;
  const char *ref;
  uint64_t lo, hi;
  const orc_tables *tb;
  int flavour;
} orc_model_job;

static void *orc_model_worker(void *arg) {
  const orc_model_job *jb = arg;
  orc_calc_gt_prob_array(jb->gt + jb->lo, jb->ref + jb->lo, jb->hi - jb->lo, jb->tb, jb->flavour);
  return NULL;
}

/* (i) calc_gt_prob() only over prepared records, nthreads workers on contiguous ranges */
int orc_calc_gt_prob_array_mt(orc_gt_meth *gt, const char *ref, uint64_t n, const orc_tables *tb, int flavour, int nthreads) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  orc_model_job jobs[256];
  pthread_t thr[256];
  for (int t = 0; t < nthreads; t++) {
    orc_model_job jb = {gt, ref, n * (uint64_t)t / (uint64_t)nthreads, n * (uint64_t)(t + 1) / (uint64_t)nthreads, tb, flavour};
    jobs[t] = jb;
  }
  if (nthreads == 1) {
    orc_model_worker(&jobs[0]);
    return 0;
  }
  for (int t = 0; t < nthreads; t++)
    if (pthread_create(&thr[t], NULL, orc_model_worker, &jobs[t])) return -1;
  for (int t = 0; t < nthreads; t++) pthread_join(thr[t], NULL);
  return 0;
}

typedef struct {
  const orc_template *tpl;
  uint32_t nr;
  const uint8_t *seq;
  uint32_t x, y;
  int min_qual, reps, rc;
  orc_pileup *out;
} orc_acc_job;

static void *orc_acc_worker(void *arg) {
  orc_acc_job *jb = arg;
  for (int r = 0; r < jb->reps; r++) jb->rc |= orc_accumulate(jb->tpl, jb->nr, jb->seq, jb->x, jb->y, jb->min_qual, jb->out);
  return NULL;
}

/* (iii) HOT LOOP A: nthreads workers, each accumulating the same block `reps` times into its OWN pile-up array
 * (out + t * (y - x + 1)): the aggregate rate T independent blocks would reach; the reference runs the loop on one
 * thread (src/call_genotypes.c:180-226, on the process thread). */
int orc_accumulate_mt(const orc_template *tpl, uint32_t nr, const uint8_t *seq, uint32_t x, uint32_t y, int min_qual,
                      orc_pileup *out, int nthreads, int reps) {
  if (nthreads < 1) nthreads = 1;
  if (nthreads > 256) nthreads = 256;
  orc_acc_job jobs[256];
  pthread_t thr[256];
  const uint64_t sz = (uint64_t)y - x + 1;
  for (int t = 0; t < nthreads; t++) {
    orc_acc_job jb = {tpl, nr, seq, x, y, min_qual, reps, 0, out + sz * (uint64_t)t};
    jobs[t] = jb;
  }
  if (nthreads == 1) {
    orc_acc_worker(&jobs[0]);
    return jobs[0].rc;
  }
  for (int t = 0; t < nthreads; t++)
    if (pthread_create(&thr[t], NULL, orc_acc_worker, &jobs[t])) return -1;
  int rc = 0;
  for (int t = 0; t < nthreads; t++) {
    pthread_join(thr[t], NULL);
    rc |= jobs[t].rc;
  }
  return rc;
}
