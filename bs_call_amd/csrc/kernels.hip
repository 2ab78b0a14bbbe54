/*
 * kernels.hip — gfx950 (MI355X / CDNA4) kernels of the bs_call per-site calling path.
 *
 *   bsc_call_kernel    pile-up -> gt_meth for a batch of genome positions: the body of the reference's
 *                      calc-thread loop (src/call_genotypes.c:44-60,109-113) and calc_gt_prob()
 *                      (src/genotype_model.c:44-246, get_Z :23-42).  Heterozygous calls are appended to a
 *                      compact list instead of running the divergent Fisher walk in this kernel.
 *   bsc_fisher_kernel  strand table + fisher() (src/call_genotypes.c:61-108, src/stats_utils.c:25-91) over
 *                      the compacted heterozygous sites.
 *   bsc_synth_kernel   synthetic L-pileup generator (synth.h) — bench/test support.
 *
 * Design (DESIGN.md has the numbers): the kernel is a streaming scan, 104 B + 1 B in and 200 B out per
 * site with no reuse, so HBM bandwidth bounds it (FP64 VALU is the second bound, within 2x).
 * The reference's records are arrays of structs; a wave reading its 64 structs directly would touch
 * each 128-B line from 2 lanes in 7 separate instructions.  Instead each 256-thread workgroup moves a tile
 * of 256 sites with fully coalesced 16-B-per-lane loads into LDS, every lane then picks its own record
 * out of LDS (13 x ds_read_b64 at a 26-dword stride: conflict-free), computes in registers, writes its
 * 200-B result back to LDS (25 x ds_write_b64 at a 50-dword stride: conflict-free) and the tile leaves
 * with coalesced 16-B-per-lane stores.
 *
 * Numerics: FP64 throughout, no contraction (-ffp-contract=off), the transcendental functions are
 * bsmath.h (fixed operation order, shared with the host), tables come verbatim from the host.  The order
 * of the additions into ll[g] is the reference's: prior first, then one term per class in class order
 * 0..7; a class with n == 0 contributes +0.0, which leaves ll[g] unchanged exactly as skipping does
 * (no ll[g] can be -0.0: every term is n*ln(...) with n > 0 and no ln() argument path yields -0).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bsmath.h"
#include "devtables.h"
#include "synth.h"

#define TILE 256
#define IN_DW 26  /* dwords per pileup  (104 B) */
#define OUT_DW 50 /* dwords per gt_meth (200 B) */
#define MAX_OUT_DW 52 /* gt_vcf stride (208 B) */

/* term kinds of the likelihood matrix */
enum { T_LNK = 0, T_ONE = 1, T_HALF = 2, T_ZA = 3, T_ZB = 4, T_ZC = 5 };

/* TERM[c][g]: which of class c's values goes into genotype g; g order AA AC AG AT CC CG CT GG GT TT */
__device__ static constexpr unsigned char TERM[8][10] = {
    /*            AA      AC      AG      AT      CC      CG      CT      GG      GT      TT   */
    /* 0 A   */ {T_ONE, T_HALF, T_HALF, T_HALF, T_LNK, T_LNK, T_LNK, T_LNK, T_LNK, T_LNK},
    /* 1 C   */ {T_LNK, T_HALF, T_LNK, T_LNK, T_ONE, T_HALF, T_HALF, T_LNK, T_LNK, T_LNK},
    /* 2 G   */ {T_LNK, T_LNK, T_HALF, T_LNK, T_LNK, T_HALF, T_LNK, T_ONE, T_HALF, T_LNK},
    /* 3 T   */ {T_LNK, T_LNK, T_LNK, T_HALF, T_LNK, T_LNK, T_HALF, T_LNK, T_HALF, T_ONE},
    /* 4 A*  */ {T_ONE, T_HALF, T_ZA, T_HALF, T_LNK, T_ZC, T_LNK, T_ZB, T_ZC, T_LNK},
    /* 5 C*  */ {T_LNK, T_ZC, T_LNK, T_LNK, T_ZA, T_ZC, T_ZB, T_LNK, T_LNK, T_LNK},
    /* 6 G*  */ {T_LNK, T_LNK, T_ZB, T_LNK, T_LNK, T_ZC, T_LNK, T_ZA, T_ZC, T_LNK},
    /* 7 T*  */ {T_LNK, T_ZC, T_LNK, T_HALF, T_ZA, T_ZC, T_ZB, T_LNK, T_HALF, T_ONE},
};

/* x / ln(10) as a true IEEE division (the reference divides by the LOG10 macro, genotype_model.c:244). */
__device__ static __forceinline__ double div_ln10(double x) { return x / BSM_LN10; }

/* get_Z (src/genotype_model.c:23-42); the caller discards the result when x1 + x2 == 0. */
__device__ static __forceinline__ void get_Z(double x1, double x2, double k1, double k2, double l, double t, double &Z0,
                                             double &Z1, double &Z2) {
  double lpt = l + t;
  double lmt = l - t;
  double d = (x1 + x2) * lmt;
  double a2 = 2.0 - lpt;
  double s0 = (x1 * (lpt + 2.0 * k2) - x2 * (a2 + 2.0 * k1)) / d;
  s0 = s0 < -1.0 ? -1.0 : (s0 > 1.0 ? 1.0 : s0);
  Z0 = 0.5 * (lmt * s0 + 2.0 - lpt);
  double s1 = (x1 * (2.0 + lpt + 4.0 * k2) - x2 * (a2 + 4.0 * k1)) / d;
  s1 = s1 < -1.0 ? -1.0 : (s1 > 1.0 ? 1.0 : s1);
  Z1 = 0.5 * (lmt * s1 + 2.0 - lpt);
  double s2 = (x1 * (lpt + 4.0 * k2) - x2 * (a2 + 4.0 * k1)) / d;
  s2 = s2 < -1.0 ? -1.0 : (s2 > 1.0 ? 1.0 : s2);
  Z2 = 0.5 * (lmt * s2 + 2.0 - lpt);
}

/* n * ln(arg) for a Z-dependent term; +0.0 when the class is empty (arg may then be garbage / negative). */
__device__ static __forceinline__ double zterm(bool has, double arg, double n, const double *logtab) {
  double v = bsm_log_t(has ? arg : 2.0, logtab) * n;
  return has ? v : 0.0;
}

extern "C" __global__ __launch_bounds__(TILE) void bsc_call_kernel(const uint32_t *__restrict__ cts,
                                                                   const uint8_t *__restrict__ ref, uint64_t n_sites,
                                                                   uint32_t *__restrict__ out, uint32_t out_dw,
                                                                   uint8_t *__restrict__ skip,
                                                                   const bsc_dev_tables *__restrict__ tb,
                                                                   uint32_t *__restrict__ het_list,
                                                                   unsigned long long *__restrict__ counters) {
  /* LDS: staging tile (input then output alias the same bytes) + the q_prob columns + block counters.
   * Static, 16-byte aligned (no dynamic LDS behind static arrays: programming guide, guideline 17). */
  __shared__ __attribute__((aligned(16))) uint32_t lds_tile[TILE * MAX_OUT_DW];
  __shared__ double s_k[44], s_lnk[44], s_half[44], s_one[44];
  __shared__ double s_logtab[256];
  __shared__ unsigned long long s_exptab[256];
  __shared__ unsigned int s_cnt[12]; /* covered, hist[10], het */

  const unsigned tid = threadIdx.x;
  if (tid < 44) {
    s_k[tid] = tb->k[tid];
    s_lnk[tid] = tb->ln_k[tid];
    s_half[tid] = tb->ln_k_half[tid];
    s_one[tid] = tb->ln_k_one[tid];
  }
  s_logtab[tid] = tb->log_tab[tid];
  s_exptab[tid] = tb->exp_tab[tid];
  if (tid < 12) s_cnt[tid] = 0;
  const double l = 1.0 - tb->under_conv;
  const double t = tb->over_conv;
  const double lrb = tb->lrb, lrb1 = tb->lrb1;
  __syncthreads();

  const uint64_t n_tiles = (n_sites + TILE - 1) / TILE;
  for (uint64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint64_t site0 = tile * TILE;
    const unsigned nvalid = (unsigned)((n_sites - site0) < TILE ? (n_sites - site0) : TILE);

    /* ---- stage in: coalesced 16 B per lane ---- */
    {
      const uint4 *src = reinterpret_cast<const uint4 *>(cts + site0 * IN_DW);
      uint4 *dst = reinterpret_cast<uint4 *>(lds_tile);
      const unsigned nvec = nvalid * IN_DW / 4; /* 104 B = 6.5 x 16 B; nvalid*26 is even, so /4 may leave 2 dwords */
      for (unsigned v = tid; v < nvec; v += TILE) dst[v] = src[v];
      if ((nvalid * IN_DW) & 3u) { /* odd number of sites: last 8 bytes */
        if (tid == 0) {
          const unsigned o = nvec * 4;
          lds_tile[o] = cts[site0 * IN_DW + o];
          lds_tile[o + 1] = cts[site0 * IN_DW + o + 1];
        }
      }
    }
    const uint64_t site = site0 + tid;
    const bool valid = tid < nvalid;
    const unsigned rf = valid ? ref[site] : 0u;
    __syncthreads();

    /* ---- my record: 13 x ds_read_b64 ---- */
    uint32_t c0[8], c1[8];
    uint32_t n_reads = 0;
    float qsum[8], mapq2 = 0.f;
    {
      const uint2 *rec = reinterpret_cast<const uint2 *>(lds_tile + tid * IN_DW);
      uint32_t w[IN_DW];
#pragma unroll
      for (int i = 0; i < IN_DW / 2; i++) {
        uint2 v = valid ? rec[i] : make_uint2(0u, 0u);
        w[2 * i] = v.x;
        w[2 * i + 1] = v.y;
      }
#pragma unroll
      for (int j = 0; j < 8; j++) {
        c0[j] = w[j];
        c1[j] = w[8 + j];
        qsum[j] = __uint_as_float(w[17 + j]);
      }
      n_reads = w[16];
      mapq2 = __uint_as_float(w[25]);
    }
    __syncthreads(); /* everyone has its input in registers: the tile may be overwritten with results */

    const bool covered = valid && n_reads != 0;

    /* ---- per-site summary (src/call_genotypes.c:45-59) ---- */
    int qual[8];
    double nd[8];
    uint32_t cnt[8];
    float tot_qual = 0.0f;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      cnt[j] = c0[j] + c1[j];
      float nn = (float)cnt[j];
      int q = 0;
      if (nn > 0) {
        tot_qual += qsum[j];
        /* f32 divide, promoted to f64 for the +0.5, rounded back to f32 by floorf's parameter */
        q = (int)floorf((float)(0.5 + (double)(qsum[j] / nn)));
      }
      qual[j] = q;
      nd[j] = (double)cnt[j];
    }
    const float nf = covered ? (float)n_reads : 1.0f;
    const int aq = (int)floorf((float)(0.5 + (double)(tot_qual / nf)));
    const int mq = (int)(0.5 + sqrt((double)(mapq2 / nf)));

    /* table index: qual is in [0,43] for real data (q <= 43 per base); clamp so garbage cannot read outside */
    int qi[8];
#pragma unroll
    for (int j = 0; j < 8; j++) qi[j] = qual[j] < 0 ? 0 : (qual[j] > 43 ? 43 : qual[j]);

    /* ---- calc_gt_prob ---- */
    double ll[10];
#pragma unroll
    for (int g = 0; g < 10; g++) ll[g] = 0.0;
    /* prior from the reference base (src/genotype_model.c:87-108) */
    {
      const bool rA = rf == 1, rC = rf == 2, rG = rf == 3, rT = rf == 4;
      ll[0] = rA ? lrb : 0.0;
      ll[4] = rC ? lrb : 0.0;
      ll[7] = rG ? lrb : 0.0;
      ll[9] = rT ? lrb : 0.0;
      ll[1] = (rA || rC) ? lrb1 : 0.0; /* AC */
      ll[2] = (rA || rG) ? lrb1 : 0.0; /* AG */
      ll[3] = (rA || rT) ? lrb1 : 0.0; /* AT */
      ll[5] = (rC || rG) ? lrb1 : 0.0; /* CG */
      ll[6] = (rC || rT) ? lrb1 : 0.0; /* CT */
      ll[8] = (rG || rT) ? lrb1 : 0.0; /* GT */
    }
    /* classes 0..3: table terms only */
#pragma unroll
    for (int c = 0; c < 4; c++) {
      const bool has = cnt[c] != 0;
      double v[3];
      v[T_LNK] = has ? nd[c] * s_lnk[qi[c]] : 0.0;
      v[T_ONE] = has ? nd[c] * s_one[qi[c]] : 0.0;
      v[T_HALF] = has ? nd[c] * s_half[qi[c]] : 0.0;
#pragma unroll
      for (int g = 0; g < 10; g++) ll[g] += v[TERM[c][g]];
    }
    /* methylation estimates (src/genotype_model.c:165-171) */
    const double k4 = s_k[qi[4]], k5 = s_k[qi[5]], k6 = s_k[qi[6]], k7 = s_k[qi[7]];
    double Z0, Z1, Z2, Z3, Z4, Z5;
    get_Z(nd[5], nd[7], k5, k7, l, t, Z0, Z1, Z2);
    get_Z(nd[6], nd[4], k6, k4, l, t, Z3, Z4, Z5);
    /* (Z is only read by classes that are non-empty, which implies its get_Z ran on a non-zero divisor) */
    {
      const bool has = cnt[4] != 0; /* class 4: A on G2A reads, :173-187 */
      double v[6];
      v[T_LNK] = has ? nd[4] * s_lnk[qi[4]] : 0.0;
      v[T_ONE] = has ? nd[4] * s_one[qi[4]] : 0.0;
      v[T_HALF] = has ? nd[4] * s_half[qi[4]] : 0.0;
      v[T_ZA] = zterm(has, 1.0 - 0.5 * Z4 + k4, nd[4], s_logtab);   /* AG */
      v[T_ZB] = zterm(has, 1.0 - Z3 + k4, nd[4], s_logtab);         /* GG */
      v[T_ZC] = zterm(has, 0.5 * (1.0 - Z5) + k4, nd[4], s_logtab); /* CG, GT */
#pragma unroll
      for (int g = 0; g < 10; g++) ll[g] += v[TERM[4][g]];
    }
    {
      const bool has = cnt[5] != 0; /* class 5: C on C2T reads, :188-201 */
      double v[6];
      v[T_LNK] = has ? nd[5] * s_lnk[qi[5]] : 0.0;
      v[T_ONE] = 0.0;
      v[T_HALF] = 0.0;
      v[T_ZA] = zterm(has, Z0 + k5, nd[5], s_logtab);       /* CC */
      v[T_ZB] = zterm(has, 0.5 * Z1 + k5, nd[5], s_logtab); /* CT */
      v[T_ZC] = zterm(has, 0.5 * Z2 + k5, nd[5], s_logtab); /* AC, CG */
#pragma unroll
      for (int g = 0; g < 10; g++) ll[g] += v[TERM[5][g]];
    }
    {
      const bool has = cnt[6] != 0; /* class 6: G on G2A reads, :202-215 */
      double v[6];
      v[T_LNK] = has ? nd[6] * s_lnk[qi[6]] : 0.0;
      v[T_ONE] = 0.0;
      v[T_HALF] = 0.0;
      v[T_ZA] = zterm(has, Z3 + k6, nd[6], s_logtab);       /* GG */
      v[T_ZB] = zterm(has, 0.5 * Z4 + k6, nd[6], s_logtab); /* AG */
      v[T_ZC] = zterm(has, 0.5 * Z5 + k6, nd[6], s_logtab); /* CG, GT */
#pragma unroll
      for (int g = 0; g < 10; g++) ll[g] += v[TERM[6][g]];
    }
    {
      const bool has = cnt[7] != 0; /* class 7: T on C2T reads, :216-230 */
      double v[6];
      v[T_LNK] = has ? nd[7] * s_lnk[qi[7]] : 0.0;
      v[T_ONE] = has ? nd[7] * s_one[qi[7]] : 0.0;
      v[T_HALF] = has ? nd[7] * s_half[qi[7]] : 0.0;
      v[T_ZA] = zterm(has, 1.0 - Z0 + k7, nd[7], s_logtab);         /* CC */
      v[T_ZB] = zterm(has, 1.0 - 0.5 * Z1 + k7, nd[7], s_logtab);   /* CT */
      v[T_ZC] = zterm(has, 0.5 * (1.0 - Z2) + k7, nd[7], s_logtab); /* AC, CG */
#pragma unroll
      for (int g = 0; g < 10; g++) ll[g] += v[TERM[7][g]];
    }
    /* first-max argmax (:231-239) */
    double mx = ll[0];
    int mxi = 0;
#pragma unroll
    for (int g = 1; g < 10; g++) {
      const bool gt = ll[g] > mx;
      mx = gt ? ll[g] : mx;
      mxi = gt ? g : mxi;
    }
    /* normalise (:240-245) */
    double sum = 0.0;
#pragma unroll
    for (int g = 0; g < 10; g++) sum += bsm_exp_t(ll[g] - mx, (const uint64_t *)s_exptab);
    const double lsum = bsm_log_t(sum, s_logtab);
    double gp[10];
#pragma unroll
    for (int g = 0; g < 10; g++) gp[g] = div_ln10(ll[g] - mx - lsum);

    /* ---- heterozygous calls go to the Fisher list; block counters ---- */
    const bool het = covered && ((0x16Eu >> mxi) & 1u); /* gt_het: AC AG AT CG CT GT = bits 1,2,3,5,6,8 */
    if (covered) {
      atomicAdd(&s_cnt[0], 1u);
      atomicAdd(&s_cnt[1 + mxi], 1u);
    }
    {
      const unsigned long long m = __ballot(het);
      if (m) {
        const unsigned lane = tid & 63u;
        unsigned base = 0;
        if (lane == 0) base = atomicAdd((unsigned int *)&counters[BSC_CNT_HET_LIST], (unsigned)__popcll(m));
        base = __shfl(base, 0);
        if (het) het_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (uint32_t)site;
        if (lane == 0) atomicAdd(&s_cnt[11], (unsigned)__popcll(m));
      }
    }

    /* ---- my result record: 25 x ds_write_b64 ---- */
    if (valid) {
      uint2 *rec = reinterpret_cast<uint2 *>(lds_tile + tid * out_dw);
      if (covered) {
#pragma unroll
        for (int j = 0; j < 8; j++) rec[j] = make_uint2(cnt[j], 0u); /* counts[j] as u64 */
#pragma unroll
        for (int j = 0; j < 4; j++) rec[8 + j] = make_uint2((uint32_t)qual[2 * j], (uint32_t)qual[2 * j + 1]);
#pragma unroll
        for (int g = 0; g < 10; g++) {
          const uint64_t b = bsm_bits(gp[g]);
          rec[12 + g] = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
        }
        rec[22] = make_uint2(0u, 0u); /* fisher_strand = 0.0; bsc_fisher_kernel fills heterozygous sites */
        rec[23] = make_uint2((uint32_t)mq, (uint32_t)aq);
        rec[24] = make_uint2((uint32_t)mxi, 0u); /* max_gt + zero padding */
      } else {
#pragma unroll
        for (int j = 0; j < 25; j++) rec[j] = make_uint2(0u, 0u); /* skipped site: the reference's memset, :179 */
      }
      /* out_dw > 50 (e.g. 52 = gt_vcf): bytes 200.. = {ready = 0, skip, pad} */
      for (unsigned j = 25; j < out_dw / 2; j++) rec[j] = make_uint2(j == 25 ? (covered ? 0u : 0x100u) : 0u, 0u);
      skip[site] = covered ? 0 : 1;
    }
    __syncthreads();

    /* ---- stage out: coalesced 16 B per lane ---- */
    {
      uint4 *dst = reinterpret_cast<uint4 *>(out + site0 * out_dw);
      const uint4 *src = reinterpret_cast<const uint4 *>(lds_tile);
      const unsigned ndw = nvalid * out_dw;
      const unsigned nvec = ndw / 4;
      for (unsigned v = tid; v < nvec; v += TILE) dst[v] = src[v];
      if (ndw & 3u) {
        if (tid == 0) {
          const unsigned o = nvec * 4;
          out[site0 * out_dw + o] = lds_tile[o];
          out[site0 * out_dw + o + 1] = lds_tile[o + 1];
        }
      }
    }
    __syncthreads(); /* tile is free again */
  }

  /* block counters -> global */
  __syncthreads();
  if (tid < 12 && s_cnt[tid]) atomicAdd(&counters[BSC_CNT_COVERED + tid], (unsigned long long)s_cnt[tid]);
}

/* lfact2 (include/bs_call.h:335) */
__device__ static __forceinline__ double lfact_dev(int x, const double *lf, const double *logtab) {
  return x < 256 ? lf[x] : bsm_lfact_big_t(x, logtab);
}

/* fisher() (src/stats_utils.c:25-91) */
__device__ static double fisher_dev(int c0, int c1, int c2, int c3, const double *lf, const double *logtab,
                                    const uint64_t *exptab) {
#define LF(x) lfact_dev((x), lf, logtab)
  const int row0 = c0 + c1, row1 = c2 + c3, col0 = c0 + c2, col1 = c1 + c3;
  const int n = row0 + row1;
  if (n == 0) return 1.0;
  const double delta = (double)c0 - (double)(row0 * col0) / (double)n;
  const double knst = LF(col0) + LF(col1) + LF(row0) + LF(row1) - LF(n);
  double l = bsm_exp_t(knst - LF(c0) - LF(c1) - LF(c2) - LF(c3), exptab);
  double p = l;
  if (delta > 0.0) {
    int mn = c1 < c2 ? c1 : c2;
    for (int i = 0; i < mn; i++) {
      l *= (double)((c1 - i) * (c2 - i)) / (double)((c0 + i + 1) * (c3 + i + 1));
      p += l;
    }
    mn = c0 < c3 ? c0 : c3;
    const int k = (int)ceil(2.0 * delta);
    if (k <= mn) {
      c0 -= k; c3 -= k; c1 += k; c2 += k;
      l = bsm_exp_t(knst - LF(c0) - LF(c1) - LF(c2) - LF(c3), exptab);
      p += l;
      for (int i = 0; i < mn - k; i++) {
        l *= (double)((c0 - i) * (c3 - i)) / (double)((c1 + i + 1) * (c2 + i + 1));
        p += l;
      }
    }
  } else {
    int mn = c0 < c3 ? c0 : c3;
    for (int i = 0; i < mn; i++) {
      l *= (double)((c0 - i) * (c3 - i)) / (double)((c1 + i + 1) * (c2 + i + 1));
      p += l;
    }
    mn = c1 < c2 ? c1 : c2;
    int k = (int)ceil(-2.0 * delta);
    if (!k) k = 1;
    if (k <= mn) {
      c0 += k; c3 += k; c1 -= k; c2 -= k;
      l = bsm_exp_t(knst - LF(c0) - LF(c1) - LF(c2) - LF(c3), exptab);
      p += l;
      for (int i = 0; i < mn - k; i++) {
        l *= (double)((c1 - i) * (c2 - i)) / (double)((c0 + i + 1) * (c3 + i + 1));
        p += l;
      }
    }
  }
  return p;
#undef LF
}

/* One thread per heterozygous site of the compact list. */
extern "C" __global__ __launch_bounds__(256) void bsc_fisher_kernel(const uint32_t *__restrict__ cts,
                                                                    uint32_t *__restrict__ out, uint32_t out_dw,
                                                                    const bsc_dev_tables *__restrict__ tb,
                                                                    const uint32_t *__restrict__ het_list,
                                                                    const unsigned long long *__restrict__ counters) {
  __shared__ double s_lf[256];
  __shared__ double s_logtab[256];
  __shared__ unsigned long long s_exptab[256];
  s_lf[threadIdx.x] = tb->lfact[threadIdx.x];
  s_logtab[threadIdx.x] = tb->log_tab[threadIdx.x];
  s_exptab[threadIdx.x] = tb->exp_tab[threadIdx.x];
  __syncthreads();
  const unsigned nhet = (unsigned)counters[BSC_CNT_HET_LIST];
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < nhet; i += gridDim.x * blockDim.x) {
    const uint64_t site = het_list[i];
    const uint32_t *p = cts + site * IN_DW;
    uint32_t f[8], r[8]; /* counts[0][*] forward, counts[1][*] reverse */
#pragma unroll
    for (int j = 0; j < 8; j++) {
      f[j] = p[j];
      r[j] = p[8 + j];
    }
    uint32_t *rec = out + site * out_dw;
    const unsigned mxi = rec[48] & 0xffu; /* max_gt at byte 192 */
    int t0 = 0, t1 = 0, t2 = 0, t3 = 0;
    switch (mxi) { /* src/call_genotypes.c:64-100 */
      case 1: /* AC */
        t0 = f[0] + f[4]; t1 = f[1] + f[5] + f[7]; t2 = r[0] + r[4]; t3 = r[1] + r[5] + r[7];
        break;
      case 2: /* AG */
        t0 = f[0]; t1 = f[2] + f[6]; t2 = r[0]; t3 = r[2] + r[6];
        break;
      case 3: /* AT */
        t0 = f[0] + f[4]; t1 = f[3] + f[7]; t2 = r[0] + r[4]; t3 = r[3] + r[7];
        break;
      case 5: /* CG */
        t0 = f[1] + f[5] + f[7]; t1 = f[2] + f[4] + f[6]; t2 = r[1] + r[5] + r[7]; t3 = r[2] + r[4] + r[6];
        break;
      case 6: /* CT */
        t0 = f[1] + f[5]; t1 = f[3]; t2 = r[1] + r[5]; t3 = r[3];
        break;
      case 8: /* GT: the reverse row uses the FORWARD class-6 count, as the reference does (:98) */
        t0 = f[2] + f[4] + f[6]; t1 = f[3] + f[7]; t2 = r[2] + r[4] + f[6]; t3 = r[3] + r[7];
        break;
      default:
        break;
    }
    double z = fisher_dev(t0, t1, t2, t3, s_lf, s_logtab, (const uint64_t *)s_exptab);
    if (z < 1.0e-20) z = 1.0e-20;
    const double fs = bsm_log_t(z, s_logtab) / BSM_LN10;
    const uint64_t b = bsm_bits(fs);
    reinterpret_cast<uint2 *>(rec)[22] = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
  }
}

/* Synthetic L-pileup generator: one thread per site (not on the timed path). */
extern "C" __global__ __launch_bounds__(256) void bsc_synth_kernel(uint64_t seed, uint64_t first_site, uint64_t n,
                                                                   uint32_t coverage, uint32_t flags,
                                                                   uint32_t *__restrict__ cts,
                                                                   uint8_t *__restrict__ ref) {
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x) {
    uint32_t counts[16], nr, rf;
    float q[8], m2;
    syn_site(seed, first_site + i, coverage, flags, counts, &nr, q, &m2, &rf);
    uint32_t *p = cts + i * IN_DW;
#pragma unroll
    for (int j = 0; j < 16; j++) p[j] = counts[j];
    p[16] = nr;
#pragma unroll
    for (int j = 0; j < 8; j++) p[17 + j] = __float_as_uint(q[j]);
    p[25] = __float_as_uint(m2);
    ref[i] = (uint8_t)rf;
  }
}

/* ---- launchers (called from the C host code in bscall_api.c) ---------------------------------------- */

/* pile-up -> gt_meth for n sites (n < 2^32), then the Fisher pass over the heterozygous list.
 * counters[BSC_CNT_HET_LIST] must be zero on entry (the host queues a memset in front). */
extern "C" int bsc_dev_launch_call(const void *cts, const void *ref, uint64_t n, void *out, uint32_t out_dw, void *skip,
                                   const void *tb, void *het_list, void *counters, int num_cus, void *stream,
                                   void *ev_start, void *ev_mid, void *ev_stop) {
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, s);
  const uint64_t n_tiles = (n + TILE - 1) / TILE;
  /* 3 workgroups fit one CU (53 KB LDS each); 8 rounds of them keep the tail short */
  uint64_t grid = (uint64_t)num_cus * 3u * 8u;
  if (grid > n_tiles) grid = n_tiles;
  hipLaunchKernelGGL(bsc_call_kernel, dim3((unsigned)grid), dim3(TILE), 0, s, (const uint32_t *)cts,
                     (const uint8_t *)ref, n, (uint32_t *)out, out_dw, (uint8_t *)skip, (const bsc_dev_tables *)tb,
                     (uint32_t *)het_list, (unsigned long long *)counters);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  if (ev_mid) (void)hipEventRecord((hipEvent_t)ev_mid, s);
  hipLaunchKernelGGL(bsc_fisher_kernel, dim3((unsigned)(num_cus * 2)), dim3(256), 0, s, (const uint32_t *)cts,
                     (uint32_t *)out, out_dw, (const bsc_dev_tables *)tb, (const uint32_t *)het_list,
                     (const unsigned long long *)counters);
  e = hipGetLastError();
  if (ev_stop) (void)hipEventRecord((hipEvent_t)ev_stop, s);
  return (int)e;
}

extern "C" int bsc_dev_launch_synth(uint64_t seed, uint64_t first_site, uint64_t n, uint32_t coverage, uint32_t flags,
                                    void *cts, void *ref, int num_cus, void *stream) {
  if (n == 0) return 0;
  uint64_t grid = (n + 255) / 256;
  const uint64_t cap = (uint64_t)num_cus * 32u;
  if (grid > cap) grid = cap;
  hipLaunchKernelGGL(bsc_synth_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, seed, first_site, n,
                     coverage, flags, (uint32_t *)cts, (uint8_t *)ref);
  return (int)hipGetLastError();
}
