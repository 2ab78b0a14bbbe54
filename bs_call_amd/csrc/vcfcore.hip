/*
 * vcfcore.hip — VCF record formation on the device: everything bs_call's print thread derives from a gt_meth and
 * its neighbours before handing the record to htslib (reference src/print_vcf.c:32-381, with the 5-site window
 * of print_vcf_entry / flush_vcf_entries, :529-594), as one fixed 64-byte bsc_vcf_core record per position.
 *
 * The reference walks a block position by position with static window state.  Inside a block that state is a pure
 * function of the position, so the work is per-site parallel:
 *   pass 1 (fused, per 256-position tile + 2 halo positions each side, through LDS):
 *                                 g[i] = 0 for a skipped position, else 1 + first-max argmax of gt_prob[] (the
 *                                 printer recomputes the argmax from gt_prob, :584-591);
 *   pass 2:                       window of called genotypes [g(i-2) .. g(i+2)], where positions outside the block
 *                                 read 0 — except that the two positions flushed at the end of a block see the LAST
 *                                 genotype repeated to their right (flush_vcf_entries shifts its 5-byte window with a
 *                                 4-byte memmove and never clears the vacated slot, :540);
 *                                 reference context ref[i-2 .. i+2], read through the reference's strncpy of a 7-base
 *                                 window that starts 4 positions before the look-ahead position: an N (code 0) in
 *                                 the window blanks every base after it (:570-577);
 *                                 then phred / FS / QD / FILTER bits / mac1 / GT / ALT / GL selection / CG / CX.
 * exp() and log() in the phred computation are bsmath.h (bit-exact glibc), so QUAL/GQ equal the reference's.
 * A block that starts below position 5 right after another block would see that block's leftovers in the
 * reference's window (store_x arithmetic, :561-568); here such a block starts from an empty window, as the first
 * block of a run does.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bsmath.h"
#include "devtables.h"

struct bsc_vcf_core_dev {
  uint32_t pos;
  uint8_t emit, gt, ref_code, gt_enc, flt, phred, n_gl;
  char cg;
  char alt[2];
  char cx_ref[5];
  char cx_gt[5];
  int32_t fs;
  uint32_t qd;
  uint32_t dp;
  float gl[6];
  uint32_t _pad;
};
static_assert(sizeof(bsc_vcf_core_dev) == 64, "bsc_vcf_core is 64 bytes");

/* genotype -> its two alleles as base codes 1..4 (AA AC AG AT CC CG CT GG GT TT) */
__device__ static __forceinline__ void alleles(int g, int &a, int &b) {
  a = g < 4 ? 1 : (g < 7 ? 2 : (g < 9 ? 3 : 4));
  b = g < 4 ? g + 1 : (g < 7 ? g - 2 : (g < 9 ? g - 4 : 4));
}
__device__ static __forceinline__ bool has_c(int g) { int a, b; alleles(g, a, b); return a == 2 || b == 2; }
__device__ static __forceinline__ bool has_g(int g) { int a, b; alleles(g, a, b); return a == 3 || b == 3; }

/* 0 for a skipped position, else 1 + first-max argmax of gt_prob[] (src/print_vcf.c:584-591) */
__device__ static __forceinline__ int called_gt(const uint8_t *__restrict__ gtm, uint32_t stride,
                                                const uint8_t *__restrict__ skip, uint32_t i) {
  if (skip[i]) return 0;
  const double *gp = reinterpret_cast<const double *>(gtm + (uint64_t)i * stride + 96);
  double z = gp[0];
  int gt = 0;
#pragma unroll
  for (int k = 1; k < 10; k++) {
    const double v = gp[k];
    if (v > z) { z = v; gt = k; }
  }
  return gt + 1;
}

#define VT 256 /* positions per workgroup */

extern "C" __global__ __launch_bounds__(VT) void bsc_vcf_core_kernel(
    const uint8_t *__restrict__ gtm, uint32_t stride, const uint8_t *__restrict__ skip, const uint8_t *__restrict__ ref,
    const uint8_t *__restrict__ dbsnp, uint32_t n, uint32_t x, int all_positions, uint32_t reg_start, uint32_t reg_stop,
    const bsc_dev_tables *__restrict__ tb, bsc_vcf_core_dev *__restrict__ out) {
  __shared__ double s_logtab[256];
  __shared__ unsigned long long s_exptab[256];
  __shared__ uint8_t s_g[VT + 4]; /* called genotypes of the tile and of 2 positions on either side */
  s_logtab[threadIdx.x] = tb->log_tab[threadIdx.x];
  s_exptab[threadIdx.x] = tb->exp_tab[threadIdx.x];
  const uint32_t n_tiles = (n + VT - 1) / VT;
  for (uint32_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const uint32_t base = tile * VT;
    const uint32_t i = base + threadIdx.x;
    __syncthreads(); /* tables on the first pass; s_g free again on later ones */
    /* pass 1 (fused): every thread calls its own position; four threads also call the halo */
    s_g[threadIdx.x + 2] = i < n ? (uint8_t)called_gt(gtm, stride, skip, i) : 0;
    if (threadIdx.x < 4) {
      const int64_t j = threadIdx.x < 2 ? (int64_t)base - 2 + threadIdx.x : (int64_t)base + VT + (threadIdx.x - 2);
      s_g[threadIdx.x < 2 ? threadIdx.x : VT + threadIdx.x] = (j >= 0 && j < (int64_t)n) ? (uint8_t)called_gt(gtm, stride, skip, (uint32_t)j) : 0;
    }
    __syncthreads();
    if (i >= n) continue;
#define G(j) s_g[(int64_t)(j) - (int64_t)base + 2] /* valid for base-2 <= j < base+VT+2 */
    bsc_vcf_core_dev o;
    {
      uint4 *z4 = reinterpret_cast<uint4 *>(&o);
      z4[0] = z4[1] = z4[2] = z4[3] = make_uint4(0, 0, 0, 0);
    }
    const int gt1 = G(i);
    const uint8_t *rec = gtm + (uint64_t)i * stride;
    const uint64_t *counts = reinterpret_cast<const uint64_t *>(rec);
    uint32_t dp1 = 0, d_inf = 0;
#pragma unroll
    for (int k = 0; k < 4; k++) dp1 += (uint32_t)counts[k];
#pragma unroll
    for (int k = 4; k < 8; k++) d_inf += (uint32_t)counts[k];
    if (gt1 != 0 && dp1 + d_inf != 0) {
      const int gt = gt1 - 1;
      /* ---- windows ---- */
      int gs[5];
      const uint32_t last = n - 1;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        const int64_t j = (int64_t)i - 2 + k;
        int v = (j >= 0 && j <= (int64_t)last) ? G(j) : 0;
        if (j > (int64_t)last && i + 2 > last) v = G(last); /* the flushed positions see the last genotype repeated
                                                               (last is i or i+1 here: inside this tile's halo) */
        gs[k] = v;
      }
      int rc[5];
      {
        const uint32_t la = i + 2 < last ? i + 2 : last;        /* look-ahead position whose strncpy filled the window */
        const int64_t w0 = la >= 4 ? (int64_t)la - 4 : 0;       /* first base of that copy */
        bool blank = false;
        for (int64_t j = w0; j < (int64_t)i - 2; j++) blank |= ref[j] == 0; /* an N before the 5-base part */
#pragma unroll
        for (int k = 0; k < 5; k++) {
          const int64_t j = (int64_t)i - 2 + k;
          int v = 0;
          if (j >= 0) {
            v = ref[j];
            blank |= v == 0;
            if (blank) v = 0;
          }
          rc[k] = v;
        }
      }
      const int rfix = rc[2];
      const uint8_t rs_found = dbsnp ? dbsnp[i] : 0;
      int ga, gb;
      alleles(gt, ga, gb);
      const bool het = ga != gb;
      bool skip = !all_positions && !(rs_found & 2) && ((gt == 0 && rfix == 1) || (gt == 9 && rfix == 4));
      /* ---- phred (:140-148) ---- */
      const double *gp = reinterpret_cast<const double *>(rec + 96);
      const double z1 = bsm_exp_t(gp[gt] * BSM_LN10, (const uint64_t *)s_exptab);
      int phred;
      if (z1 >= 1.0) phred = 255;
      else {
        phred = (int)(-10.0 * bsm_log_t(1.0 - z1, s_logtab) / BSM_LN10);
        if (phred > 255) phred = 255;
      }
      const double fisher = *reinterpret_cast<const double *>(rec + 176);
      const int mq = *reinterpret_cast<const int32_t *>(rec + 184);
      const int fs = (int)(-fisher * 10.0 + 0.5);
      const uint32_t qd = dp1 > 0 ? (uint32_t)phred / dp1 : (uint32_t)phred;
      const uint32_t pos = x + i;
      if (!skip) skip = pos < reg_start || pos > reg_stop;
      /* ---- CpG status and IUPAC context (:227-266) ---- */
      const char iupac[12] = "NAMRWCSYGKT";
      const char pbase[6] = "NACGT";
      char cg = '.';
      {
        const int c = gs[2], nx = gs[3], pv = gs[1];
        if ((c == 5 && nx == 8) || (c == 8 && pv == 5)) cg = 'C'; /* "CG" */
        else if (c == 5) cg = nx ? (has_g(nx - 1) ? 'H' : 'N') : '?';
        else if (c == 8) cg = pv ? (has_c(pv - 1) ? 'H' : 'N') : '?';
        else if (has_c(c - 1)) cg = nx ? (has_g(nx - 1) ? 'H' : 'N') : '?';
        else if (has_g(c - 1)) cg = pv ? (has_c(pv - 1) ? 'H' : 'N') : '.';
      }
      o.pos = pos;
      o.gt = (uint8_t)gt;
      o.ref_code = (uint8_t)rfix;
      o.phred = (uint8_t)phred;
      o.fs = fs;
      o.qd = qd;
      o.dp = dp1;
      o.cg = cg;
#pragma unroll
      for (int k = 0; k < 5; k++) {
        o.cx_ref[k] = pbase[rc[k]];
        o.cx_gt[k] = iupac[gs[k]];
      }
      if (!skip) {
        uint32_t flt = 0;
        if (phred < 20) flt |= 1;
        if (qd < 2) flt |= 2;
        if (fs > 60) flt |= 4;
        if (mq < 40) flt |= 8;
        if (!flt && het) { /* mac1 (:191-214): either allele supported by at most one base */
          const uint64_t sA = counts[0] + counts[4];                 /* A: classes 0 and 4 (for AC, AT) */
          const uint64_t sC = counts[1] + counts[5] + counts[7];     /* C incl. converted */
          const uint64_t sG = counts[2] + counts[6] + counts[4];     /* G incl. converted */
          const uint64_t sT = counts[3] + counts[7];
          bool mac1 = false;
          switch (gt) {
            case 1: mac1 = sC <= 1 || sA <= 1; break;                               /* AC */
            case 2: mac1 = counts[2] + counts[6] <= 1 || counts[0] <= 1; break;     /* AG */
            case 3: mac1 = sT <= 1 || sA <= 1; break;                               /* AT */
            case 5: mac1 = sG <= 1 || sC <= 1; break;                               /* CG */
            case 6: mac1 = counts[3] <= 1 || counts[1] + counts[5] <= 1; break;     /* CT */
            case 8: mac1 = sT <= 1 || sG <= 1; break;                               /* GT */
          }
          if (mac1) flt |= 128;
        }
        o.emit = 1;
        o.flt = (uint8_t)flt;
        /* ALT alleles and GT codes: the genotype's alleles that differ from the reference base */
        int aix0 = 0, aix1 = 0;
        if (ga != rfix) aix0 = ga;
        if (gb != ga && gb != rfix) { if (aix0) aix1 = gb; else aix0 = gb; }
        o.alt[0] = aix0 ? pbase[aix0] : 0;
        o.alt[1] = aix1 ? pbase[aix1] : 0;
        o.gt_enc = het ? ((ga == rfix || gb == rfix) ? 0x24 : 0x48) : (ga == rfix ? 0x22 : 0x44);
        /* GL (:319-347) */
        float gl[6];
        int ngl = 1;
        {
          double z = -99.999;
          if (rfix) {
            z = gp[rfix * (9 - rfix) / 2 + rfix - 5];
            if (z < -99.999) z = -99.999;
          }
          gl[0] = (float)z;
          const int aix[2] = {aix0, aix1};
#pragma unroll
          for (int k = 0; k < 2; k++) {
            const int a = aix[k];
            if (a > 0 && (k == 0 || aix[0] > 0)) {
              if (rfix) {
                const int j = rfix < a ? rfix * (9 - rfix) / 2 + a - 5 : a * (9 - a) / 2 + rfix - 5;
                z = gp[j];
                if (z < -99.999) z = -99.999;
                gl[ngl++] = (float)z;
              }
              z = gp[a * (9 - a) / 2 + a - 5];
              if (z < -99.999) z = -99.999;
              gl[ngl++] = (float)z;
            }
          }
        }
        o.n_gl = (uint8_t)ngl;
        for (int k = 0; k < 6; k++) o.gl[k] = k < ngl ? gl[k] : 0.0f;
      }
    }
    {
      uint4 *d4 = reinterpret_cast<uint4 *>(out + i);
      const uint4 *s4 = reinterpret_cast<const uint4 *>(&o);
      d4[0] = s4[0]; d4[1] = s4[1]; d4[2] = s4[2]; d4[3] = s4[3];
    }
  }
}

#undef G

extern "C" int bsc_dev_launch_vcf(const void *gtm, uint32_t stride, const void *skip, const void *ref, const void *dbsnp,
                                  uint32_t n, uint32_t x, int all_positions, uint32_t reg_start, uint32_t reg_stop,
                                  const void *tb, void *g_unused, void *out, int num_cus, void *stream) {
  (void)g_unused;
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  unsigned grid = (n + VT - 1u) / VT;
  if (grid > (unsigned)num_cus * 16u) grid = (unsigned)num_cus * 16u;
  hipLaunchKernelGGL(bsc_vcf_core_kernel, dim3(grid), dim3(VT), 0, s, (const uint8_t *)gtm, stride, (const uint8_t *)skip,
                     (const uint8_t *)ref, (const uint8_t *)dbsnp, n, x, all_positions, reg_start, reg_stop,
                     (const bsc_dev_tables *)tb, (bsc_vcf_core_dev *)out);
  return (int)hipGetLastError();
}
