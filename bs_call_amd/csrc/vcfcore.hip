/*
 * vcfcore.hip — VCF record formation on the device: everything bs_call's print thread derives from a gt_meth and
 * its neighbours before handing the record to htslib (reference src/print_vcf.c:32-381, with the 5-site window
 * of print_vcf_entry / flush_vcf_entries, :529-594), as one fixed 64-byte bsc_vcf_core record per position.
 *
 * The reference walks a block position by position with static window state.  Inside a block that state is a pure
 * function of the position, so the work is per-site parallel:
 *   pass 1 (fused, per 64-position wave-tile + 2 halo positions each side, through LDS):
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

/* first-max argmax of gt_prob[] + 1 (src/print_vcf.c:584-591); gp points at the record's gt_prob (LDS or global) */
template <typename P>
__device__ static __forceinline__ int argmax1(P gp) {
  double z = gp[0];
  int gt = 0;
#pragma unroll
  for (int k = 1; k < 10; k++) {
    const double v = gp[k];
    if (v > z) { z = v; gt = k; }
  }
  return gt + 1;
}

/* 0 for a skipped or out-of-block position, else the called genotype + 1, read from global memory (halo positions) */
__device__ static __forceinline__ int called_gt_global(const uint8_t *__restrict__ gtm, uint32_t stride,
                                                       const uint8_t *__restrict__ skip, int64_t j, uint32_t n) {
  if (j < 0 || j >= (int64_t)n || skip[j]) return 0;
  return argmax1(reinterpret_cast<const double *>(gtm + (uint64_t)j * stride + 96));
}

#define VW 4                 /* waves per workgroup */
#define VSLOT_DW (64 * 52)   /* per-wave record slot: 64 records of up to 208 bytes */

/* LDS-DMA: 16 bytes per lane, global (per-lane address) -> LDS (wave-uniform base + lane * 16); nt: read once */
__device__ static __forceinline__ void vdma16(const void *g, void *lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                   (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 2);
}

/*
 * One wave per 64-position wave-tile, no workgroup barrier after the table set-up.  The tile's 64 records (12 800 or
 * 13 312 contiguous bytes) arrive in the wave's LDS slot by LDS-DMA and each lane works on its own record there (stride
 * 50 / 52 dwords: at most 2-way conflicts); the 64-byte results are collected in the slot afterwards and leave with
 * 16-byte-per-lane stores.  The called genotypes of the tile go through 68 LDS bytes (2 halo positions each side, which
 * four lanes fetch from global memory).
 */
extern "C" __global__ __launch_bounds__(64 * VW) void bsc_vcf_core_kernel(
    const uint8_t *__restrict__ gtm, uint32_t stride, const uint8_t *__restrict__ skip, const uint8_t *__restrict__ ref,
    const uint8_t *__restrict__ dbsnp, uint32_t n, uint32_t x, int all_positions, uint32_t reg_start, uint32_t reg_stop,
    const bsc_dev_tables *__restrict__ tb, bsc_vcf_core_dev *__restrict__ out) {
  __shared__ __attribute__((aligned(16))) uint32_t s_slot[VW][VSLOT_DW];
  __shared__ uint8_t s_gw[VW][72];
  /* LDS holds nothing but the record slots: 53.5 KB per workgroup, so that three workgroups (12 waves) share a CU —
   * the kernel waits for memory two thirds of its time and lives on waves in flight.  The exp / log tables of the one
   * QUAL evaluation per record are read from global memory (4 KB, cache resident), the 64-byte results are staged in
   * the slot itself once every lane has read its record (a wave executes in lockstep). */
  const double *const g_logtab = tb->log_tab;
  const uint64_t *const g_exptab = reinterpret_cast<const uint64_t *>(tb->exp_tab);
  const unsigned lane = threadIdx.x & 63u;
  const unsigned wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  uint32_t *slot = s_slot[wid];
  uint8_t *sg = s_gw[wid];
  const uint32_t sdw = stride / 4u;
  const uint32_t n_wt = (n + 63u) / 64u;
  const bool aligned = ((uintptr_t)gtm & 15u) == 0;
  for (uint32_t wt = blockIdx.x * VW + wid; wt < n_wt; wt += gridDim.x * VW) {
    const uint32_t site0 = wt * 64u;
    const uint32_t i = site0 + lane;
    const bool valid = i < n;
    const bool full = site0 + 64u <= n && aligned; /* wave-uniform */
    /* ---- the tile's records -> slot ---- */
    if (full) {
      const char *src = reinterpret_cast<const char *>(gtm + (uint64_t)site0 * stride) + lane * 16;
      const uint32_t n16 = 64u * stride / 16u; /* 800 or 832 */
#pragma unroll
      for (int j = 0; j < 13; j++)
        if (j * 64u + lane < n16) vdma16(src + j * 1024, slot + j * 256);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else if (valid) { /* ragged last tile / unaligned base: each lane copies its own record */
      const uint32_t *g32 = reinterpret_cast<const uint32_t *>(gtm + (uint64_t)i * stride);
      for (uint32_t k = 0; k < 50u; k++) slot[lane * sdw + k] = g32[k];
    }
    const uint32_t *rec32 = slot + lane * sdw;
    const double *gp = reinterpret_cast<const double *>(rec32 + 24);
    /* ---- called genotypes of the tile + halo ---- */
    const int g_own = (valid && !skip[i]) ? argmax1(gp) : 0;
    sg[lane + 2] = (uint8_t)g_own;
    if (lane < 4) {
      const int64_t j = lane < 2 ? (int64_t)site0 - 2 + lane : (int64_t)site0 + 62 + lane;
      sg[lane < 2 ? lane : 64 + lane] = (uint8_t)called_gt_global(gtm, stride, skip, j, n);
    }
#define G(j) sg[(int64_t)(j) - (int64_t)site0 + 2] /* valid for site0-2 <= j < site0+66 */
    bsc_vcf_core_dev o;
    {
      uint4 *z4 = reinterpret_cast<uint4 *>(&o);
      z4[0] = z4[1] = z4[2] = z4[3] = make_uint4(0, 0, 0, 0);
    }
    const int gt1 = valid ? (int)G(i) : 0;
    const uint8_t *rec = reinterpret_cast<const uint8_t *>(rec32); /* my record, in LDS */
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
      const double z1 = bsm_exp_t(gp[gt] * BSM_LN10, g_exptab);
      int phred;
      if (z1 >= 1.0) phred = 255;
      else {
        phred = (int)(-10.0 * bsm_log_t(1.0 - z1, g_logtab) / BSM_LN10);
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
      /* QUAL / QD of a position whose record is not written: the printer computes them (:140-152) and never reads them again
       * (flags and statistics are behind `skip`, :185-217,382-398) — left 0, so that the fused chain (fused.hip) need not make the
       * gt_prob[] of such a position */
      o.phred = skip ? (uint8_t)0 : (uint8_t)phred;
      o.fs = fs;
      o.qd = skip ? 0u : qd;
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
    /* ---- results: own 64-byte record -> the slot (all record reads are done), then the tile's 4 KiB leave contiguously ---- */
    {
      /* Lane L's result lands on bytes that still hold other lanes' input records: every lane's record reads must be
       * issued before any lane's staging store, and the staging stores before the cross-lane copy-out.  A wave runs in
       * lockstep, so what has to be ruled out is the COMPILER sinking a load or hoisting a store across these points:
       * s_wave_barrier is convergent and a scheduling barrier, and costs nothing at run time. */
      __builtin_amdgcn_wave_barrier();
      uint4 *so = reinterpret_cast<uint4 *>(slot);
      const uint4 *s4 = reinterpret_cast<const uint4 *>(&o);
#pragma unroll
      for (int k = 0; k < 4; k++) so[lane * 4 + k] = s4[k];
      __builtin_amdgcn_wave_barrier();
      const uint32_t nvec = (n - site0 < 64u ? n - site0 : 64u) * 4u;
      uint4 *d4 = reinterpret_cast<uint4 *>(out + site0);
#pragma unroll
      for (int k = 0; k < 4; k++) {
        const uint32_t idx = k * 64u + lane;
        if (idx < nvec) d4[idx] = so[idx];
      }
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
  const unsigned n_wt = (n + 63u) / 64u;
  unsigned grid = (n_wt + VW - 1u) / VW;
  if (grid > (unsigned)num_cus * 3u * 8u) grid = (unsigned)num_cus * 3u * 8u; /* 3 workgroups (53.5 KB LDS) per CU */
  hipLaunchKernelGGL(bsc_vcf_core_kernel, dim3(grid), dim3(64 * VW), 0, s, (const uint8_t *)gtm, stride,
                     (const uint8_t *)skip, (const uint8_t *)ref, (const uint8_t *)dbsnp, n, x, all_positions, reg_start,
                     reg_stop, (const bsc_dev_tables *)tb, (bsc_vcf_core_dev *)out);
  return (int)hipGetLastError();
}
