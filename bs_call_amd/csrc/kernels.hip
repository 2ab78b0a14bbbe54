/*
 * kernels.hip — gfx950 (MI355X / CDNA4) kernels of the bs_call per-site calling path.
 *
 *   bsc_call_kernel    pile-up -> gt_meth for a batch of genome positions: the body of the reference's
 *                      calc-thread loop (src/call_genotypes.c:44-60,109-113) and calc_gt_prob()
 *                      (src/genotype_model.c:44-246, get_Z :23-42).  Heterozygous calls are marked in a 64-bit
 *                      mask per wave-tile instead of running the divergent Fisher walk in this kernel.
 *   bsc_fisher_kernel  strand table + fisher() (src/call_genotypes.c:61-108, src/stats_utils.c:25-91) over
 *                      the marked sites, compacted chunk by chunk in LDS.
 *   bsc_synth_kernel   synthetic L-pileup generator (synth.h) — bench/test support.
 *
 * Design (DESIGN.md has the numbers): the kernel is a streaming scan, 104 B + 1 B in and 200 B out per
 * site with no reuse, so HBM bandwidth is its roofline; in practice instruction issue (mostly FP64 VALU) is the
 * nearer bound, so the structure below is about issuing fewer instructions as much as about moving bytes well.
 * The reference's records are arrays of structs; a wave reading its 64 structs directly would touch
 * each 128-B line from 2 lanes in 7 separate instructions.  Instead each wave moves its 64 records with
 * fully coalesced 16-B-per-lane transfers through a private LDS slot (see bsc_call_kernel).
 *
 * Numerics: FP64 throughout, no contraction (-ffp-contract=off), the transcendental functions are
 * bsmath.h (bit-exact replicas of glibc's log/exp/lgamma, shared with the host), tables come verbatim from the host.  The order
 * of the additions into ll[g] is the reference's: prior first, then one term per class in class order
 * 0..7; a class with n == 0 contributes +0.0, which leaves ll[g] unchanged exactly as skipping does
 * (no ll[g] can be -0.0: every term is n*ln(...) with n > 0 and no ln() argument path yields -0).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bsmath.h"
#include "callmath.h"
#include "devtables.h"
#include "synth.h"

#ifndef TILE
#define TILE 256 /* threads per workgroup of bsc_call_kernel (a multiple of 64) */
#endif
#ifndef BSC_TILES_PER_WAVE
#define BSC_TILES_PER_WAVE 16 /* launch heuristic: wave-tiles per wave (round 3, tools/ab_call.py: 16 is 1 % ahead of 8 at 50 M positions, 2.5 % at 25 M, level below; 2-4 lose 4-18 % at 10 M, 64 loses 4 % at 50 M) */
#endif
#ifndef BSC_WAVES_PER_SIMD
#define BSC_WAVES_PER_SIMD 4 /* occupancy target of bsc_call_kernel: bounds its VGPR budget (512 / waves) */
#endif

/*
 * The calling kernel.  No workgroup barrier after the table set-up: every wave owns a 6 656-byte LDS slot
 * and walks its own 64-site wave-tiles.
 *   in : the wave-tile's 64 pile-ups are 6 656 contiguous bytes in HBM; 6.5 LDS-DMA instructions (1 KiB
 *        each, no VGPRs) land them in the slot; lane i then reads record i (13 x ds_read_b64, stride 26
 *        dwords: conflict-free).
 *   out: results leave in two halves of 32 records (6 400 contiguous bytes, or 6 656 for gt_vcf stride):
 *        the owning lanes write their record to the slot (25 x ds_write_b64, stride 50 dwords:
 *        conflict-free), then all 64 lanes copy the slot out with 16-byte stores (6.25 KiB-wide
 *        instructions).
 * LDS of one wave is touched only by that wave and a wave's LDS operations execute in order, so the
 * hand-over inside a wave needs no barrier — only the vmcnt wait that retires the DMA.
 */
/* FULL = true: the launch's complete 64-position wave-tiles (wt_begin = 0, wt_end = n_sites / 64) — every lane valid,
 * no guarded paths, and the staging pointers are known to be LDS (ds_write instead of flat stores, which would issue to
 * the LDS and the vector-memory pipe both: profiles/r01_h counted 26 such stores per tile).  FULL = false: the same
 * code with the guards, launched for the last, partial wave-tile only. */
template <bool FULL>
__global__ __launch_bounds__(TILE, BSC_WAVES_PER_SIMD) void bsc_call_kernel_t(const uint32_t *__restrict__ cts,
                                                                      const uint8_t *__restrict__ ref,
                                                                      uint64_t n_sites, uint64_t wt_begin,
                                                                      uint64_t wt_end, uint32_t *__restrict__ out,
                                                                      uint32_t out_dw, uint8_t *__restrict__ skip,
                                                                      const bsc_dev_tables *__restrict__ tb,
                                                                      uint32_t *__restrict__ het_list,
                                                                      unsigned long long *__restrict__ counters) {
  __shared__ __attribute__((aligned(16))) uint32_t lds_slot[TILE / 64][SLOT_DW];
  __shared__ double s_k[44], s_lnk[44], s_half[44], s_one[44];
#ifdef BSC_TABLES_GLOBAL /* A/B variant (tools/ab_variants.py): the log / exp tables read through L1 instead of LDS */
  const double *const s_logtab = tb->log_tab;
  const unsigned long long *const s_exptab = tb->exp_tab;
#else
  __shared__ double s_logtab[256];
  __shared__ unsigned long long s_exptab[256];
#endif
  __shared__ unsigned int s_cnt[12]; /* covered, hist[10], het */
  __shared__ double s_ptab[PT_WORDS]; /* logs of the methylation arguments of a class whose partner class is empty (callmath.h) */
  __shared__ uint8_t s_pairs[TILE / 64][256]; /* per wave: the (lane, class) pairs whose logs are needed */

  const unsigned tid = threadIdx.x;
  const unsigned lane = tid & 63u;
  const unsigned wid = __builtin_amdgcn_readfirstlane(tid >> 6); /* wave-uniform: the slot base lives in scalar registers */
  if (tid < 44) {
    s_k[tid] = tb->k[tid];
    s_lnk[tid] = tb->ln_k[tid];
    s_half[tid] = tb->ln_k_half[tid];
    s_one[tid] = tb->ln_k_one[tid];
  }
#ifndef BSC_TABLES_GLOBAL
  for (unsigned i = tid; i < 256; i += TILE) {
    s_logtab[i] = tb->log_tab[i];
    s_exptab[i] = tb->exp_tab[i];
  }
#endif
  if (tid < 12) s_cnt[tid] = 0;
  const double l = 1.0 - tb->under_conv;
  const double t = tb->over_conv;
  const double lrb = tb->lrb, lrb1 = tb->lrb1;
  __syncthreads();
  for (unsigned i = tid; i < PT_WORDS; i += TILE) s_ptab[i] = pure_log_entry(i, l, t, s_k, s_logtab); /* callmath.h PT_* */
  __syncthreads();

  uint32_t *slot = lds_slot[wid];
  /* wave-tile indices fit 32 bits (a launch has < 2^31 positions): 64-bit arithmetic only where addresses are formed */
  const uint32_t wave_stride = gridDim.x * (TILE / 64);
  const uint32_t wt_last = (uint32_t)wt_end;
  for (uint32_t wt = (uint32_t)wt_begin + blockIdx.x * (TILE / 64) + wid; wt < wt_last; wt += wave_stride) {
    const uint64_t site0 = (uint64_t)wt * 64;
    const unsigned nvalid = FULL ? 64u : (unsigned)((n_sites - site0) < 64 ? (n_sites - site0) : 64);
    const bool full = FULL; /* compile-time */
    const uint64_t site = site0 + lane;
    const bool valid = FULL || lane < nvalid;
    const unsigned rf = valid ? ref[site] : 0u;

    /* ---- my record ---- */
    uint32_t w[IN_DW];
    if (full) {
      const char *src = reinterpret_cast<const char *>(cts + site0 * IN_DW) + lane * 16;
#pragma unroll
      for (int j = 0; j < 6; j++) dma16(src + j * 1024, slot + j * 256);
      if (lane < 32) dma16(src + 6 * 1024, slot + 6 * 256);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const uint2 *rec = reinterpret_cast<const uint2 *>(slot + lane * IN_DW);
#pragma unroll
      for (int i = 0; i < IN_DW / 2; i++) {
        const uint2 v = rec[i];
        w[2 * i] = v.x;
        w[2 * i + 1] = v.y;
      }
    } else { /* last, partial wave-tile of the launch: plain guarded loads (no 16-byte over-read) */
#pragma unroll
      for (int i = 0; i < IN_DW; i++) w[i] = valid ? cts[site * IN_DW + i] : 0u;
    }
#include "call_body.inc"
    /* ---- heterozygous calls go to the Fisher list; block counters ---- */
    const bool het = covered && ((0x16Eu >> mxi) & 1u); /* gt_het: AC AG AT CG CT GT = bits 1,2,3,5,6,8 */
    if (covered) {
      atomicAdd(&s_cnt[0], 1u);
      atomicAdd(&s_cnt[1 + mxi], 1u);
    }
    { /* one 64-bit mask per wave-tile, written whether or not it has a bit set (nothing to clear between launches).  Rounds
       * 1-2 appended to ONE compact list through an atomic on its length: at 10x, where four tiles in five hold a
       * heterozygous call, the 1.3e5 returning atomics on that one address serialised to 1 ms per 10 M positions — 2.5x the
       * kernel's own time (tools/ab_call.py). */
      const unsigned long long m = __ballot(het);
      if (lane == 0) {
        reinterpret_cast<unsigned long long *>(het_list)[wt] = m;
        if (m) atomicAdd(&s_cnt[11], (unsigned)__popcll(m));
      }
    }
    if (valid) skip[site] = covered ? 0 : 1;

    /* ---- results: two halves of 32 records through the slot ----
     * A lane's record lands on bytes that other lanes used as their la[] areas, and the second half on bytes the
     * copy-out of the first half reads: a wave runs in lockstep, so only the COMPILER could reorder these LDS accesses
     * across lanes' program order; s_wave_barrier (convergent, a scheduling barrier, free at run time) pins them. */
#pragma unroll 1
    for (unsigned half = 0; half < 2; half++) {
      __builtin_amdgcn_wave_barrier();
      const bool mine = valid && (lane >> 5) == half;
      auto write_record = [&](auto *rec) {
        if (covered) {
#pragma unroll
          for (int j = 0; j < 8; j++) rec[j] = make_uint2(cnt[j], 0u); /* counts[j] as u64 */
          rec[8] = make_uint2(qpack0 & 0xffu, (qpack0 >> 8) & 0xffu); /* qual[0..7] as i32 */
          rec[9] = make_uint2((qpack0 >> 16) & 0xffu, qpack0 >> 24);
          rec[10] = make_uint2(qpack1 & 0xffu, (qpack1 >> 8) & 0xffu);
          rec[11] = make_uint2((qpack1 >> 16) & 0xffu, qpack1 >> 24);
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
        /* out_dw == 52 (gt_vcf): bytes 200.. = {ready = 0, skip, pad} */
        if (out_dw > OUT_DW) rec[25] = make_uint2(covered ? 0u : 0x100u, 0u);
      };
      if (mine) {
        if (full) write_record(reinterpret_cast<uint2 *>(slot + (lane & 31u) * out_dw)); /* LDS */
        else write_record(reinterpret_cast<uint2 *>(out + site * out_dw));                /* global */
      }
      if (full) { /* copy the 32 records out: 16 bytes per lane, contiguous */
        __builtin_amdgcn_wave_barrier(); /* the staging stores above must not sink below the cross-lane reads */
        const unsigned nvec = 32u * out_dw / 4u; /* 400 or 416 */
        uint4 *dst = reinterpret_cast<uint4 *>(out + (site0 + half * 32u) * out_dw);
        const uint4 *srcv = reinterpret_cast<const uint4 *>(slot);
#pragma unroll
        for (unsigned v = 0; v < 7; v++) {
          const unsigned idx = v * 64u + lane;
          if (idx < nvec) { /* written once, never re-read by this kernel: non-temporal (-2 % kernel time) */
            typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
            __builtin_nontemporal_store(reinterpret_cast<const u32x4 *>(srcv)[idx], reinterpret_cast<u32x4 *>(dst) + idx);
          }
        }
      }
    }
  }

  /* block counters -> global */
  __syncthreads();
  if (tid < 12 && s_cnt[tid]) atomicAdd(&counters[BSC_CNT_COVERED + tid], (unsigned long long)s_cnt[tid]);
}

/*
 * Fisher's test for the heterozygous calls of a launch, found through the per-tile masks the calling kernel left.  A
 * workgroup takes a chunk of consecutive wave-tiles at a time: its threads count the set bits of their share of the masks,
 * a prefix sum over the 256 shares orders them, every thread writes the positions of its own calls into one compact list in
 * LDS (FI_LIST at a time when a chunk holds more), and the threads then take the listed calls one each — all lanes busy
 * whatever the density.  The launcher sizes the chunk so that every workgroup has about one (dense input: several rounds
 * of 256 per chunk; WGBS at 30x: ~100 calls per chunk, one round).
 */
#define FI_THREADS 256
#define FI_LIST 2048
extern "C" __global__ __launch_bounds__(FI_THREADS) void bsc_fisher_kernel(const uint32_t *__restrict__ cts,
                                                                           uint32_t *__restrict__ out, uint32_t out_dw,
                                                                           const bsc_dev_tables *__restrict__ tb,
                                                                           const unsigned long long *__restrict__ masks,
                                                                           uint32_t n_tiles, uint32_t chunk_tiles) {
  __shared__ double s_lf[256];
  __shared__ double s_logtab[256];
  __shared__ unsigned long long s_exptab[256];
  __shared__ uint32_t s_scan[FI_THREADS + 1];
  __shared__ uint32_t s_list[FI_LIST];
  const unsigned tid = threadIdx.x;
  s_lf[tid] = tb->lfact[tid];
  s_logtab[tid] = tb->log_tab[tid];
  s_exptab[tid] = tb->exp_tab[tid];
  const uint32_t per = (chunk_tiles + FI_THREADS - 1) / FI_THREADS; /* masks per thread */
  for (uint32_t c0 = blockIdx.x * chunk_tiles; c0 < n_tiles; c0 += gridDim.x * chunk_tiles) {
    const uint32_t c1 = min(c0 + chunk_tiles, n_tiles);
    const uint32_t t0 = min(c0 + tid * per, c1), t1 = min(t0 + per, c1); /* this thread's masks */
    uint32_t mine = 0;
    for (uint32_t t = t0; t < t1; t++) mine += (uint32_t)__popcll(masks[t]);
    __syncthreads(); /* the previous chunk's list and scan are done with (first pass: the tables are in place) */
    s_scan[tid + 1] = mine;
    if (tid == 0) s_scan[0] = 0;
    __syncthreads();
    if (tid < 64) { /* inclusive scan of 256 counts by one wave: four per lane, then across lanes */
      uint32_t a = s_scan[4 * tid + 1], b = a + s_scan[4 * tid + 2], c = b + s_scan[4 * tid + 3], d = c + s_scan[4 * tid + 4];
      uint32_t run = d;
#pragma unroll
      for (int off = 1; off < 64; off <<= 1) {
        const uint32_t v = __shfl_up(run, off);
        if ((int)tid >= off) run += v;
      }
      const uint32_t base = run - d;
      s_scan[4 * tid + 1] = base + a;
      s_scan[4 * tid + 2] = base + b;
      s_scan[4 * tid + 3] = base + c;
      s_scan[4 * tid + 4] = base + d;
    }
    __syncthreads();
    const uint32_t total = s_scan[FI_THREADS], first = s_scan[tid];
    for (uint32_t w0 = 0; w0 < total; w0 += FI_LIST) { /* the chunk's calls, FI_LIST at a time */
      if (w0) __syncthreads();
      if (first < w0 + FI_LIST && first + mine > w0) {
        uint32_t k = first;
        for (uint32_t t = t0; t < t1; t++) {
          unsigned long long m = masks[t];
          while (m) {
            const unsigned bit = (unsigned)__builtin_ctzll(m);
            m &= m - 1ull;
            if (k >= w0 && k < w0 + FI_LIST) s_list[k - w0] = t * 64u + bit;
            k++;
          }
        }
      }
      __syncthreads();
      const uint32_t cnt = min((uint32_t)FI_LIST, total - w0);
      for (uint32_t i = tid; i < cnt; i += FI_THREADS) {
        const uint64_t site = s_list[i];
        const uint32_t *p = cts + site * IN_DW;
        uint32_t f[8], r[8]; /* counts[0][*] forward, counts[1][*] reverse */
#pragma unroll
        for (int j = 0; j < 8; j++) {
          f[j] = p[j];
          r[j] = p[8 + j];
        }
        uint32_t *rec = out + site * out_dw;
        const unsigned mxi = rec[48] & 0xffu; /* max_gt at byte 192 */
        int t0_, t1_, t2_, t3_;
        strand_table(mxi, f, r, t0_, t1_, t2_, t3_);
        double z = fisher_dev(t0_, t1_, t2_, t3_, s_lf, s_logtab, (const uint64_t *)s_exptab);
        if (z < 1.0e-20) z = 1.0e-20;
        const double fs = bsm_log_t(z, s_logtab) / BSM_LN10;
        const uint64_t b = bsm_bits(fs);
        reinterpret_cast<uint2 *>(rec)[22] = make_uint2((uint32_t)b, (uint32_t)(b >> 32));
      }
    }
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
 * het_list: room for one 64-bit mask per wave-tile, (n + 63) / 64 of them. */
extern "C" int bsc_dev_launch_call(const void *cts, const void *ref, uint64_t n, void *out, uint32_t out_dw, void *skip,
                                   const void *tb, void *het_list, void *counters, int num_cus, void *stream,
                                   void *ev_start, void *ev_mid, void *ev_stop) {
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  if (ev_start) (void)hipEventRecord((hipEvent_t)ev_start, s);
  /* Grid: the workgroups resident at once (BSC_WAVES_PER_SIMD per CU: 33 KB LDS, <= 128 VGPRs each) times a number of
   * rounds chosen so that a wave walks about BSC_TILES_PER_WAVE wave-tiles.  Measured (gpurun_out/ab*.txt):
   * a fully persistent grid (1 round) is best for small blocks (the table set-up is paid once per wave slot) but 7 %
   * slower at 50 M positions than 8-16 rounds, whose workgroup turnover keeps the waves of a CU out of phase;
   * beyond 32 rounds the set-up cost shows again. */
  uint64_t resident = (uint64_t)num_cus * (BSC_WAVES_PER_SIMD * 4) / (TILE / 64); /* workgroups that fit the chip at once */
  if (resident < 1) resident = 1;
  const uint64_t n_full = n / 64; /* complete wave-tiles: the specialised kernel; the ragged rest: the guarded one */
  hipError_t e = hipSuccess;
  if (n_full) {
    const uint64_t n_tiles = (n_full + (TILE / 64) - 1) / (TILE / 64);
    uint64_t rounds = n_full / (resident * (TILE / 64) * BSC_TILES_PER_WAVE);
    rounds = rounds < 1 ? 1 : (rounds > 16 ? 16 : rounds);
    uint64_t grid = resident * rounds;
    if (grid > n_tiles) grid = n_tiles;
    hipLaunchKernelGGL(bsc_call_kernel_t<true>, dim3((unsigned)grid), dim3(TILE), 0, s, (const uint32_t *)cts,
                       (const uint8_t *)ref, n, (uint64_t)0, n_full, (uint32_t *)out, out_dw, (uint8_t *)skip,
                       (const bsc_dev_tables *)tb, (uint32_t *)het_list, (unsigned long long *)counters);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (n & 63u) {
    hipLaunchKernelGGL(bsc_call_kernel_t<false>, dim3(1), dim3(TILE), 0, s, (const uint32_t *)cts, (const uint8_t *)ref, n,
                       n_full, n_full + 1, (uint32_t *)out, out_dw, (uint8_t *)skip, (const bsc_dev_tables *)tb,
                       (uint32_t *)het_list, (unsigned long long *)counters);
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  if (ev_mid) (void)hipEventRecord((hipEvent_t)ev_mid, s);
  {
    const uint32_t n_tiles = (uint32_t)((n + 63) / 64);
    const uint32_t grid = (uint32_t)num_cus * 2u;
    uint32_t chunk = (n_tiles + grid - 1) / grid; /* about one chunk per workgroup ... */
    chunk = chunk < 64u ? 64u : (chunk > 4096u ? 4096u : chunk); /* ... of 4 K .. 256 K positions */
    const uint32_t n_chunks = (n_tiles + chunk - 1) / chunk;
    hipLaunchKernelGGL(bsc_fisher_kernel, dim3(n_chunks < grid ? n_chunks : grid), dim3(FI_THREADS), 0, s, (const uint32_t *)cts,
                       (uint32_t *)out, out_dw, (const bsc_dev_tables *)tb, (const unsigned long long *)het_list, n_tiles, chunk);
  }
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
