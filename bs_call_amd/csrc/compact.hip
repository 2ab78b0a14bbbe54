/*
 * compact.hip — the written records of a block, packed: what crosses PCIe when the host only wants what the printer
 * would write.  A block's bsc_vcf_core records mark the positions the reference writes a record for (emit); for those
 * — about half of a WGBS genome, every C and G — the 64-byte record and the fields of gt_meth the encoder still needs
 * (MC8 counts, AMQ qualities, MQ; src/print_vcf.c:306-359) are gathered into one 128-byte bsc_vcf_rec, in position
 * order: 64 bytes per position on average instead of the 201 of a gt_vcf entry.
 *
 *   bsc_tile_emit_kernel   one wave per 64-position tile: number of written records (ballot + popcount)
 *   (exclusive scan of the tile counts: rocPRIM, sort.hip)
 *   bsc_compact_kernel     one wave per tile: lane rank among the tile's written records -> its slot; the record is
 *                          assembled in registers and leaves as eight 16-byte stores per lane
 * Round 5, for records that come with the chain's aux array: the flags as a byte per position (left by the chain kernel, or by the counting
 * pass), and bsc_compact_aux_emit_kernel — eight lanes per written record, whole sectors in, whole lines out (20 M positions: 1.13 -> 0.62 ms).
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bscall_amd.h"

static_assert(sizeof(bsc_vcf_rec) == 128, "bsc_vcf_rec is 128 bytes");

/* emit_ws: NULL, or a byte per position that receives the flag (what the chain kernel's emit_out would have held: the packing pass then
 * takes its eight-lanes-per-record form, bsc_compact_aux_emit_kernel) */
extern "C" __global__ __launch_bounds__(256) void bsc_tile_emit_kernel(const uint8_t *__restrict__ core, uint32_t n,
                                                                       uint32_t *__restrict__ tile_cnt, uint8_t *__restrict__ emit_ws) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t n_tiles = (n + 63u) / 64u;
  for (uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * 4u) {
    const uint32_t i = tile * 64u + lane;
    const bool emit = i < n && core[(uint64_t)i * 64u + 4u] != 0; /* bsc_vcf_core.emit */
    if (emit_ws && i < n) emit_ws[i] = emit ? 1 : 0;
    const unsigned long long m = __ballot(emit);
    if (lane == 0) tile_cnt[tile] = (uint32_t)__popcll(m);
  }
}

extern "C" __global__ __launch_bounds__(256) void bsc_compact_kernel(const uint8_t *__restrict__ core,
                                                                     const uint8_t *__restrict__ gtm, uint32_t gtm_stride,
                                                                     const uint8_t *__restrict__ dbsnp, uint32_t n,
                                                                     const uint32_t *__restrict__ tile_off,
                                                                     const uint32_t *__restrict__ tile_cnt,
                                                                     uint8_t *__restrict__ out, uint64_t out_cap,
                                                                     unsigned long long *__restrict__ total) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t n_tiles = (n + 63u) / 64u;
  for (uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * 4u) {
    const uint32_t i = tile * 64u + lane;
    const uint32_t off = tile_off[tile];
    if (tile == n_tiles - 1u && lane == 0) *total = (unsigned long long)off + tile_cnt[tile];
    uint4 c[4];
    bool emit = false;
    if (i < n) {
      c[0] = *reinterpret_cast<const uint4 *>(core + (uint64_t)i * 64u);
      emit = ((c[0].y) & 0xffu) != 0;
    }
    const unsigned long long m = __ballot(emit);
    if (!emit) continue;
    const uint64_t slot = (uint64_t)off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (slot >= out_cap) continue; /* the host reports the overflow from *total */
#pragma unroll
    for (int k = 1; k < 4; k++) c[k] = *reinterpret_cast<const uint4 *>(core + (uint64_t)i * 64u + 16u * k);
    const uint8_t *g = gtm + (uint64_t)i * gtm_stride;
    const uint64_t *cnt = reinterpret_cast<const uint64_t *>(g);
    const int32_t *ql = reinterpret_cast<const int32_t *>(g + 64);
    uint32_t cc[8], qq[2] = {0u, 0u};
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const uint64_t v = cnt[k];
      cc[k] = v > 0xffffffffull ? 0xffffffffu : (uint32_t)v;
      qq[k >> 2] |= ((uint32_t)ql[k] & 0xffu) << (8 * (k & 3));
    }
    const int32_t mq = *reinterpret_cast<const int32_t *>(g + 184), aq = *reinterpret_cast<const int32_t *>(g + 188);
    const uint32_t tail = (uint32_t)g[192] | ((dbsnp ? (uint32_t)dbsnp[i] : 0u) << 8); /* max_gt, rs_found */
    uint4 *o = reinterpret_cast<uint4 *>(out + slot * 128u);
    o[0] = c[0];
    o[1] = c[1];
    o[2] = c[2];
    o[3] = c[3];
    o[4] = make_uint4(cc[0], cc[1], cc[2], cc[3]);
    o[5] = make_uint4(cc[4], cc[5], cc[6], cc[7]);
    o[6] = make_uint4(qq[0], qq[1], (uint32_t)mq, (uint32_t)aq);
    o[7] = make_uint4(tail, 0u, 0u, 0u);
  }
}

/* The same packing when the encoder's fields come from the chain kernel itself (fused.hip, aux_out): aux[i] is already the
 * second half of position i's bsc_vcf_rec (counts, qualities, MQ, mean quality, max_gt, rs_found) — the chain formed it from
 * the pile-up in its registers, there is no gt_meth to read. */
extern "C" __global__ __launch_bounds__(256) void bsc_compact_aux_kernel(const uint8_t *__restrict__ core,
                                                                         const uint8_t *__restrict__ aux, uint32_t n,
                                                                         const uint32_t *__restrict__ tile_off,
                                                                         const uint32_t *__restrict__ tile_cnt,
                                                                         uint8_t *__restrict__ out, uint64_t out_cap,
                                                                         unsigned long long *__restrict__ total) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t n_tiles = (n + 63u) / 64u;
  for (uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * 4u) {
    const uint32_t i = tile * 64u + lane;
    const uint32_t off = tile_off[tile];
    if (tile == n_tiles - 1u && lane == 0) *total = (unsigned long long)off + tile_cnt[tile];
    uint4 c0 = make_uint4(0u, 0u, 0u, 0u);
    if (i < n) c0 = *reinterpret_cast<const uint4 *>(core + (uint64_t)i * 64u);
    const bool emit = (c0.y & 0xffu) != 0;
    const unsigned long long m = __ballot(emit);
    if (!emit) continue;
    const uint64_t slot = (uint64_t)off + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
    if (slot >= out_cap) continue; /* the host reports the overflow from *total */
    const uint4 *cs = reinterpret_cast<const uint4 *>(core + (uint64_t)i * 64u);
    const uint4 *as = reinterpret_cast<const uint4 *>(aux + (uint64_t)i * 64u);
    uint4 *o = reinterpret_cast<uint4 *>(out + slot * 128u);
    o[0] = c0;
#pragma unroll
    for (int k = 1; k < 4; k++) o[k] = cs[k];
#pragma unroll
    for (int k = 0; k < 4; k++) o[4 + k] = as[k];
  }
}

/* The same two passes when the chain kernel has left the records' emit flags as a byte per position (fused.hip, emit_out): the count reads
 * 64 bytes per tile instead of one byte of each of its 64 records, and the packing fetches the records that are written and no others. */
extern "C" __global__ __launch_bounds__(256) void bsc_tile_emit_bytes_kernel(const uint8_t *__restrict__ emit, uint32_t n, uint32_t *__restrict__ tile_cnt) {
  const unsigned lane = threadIdx.x & 63u;
  const uint32_t n_tiles = (n + 63u) / 64u;
  for (uint32_t tile = blockIdx.x * 4u + (threadIdx.x >> 6); tile < n_tiles; tile += gridDim.x * 4u) {
    const uint32_t i = tile * 64u + lane;
    const unsigned long long m = __ballot(i < n && emit[i] != 0);
    if (lane == 0) tile_cnt[tile] = (uint32_t)__popcll(m);
  }
}

/* Eight lanes per written record: the tile's written positions are listed by rank in the wave's LDS (a byte each), and group g of a round
 * moves the record of rank base + g — lanes 0-3 its 64-byte core record, lanes 4-7 its 64 bytes of the aux array, 16 bytes a lane — so that
 * a load instruction fetches 16 whole 64-byte sectors and a store instruction writes 8 whole 128-byte lines (one lane per record: every
 * instruction touched a quarter of 64 sectors and an eighth of 64 lines). */
extern "C" __global__ __launch_bounds__(256) void bsc_compact_aux_emit_kernel(const uint8_t *__restrict__ core, const uint8_t *__restrict__ aux,
                                                                              const uint8_t *__restrict__ emit_b, uint32_t n,
                                                                              const uint32_t *__restrict__ tile_off, const uint32_t *__restrict__ tile_cnt,
                                                                              uint8_t *__restrict__ out, uint64_t out_cap,
                                                                              unsigned long long *__restrict__ total) {
  __shared__ uint8_t s_idx[4][64];
  const unsigned lane = threadIdx.x & 63u, wid = threadIdx.x >> 6;
  const unsigned sub = lane & 7u, grp = lane >> 3;
  uint8_t *const idx = s_idx[wid];
  const uint32_t n_tiles = (n + 63u) / 64u;
  for (uint32_t tile = blockIdx.x * 4u + wid; tile < n_tiles; tile += gridDim.x * 4u) {
    const uint32_t i = tile * 64u + lane;
    const uint32_t off = tile_off[tile];
    if (tile == n_tiles - 1u && lane == 0) *total = (unsigned long long)off + tile_cnt[tile];
    const bool emit = i < n && emit_b[i] != 0;
    const unsigned long long m = __ballot(emit);
    const unsigned cnt = (unsigned)__popcll(m); /* wave-uniform */
    if (!cnt) continue;
    if (emit) idx[__popcll(m & ((1ull << lane) - 1ull))] = (uint8_t)lane;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    for (unsigned base = 0; base < cnt; base += 8u) {
      const unsigned r = base + grp;
      const uint64_t slot = (uint64_t)off + r;
      if (r < cnt && slot < out_cap) { /* beyond out_cap: the host reports the overflow from *total */
        const uint64_t p = (uint64_t)tile * 64u + idx[r];
        const uint8_t *src = sub < 4u ? core + p * 64u + 16u * sub : aux + p * 64u + 16u * (sub - 4u);
        const uint4 v = *reinterpret_cast<const uint4 *>(src);
        *reinterpret_cast<uint4 *>(out + slot * 128u + 16u * sub) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier(); /* the list is rewritten for the wave's next tile */
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

extern "C" int bsc_dev_scan_u32(const void *in, void *out, uint32_t n, void *tmp, size_t tmp_bytes, void *stream); /* sort.hip */

/* tile_cnt / tile_off: (n + 63) / 64 words each; total: one u64; gtm_stride = 0: gtm is the chain's aux array (64 bytes per position) */
/* emit (gtm_stride == 0 only): the records' emit flags as a byte per position where the chain kernel has left them, or NULL; emit_ws: n bytes
 * of workspace the counting pass then leaves them in itself (NULL: the lane-per-record form of round 3) */
extern "C" int bsc_dev_launch_compact(const void *core, const void *gtm, uint32_t gtm_stride, const void *dbsnp, uint32_t n,
                                      void *tile_cnt, void *tile_off, void *scan_tmp, size_t scan_tmp_bytes, void *out,
                                      uint64_t out_cap, void *total, const void *emit, void *emit_ws, int num_cus, void *stream) {
  if (n == 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const unsigned n_tiles = (n + 63u) / 64u;
  unsigned grid = (n_tiles + 3u) / 4u;
  if (grid > (unsigned)num_cus * 32u) grid = (unsigned)num_cus * 32u;
  if (gtm_stride != 0) emit = emit_ws = NULL;
  if (emit)
    hipLaunchKernelGGL(bsc_tile_emit_bytes_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t *)emit, n, (uint32_t *)tile_cnt);
  else
    hipLaunchKernelGGL(bsc_tile_emit_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t *)core, n, (uint32_t *)tile_cnt, (uint8_t *)emit_ws);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return (int)e;
  int rc = bsc_dev_scan_u32(tile_cnt, tile_off, n_tiles, scan_tmp, scan_tmp_bytes, stream);
  if (rc) return rc;
  if (emit || emit_ws)
    hipLaunchKernelGGL(bsc_compact_aux_emit_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t *)core, (const uint8_t *)gtm,
                       (const uint8_t *)(emit ? emit : emit_ws), n, (const uint32_t *)tile_off, (const uint32_t *)tile_cnt, (uint8_t *)out, out_cap,
                       (unsigned long long *)total);
  else if (gtm_stride == 0) /* gtm = the chain's aux array */
    hipLaunchKernelGGL(bsc_compact_aux_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t *)core, (const uint8_t *)gtm, n,
                       (const uint32_t *)tile_off, (const uint32_t *)tile_cnt, (uint8_t *)out, out_cap, (unsigned long long *)total);
  else
    hipLaunchKernelGGL(bsc_compact_kernel, dim3(grid), dim3(256), 0, s, (const uint8_t *)core, (const uint8_t *)gtm, gtm_stride,
                       (const uint8_t *)dbsnp, n, (const uint32_t *)tile_off, (const uint32_t *)tile_cnt, (uint8_t *)out,
                       out_cap, (unsigned long long *)total);
  return (int)hipGetLastError();
}
