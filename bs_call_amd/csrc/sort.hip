/*
 * sort.hip — the two library primitives of the device path (rocPRIM, ROCm's own primitive library): the ordering of a
 * block's templates by leftmost position, and (at the end of the file) the prefix sum that packs written records.
 *
 * The reference walks its align_list in whatever order the reads arrived (src/call_genotypes.c:181) — the sums do not
 * depend on it.  The accumulate kernels want the block's READS ordered by first countable position so that a
 * 64-position tile only looks at the window of reads that can reach it (ordering reads rather than templates keeps
 * that window one read long whatever the distance between mates); align_lists are nearly but not exactly in that
 * order, so every block is ordered here: keys = first countable position relative to the block start
 * (bsc_prep_reads_kernel, accumulate.hip), values = read index, rocPRIM's device radix sort over just the bits the
 * block's length needs (ROCm's own primitive library; 2.4 million pairs take about 0.15 ms, the host qsort of the
 * templates it replaced took 25 ms).
 */
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

/* bytes of temporary storage bsc_dev_sort_templates needs for nr pairs */
extern "C" int bsc_dev_sort_tmp_bytes(uint32_t nr, size_t *bytes) {
  *bytes = 0;
  if (nr == 0) return 0;
  return (int)rocprim::radix_sort_pairs(nullptr, *bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr,
                                        rocprim::counting_iterator<uint32_t>(0), (uint32_t *)nullptr, nr, 0, 32,
                                        (hipStream_t)0);
}

/* keys[nr] -> keys_sorted[nr] ascending (stable), perm[i] = index of the template that comes i-th */
extern "C" int bsc_dev_sort_templates(const void *keys, void *keys_sorted, void *perm, uint32_t nr, unsigned key_bits,
                                      void *tmp, size_t tmp_bytes, void *stream) {
  if (nr == 0) return 0;
  if (key_bits < 1) key_bits = 1;
  if (key_bits > 32) key_bits = 32;
  return (int)rocprim::radix_sort_pairs(tmp, tmp_bytes, (const uint32_t *)keys, (uint32_t *)keys_sorted,
                                        rocprim::counting_iterator<uint32_t>(0), (uint32_t *)perm, nr, 0, key_bits,
                                        (hipStream_t)stream);
}

/* exclusive prefix sum of n u32 values (compact.hip: record counts per tile -> first slot of each tile) */
extern "C" int bsc_dev_scan_tmp_bytes(uint32_t n, size_t *bytes) {
  *bytes = 0;
  if (n == 0) return 0;
  return (int)rocprim::exclusive_scan(nullptr, *bytes, (const uint32_t *)nullptr, (uint32_t *)nullptr, 0u, n,
                                      rocprim::plus<uint32_t>(), (hipStream_t)0);
}
extern "C" int bsc_dev_scan_u32(const void *in, void *out, uint32_t n, void *tmp, size_t tmp_bytes, void *stream) {
  if (n == 0) return 0;
  return (int)rocprim::exclusive_scan(tmp, tmp_bytes, (const uint32_t *)in, (uint32_t *)out, 0u, n,
                                      rocprim::plus<uint32_t>(), (hipStream_t)stream);
}
