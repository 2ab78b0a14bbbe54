/*
 * sort.hip — the one library primitive of the device path (rocPRIM, ROCm's own primitive library): an exclusive prefix sum,
 * used to turn per-bin read counts into bin offsets (accumulate.hip) and per-tile record counts into packing slots
 * (compact.hip).  Round 3: the radix sort of the block's reads that lived here is gone — the reads are grouped by
 * 64-position bin with a count / scan / scatter of our own (accumulate.hip), which is all the tiles' walk needs.
 */
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <stdint.h>

/* exclusive prefix sum of n u32 values */
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

/* the same over n u64 values (bsc_prepare_templates_device: where every prepared read lands in the output buffer) */
extern "C" int bsc_dev_scan_tmp_bytes_u64(uint32_t n, size_t *bytes) {
  *bytes = 0;
  if (n == 0) return 0;
  return (int)rocprim::exclusive_scan(nullptr, *bytes, (const unsigned long long *)nullptr, (unsigned long long *)nullptr, 0ull, n,
                                      rocprim::plus<unsigned long long>(), (hipStream_t)0);
}
extern "C" int bsc_dev_scan_u64(const void *in, void *out, uint32_t n, void *tmp, size_t tmp_bytes, void *stream) {
  if (n == 0) return 0;
  return (int)rocprim::exclusive_scan(tmp, tmp_bytes, (const unsigned long long *)in, (unsigned long long *)out, 0ull, n,
                                      rocprim::plus<unsigned long long>(), (hipStream_t)stream);
}

/* inclusive running maximum of n u32 values (the read profile's vector length after every template); the scratch of the u64 scan
 * over 2 n + 1 values is more than this one needs */
extern "C" int bsc_dev_scan_max_u32(const void *in, void *out, uint32_t n, void *tmp, size_t tmp_bytes, void *stream) {
  if (n == 0) return 0;
  size_t need = 0;
  hipError_t e = rocprim::inclusive_scan(nullptr, need, (const uint32_t *)nullptr, (uint32_t *)nullptr, n, rocprim::maximum<uint32_t>(), (hipStream_t)0);
  if (e != hipSuccess) return (int)e;
  if (need > tmp_bytes) return (int)hipErrorInvalidValue;
  return (int)rocprim::inclusive_scan(tmp, need, (const uint32_t *)in, (uint32_t *)out, n, rocprim::maximum<uint32_t>(), (hipStream_t)stream);
}
