/*
 * probe.hip — measurement support: what the memory system delivers for the calling kernel's traffic and nothing else.
 * One wave per 64-position tile, exactly the calling kernel's shape — 6 656 contiguous bytes in (+ 64 reference codes),
 * 12 800 contiguous bytes out (+ 64 skip bytes), 16 bytes per lane, non-temporal — with no arithmetic in between.  The
 * time of this kernel on the benchmark's own buffers is the practical ceiling for any kernel that has to move the
 * reference's 104-byte pile-ups in and 200-byte gt_meth records out: HBM3E writes stream slower than reads, so a
 * 1 : 2 read : write mix tops out near 5.5 TB/s on MI355X, not at the 8 TB/s headline.  bench.py reports the calling
 * kernel against both.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned int probe_u4 __attribute__((ext_vector_type(4)));

extern "C" __global__ __launch_bounds__(256) void bsc_stream_probe_kernel(const probe_u4 *__restrict__ cts,
                                                                          const uint8_t *__restrict__ ref,
                                                                          probe_u4 *__restrict__ out,
                                                                          uint8_t *__restrict__ skip, uint64_t n_tiles) {
  const unsigned lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const uint64_t n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t t = wave; t < n_tiles; t += n_waves) {
    const probe_u4 *src = cts + t * 416u; /* 64 x 104 B */
    probe_u4 *dst = out + t * 800u;       /* 64 x 200 B */
    probe_u4 v[7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const unsigned idx = j * 64u + lane;
      v[j] = idx < 416u ? __builtin_nontemporal_load(src + idx) : (probe_u4){0u, 0u, 0u, 0u};
    }
    const unsigned r = ref[t * 64u + lane];
#pragma unroll
    for (int k = 0; k < 13; k++) {
      const unsigned idx = k * 64u + lane;
      probe_u4 w = v[k % 7];
      w.x ^= (unsigned)k + r;
      if (idx < 800u) __builtin_nontemporal_store(w, dst + idx);
    }
    skip[t * 64u + lane] = (uint8_t)(v[0].x & 1u);
  }
}

/* n must be a multiple of 64; the buffers are those of bsc_call_sites_device (out_stride 200) */
extern "C" int bsc_dev_launch_stream_probe(const void *cts, const void *ref, uint64_t n, void *out, void *skip, int num_cus,
                                           void *stream) {
  if (n < 64) return 0;
  hipLaunchKernelGGL(bsc_stream_probe_kernel, dim3((unsigned)num_cus * 16u), dim3(256), 0, (hipStream_t)stream,
                     (const probe_u4 *)cts, (const uint8_t *)ref, (probe_u4 *)out, (uint8_t *)skip, n / 64u);
  return (int)hipGetLastError();
}
