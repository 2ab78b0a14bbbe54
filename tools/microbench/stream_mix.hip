// stream_mix.hip — what the memory system gives for the calling kernel's traffic mix: every position reads 104 + 1 bytes
// and writes 200 + 1 bytes, nothing else.  Coalesced 16-byte loads / stores (plain or non-temporal), grid-stride.
// build: hipcc -O3 --offload-arch=gfx950 tools/microbench/stream_mix.hip -o /tmp/stream_mix ; run: /tmp/stream_mix [positions]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef unsigned int u4 __attribute__((ext_vector_type(4)));

// one wave per 64-position tile, exactly the calling kernel's shape: 6 656 contiguous bytes in (6.5 x 64 lanes x 16 B),
// 12 800 contiguous bytes out (12.5 x 64 x 16 B); tiles dealt to waves grid-stride
// MODE 0: read + write (the calling kernel's mix), 1: reads only (one store per wave keeps them alive), 2: writes only
template <bool NT, int MODE = 0>
__global__ __launch_bounds__(256) void mix_kernel(const u4 *__restrict__ in, u4 *__restrict__ out, uint64_t n_tiles) {
  const unsigned lane = threadIdx.x & 63u;
  const uint64_t wave = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = ((uint64_t)gridDim.x * blockDim.x) >> 6;
  for (uint64_t t = wave; t < n_tiles; t += n_waves) {
    const u4 *src = in + t * 416u;
    u4 *dst = out + t * 800u;
    u4 v[7];
#pragma unroll
    for (int j = 0; j < 7; j++) {
      const unsigned idx = j * 64u + lane;
      v[j] = (MODE != 2 && idx < 416u) ? (NT ? __builtin_nontemporal_load(src + idx) : src[idx]) : (u4){(unsigned)t, 0u, 0u, 0u};
    }
    if (MODE == 1) {
      u4 acc = v[0];
#pragma unroll
      for (int j = 1; j < 7; j++) acc.x ^= v[j].x ^ v[j].y ^ v[j].z ^ v[j].w;
      if (acc.x == 0x12345678u) dst[lane] = acc; /* never true for the test pattern */
      continue;
    }
#pragma unroll
    for (int k = 0; k < 13; k++) {
      const unsigned idx = k * 64u + lane;
      u4 w = v[k % 7];
      w.x ^= (unsigned)k;
      if (idx < 800u) {
        if (NT) __builtin_nontemporal_store(w, dst + idx); else dst[idx] = w;
      }
    }
  }
}

int main(int argc, char **argv) {
  const uint64_t n = argc > 1 ? strtoull(argv[1], 0, 10) : 50000000ull;
  const uint64_t in_b = n * 104, out_b = n * 200;
  u4 *in, *out;
  hipMalloc(&in, in_b);
  hipMalloc(&out, out_b);
  hipMemset(in, 1, in_b);
  hipMemset(out, 0, out_b);
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  int cus = 256;
  for (int nt = 0; nt < 2; nt++)
    for (int mult = 4; mult <= 32; mult *= 2) {
      float best = 1e9f;
      for (int r = 0; r < 5; r++) {
        hipEventRecord(a);
        if (nt) hipLaunchKernelGGL(mix_kernel<true>, dim3(cus * mult), dim3(256), 0, 0, in, out, n / 64);
        else hipLaunchKernelGGL(mix_kernel<false>, dim3(cus * mult), dim3(256), 0, 0, in, out, n / 64);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
      }
      printf("%s stores/loads, %4d workgroups per CU-set (x%d): %.3f ms  -> %.2f TB/s (read %.2f GB + write %.2f GB)\n",
             nt ? "nt   " : "plain", cus * mult, mult, best, (in_b + out_b) / best / 1e9, in_b / 1e9, out_b / 1e9);
    }
  for (int mode = 1; mode <= 2; mode++) {
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
      hipEventRecord(a);
      if (mode == 1) hipLaunchKernelGGL((mix_kernel<true, 1>), dim3(cus * 8), dim3(256), 0, 0, in, out, n / 64);
      else hipLaunchKernelGGL((mix_kernel<true, 2>), dim3(cus * 8), dim3(256), 0, 0, in, out, n / 64);
      hipEventRecord(b);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      if (ms < best) best = ms;
    }
    const double bytes = mode == 1 ? (double)in_b : (double)out_b;
    printf("%s only (nt): %.3f ms -> %.2f TB/s\n", mode == 1 ? "reads " : "writes", best, bytes / best / 1e9);
  }
  return 0;
}
