// Micro-benchmark: what does a byte-misaligned dword access cost on gfx950?  Every lane copies 4 bytes; source and destination are
// shifted by 0..3 bytes.  build: hipcc --offload-arch=gfx950 -O3 -o /tmp/unaligned_copy tools/micro/unaligned_copy.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k(const uint8_t *__restrict__ a, uint8_t *__restrict__ b, size_t n4, int sa, int sb) {
  for (size_t i = blockIdx.x * 256ull + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256ull) {
    uint32_t v;
    __builtin_memcpy(&v, a + 4 * i + sa, 4);
    __builtin_memcpy(b + 4 * i + sb, &v, 4);
  }
}
int main() {
  const size_t n = 1536ull << 20;
  uint8_t *a, *b;
  hipMalloc(&a, n + 64);
  hipMalloc(&b, n + 64);
  hipMemset(a, 1, n + 64);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int sa = 0; sa < 2; sa++)
    for (int sb = 0; sb < 4; sb += 1) {
      float best = 1e9;
      for (int r = 0; r < 4; r++) {
        hipEventRecord(e0);
        k<<<256 * 8, 256>>>(a, b, n / 4, sa, sb);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
      }
      printf("src shift %d dst shift %d: %.3f ms  %.0f GB/s (read + write)\n", sa, sb, best, 2.0 * n / best / 1e6);
    }
  return 0;
}
