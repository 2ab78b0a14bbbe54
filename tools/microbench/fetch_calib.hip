/*
 * fetch_calib.hip — what rocprofv3's FETCH_SIZE reports for the access patterns of the reads-in kernels, on known byte counts
 * (MI355X_MICROARCH.md: FETCH_SIZE is calibrated for 16 B/lane streaming reads only — it shows half of those — "other access
 * widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Each kernel reads N bytes exactly once:
 *   k_bytes      one byte per lane, 64 consecutive bytes per wave instruction (the walk's buffer_load_ubyte over a read)
 *   k_desc24     24-byte records, one per lane (the candidate descriptors: dwordx4 + dwordx2 at stride 24)
 *   k_tpl40      40-byte records, one per lane (the templates)
 *   k_wide16     16 bytes per lane (the calibrated case, for reference)
 * build: hipcc -O3 --offload-arch=gfx950 tools/microbench/fetch_calib.hip -o gpurun_out/fetch_calib
 * run:   rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- gpurun_out/fetch_calib
 * FETCH_SIZE is in KiB; N = 1 GiB here, so a kernel that shows 1 048 576 counts every byte once.
 */
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#define N (1ull << 30)

__global__ void k_bytes(const uint8_t *p, uint32_t *sink) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += stride) acc += p[i];
  if (acc == 0x12345678u) sink[0] = acc;
}
struct d24 { uint32_t a, b; int64_t base; uint32_t meta, lut; };
__global__ void k_desc24(const d24 *p, uint32_t *sink) {
  const uint64_t n = N / 24, stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const d24 e = p[i];
    acc += e.a ^ e.b ^ (uint32_t)e.base ^ e.meta ^ e.lut;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
struct t40 { uint32_t pos[2], len[2]; uint64_t off[2]; uint8_t mapq[2], ori, bs; uint32_t pad; };
__global__ void k_tpl40(const t40 *p, uint32_t *sink) {
  const uint64_t n = N / 40, stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const t40 e = p[i];
    acc += e.pos[0] ^ e.pos[1] ^ e.len[0] ^ e.len[1] ^ (uint32_t)e.off[0] ^ (uint32_t)e.off[1] ^ e.mapq[0] ^ e.ori ^ e.pad;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
__global__ void k_wide16(const uint4 *p, uint32_t *sink) {
  const uint64_t n = N / 16, stride = (uint64_t)gridDim.x * blockDim.x;
  uint32_t acc = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const uint4 e = p[i];
    acc += e.x ^ e.y ^ e.z ^ e.w;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}

int main() {
  void *p;
  uint32_t *sink;
  if (hipMalloc(&p, N) != hipSuccess || hipMalloc((void **)&sink, 64) != hipSuccess) return 1;
  hipMemset(p, 1, N);
  hipDeviceSynchronize();
  for (int rep = 0; rep < 2; rep++) {
    hipLaunchKernelGGL(k_bytes, dim3(256 * 16), dim3(256), 0, 0, (const uint8_t *)p, sink);
    hipLaunchKernelGGL(k_desc24, dim3(256 * 16), dim3(256), 0, 0, (const d24 *)p, sink);
    hipLaunchKernelGGL(k_tpl40, dim3(256 * 16), dim3(256), 0, 0, (const t40 *)p, sink);
    hipLaunchKernelGGL(k_wide16, dim3(256 * 16), dim3(256), 0, 0, (const uint4 *)p, sink);
  }
  if (hipDeviceSynchronize() != hipSuccess) return 2;
  printf("done: every kernel read %llu bytes once\n", (unsigned long long)N);
  return 0;
}
