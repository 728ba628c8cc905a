// Micro-benchmark: sustained integer VALU issue rate on gfx950 (wave-instructions per second), used to price the
// VALU-bound kernels (fast_cells, orient_describe).  Build: hipcc --offload-arch=gfx950 -O3 valu_bench.hip -o valu_bench
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ __launch_bounds__(256) void k_int(unsigned* out, int iters) {
  unsigned a = threadIdx.x, b = blockIdx.x, c = 3, d = 5, e = 7, f = 11, g = 13, h = 17;
  for (int i = 0; i < iters; i++) {
#pragma unroll
    for (int u = 0; u < 16; u++) {
      a = (a - b) | c; b = (b - c) & d; c = (c - d) | e; d = (d - e) & f;
      e = (e - f) | g; f = (f - g) & h; g = (g - h) | a; h = (h - a) & b;
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h;
}
int main() {
  unsigned* d;
  const int blocks = 256 * 8 * 4, iters = 2000;
  hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  k_int<<<blocks, 256>>>(d, 10);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k_int<<<blocks, 256>>>(d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  // per loop body: 16 * 8 * 2 = 256 VALU ops (sub + and/or; may fuse into v_sub + v_and_or: count ~ 2 per statement)
  double winstr = (double)blocks * 4 * iters * 256;
  printf("int VALU: %.3f ms, %.3e wave-instr/s (if 2 instr per statement), per SIMD per clock @2.4GHz: %.3f\n", ms,
         winstr / (ms * 1e-3), winstr / (ms * 1e-3) / 1024 / 2.4e9);
  return 0;
}
