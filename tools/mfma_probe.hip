// Checks the operand / result lane layout of v_mfma_i32_16x16x64_i8 that gauss_blur7_mfma_kernel relies on:
//   A: lane (q = lane >> 4, m = lane & 15) holds 16 bytes of row m;  B: lane (q, n) holds 16 bytes of column n;
//   byte j of lane group q of A pairs with byte j of lane group q of B (whatever k that is);
//   D: lane (q', n), register i  <->  D[m = 4q' + i][n].
// hipcc --offload-arch=gfx950 -O2 tools/mfma_probe.hip -o tools/mfma_probe && tools/mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef int v4i __attribute__((ext_vector_type(4)));
__global__ void probe(const v4i* a, const v4i* b, v4i* d) {
  const v4i z = {0, 0, 0, 0};
  d[threadIdx.x] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[threadIdx.x], b[threadIdx.x], z, 0, 0, 0);
}
int main() {
  int8_t ha[64][16], hb[64][16];
  int32_t hd[64][4];
  srand(7);
  for (int l = 0; l < 64; l++)
    for (int j = 0; j < 16; j++) { ha[l][j] = (int8_t)(rand() % 256 - 128); hb[l][j] = (int8_t)(rand() % 256 - 128); }
  v4i *da, *db, *dd;
  hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dd, 1024);
  hipMemcpy(da, ha, 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, da, db, dd);
  if (hipMemcpy(hd, dd, 1024, hipMemcpyDeviceToHost) != hipSuccess) { printf("mfma_probe: HIP error\n"); return 2; }
  int bad = 0;
  for (int qp = 0; qp < 4; qp++)
    for (int n = 0; n < 16; n++)
      for (int i = 0; i < 4; i++) {
        const int m = 4 * qp + i;
        int acc = 0;
        for (int q = 0; q < 4; q++)
          for (int j = 0; j < 16; j++) acc += (int)ha[q * 16 + m][j] * (int)hb[q * 16 + n][j];
        if (acc != hd[qp * 16 + n][i]) bad++;
      }
  printf("mfma_probe: v_mfma_i32_16x16x64_i8 layout %s (%d of 1024 elements differ)\n", bad ? "DIFFERS" : "as assumed", bad);
  return bad ? 1 : 0;
}
