// Operand lane map of v_mfma_i32_32x32x32_i8 on gfx950, checked with exact integer data: is byte j of lane (r = l & 31,
// h = l >> 5) element [row r][k = 16 h + j] of A (and [k][col r] of B)?  C/D: col = lane & 31, row = (i & 3) + 8 (i >> 2) + 4 h.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_i8_layout.hip -o /tmp/probe_i8 && /tmp/probe_i8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

__global__ void k(const signed char* A, const signed char* B, int* Cout) {  // A[32][32] row-major (row, k); B[32][32] (k, col)
  const int l = threadIdx.x, r = l & 31, h = l >> 5;
  i32x4 a, b;
  for (int w = 0; w < 4; ++w) {
    unsigned wa = 0, wb = 0;
    for (int j = 0; j < 4; ++j) {
      const int kk = 16 * h + 4 * w + j;
      wa |= ((unsigned)(unsigned char)A[r * 32 + kk]) << (8 * j);
      wb |= ((unsigned)(unsigned char)B[kk * 32 + r]) << (8 * j);
    }
    a[w] = (int)wa; b[w] = (int)wb;
  }
  i32x16 c;
  for (int i = 0; i < 16; ++i) c[i] = 0;
  c = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
  for (int i = 0; i < 16; ++i) Cout[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

int main() {
  signed char hA[1024], hB[1024];
  srand(7);
  for (int i = 0; i < 1024; ++i) { hA[i] = (signed char)(rand() % 255 - 127); hB[i] = (signed char)(rand() % 255 - 127); }
  signed char *dA, *dB; int* dC;
  hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096);
  hipMemcpy(dA, hA, 1024, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 1024, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dC);
  int hC[1024];
  hipMemcpy(hC, dC, 4096, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      int s = 0;
      for (int kk = 0; kk < 32; ++kk) s += (int)hA[i * 32 + kk] * (int)hB[kk * 32 + j];
      bad += s != hC[i * 32 + j];
    }
  printf("v_mfma_i32_32x32x32_i8 with byte j of lane (r, h) = k 16 h + j: %d of 1024 outputs wrong -> %s\n", bad, bad ? "MAP WRONG" : "map confirmed");
  return bad != 0;
}
