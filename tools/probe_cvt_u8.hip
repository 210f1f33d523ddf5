// v_cvt_pk_u8_f32 on gfx950: what it rounds to, how it saturates, and how many issue cycles it takes next to
// v_exp_f32 / v_cvt_pk_fp8_f32 / v_add_f32 (one wave, 64 independent instructions per iteration, s_memtime).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_cvt_u8.hip -o /tmp/probe_cvt_u8 && /tmp/probe_cvt_u8
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>

__global__ void sem(const float* in, unsigned* out, int n) {
  const int i = threadIdx.x;
  if (i < n) out[i] = __builtin_amdgcn_cvt_pk_u8_f32(in[i], 1, 0xAABBCCDDu);
}

template <int KIND>
__global__ __launch_bounds__(64) void rate(long long* out, int iters) {
  float x[8];
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 0.01f + i;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 64; ++u) {
      float& v = x[u & 7];
      if (KIND == 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(v));
      else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v));
      else if (KIND == 2) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %1" : "+v"(v) : "v"(x[(u + 1) & 7]));
      else if (KIND == 3) asm volatile("v_cvt_pk_u8_f32 %0, %1, 1, %0" : "+v"(v) : "v"(x[(u + 1) & 7]));
      else if (KIND == 4) asm volatile("v_perm_b32 %0, %1, %0, %1" : "+v"(v) : "v"(x[(u + 1) & 7]));
      else if (KIND == 5) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v) : "v"(x[(u + 1) & 7]));
      else if (KIND == 6) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(v));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) s += x[i];
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; }
}

int main() {
  const float h[] = {0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 2.51f, 3.5f, -3.f, 254.6f, 300.f, -INFINITY, NAN, INFINITY, 126.49f};
  const int n = sizeof(h) / sizeof(h[0]);
  float* di; unsigned* dout;
  hipMalloc(&di, sizeof(h)); hipMalloc(&dout, n * 4);
  hipMemcpy(di, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(sem, dim3(1), dim3(64), 0, 0, di, dout, n);
  unsigned ho[32];
  hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("cvt_pk_u8_f32(%g, byte 1, 0xAABBCCDD) = 0x%08x -> %u\n", h[i], ho[i], (ho[i] >> 8) & 255);
  long long* d; hipMalloc(&d, 16);
  const char* names[] = {"v_add_f32", "v_exp_f32", "v_cvt_pk_fp8_f32", "v_cvt_pk_u8_f32", "v_perm_b32", "v_fma_f32", "v_cvt_u32_f32"};
#define RUN(K) { hipLaunchKernelGGL((rate<K>), dim3(1), dim3(64), 0, 0, d, 2000); long long r[2]; hipMemcpy(r, d, 16, hipMemcpyDeviceToHost); \
    printf("%-18s %.2f memtime ticks per instruction (one wave)\n", names[K], (double)r[0] / (2000.0 * 64)); }
  RUN(0) RUN(1) RUN(2) RUN(3) RUN(4) RUN(5) RUN(6)
  return 0;
}
