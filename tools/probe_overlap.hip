// How much independent VALU work hides under one MFMA on gfx950?  For the fp8 32x32x64 (64-cycle) and the bf16
// 32x32x16 (32-cycle) MFMA: cycles per {1 MFMA + NV independent VALU instructions} with one and two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_overlap.hip -o /tmp/probe_overlap && /tmp/probe_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

typedef __attribute__((ext_vector_type(4))) int pi4;
typedef __attribute__((ext_vector_type(2))) int pi2;
template <int NV, int KIND>
__device__ __forceinline__ void valu(float& x0, float& x1, float& x2, float& x3) {
  pi4 ldsv; pi2 ldsv2;
  typedef __attribute__((ext_vector_type(2))) float pf2;
  pf2 xp = {x0, x1};
  const int ldsa = (threadIdx.x & 63) * 16;
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    float& x = (i & 3) == 0 ? x0 : (i & 3) == 1 ? x1 : (i & 3) == 2 ? x2 : x3;
    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
    else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
    else if (KIND == 2) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(x));
    else if (KIND == 3) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %1" : "+v"(x) : "v"(x1));
    else if (KIND == 4) asm volatile("v_max3_i32 %0, %0, %1, %1" : "+v"(x) : "v"(x1));
    else if (KIND == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(ldsv) : "v"(ldsa));
    else if (KIND == 6) asm volatile("ds_read_b64_tr_b8 %0, %1" : "=v"(ldsv2) : "v"(ldsa));
    else if (KIND == 7) asm volatile("v_cvt_pk_fp8_f32 %0, %1, %1 op_sel:[0,0,1]" : "+v"(x) : "v"(x1));
    else if (KIND == 8) asm volatile("v_dot2_f32_f16 %0, %1, %1, %0" : "+v"(x) : "v"(x1));
    else if (KIND == 9) asm volatile("v_dot2_f32_bf16 %0, %1, %1, %0" : "+v"(x) : "v"(x1));
    else if (KIND == 10) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(xp) : "v"(xp));
    else if (KIND == 11) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %1" : "+v"(x) : "v"(x1));
    else if (KIND == 12) asm volatile("v_cvt_pk_f16_f32 %0, %1, %1" : "+v"(x) : "v"(x1));
    else if (KIND == 13) asm volatile("v_dot2c_f32_bf16 %0, %1, %1" : "+v"(x) : "v"(x1));
  }
  if (KIND == 5 || KIND == 6) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

template <int MODE, int NV, int KIND>
__global__ __launch_bounds__(512) void k(long long* out, int iters) {
  __shared__ char lds_dummy[4096];
  if (iters < 0) lds_dummy[threadIdx.x] = 1;
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i & 1); b[i] = 0x3c3c3c3c ^ (threadIdx.x << (i & 3)); }
  bf16x8 ha, hb;
  for (int i = 0; i < 8; ++i) { ha[i] = (__bf16)(1.0f + threadIdx.x * 0.01f); hb[i] = (__bf16)(0.5f + i * 0.1f); }
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  float x0 = threadIdx.x * 1e-3f, x1 = 0.5f, x2 = 0.25f, x3 = 0.125f;
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0, 0, 0);
      else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, c0, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      valu<NV, KIND>(x0, x1, x2, x3);
      __builtin_amdgcn_sched_barrier(0);
      if (MODE == 0) c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 0, 0, 0, 0, 0, 0);
      else c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb, ha, c1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      valu<NV, KIND>(x0, x1, x2, x3);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = x0 + x1 + x2 + x3;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  if (threadIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; }
}

template <int MODE, int NV, int KIND>
void run(const char* name, long long* d) {
  const int iters = 500;
  for (int threads : {256, 512}) {
    long long h[2];
    hipLaunchKernelGGL((k<MODE, NV, KIND>), dim3(1), dim3(threads), 0, 0, d, iters);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    // per SIMD: threads/256 waves, each issuing 8 MFMAs per iteration
    const double per_group = (double)h[0] / (iters * 8);
    printf("%-5s %s NV=%2d  %d waves/SIMD: %6.1f cycles per {MFMA + NV valu} per wave  (%.1f per SIMD-MFMA)\n",
           MODE == 0 ? "fp8" : "bf16", name, NV, threads / 256, per_group, per_group / (threads / 256));
  }
}

// whole-chip rate: `grid` workgroups of `threads` threads, wall time by HIP events
template <int MODE>
void chip(long long* d, int grid, int threads) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE, 0, 0>), dim3(grid), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE, 0, 0>), dim3(grid), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  const double flop_per = MODE == 0 ? 2.0 * 32 * 32 * 64 : 2.0 * 32 * 32 * 16;
  const double flops = (double)grid * (threads / 64) * iters * 8 * flop_per;
  printf("%-5s grid %4d x %3d threads: %.3f ms  %.0f TFLOP/s; block 0 wave 0: %.1f cycles per MFMA -> clock %.2f GHz\n",
         MODE == 0 ? "fp8" : "bf16", grid, threads, ms, flops / ms / 1e9, (double)h[0] / (iters * 8),
         (double)h[0] / (ms * 1e6));
}

// whole-chip: {1 MFMA + NV valu} groups with 1 and 2 waves per SIMD; cycles per group per SIMD from the wall time and
// the in-kernel clock of a 1-wave-per-SIMD block (s_memtime of wave 0 is only meaningful there)
template <int MODE, int NV, int KIND>
void chip_nv(long long* d, const char* name) {
  const int iters = 2000;
  double ms_[2];
  double clk = 0;
  for (int t = 0; t < 2; ++t) {
    const int threads = t == 0 ? 256 : 512;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV, KIND>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV, KIND>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms_[t] = ms;
    if (t == 0) { long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost); clk = (double)h[0] / (ms * 1e6); }
  }
  // groups per SIMD: iters * 8 * waves_per_simd
  const double c1 = ms_[0] * 1e6 * clk / (iters * 8.0), c2 = ms_[1] * 1e6 * clk / (iters * 8.0 * 2);
  printf("%-5s %s NV=%2d chip-wide: 1 wave/SIMD %6.1f cycles per group; 2 waves/SIMD %6.1f cycles per group (clock %.2f GHz)\n",
         MODE == 0 ? "fp8" : "bf16", name, NV, c1, c2, clk);
}

int main() {
  long long* d;
  hipMalloc(&d, 16);
#define CROW(MODE, KIND, NAME) chip_nv<MODE, 0, KIND>(d, NAME); chip_nv<MODE, 4, KIND>(d, NAME); chip_nv<MODE, 8, KIND>(d, NAME); \
  chip_nv<MODE, 12, KIND>(d, NAME); chip_nv<MODE, 16, KIND>(d, NAME); chip_nv<MODE, 24, KIND>(d, NAME);
  CROW(1, 8, "v_dot2_f32_f16")
  CROW(1, 9, "v_dot2_f32_bf16")
  CROW(1, 13, "v_dot2c_f32_bf16")
  CROW(1, 10, "v_pk_add_f32")
  CROW(1, 11, "v_cvt_pk_bf16_f32")
  CROW(1, 12, "v_cvt_pk_f16_f32")
  CROW(1, 0, "v_add_f32")
  hipDeviceSynchronize();
  return 0;
  for (int threads : {256, 512, 1024}) { chip<0>(d, 256, threads); chip<1>(d, 256, threads); }
  chip<0>(d, 512, 256); chip<1>(d, 512, 256);
#define ROW(MODE, KIND, NAME) run<MODE, 0, KIND>(NAME, d); run<MODE, 4, KIND>(NAME, d); run<MODE, 8, KIND>(NAME, d); \
  run<MODE, 12, KIND>(NAME, d); run<MODE, 16, KIND>(NAME, d); run<MODE, 24, KIND>(NAME, d);
  ROW(0, 0, "v_add ")
  ROW(0, 1, "v_exp ")
  ROW(0, 2, "v_max3")
  ROW(1, 0, "v_add ")
  ROW(1, 1, "v_exp ")
  return 0;
}
