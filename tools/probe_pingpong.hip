// Can one wave's VALU stream run under ANOTHER wave's MFMA stream on the same SIMD?  512-thread workgroups (waves w and
// w+4 share a SIMD), every CU busy: waves 0-3 issue only fp8 32x32x64 MFMAs, waves 4-7 only VALU of one kind; timed
// alone and together (wall clock by HIP events).
//   hipcc --offload-arch=gfx950 -O3 tools/probe_pingpong.hip -o /tmp/probe_pingpong && /tmp/probe_pingpong
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

// mode bit0: first half of the waves runs MFMAs; bit1: second half runs VALU of kind KIND
template <int KIND>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode, int swap) {
  const int wave = threadIdx.x >> 6;
  const bool first = (wave < 4) != (swap != 0);
  i32x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = 0x38383838 + threadIdx.x * 0x01010101 * (i & 1); b[i] = 0x3c3c3c3c ^ (threadIdx.x << (i & 3)); }
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  float x0 = threadIdx.x * 1e-3f, x1 = 0.5f, x2 = 0.25f, x3 = 0.125f;
  if (first) {
    if (mode & 1)
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c0, 0, 0, 0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b, a, c1, 0, 0, 0, 0, 0, 0);
        }
      }
  } else if (mode & 2) {
    // the same wall time as the MFMA stream if it ran alone: 8 MFMAs x 64 cycles = 512 cycles per iteration
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < (KIND == 0 ? 112 : 56); ++u) {
        float& x = (u & 3) == 0 ? x0 : (u & 3) == 1 ? x1 : (u & 3) == 2 ? x2 : x3;
        if (KIND == 0) asm volatile("v_add_f32 %0, %0, %0" : "+v"(x));
        else if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
        else asm volatile("v_cvt_pk_fp8_f32 %0, %1, %1" : "+v"(x) : "v"(x1));
      }
    }
  }
  float s = x0 + x1 + x2 + x3;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char* name, float* d) {
  const int iters = 20000;
  for (int swap = 0; swap < 2; ++swap) {
    float ms[4] = {0, 0, 0, 0};
    for (int mode = 1; mode <= 3; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0); hipEventCreate(&e1);
      hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, d, iters, mode, swap);
      hipEventRecord(e0);
      hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(512), 0, 0, d, iters, mode, swap);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms[mode], e0, e1);
    }
    printf("%-12s %s: MFMA waves alone %.3f ms, VALU waves alone %.3f ms, together %.3f ms  (max %.3f, sum %.3f)\n", name,
           swap ? "MFMA on waves 4-7" : "MFMA on waves 0-3", ms[1], ms[2], ms[3], ms[1] > ms[2] ? ms[1] : ms[2], ms[1] + ms[2]);
  }
}

int main() {
  float* d;
  hipMalloc(&d, 16);
  run<0>("v_add_f32", d);
  run<1>("v_exp_f32", d);
  run<2>("cvt_pk_fp8", d);
  return 0;
}
