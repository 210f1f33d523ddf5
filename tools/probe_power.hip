// What does the chip sustain on RANDOM operands?  Pure MFMA loops (operands in registers, 4 operand sets cycled so the
// multipliers toggle), every CU busy, 1 and 2 waves per SIMD; wall time by HIP events, clock from s_memtime.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_power.hip -o /tmp/probe_power && /tmp/probe_power
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x; }

template <int MODE, int RANDOM>
__global__ __launch_bounds__(512) void k(long long* out, int iters) {
  i32x8 a[4], b[4];
  bf16x8 ha[4], hb[4];
  f16x8 fa[4], fb[4];
  for (int s = 0; s < 4; ++s)
    for (int i = 0; i < 8; ++i) {
      unsigned r0 = hash(threadIdx.x * 64 + s * 16 + i), r1 = hash(r0 + 12345);
      // e4m3 bytes with exponent field <= 12: finite, magnitudes up to 60
      a[s][i] = RANDOM ? (int)(r0 & 0xe7e7e7e7u) : 0x38383838;
      b[s][i] = RANDOM ? (int)(r1 & 0xe7e7e7e7u) : 0x38383838;
      ha[s][i] = RANDOM ? (__bf16)(((int)(r0 & 0xffff) - 32768) * (1.0f / 8192)) : (__bf16)1.0f;
      hb[s][i] = RANDOM ? (__bf16)(((int)(r1 & 0xffff) - 32768) * (1.0f / 8192)) : (__bf16)1.0f;
      fa[s][i] = RANDOM ? (_Float16)(((int)(r0 & 0xffff) - 32768) * (1.0f / 8192)) : (_Float16)1.0f;
      fb[s][i] = RANDOM ? (_Float16)(((int)(r1 & 0xffff) - 32768) * (1.0f / 8192)) : (_Float16)1.0f;
    }
  f32x16 c0, c1;
  for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (MODE == 0) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[u], b[u], c0, 0, 0, 0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b[u], a[(u + 1) & 3], c1, 0, 0, 0, 0, 0, 0);
      } else if (MODE == 1) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha[u], hb[u], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(hb[u], ha[(u + 1) & 3], c1, 0, 0, 0);
      } else {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[u], fb[u], c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[u], fa[(u + 1) & 3], c1, 0, 0, 0);
      }
    }
    if (RANDOM && (it & 63) == 63) {  // keep the accumulators finite
      for (int i = 0; i < 16; ++i) { c0[i] *= 1e-3f; c1[i] *= 1e-3f; }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 16; ++i) s += c0[i] + c1[i];
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = (long long)s; }
}

template <int MODE, int RANDOM>
void run(long long* d, int grid, int threads) {
  const int iters = 40000;  // ~10+ ms per launch so the clock settles
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(grid), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e0);
  const int reps = 5;
  for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<MODE, RANDOM>), dim3(grid), dim3(threads), 0, 0, d, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  long long h[2]; hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
  const double flop_per = MODE == 0 ? 2.0 * 32 * 32 * 64 : 2.0 * 32 * 32 * 16;
  const double flops = (double)grid * (threads / 64) * iters * 8 * flop_per;
  const double clk = threads == 256 ? (double)h[0] / (ms * 1e6) : 0.0;
  printf("%-5s %-8s grid %4d x %3d: %8.3f ms  %6.0f TFLOP/s  clock %.2f GHz\n", MODE == 0 ? "fp8" : MODE == 1 ? "bf16" : "fp16",
         RANDOM ? "random" : "constant", grid, threads, ms, flops / ms / 1e9, clk);
}

int main() {
  long long* d;
  hipMalloc(&d, 16);
  for (int rep = 0; rep < 2; ++rep) {
    run<0, 0>(d, 256, 256); run<0, 1>(d, 256, 256); run<0, 1>(d, 256, 512);
    run<1, 0>(d, 256, 256); run<1, 1>(d, 256, 256); run<1, 1>(d, 256, 512);
    run<2, 0>(d, 256, 256); run<2, 1>(d, 256, 256); run<2, 1>(d, 256, 512);
  }
  return 0;
}
