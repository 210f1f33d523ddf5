// What does ONE tile request cost the wave that issues it?  (round 4: the int8-score step spends ~180 of its ~1 600 cycles per key
// block on two LDS-DMA requests per wave, wherever they stand -- profiles/r04_i8_ablation.txt.)  512-thread workgroups, one per CU,
// two waves per SIMD as in the attention kernels; every wave runs a loop of {NV independent v_fma_f32, one request}; the request
// is  0: nothing | 1: buffer_load_dwordx4 ... lds (1 KiB per wave, as the kernels) | 2: global_load_dwordx4 into registers,
// stored with ds_write_b128 one iteration later | 3: the load alone (never stored) | 4: the ds_write_b128 alone.
// Cycles per iteration (s_memtime) minus kind 0 = the cost of the request to its wave.  Rows come from a 64 MiB region (L2 /
// Infinity Cache hits mostly), a different row block per wave and iteration.
//   hipcc --offload-arch=gfx950 -O3 tools/probe_request_cost.hip -o /tmp/probe_req && /tmp/probe_req
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((ext_vector_type(4))) int i32x4;
#define LDS_AS __attribute__((address_space(3)))

template <int KIND, int NV>
__global__ __launch_bounds__(512, 2) void k(long long* out, const char* src, int rows, int iters) {
  __shared__ __attribute__((aligned(16))) char smem[2 * 8 * 1024];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, 0x7fffffff, 0x00020000);
  float x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3f + i;
  unsigned row = (blockIdx.x * 977u + wave * 131u) % (unsigned)rows;
  i32x4 hold = {0, 0, 0, 0};
  __syncthreads();
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    const int off = (int)(row * 128u) + ((lane & 7) << 4) + (lane >> 3) * 128;  // 8 rows of 128 bytes per request
    if (KIND == 1) {
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (LDS_AS void*)(smem + (it & 1) * 8192 + wave * 1024), 16, off, 0, 0, 0);
    } else if (KIND == 2 || KIND == 3) {
      if (KIND == 2) *(i32x4*)(smem + (it & 1) * 8192 + wave * 1024 + lane * 16) = hold;  // last iteration's data
      hold = *(const i32x4*)(src + off);
    } else if (KIND == 4) {
      *(i32x4*)(smem + (it & 1) * 8192 + wave * 1024 + lane * 16) = hold;
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NV; ++i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(x[i & 7]) : "v"(x[(i + 1) & 7]));
    __builtin_amdgcn_sched_barrier(0);
    row += 8 * 8 * 256;
    if (row >= (unsigned)rows) row -= (unsigned)rows;
    if ((it & 7) == 7) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // the kernels drain once per key block
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  s += (float)hold[0] + (float)smem[lane];
  if (lane == 0) { out[(blockIdx.x * 8 + wave) * 2] = t1 - t0; out[(blockIdx.x * 8 + wave) * 2 + 1] = (long long)s; }
}

template <int KIND, int NV>
double run(long long* d, const char* src, int rows) {
  const int G = 256, iters = 4000;
  hipLaunchKernelGGL((k<KIND, NV>), dim3(G), dim3(512), 0, 0, d, src, rows, iters);
  hipLaunchKernelGGL((k<KIND, NV>), dim3(G), dim3(512), 0, 0, d, src, rows, iters);
  hipDeviceSynchronize();
  static long long h[256 * 8 * 2];
  hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  double tot = 0;
  for (int i = 0; i < G * 8; ++i) tot += (double)h[2 * i];
  return tot / (G * 8) / iters;
}

template <int NV>
void line(long long* d, const char* src, int rows) {
  const double b = run<0, NV>(d, src, rows);
  const double a1 = run<1, NV>(d, src, rows), a2 = run<2, NV>(d, src, rows), a3 = run<3, NV>(d, src, rows), a4 = run<4, NV>(d, src, rows);
  printf("%3d v_fma per iteration: none %7.1f | LDS-DMA %+7.1f | load + ds_write %+7.1f | load alone %+7.1f | ds_write alone %+7.1f  cycles\n",
         NV, b, a1 - b, a2 - b, a3 - b, a4 - b);
}

int main() {
  long long* d;
  char* src;
  const int rows = 1 << 19;  // x 128 B = 64 MiB
  hipMalloc(&d, sizeof(long long) * 256 * 8 * 2);
  hipMalloc(&src, (size_t)rows * 128 + 65536);
  hipMemset(src, 1, (size_t)rows * 128 + 65536);
  printf("cost of one tile request to the wave that issues it (every wave of the chip issuing; cycles per iteration above the request-free loop)\n");
  line<8>(d, src, rows);
  line<24>(d, src, rows);
  line<48>(d, src, rows);
  line<96>(d, src, rows);
  return 0;
}
