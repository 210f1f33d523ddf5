// Score-weighted sum of the three experts' outputs for gfx950: include/vorta_hip.h vorta_mix_experts.
//
// The training-time forward of the reference runs every head through all three experts and mixes the results
// (`_combine_attn_outputs`, hunyuan.py:509-513, wan.py:296-300: stack to (B,H,3,S,D), multiply by the scores,
// sum over the expert axis -- two more full-size temporaries).  Here: one pass, three reads + one write per
// element, fp32 accumulation, one rounding.  HBM-bound (4 x rows x 256 B per head); a 16-lane quarter wave
// per 256-byte row.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

namespace {

struct MParams {
  const char* x[3]; int64_t x_sh[3], x_ss[3];  // bytes
  char* o; int64_t o_sh, o_ss;
  const void* scores;  // [heads][3] in the I/O dtype (batch item 0)
  int heads, n_rows;
};

template <typename T>
__global__ __launch_bounds__(256) void mix_experts_kernel(const MParams p) {
  typedef __attribute__((ext_vector_type(8))) T V8;
  const int64_t item = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  if (item >= (int64_t)p.heads * p.n_rows) return;
  const int head = (int)(item / p.n_rows);
  const int64_t row = item - (int64_t)head * p.n_rows;
  const T* sc = (const T*)p.scores + head * 3;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int e = 0; e < 3; ++e) {
    const float w = (float)sc[e];
    const V8 xv = *(const V8*)(p.x[e] + (int64_t)head * p.x_sh[e] + row * p.x_ss[e] + sub * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] += w * (float)xv[i];
  }
  V8 ov;
#pragma unroll
  for (int i = 0; i < 8; ++i) ov[i] = (T)acc[i];
  *(V8*)(p.o + (int64_t)head * p.o_sh + row * p.o_ss + sub * 16) = ov;
}

}  // namespace

extern "C" int vorta_mix_experts(const vorta_mix_args* a, void* hip_stream) {
  if (!a || a->struct_size != sizeof(vorta_mix_args)) return VORTA_EINVAL;
  if (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16) return VORTA_EUNSUPPORTED;
  if (a->head_dim != 128 || a->n_experts != 3) return VORTA_EUNSUPPORTED;
  if (a->heads <= 0 || a->n_rows < 0 || !a->scores || !a->out.ptr) return VORTA_EINVAL;
  if (a->n_rows == 0) return VORTA_OK;
  MParams p;
  for (int e = 0; e < 3; ++e) {
    const vorta_tensor& t = a->x[e];
    if (!t.ptr || ((uintptr_t)t.ptr & 15) || (t.stride_s % 8) || (t.stride_h % 8)) return VORTA_EINVAL;
    p.x[e] = (const char*)t.ptr; p.x_sh[e] = t.stride_h * 2; p.x_ss[e] = t.stride_s * 2;
  }
  if (((uintptr_t)a->out.ptr & 15) || (a->out.stride_s % 8) || (a->out.stride_h % 8)) return VORTA_EINVAL;
  p.o = (char*)a->out.ptr; p.o_sh = a->out.stride_h * 2; p.o_ss = a->out.stride_s * 2;
  p.scores = a->scores; p.heads = a->heads; p.n_rows = a->n_rows;
  const int64_t items = (int64_t)p.heads * p.n_rows;
  if (items > 0x7fffffff0ll) return VORTA_EINVAL;
  hipStream_t st = (hipStream_t)hip_stream;
  const dim3 grid((unsigned)((items + 15) / 16));
  if (a->dtype == VORTA_BF16) hipLaunchKernelGGL(mix_experts_kernel<__bf16>, grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL(mix_experts_kernel<_Float16>, grid, dim3(256), 0, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}
