// vorta_fp8_quantize_qkv (include/vorta_hip.h): post-RoPE q,k,v (bf16 / fp16) -> e4m3 copies for the fp8 attention
// kernels, with the softmax scale and log2(e) folded into the q/k multipliers.  HBM-bound: the abs-max pass reads
// every element once (2 B), the convert pass reads it again and writes 1 B.  Three launches, no host round trip.
//
// Workspace (floats): amax_q[H] | amax_k[H] | amax_v[H][D] | qmul[H] | kmul[H] | vmul[H][D]
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

namespace {

constexpr int D = 128;
constexpr float V_TARGET = 240.f;  // amax of a v channel maps here (e4m3 max 448; relative precision is range-independent)
constexpr float E4M3_MAX = 448.f;

struct QParams {
  const char* x[3]; int64_t x_sh[3], x_ss[3];  // inputs (bytes)
  char* y[3]; int64_t y_sh[3], y_ss[3];        // outputs (bytes)
  int heads, n_tokens, rows_per_block;
  float c0;  // qk_scale * log2(e)
  float* ws; float* v_descale;
  int v_per_head;
};

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }

// grid (row chunks, heads, 3): per-head abs-max of q and k, per-(head, channel) abs-max of v
template <typename T>
__global__ __launch_bounds__(256) void fp8_absmax_kernel(const QParams p) {
  typedef __attribute__((ext_vector_type(8))) T T8;
  const int which = blockIdx.z, h = blockIdx.y;
  const int t = threadIdx.x, cc = t & 15, rl = t >> 4;
  const int r0 = blockIdx.x * p.rows_per_block, r1 = min(r0 + p.rows_per_block, p.n_tokens);
  const char* base = p.x[which] + (int64_t)h * p.x_sh[which] + cc * 16;
  float m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = 0.f;
  for (int r = r0 + rl; r < r1; r += 16) {
    const T8 v = *(const T8*)(base + (int64_t)r * p.x_ss[which]);
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], fabsf(to_f(v[i])));
  }
  __shared__ float red[16][D + 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cc * 8 + i] = m[i];
  __syncthreads();
  float cm = 0.f;
  if (t < D) {
#pragma unroll
    for (int j = 0; j < 16; ++j) cm = fmaxf(cm, red[j][t]);
  }
  unsigned* ws = (unsigned*)p.ws;
  const int H = p.heads;
  if (which == 2) {
    if (t < D) atomicMax(ws + 2 * H + h * D + t, __float_as_uint(cm));
    return;
  }
  // q / k: one value per head
  __syncthreads();
  if (t < D) red[0][t] = cm;
  __syncthreads();
  if (t < 64) {
    float x = fmaxf(red[0][t], red[0][t + 64]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) x = fmaxf(x, __shfl_xor(x, off));
    if (t == 0) atomicMax(ws + which * H + h, __float_as_uint(x));
  }
}

// grid (heads), 128 threads: multipliers from the abs-max slots
__global__ __launch_bounds__(128) void fp8_scales_kernel(const QParams p) {
  const int h = blockIdx.x, d = threadIdx.x, H = p.heads;
  const float mq = p.ws[h], mk = p.ws[H + h];
  float* qmul = p.ws + 2 * H + H * D;
  float* kmul = qmul + H;
  float* vmul = kmul + H;
  if (d == 0) {
    // q8 . k8 = c0 * q . k with both operand maxima at sqrt(c0 * mq * mk)
    float t = 1.f;
    if (mq > 0.f && mk > 0.f) t = sqrtf(mk / (p.c0 * mq));
    qmul[h] = p.c0 * t;
    kmul[h] = 1.f / t;
  }
  float mv = p.ws[2 * H + h * D + d];
  if (p.v_per_head) {
    __shared__ float red[D];
    red[d] = mv;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
      if (d < s) red[d] = fmaxf(red[d], red[d + s]);
      __syncthreads();
    }
    mv = red[0];
  }
  vmul[h * D + d] = mv > 0.f ? V_TARGET / mv : 0.f;
  p.v_descale[h * D + d] = mv / V_TARGET;
}

__device__ __forceinline__ float clamp448(float x) { return __builtin_amdgcn_fmed3f(x, -E4M3_MAX, E4M3_MAX); }

// grid (row chunks, heads, 3): thread = 16 channels of a row (32 B in, 16 B out); 8 threads per row, 32 rows per pass
template <typename T>
__global__ __launch_bounds__(256) void fp8_convert_kernel(const QParams p) {
  typedef __attribute__((ext_vector_type(8))) T T8;
  const int which = blockIdx.z, h = blockIdx.y, H = p.heads;
  const int t = threadIdx.x, cc = t & 7, rl = t >> 3;
  const int r0 = blockIdx.x * p.rows_per_block, r1 = min(r0 + p.rows_per_block, p.n_tokens);
  const float* qmul = p.ws + 2 * H + H * D;
  float mul[16];
  if (which == 2) {
    const float* vm = qmul + 2 * H + h * D + cc * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) mul[i] = vm[i];
  } else {
    const float s = qmul[which * H + h];
#pragma unroll
    for (int i = 0; i < 16; ++i) mul[i] = s;
  }
  const char* src = p.x[which] + (int64_t)h * p.x_sh[which] + cc * 32;
  char* dst = p.y[which] + (int64_t)h * p.y_sh[which] + cc * 16;
  for (int r = r0 + rl; r < r1; r += 32) {
    const T8 a = *(const T8*)(src + (int64_t)r * p.x_ss[which]);
    const T8 b = *(const T8*)(src + (int64_t)r * p.x_ss[which] + 16);
    float f[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) { f[i] = clamp448(to_f(a[i]) * mul[i]); f[8 + i] = clamp448(to_f(b[i]) * mul[8 + i]); }
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * w], f[4 * w + 1], 0, false);
      o[w] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[4 * w + 2], f[4 * w + 3], lo, true);
    }
    *(u32x4*)(dst + (int64_t)r * p.y_ss[which]) = o;
  }
}

}  // namespace

extern "C" int vorta_fp8_quant_ws_floats(int32_t heads, int32_t head_dim) {
  if (heads <= 0 || head_dim != D) return VORTA_EINVAL;
  return 2 * (2 * heads + heads * head_dim);
}

extern "C" int vorta_fp8_quantize_qkv(const vorta_fp8_quant_args* a, void* hip_stream) {
  if (!a || a->struct_size != sizeof(vorta_fp8_quant_args)) return VORTA_EINVAL;
  if (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16) return VORTA_EUNSUPPORTED;
  if (a->head_dim != D) return VORTA_EUNSUPPORTED;
  if (a->heads < 0 || a->n_tokens < 0) return VORTA_EINVAL;
  if (a->heads == 0 || a->n_tokens == 0) return VORTA_OK;
  if (!a->ws || !a->v_descale || !(a->qk_scale > 0.f)) return VORTA_EINVAL;
  const vorta_tensor* in[3] = {&a->q, &a->k, &a->v};
  const vorta_tensor* out[3] = {&a->q8, &a->k8, &a->v8};
  QParams p{};
  for (int i = 0; i < 3; ++i) {
    if (!in[i]->ptr || !out[i]->ptr) return VORTA_EINVAL;
    if (((uintptr_t)in[i]->ptr & 15) || (in[i]->stride_s % 8) || (in[i]->stride_h % 8) || in[i]->stride_s < D) return VORTA_EINVAL;
    if (((uintptr_t)out[i]->ptr & 15) || (out[i]->stride_s % 16) || (out[i]->stride_h % 16) || out[i]->stride_s < D) return VORTA_EINVAL;
    p.x[i] = (const char*)in[i]->ptr; p.x_sh[i] = in[i]->stride_h * 2; p.x_ss[i] = in[i]->stride_s * 2;
    p.y[i] = (char*)out[i]->ptr; p.y_sh[i] = out[i]->stride_h; p.y_ss[i] = out[i]->stride_s;
  }
  p.heads = a->heads; p.n_tokens = a->n_tokens;
  p.c0 = a->qk_scale * 1.4426950408889634f;
  p.ws = a->ws; p.v_descale = a->v_descale;
  p.v_per_head = a->flags & 1;
  hipStream_t st = (hipStream_t)hip_stream;
  const int H = a->heads;
  hipError_t e = hipMemsetAsync(a->ws, 0, sizeof(float) * (2 * H + H * D), st);
  if (e != hipSuccess) return vorta_set_hip_error(e);
  // enough workgroups to fill the chip several times over, few enough that the atomics stay cheap
  p.rows_per_block = 1024;
  const unsigned chunks = (unsigned)((a->n_tokens + p.rows_per_block - 1) / p.rows_per_block);
  const dim3 grid(chunks, (unsigned)H, 3);
  if (a->dtype == VORTA_BF16) hipLaunchKernelGGL((fp8_absmax_kernel<__bf16>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((fp8_absmax_kernel<_Float16>), grid, dim3(256), 0, st, p);
  e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  hipLaunchKernelGGL(fp8_scales_kernel, dim3((unsigned)H), dim3(128), 0, st, p);
  e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  if (a->dtype == VORTA_BF16) hipLaunchKernelGGL((fp8_convert_kernel<__bf16>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((fp8_convert_kernel<_Float16>), grid, dim3(256), 0, st, p);
  e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  return VORTA_OK;
}
