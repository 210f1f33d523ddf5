// Fused q/k RMSNorm + rotary embedding for gfx950: include/vorta_hip.h vorta_qk_norm_rope.
//
// The step right before the attention boundary.  The reference runs it as separate torch
// ops -- RMSNorm (hunyuan.py:62-73 per head, wan.py:85-89 across heads) then RoPE (hunyuan.py:75-104 via
// diffusers' apply_rotary_emb, wan.py:34-37,96-100 as a float64 complex product) -- i.e. roughly ten full
// passes over Q and K.  Here each of Q and K is read once and written once, in place: HBM-bound
// (2 x rows x D x 2 B per tensor), one 16-lane quarter wave per 256-byte row.
//   y = x * rsqrt(mean(x^2) + eps) * w          (fp32; mean over D, or over all H*D channels of the token)
//   out[2i]   = y[2i] * cos[2i]   - y[2i+1] * sin[2i]
//   out[2i+1] = y[2i+1]*cos[2i+1] + y[2i]   * sin[2i+1]
// (interleaved pairs: diffusers use_real_unbind_dim=-1, identical to wan's view_as_complex product).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

namespace {

struct NParams {
  char* x; int64_t x_sh, x_ss;   // bytes
  const void* w;                 // [D] (per head) or [H*D] (across heads), same dtype as x; may be NULL
  const float* cs; const float* sn;  // [n_tokens][D] fp32 or NULL (no rotation)
  int heads, n_tokens, token_offset;
  int rope_tokens;               // tokens >= rope_tokens (relative) are normalised but not rotated (text tail)
  float eps;
  int across_heads;
};

__device__ __forceinline__ float q16_sum(float v) {
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

template <typename T>
__device__ __forceinline__ void rope8(float (&y)[8], const NParams& p, int token, int d0) {
  const float* c = p.cs + (int64_t)token * 128 + d0;
  const float* s = p.sn + (int64_t)token * 128 + d0;
  const f32x4 c0 = *(const f32x4*)c, c1 = *(const f32x4*)(c + 4);
  const f32x4 s0 = *(const f32x4*)s, s1 = *(const f32x4*)(s + 4);
  const float cc[8] = {c0[0], c0[1], c0[2], c0[3], c1[0], c1[1], c1[2], c1[3]};
  const float ss[8] = {s0[0], s0[1], s0[2], s0[3], s1[0], s1[1], s1[2], s1[3]};
  float o[8];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    o[2 * i] = y[2 * i] * cc[2 * i] - y[2 * i + 1] * ss[2 * i];
    o[2 * i + 1] = y[2 * i + 1] * cc[2 * i + 1] + y[2 * i] * ss[2 * i + 1];
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) y[i] = o[i];
}

// per-head normalisation (HunyuanVideo): one quarter wave per (token, head) row, heads fastest
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_head_kernel(const NParams p) {
  typedef __attribute__((ext_vector_type(8))) T V8;
  const int64_t row = (int64_t)blockIdx.x * 16 + (threadIdx.x >> 4);
  const int sub = threadIdx.x & 15;
  if (row >= (int64_t)p.n_tokens * p.heads) return;
  const int token = (int)(row / p.heads), head = (int)(row - (int64_t)token * p.heads);
  char* xp = p.x + (int64_t)head * p.x_sh + (int64_t)(p.token_offset + token) * p.x_ss + sub * 16;
  const V8 xv = *(const V8*)xp;
  float y[8], ss = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { y[i] = (float)xv[i]; ss += y[i] * y[i]; }
  ss = q16_sum(ss);
  const float r = rsqrtf(ss * (1.f / 128.f) + p.eps);
  if (p.w) {
    const V8 wv = *(const V8*)((const char*)p.w + sub * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] = y[i] * r * (float)wv[i];
  } else {
#pragma unroll
    for (int i = 0; i < 8; ++i) y[i] *= r;
  }
  if (p.cs && token < p.rope_tokens) rope8<T>(y, p, token, sub * 8);
  V8 ov;
#pragma unroll
  for (int i = 0; i < 8; ++i) ov[i] = (T)y[i];
  *(V8*)xp = ov;
}

// normalisation across all heads of a token (Wan): one wave per token, values kept in registers
template <typename T, int MAXIT>
__global__ __launch_bounds__(256) void qk_norm_rope_token_kernel(const NParams p) {
  typedef __attribute__((ext_vector_type(8))) T V8;
  const int lane = threadIdx.x & 63;
  const int token = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (token >= p.n_tokens) return;
  const int chunks = p.heads * 16;  // 16-byte chunks per token
  float y[MAXIT][8];
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int c = it * 64 + lane;
    if (c < chunks) {
      const int head = c >> 4, sub = c & 15;
      const V8 xv = *(const V8*)(p.x + (int64_t)head * p.x_sh + (int64_t)(p.token_offset + token) * p.x_ss + sub * 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) { y[it][i] = (float)xv[i]; ss += y[it][i] * y[it][i]; }
    }
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) ss += __shfl_xor(ss, s);
  const float r = rsqrtf(ss / (float)(p.heads * 128) + p.eps);
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int c = it * 64 + lane;
    if (c < chunks) {
      const int head = c >> 4, sub = c & 15;
      if (p.w) {
        const V8 wv = *(const V8*)((const char*)p.w + (int64_t)c * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[it][i] = y[it][i] * r * (float)wv[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) y[it][i] *= r;
      }
      if (p.cs && token < p.rope_tokens) rope8<T>(y[it], p, token, sub * 8);
      V8 ov;
#pragma unroll
      for (int i = 0; i < 8; ++i) ov[i] = (T)y[it][i];
      *(V8*)(p.x + (int64_t)head * p.x_sh + (int64_t)(p.token_offset + token) * p.x_ss + sub * 16) = ov;
    }
  }
}

template <typename T>
int launch(const NParams& p, hipStream_t st) {
  if (!p.across_heads) {
    const int64_t rows = (int64_t)p.n_tokens * p.heads;
    hipLaunchKernelGGL(qk_norm_rope_head_kernel<T>, dim3((unsigned)((rows + 15) / 16)), dim3(256), 0, st, p);
  } else {
    const unsigned blocks = (unsigned)((p.n_tokens + 3) / 4);
    const int its = (p.heads * 16 + 63) / 64;
    if (its <= 3) hipLaunchKernelGGL((qk_norm_rope_token_kernel<T, 3>), dim3(blocks), dim3(256), 0, st, p);
    else if (its <= 6) hipLaunchKernelGGL((qk_norm_rope_token_kernel<T, 6>), dim3(blocks), dim3(256), 0, st, p);
    else if (its <= 10) hipLaunchKernelGGL((qk_norm_rope_token_kernel<T, 10>), dim3(blocks), dim3(256), 0, st, p);
    else return VORTA_EUNSUPPORTED;
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}

}  // namespace

extern "C" int vorta_qk_norm_rope(const vorta_norm_rope_args* a, void* hip_stream) {
  if (!a || a->struct_size != sizeof(vorta_norm_rope_args)) return VORTA_EINVAL;
  if (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16) return VORTA_EUNSUPPORTED;
  if (a->head_dim != 128) return VORTA_EUNSUPPORTED;
  if (a->heads <= 0 || a->n_tokens < 0 || a->token_offset < 0 || a->rope_tokens < 0) return VORTA_EINVAL;
  if (a->n_tokens == 0) return VORTA_OK;
  if (!a->x.ptr || ((uintptr_t)a->x.ptr & 15) || (a->x.stride_s % 8) || (a->x.stride_h % 8)) return VORTA_EINVAL;
  if ((a->cos == nullptr) != (a->sin == nullptr)) return VORTA_EINVAL;
  if (a->cos && (((uintptr_t)a->cos & 15) || ((uintptr_t)a->sin & 15))) return VORTA_EINVAL;
  if (a->weight && ((uintptr_t)a->weight & 15)) return VORTA_EINVAL;
  if ((int64_t)a->n_tokens * a->heads > 0x7fffffff0ll) return VORTA_EINVAL;
  NParams p{(char*)a->x.ptr, a->x.stride_h * 2, a->x.stride_s * 2, a->weight, a->cos, a->sin, a->heads, a->n_tokens,
            a->token_offset, a->rope_tokens, a->eps, a->across_heads};
  hipStream_t st = (hipStream_t)hip_stream;
  return a->dtype == VORTA_BF16 ? launch<__bf16>(p, st) : launch<_Float16>(p, st);
}
