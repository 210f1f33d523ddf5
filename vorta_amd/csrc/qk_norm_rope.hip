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

// per-head normalisation (HunyuanVideo): a 16-lane quarter wave owns one token and every fourth head of it.  The token's
// cos / sin (2 x 512 B of fp32, the same for all its heads) are read ONCE into registers -- read per (token, head) row
// they were 4 x the bytes of the row itself through the vector cache -- and the head rows are loaded UNR at a time before
// the first is touched.  Same box, Hunyuan-129f q (1.46 GB read + write): 4.3 -> 5.0-5.25 TB/s (tools/bench_norm_rope.py,
// profiles/r03_norm_rope_quantizer_bandwidth.txt).
template <typename T>
__global__ __launch_bounds__(256) void qk_norm_rope_head_kernel(const NParams p) {
  typedef __attribute__((ext_vector_type(8))) T V8;
  constexpr int UNR = 6;  // 24 heads: every quarter wave's six rows in flight at once (3: 4.9, 6: 5.0-5.25 TB/s)
  const int sub = threadIdx.x & 15;
  const int qw = threadIdx.x >> 4;
  const int token = blockIdx.x * 4 + (qw >> 2);
  const int hq = qw & 3;
  if (token >= p.n_tokens) return;
  const bool rot = p.cs && token < p.rope_tokens;
  float cc[8], sn[8];
  if (rot) {
    const float* c = p.cs + (int64_t)token * 128 + sub * 8;
    const float* s = p.sn + (int64_t)token * 128 + sub * 8;
    const f32x4 c0 = *(const f32x4*)c, c1 = *(const f32x4*)(c + 4);
    const f32x4 s0 = *(const f32x4*)s, s1 = *(const f32x4*)(s + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { cc[i] = c0[i]; cc[4 + i] = c1[i]; sn[i] = s0[i]; sn[4 + i] = s1[i]; }
  }
  float wf[8];
  if (p.w) {
    const V8 wv = *(const V8*)((const char*)p.w + sub * 16);
#pragma unroll
    for (int i = 0; i < 8; ++i) wf[i] = (float)wv[i];
  }
  char* base = p.x + (int64_t)(p.token_offset + token) * p.x_ss + sub * 16;
  for (int h0 = hq; h0 < p.heads; h0 += 4 * UNR) {
    V8 xv[UNR];
#pragma unroll
    for (int j = 0; j < UNR; ++j)
      if (h0 + 4 * j < p.heads) xv[j] = *(const V8*)(base + (int64_t)(h0 + 4 * j) * p.x_sh);
#pragma unroll
    for (int j = 0; j < UNR; ++j) {
      if (h0 + 4 * j >= p.heads) break;
      float y[8], ss = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i) { y[i] = (float)xv[j][i]; ss += y[i] * y[i]; }
      ss = q16_sum(ss);
      const float r = rsqrtf(ss * (1.f / 128.f) + p.eps);
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = p.w ? y[i] * r * wf[i] : y[i] * r;
      V8 ov;
      if (rot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ov[2 * i] = (T)(y[2 * i] * cc[2 * i] - y[2 * i + 1] * sn[2 * i]);
          ov[2 * i + 1] = (T)(y[2 * i + 1] * cc[2 * i + 1] + y[2 * i] * sn[2 * i + 1]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = (T)y[i];
      }
      *(V8*)(base + (int64_t)(h0 + 4 * j) * p.x_sh) = ov;
    }
  }
}

// normalisation across all heads of a token (Wan): one wave per token, values kept in registers
template <typename T, int MAXIT>
__global__ __launch_bounds__(256) void qk_norm_rope_token_kernel(const NParams p) {
  typedef __attribute__((ext_vector_type(8))) T V8;
  const int lane = threadIdx.x & 63;
  const int token = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (token >= p.n_tokens) return;
  const int chunks = p.heads * 16;  // 16-byte chunks per token
  // chunk c = it * 64 + lane of the token is channels 8 (lane & 15) .. +8 of head c >> 4: a lane meets the same channels in
  // every one of its chunks, so its cos / sin are read once
  const bool rot = p.cs && token < p.rope_tokens;
  float cc[8], sn[8];
  if (rot) {
    const float* c = p.cs + (int64_t)token * 128 + (lane & 15) * 8;
    const float* s = p.sn + (int64_t)token * 128 + (lane & 15) * 8;
    const f32x4 c0 = *(const f32x4*)c, c1 = *(const f32x4*)(c + 4);
    const f32x4 s0 = *(const f32x4*)s, s1 = *(const f32x4*)(s + 4);
#pragma unroll
    for (int i = 0; i < 4; ++i) { cc[i] = c0[i]; cc[4 + i] = c1[i]; sn[i] = s0[i]; sn[4 + i] = s1[i]; }
  }
  V8 xv[MAXIT];  // the token stays in registers in its 16-bit form (half the registers of fp32 copies: more waves per SIMD)
  float ss = 0.f;
#pragma unroll
  for (int it = 0; it < MAXIT; ++it) {
    const int c = it * 64 + lane;
    if (c < chunks) {
      const int head = c >> 4, sub = c & 15;
      xv[it] = *(const V8*)(p.x + (int64_t)head * p.x_sh + (int64_t)(p.token_offset + token) * p.x_ss + sub * 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float f = (float)xv[it][i]; ss += f * f; }
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
      float y[8];
      if (p.w) {
        const V8 wv = *(const V8*)((const char*)p.w + (int64_t)c * 16);
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = (float)xv[it][i] * r * (float)wv[i];
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) y[i] = (float)xv[it][i] * r;
      }
      V8 ov;
      if (rot) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          ov[2 * i] = (T)(y[2 * i] * cc[2 * i] - y[2 * i + 1] * sn[2 * i]);
          ov[2 * i + 1] = (T)(y[2 * i + 1] * cc[2 * i + 1] + y[2 * i] * sn[2 * i + 1]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 8; ++i) ov[i] = (T)y[i];
      }
      *(V8*)(p.x + (int64_t)head * p.x_sh + (int64_t)(p.token_offset + token) * p.x_ss + sub * 16) = ov;
    }
  }
}

template <typename T>
int launch(const NParams& p, hipStream_t st) {
  if (!p.across_heads) {
    hipLaunchKernelGGL(qk_norm_rope_head_kernel<T>, dim3((unsigned)((p.n_tokens + 3) / 4)), dim3(256), 0, st, p);
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
