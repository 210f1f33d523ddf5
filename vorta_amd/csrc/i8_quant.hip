// vorta_i8_quantize_k (include/vorta_hip.h, ABI 6): keys -> int8 rows + one float32 bias per row, for the int8-score
// attention kernel (attn_fwd_i8.hip).  No reference counterpart (the reference computes in the dtype of q, k, v:
// wan.py:243-294).
//
// Per head h, from the SAMPLE -- tokens i * stride, i < cand (~1024 of them; stride odd, so it does not lock onto a
// segment length), video tokens then tail tokens, the same tokens in the (H,S,D) view and in the segmented Ulysses
// receive layout -- in float32 with a FIXED summation order (the oracle restates it bit for bit):
//     ck[d] = mean k[d],  cq[d] = mean q[d],  s[d] = clamp((var k[d] / var q[d])^(1/4), 1/8, 8)
// then two passes over the rows (kt = (k - ck) * (1 / s)):  amax = max |kt| over the head (exact: integer atomic max of the
// float bits), and  k8 = rint(kt * (127 / amax)),  k_bias = (cq . (k - ck)) * (127 / amax)  (the dot product summed 8
// channels per lane in order, the 16 lanes of a row pairwise at distance 1, 2, 4, 8).
// Bound: HBM (2 + 2 B in, 1 B + 4/128 B out per element).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

namespace {

constexpr int D = 128;
constexpr int SAMPLES = 1024;

struct IParams {
  const char* q; const char* k; int64_t q_sh, q_ss, k_sh, k_ss;  // bytes
  char* k8; int64_t k8_sh, k8_ss;
  float* k_bias; int64_t k_bias_sh;
  float* q_prep; float* k_head_scale; float* ws;
  int heads, n_tokens, rows_per_block;
  int seg_len, chunks_per_seg, tail_first, tail_len;
  int slot_first, slot_count;
  int64_t video_tokens, total_tokens;
  int stride, cand;
  int no_smooth, no_center;
};

// grid (heads of the slot range), 1024 threads = 64 row lanes x 16 channel groups of 8.  Row lane rl sums samples rl,
// rl + 64, ... in order; the 64 partials are added pairwise (distance 32, 16, ... 1).  No fused multiply-adds: every
// product and every sum is rounded to float32 on its own (#pragma clang fp contract(off)), so numpy restates it exactly.
template <typename T>
__global__ __launch_bounds__(1024) void i8_stats_kernel(const IParams p) {
#pragma clang fp contract(off)
  typedef __attribute__((ext_vector_type(8))) T T8;
  const int h = p.slot_first + blockIdx.x;
  const int t = threadIdx.x, cg = t & 15, rl = t >> 4;
  auto row_of = [&](int64_t tok) -> int64_t {  // physical row of token `tok` of head h
    if (p.seg_len <= 0) return tok;
    if (tok < p.video_tokens) return ((tok / p.seg_len) * p.heads + h) * p.seg_len + tok % p.seg_len;
    return (int64_t)p.tail_first + (int64_t)h * p.seg_len + (tok - p.video_tokens);
  };
  const char* kb = p.k + (p.seg_len > 0 ? 0 : (int64_t)h * p.k_sh) + cg * 16;
  const char* qb = p.q + (p.seg_len > 0 ? 0 : (int64_t)h * p.q_sh) + cg * 16;
  float sk[8], sk2[8], sq[8], sq2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { sk[i] = 0.f; sk2[i] = 0.f; sq[i] = 0.f; sq2[i] = 0.f; }
  for (int i = rl; i < p.cand; i += 64) {
    const int64_t r = row_of((int64_t)i * p.stride);
    const T8 kv = *(const T8*)(kb + r * p.k_ss);
    const T8 qv = *(const T8*)(qb + r * p.q_ss);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float kf = (float)kv[e], qf = (float)qv[e];
      const float k2 = kf * kf, q2 = qf * qf;
      sk[e] = sk[e] + kf;
      sk2[e] = sk2[e] + k2;
      sq[e] = sq[e] + qf;
      sq2[e] = sq2[e] + q2;
    }
  }
  __shared__ float red[4][64][D];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    red[0][rl][cg * 8 + e] = sk[e];
    red[1][rl][cg * 8 + e] = sk2[e];
    red[2][rl][cg * 8 + e] = sq[e];
    red[3][rl][cg * 8 + e] = sq2[e];
  }
  __syncthreads();
  for (int off = 32; off > 0; off >>= 1) {
    if (rl < off) {
#pragma unroll
      for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int e = 0; e < 8; ++e) red[w][rl][cg * 8 + e] = red[w][rl][cg * 8 + e] + red[w][rl + off][cg * 8 + e];
    }
    __syncthreads();
  }
  if (t < D) {
    const float n = (float)p.cand;
    const float mean_k = red[0][0][t] / n;
    const float mk2 = red[1][0][t] / n;
    const float var_k = mk2 - mean_k * mean_k;
    const float mean_q = red[2][0][t] / n;
    const float mq2 = red[3][0][t] / n;
    const float var_q = mq2 - mean_q * mean_q;
    float s = 1.f;
    if (!p.no_smooth && var_k > 0.f && var_q > 0.f) {
      s = sqrtf(sqrtf(var_k / var_q));
      s = fminf(fmaxf(s, 0.125f), 8.f);
    }
    p.ws[h * D + t] = p.no_center ? 0.f : mean_k;
    p.ws[(p.heads + h) * D + t] = 1.f / s;
    p.q_prep[(2 * h) * D + t] = p.no_center ? 0.f : mean_q;
    p.q_prep[(2 * h + 1) * D + t] = s;
  }
  if (t == 0) ((int*)p.ws)[2 * p.heads * D + h] = 0;  // the head's abs-max slot (raised by the next launch)
}

// the rows a block works on: [r0, r1) of head `head` (physical head index `hphys` of the (H,S,D) views; 0 in the row array)
__device__ __forceinline__ void block_rows(const IParams& p, int& r0, int& r1, int& head, int& hphys) {
  if (p.seg_len > 0) {
    const int per = p.slot_count * p.chunks_per_seg;
    const int sg = blockIdx.x / per, rem = blockIdx.x - sg * per;
    const int si = rem / p.chunks_per_seg, c = rem - si * p.chunks_per_seg;
    const int seg = sg * p.heads + p.slot_first + si;
    const int s0 = seg * p.seg_len;
    r0 = s0 + c * p.rows_per_block;
    const int seg_rows = s0 >= p.tail_first ? p.tail_len : p.seg_len;
    r1 = min(min(r0 + p.rows_per_block, s0 + seg_rows), p.n_tokens);
    head = seg % p.heads;
    hphys = 0;
  } else {
    head = hphys = blockIdx.y;
    r0 = blockIdx.x * p.rows_per_block;
    r1 = min(r0 + p.rows_per_block, p.n_tokens);
  }
}

// 256 threads: 16 lanes (8 channels each) per row, 16 rows per pass, UNR passes in flight.  PASS 0: the head's abs-max of
// kt (one integer atomic max of the float bits per workgroup); PASS 1: int8 rows and the row biases.
template <typename T, int PASS>
__global__ __launch_bounds__(256) void i8_rows_kernel(const IParams p) {
#pragma clang fp contract(off)
  typedef __attribute__((ext_vector_type(8))) T T8;
  int r0, r1, h, hphys;
  block_rows(p, r0, r1, h, hphys);
  if (r0 >= r1) return;
  const int t = threadIdx.x, cg = t & 15, rl = t >> 4;
  float c[8], is[8], cq[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    c[e] = p.ws[h * D + cg * 8 + e];
    is[e] = p.ws[(p.heads + h) * D + cg * 8 + e];
    cq[e] = p.q_prep[(2 * h) * D + cg * 8 + e];
  }
  const char* src = p.k + (int64_t)hphys * p.k_sh + cg * 16;
  char* dst = p.k8 + (int64_t)hphys * p.k8_sh + cg * 8;
  float* bias = p.k_bias + (int64_t)hphys * p.k_bias_sh;
  const float am_head = PASS == 1 ? __int_as_float(((const int*)p.ws)[2 * p.heads * D + h]) : 0.f;
  const float inv = am_head > 0.f ? 127.f / am_head : 0.f;
  float am = 0.f;
  constexpr int UNR = 4;
  for (int rb = r0 + rl; rb < r1; rb += 16 * UNR) {
    T8 a[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = rb + 16 * u;
      if (r < r1) a[u] = *(const T8*)(src + (int64_t)r * p.k_ss);
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = rb + 16 * u;  // rows past the end: the lanes still take part in the exchanges below with zeros
      float f[8];
      float dot = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float d = r < r1 ? (float)a[u][e] - c[e] : 0.f;
        f[e] = d * is[e];
        if (PASS == 0) am = fmaxf(am, fabsf(f[e]));
        else { const float pr = cq[e] * d; dot = dot + pr; }
      }
      if (PASS == 1) {
#pragma unroll
        for (int m = 1; m < 16; m <<= 1) dot = dot + __shfl_xor(dot, m, 16);
        uint32_t w[2] = {0u, 0u};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          int v = (int)__builtin_rintf(f[e] * inv);
          v = max(-127, min(127, v));
          w[e >> 2] |= ((uint32_t)v & 0xffu) << (8 * (e & 3));
        }
        if (r < r1) {
          *(u32x2*)(dst + (int64_t)r * p.k8_ss) = u32x2{w[0], w[1]};
          if (cg == 0) bias[r] = dot * inv;
        }
      }
    }
  }
  if (PASS == 0) {
    __shared__ float red[256];
    red[t] = am;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if (t < s) red[t] = fmaxf(red[t], red[t + s]);
      __syncthreads();
    }
    if (t == 0) atomicMax((int*)p.ws + 2 * p.heads * D + h, __float_as_int(red[0]));  // non-negative floats order as ints
  }
}

// grid (heads of the slot range), 64 threads: sk = amax / 127 (1 for an all-zero head)
__global__ void i8_head_scale_kernel(const IParams p) {
#pragma clang fp contract(off)
  const int h = p.slot_first + blockIdx.x;
  if (threadIdx.x == 0) {
    const float am = __int_as_float(((const int*)p.ws)[2 * p.heads * D + h]);
    p.k_head_scale[h] = am > 0.f ? am * (1.f / 127.f) : 1.f;
  }
}

// grid (heads), 1024 threads = 128 row lanes x 8 column lanes of 16 bytes (one 128-byte row per 8 threads: D == 128): the sum of
// squares of ~1024 evenly spaced int8 rows of the head (row lane rl takes samples rl, rl + 128, ...; the 16 bytes of a lane in
// order; lanes pairwise at distance 512, 256, ... 1): exact in int64, so the flag does not depend on the order at all
static_assert(D == 128, "i8_tail_kernel reads a row as 8 lanes x 16 bytes and is launched with 1024 threads");
__global__ __launch_bounds__(1024) void i8_tail_kernel(const char* k8, int64_t sh, int64_t ss, const int* row_map, int n_tokens,
                                                        int stride, int cand, float min_rms, int* flags) {
  const int h = blockIdx.x, t = threadIdx.x, cg = t & 7, rl = t >> 3;  // 8 lanes x 16 bytes per row, 128 row lanes
  long long acc = 0;
  for (int i = rl; i < cand; i += 128) {
    const int64_t tok = min((int64_t)i * stride, (int64_t)n_tokens - 1);
    const int64_t r = row_map ? (int64_t)row_map[tok] : tok;  // (the zero-copy Ulysses layout: token -> row of the head's view)
    const int4 w = *(const int4*)(k8 + (int64_t)h * sh + r * ss + cg * 16);
    const int v[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int x = (int)(signed char)((v[e] >> (8 * b)) & 0xff);
        acc += x * x;
      }
  }
  __shared__ long long red[1024];
  red[t] = acc;
  __syncthreads();
  for (int off = 512; off > 0; off >>= 1) {
    if (t < off) red[t] += red[t + off];
    __syncthreads();
  }
  if (t == 0) {
    const double mean2 = (double)red[0] / ((double)cand * D);
    flags[h] = mean2 < (double)min_rms * (double)min_rms ? 1 : 0;
  }
}

// one thread: order-preserving split of a head list by a per-head flag
__global__ void split_heads_kernel(const int* head_list, const int* n_dev, int n, const int* flags, int* list0, int* list1,
                                   int* counts) {
  if (threadIdx.x || blockIdx.x) return;
  const int m = n_dev ? min(max(*n_dev, 0), n) : n;
  int c0 = 0, c1 = 0;
  for (int y = 0; y < m; ++y) {
    const int h = head_list ? head_list[y] : y;
    if (flags[h]) list1[c1++] = h;
    else list0[c0++] = h;
  }
  counts[0] = c0;
  counts[1] = c1;
}

}  // namespace

extern "C" int vorta_i8_tail_flags(const vorta_tensor* k8, int32_t heads, int32_t n_tokens, const int32_t* row_map, float min_rms,
                                   int32_t* flags, void* hip_stream) {
  if (!k8 || heads < 0 || n_tokens < 0 || !(min_rms >= 0.f)) return VORTA_EINVAL;
  if (heads == 0) return VORTA_OK;
  if (!k8->ptr || !flags || n_tokens == 0) return VORTA_EINVAL;
  if (((uintptr_t)k8->ptr & 15) || (k8->stride_s % 16) || (k8->stride_h % 16) || k8->stride_s < D) return VORTA_EINVAL;
  // token = row (no row map): the heads' row ranges must not overlap -- a head stride shorter than n_tokens rows is a wrong
  // view (with a row map the rows are the caller's: the struct carries no extent, the Python host checks the map's length)
  if (!row_map && heads > 1 && k8->stride_h != 0 && llabs((long long)k8->stride_h) < (long long)n_tokens * k8->stride_s) return VORTA_EINVAL;
  int stride = n_tokens / SAMPLES;
  if (stride < 1) stride = 1;
  stride |= 1;
  const int cand = (n_tokens + stride - 1) / stride;
  hipLaunchKernelGGL(i8_tail_kernel, dim3(heads), dim3(1024), 0, (hipStream_t)hip_stream, (const char*)k8->ptr, k8->stride_h,
                     k8->stride_s, row_map, n_tokens, stride, cand, min_rms, flags);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}

extern "C" int vorta_split_heads(const int32_t* head_list, const int32_t* n_heads_dev, int32_t n_heads, const int32_t* flags,
                                 int32_t* list0, int32_t* list1, int32_t* counts, void* hip_stream) {
  if (n_heads < 0) return VORTA_EINVAL;
  if (!flags || !list0 || !list1 || !counts) return VORTA_EINVAL;
  hipLaunchKernelGGL(split_heads_kernel, dim3(1), dim3(64), 0, (hipStream_t)hip_stream, head_list, n_heads_dev, n_heads, flags,
                     list0, list1, counts);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}

extern "C" int vorta_i8_quantize_k(const vorta_i8_quant_args* a, void* hip_stream) {
  if (!a || a->struct_size != sizeof(vorta_i8_quant_args)) return VORTA_EINVAL;
  if (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16) return VORTA_EUNSUPPORTED;
  if (a->head_dim != D) return VORTA_EUNSUPPORTED;
  if (a->heads < 0 || a->n_tokens < 0 || a->seg_len < 0 || a->tail_first < 0 || a->tail_len < 0) return VORTA_EINVAL;
  if (a->seg_len > 0 && (a->tail_first > 0 || a->tail_len > 0) && (a->tail_first % a->seg_len || a->tail_len > a->seg_len))
    return VORTA_EINVAL;
  if (a->slot_first < 0 || a->slot_count < 0 || (int64_t)a->slot_first + a->slot_count > a->heads) return VORTA_EINVAL;
  if ((a->slot_first || a->slot_count) && a->seg_len <= 0) return VORTA_EINVAL;
  if (a->heads == 0 || a->n_tokens == 0) return VORTA_OK;
  if (!a->q.ptr || !a->k.ptr || !a->k8.ptr || !a->k_bias || !a->q_prep || !a->k_head_scale || !a->ws) return VORTA_EINVAL;
  const vorta_tensor* in[2] = {&a->q, &a->k};
  for (int i = 0; i < 2; ++i)
    if (((uintptr_t)in[i]->ptr & 15) || (in[i]->stride_s % 8) || (in[i]->stride_h % 8) || in[i]->stride_s < D) return VORTA_EINVAL;
  if (((uintptr_t)a->k8.ptr & 15) || (a->k8.stride_s % 16) || (a->k8.stride_h % 16) || a->k8.stride_s < D) return VORTA_EINVAL;
  IParams p{};
  p.q = (const char*)a->q.ptr; p.k = (const char*)a->k.ptr;
  p.q_sh = a->q.stride_h * 2; p.q_ss = a->q.stride_s * 2; p.k_sh = a->k.stride_h * 2; p.k_ss = a->k.stride_s * 2;
  p.k8 = (char*)a->k8.ptr; p.k8_sh = a->k8.stride_h; p.k8_ss = a->k8.stride_s;
  p.k_bias = a->k_bias; p.k_bias_sh = a->seg_len > 0 ? 0 : a->k_bias_stride_h;
  p.q_prep = a->q_prep; p.k_head_scale = a->k_head_scale; p.ws = a->ws;
  p.heads = a->heads; p.n_tokens = a->n_tokens;
  p.seg_len = a->seg_len;
  const bool has_tail = a->seg_len > 0 && (a->tail_first > 0 || a->tail_len > 0);
  p.tail_first = has_tail ? a->tail_first : 0x7fffffff;
  p.tail_len = has_tail ? a->tail_len : 0;
  p.slot_first = a->slot_count > 0 ? a->slot_first : 0;
  p.slot_count = a->slot_count > 0 ? a->slot_count : a->heads;
  p.no_smooth = a->flags & 1; p.no_center = (a->flags >> 1) & 1;
  const int H = a->heads, hn = p.slot_count;
  int64_t head_tokens = a->n_tokens;
  p.video_tokens = head_tokens;
  if (p.seg_len > 0) {
    const int64_t data_rows = has_tail ? (int64_t)a->tail_first : (int64_t)a->n_tokens;
    p.video_tokens = (data_rows / p.seg_len / H) * p.seg_len;
    head_tokens = p.video_tokens + p.tail_len;
  }
  p.total_tokens = head_tokens;
  if (head_tokens <= 0) return VORTA_EINVAL;
  p.stride = (int)(head_tokens / SAMPLES);
  if (p.stride < 1) p.stride = 1;
  p.stride |= 1;
  p.cand = (int)((head_tokens + p.stride - 1) / p.stride);
  p.rows_per_block = 1024;
  dim3 grid;
  if (p.seg_len > 0) {
    p.chunks_per_seg = (p.seg_len + p.rows_per_block - 1) / p.rows_per_block;
    const int64_t n_seg = ((int64_t)a->n_tokens + p.seg_len - 1) / p.seg_len;
    const int64_t n_sg = (n_seg + H - 1) / H;
    if (n_sg * hn * p.chunks_per_seg > 0x7fffffffll) return VORTA_EINVAL;
    grid = dim3((unsigned)(n_sg * hn * p.chunks_per_seg), 1, 1);
  } else {
    p.chunks_per_seg = 1;
    grid = dim3((unsigned)((a->n_tokens + p.rows_per_block - 1) / p.rows_per_block), (unsigned)H, 1);
  }
  hipStream_t st = (hipStream_t)hip_stream;
  const bool bf = a->dtype == VORTA_BF16;
  if (bf) hipLaunchKernelGGL((i8_stats_kernel<__bf16>), dim3(hn), dim3(1024), 0, st, p);
  else hipLaunchKernelGGL((i8_stats_kernel<_Float16>), dim3(hn), dim3(1024), 0, st, p);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  if (bf) hipLaunchKernelGGL((i8_rows_kernel<__bf16, 0>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((i8_rows_kernel<_Float16, 0>), grid, dim3(256), 0, st, p);
  e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  hipLaunchKernelGGL(i8_head_scale_kernel, dim3(hn), dim3(64), 0, st, p);
  if (bf) hipLaunchKernelGGL((i8_rows_kernel<__bf16, 1>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((i8_rows_kernel<_Float16, 1>), grid, dim3(256), 0, st, p);
  e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  return VORTA_OK;
}
