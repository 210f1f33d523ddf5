// Router gate + top-1 dispatch for gfx950: include/vorta_hip.h vorta_router_route.
//
// Launch-latency work (3H dot products of length E): kernel 1 computes the logits with one wave per
// output row, kernel 2 (one wave) does the softmax over the 3 experts, the top-1 / tau rule and writes the
// per-expert head lists + counts to DEVICE memory, so the expert kernels can be enqueued without the
// torch.nonzero host sync of the reference (hunyuan.py:633).
// Rounding follows the reference's bf16 module (router.py:41-43): silu output, logits and scores are
// rounded to the I/O dtype; the top-1 is taken on the rounded scores of batch item 0.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

namespace {

struct RParams {
  const void* temb; const void* w; const void* b;
  int batch, E, H, NE;
  float tau;
  void* scores; int32_t* expert; int32_t* lists; int32_t* counts;
  float* logits;
};

template <typename T> __device__ __forceinline__ float rnd(float x) { return (float)(T)x; }

template <typename T>
__global__ __launch_bounds__(256) void router_logits_kernel(const RParams p) {
  typedef __attribute__((ext_vector_type(8))) T V8;
  const int lane = threadIdx.x & 63;
  const int item = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n_out = p.H * p.NE;
  if (item >= p.batch * n_out) return;
  const int b = item / n_out, o = item - b * n_out;
  const int layer = blockIdx.y;  // vorta_route_plan: one grid row per layer, same temb for all of them
  const T* x = (const T*)p.temb + (int64_t)b * p.E;
  const T* w = (const T*)p.w + ((int64_t)layer * n_out + o) * p.E;
  float acc = 0.f;
  const int e8 = p.E & ~7;
  for (int i = lane * 8; i < e8; i += 512) {
    const V8 xv = *(const V8*)(x + i);
    const V8 wv = *(const V8*)(w + i);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float t = (float)xv[j];
      acc += rnd<T>(t / (1.f + __expf(-t))) * (float)wv[j];
    }
  }
  for (int i = e8 + lane; i < p.E; i += 64) {
    const float t = (float)x[i];
    acc += rnd<T>(t / (1.f + __expf(-t))) * (float)w[i];
  }
#pragma unroll
  for (int s = 32; s > 0; s >>= 1) acc += __shfl_xor(acc, s);
  if (lane == 0)
    p.logits[(int64_t)layer * p.batch * n_out + item] = rnd<T>(acc + (float)((const T*)p.b)[(int64_t)layer * n_out + o]);
}

template <typename T>
__global__ __launch_bounds__(64) void router_route_kernel(const RParams p) {
  const int lane = threadIdx.x;
  const int layer = blockIdx.x;
  __shared__ int s_expert[1024];
  int32_t* const expert = p.expert + (int64_t)layer * p.H;
  int32_t* const lists = p.lists + (int64_t)layer * p.NE * p.H;
  for (int it = lane; it < p.batch * p.H; it += 64) {
    const int b = it / p.H, h = it - b * p.H;
    const int64_t git = (int64_t)layer * p.batch * p.H + it;
    const float* lg = p.logits + git * p.NE;
    float mx = lg[0];
    for (int e = 1; e < p.NE; ++e) mx = fmaxf(mx, lg[e]);
    float den = 0.f;
    for (int e = 0; e < p.NE; ++e) den += __expf(lg[e] - mx);
    int best = 0;
    float best_s = -1.f;
    for (int e = 0; e < p.NE; ++e) {
      const float sc = rnd<T>(__expf(lg[e] - mx) / den);
      if (p.scores) ((T*)p.scores)[git * p.NE + e] = (T)sc;
      if (sc > best_s) { best_s = sc; best = e; }  // first maximum wins (torch.topk on ties)
    }
    if (b == 0) {
      // hunyuan.py:623 `top1_score < tau_sparse`: torch compares a tensor with a Python scalar in the tensor's
      // dtype, i.e. tau is rounded to the score dtype first
      if (best_s < rnd<T>(p.tau)) best = 0;
      expert[h] = best;
      s_expert[h] = best;
    }
  }
  __syncthreads();
  if (lane < p.NE) {
    int n = 0;
    for (int h = 0; h < p.H; ++h)
      if (s_expert[h] == lane) lists[lane * p.H + n++] = h;
    p.counts[layer * p.NE + lane] = n;
  }
}

template <typename T>
__global__ __launch_bounds__(64) void route_scores_kernel(const RParams p) {
  const int lane = threadIdx.x;
  __shared__ int s_expert[1024];
  for (int h = lane; h < p.H; h += 64) {
    const T* sc = (const T*)p.scores + (int64_t)h * p.NE;  // batch item 0 (hunyuan.py:622)
    int best = 0;
    float best_s = (float)sc[0];
    for (int e = 1; e < p.NE; ++e) {
      const float v = (float)sc[e];
      if (v > best_s) { best_s = v; best = e; }
    }
    if (best_s < rnd<T>(p.tau)) best = 0;
    p.expert[h] = best;
    s_expert[h] = best;
  }
  __syncthreads();
  if (lane < p.NE) {
    int n = 0;
    for (int h = 0; h < p.H; ++h)
      if (s_expert[h] == lane) p.lists[lane * p.H + n++] = h;
    p.counts[lane] = n;
  }
}

}  // namespace

extern "C" int vorta_route_scores(const vorta_router_args* a, void* hip_stream) {
  if (!a || a->struct_size != sizeof(vorta_router_args)) return VORTA_EINVAL;
  if (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16 && a->dtype != VORTA_FP32) return VORTA_EUNSUPPORTED;
  if (a->batch <= 0 || a->heads <= 0 || a->heads > 1024 || a->n_experts <= 0 || a->n_experts > 64) return VORTA_EINVAL;
  if (!a->scores || !a->expert_of_head || !a->head_lists || !a->head_counts) return VORTA_EINVAL;
  RParams p{nullptr, nullptr, nullptr, a->batch, 0, a->heads, a->n_experts, a->tau,
            a->scores, a->expert_of_head, a->head_lists, a->head_counts, nullptr};
  hipStream_t st = (hipStream_t)hip_stream;
  if (a->dtype == VORTA_BF16) hipLaunchKernelGGL(route_scores_kernel<__bf16>, dim3(1), dim3(64), 0, st, p);
  else if (a->dtype == VORTA_FP16) hipLaunchKernelGGL(route_scores_kernel<_Float16>, dim3(1), dim3(64), 0, st, p);
  else hipLaunchKernelGGL(route_scores_kernel<float>, dim3(1), dim3(64), 0, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}

namespace {
int route_layers(const vorta_router_args* a, int n_layers, void* hip_stream);
}

extern "C" int vorta_router_route(const vorta_router_args* a, void* hip_stream) { return route_layers(a, 1, hip_stream); }

extern "C" int vorta_route_plan(const vorta_router_args* a, int32_t n_layers, void* hip_stream) {
  if (n_layers <= 0 || n_layers > 65535) return VORTA_EINVAL;
  return route_layers(a, n_layers, hip_stream);
}

namespace {
int route_layers(const vorta_router_args* a, int n_layers, void* hip_stream) {
  if (!a || a->struct_size != sizeof(vorta_router_args)) return VORTA_EINVAL;
  if (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16) return VORTA_EUNSUPPORTED;
  if (a->batch <= 0 || a->embed_dim <= 0 || a->heads <= 0 || a->heads > 1024 || a->n_experts <= 0 || a->n_experts > 64)
    return VORTA_EINVAL;
  if (!a->temb || !a->weight || !a->bias || !a->expert_of_head || !a->head_lists || !a->head_counts || !a->ws_logits)
    return VORTA_EINVAL;
  if (((uintptr_t)a->temb & 15) || ((uintptr_t)a->weight & 15) || (a->embed_dim % 8)) return VORTA_EINVAL;
  RParams p{a->temb, a->weight, a->bias, a->batch, a->embed_dim, a->heads, a->n_experts, a->tau,
            a->scores, a->expert_of_head, a->head_lists, a->head_counts, a->ws_logits};
  hipStream_t st = (hipStream_t)hip_stream;
  const int items = p.batch * p.H * p.NE;
  const dim3 lg((items + 3) / 4, n_layers);
  if (a->dtype == VORTA_BF16) {
    hipLaunchKernelGGL(router_logits_kernel<__bf16>, lg, dim3(256), 0, st, p);
    hipLaunchKernelGGL(router_route_kernel<__bf16>, dim3(n_layers), dim3(64), 0, st, p);
  } else {
    hipLaunchKernelGGL(router_logits_kernel<_Float16>, lg, dim3(256), 0, st, p);
    hipLaunchKernelGGL(router_route_kernel<_Float16>, dim3(n_layers), dim3(64), 0, st, p);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}
}  // namespace
