// Coreset ("low-res") token selection for gfx950: include/vorta_hip.h vorta_coreset_select.
//
// HBM-bound: every token row (256 B) of the selected heads is read exactly once; the output is two small
// int32 tables (keep list / drop list) that the attention kernel consumes as row indirection, so the
// reference's pooled copies (coreset_select.py:91-93,118-123) and its three scatters (:169-184) never
// touch memory.  One wave per window group: the four 16-lane quarters each own one margin token at a
// time (16 lanes x 16 B = one row), the centre row is shared.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

namespace {

constexpr int MAX_G = 64;  // tokens per window group

struct CParams {
  const char* x; int64_t x_sh, x_ss;
  const int32_t* head_list; const int32_t* n_heads_dev;
  int n_heads;
  int lat[3], grp[3], ng[3];
  int n_groups, g, centre, n_keep;
  int tail_first, n_tail;
  const int32_t* row_map;
  int32_t* keep_rows; int64_t keep_sh;
  int32_t* drop_rows; int64_t drop_sh;
  int32_t* kv_rows; int64_t kv_sh;  // optional second keep list in group-major, token-ascending order (key side)
};

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<__bf16>(__bf16 v) { return (float)v; }
template <> __device__ __forceinline__ float to_f<_Float16>(_Float16 v) { return (float)v; }

__device__ __forceinline__ float quarter_sum(float v) {
  // sum over the 16 lanes of a quarter wave
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 2);
  v += __shfl_xor(v, 1);
  return v;
}

template <typename T>
__global__ __launch_bounds__(256) void coreset_select_kernel(const CParams p) {
  typedef __attribute__((ext_vector_type(8))) T V8;
  __shared__ float sims[4][MAX_G];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t item = (int64_t)blockIdx.x * 4 + wave;
  const int y = (int)(item / p.n_groups);
  const int gidx = (int)(item - (int64_t)y * p.n_groups);
  const bool valid = y < p.n_heads && !(p.n_heads_dev && y >= *p.n_heads_dev);  // wave-uniform
  const int head = valid ? (p.head_list ? p.head_list[y] : y) : 0;

  // group coordinates -> first token of the window
  const int gw_ = gidx % p.ng[2];
  const int gh_ = (gidx / p.ng[2]) % p.ng[1];
  const int gf_ = gidx / (p.ng[2] * p.ng[1]);
  const int f0 = gf_ * p.grp[0], h0 = gh_ * p.grp[1], w0 = gw_ * p.grp[2];
  auto token_of = [&](int idx) {  // idx: in-window raster index
    const int dw = idx % p.grp[2];
    const int dh = (idx / p.grp[2]) % p.grp[1];
    const int df = idx / (p.grp[2] * p.grp[1]);
    return ((f0 + df) * p.lat[1] + (h0 + dh)) * p.lat[2] + (w0 + dw);
  };
  auto row_of = [&](int tok) { return p.row_map ? p.row_map[tok] : tok; };

  const int qtr = lane >> 4, sub = lane & 15;
  const int nm = p.g - 1;
  int ctok = 0;
  if (valid) {
  const char* xb = p.x + (int64_t)head * p.x_sh + sub * 16;
  ctok = token_of(p.centre);
  // the centre row and the quarter's margin rows, UNR at a time, are requested before anything is computed (one row per
  // quarter in flight read 3.2 TB/s)
  constexpr int UNR = 5;  // window (3,3,2): 17 margins = 5 per quarter
  const V8 cv = *(const V8*)(xb + (int64_t)row_of(ctok) * p.x_ss);
  V8 mv[UNR];
  auto request = [&](int m0) {
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int m = m0 + 4 * u;
      if (m < nm) mv[u] = *(const V8*)(xb + (int64_t)row_of(token_of(m < p.centre ? m : m + 1)) * p.x_ss);
    }
  };
  request(qtr);
  float cf[8], cn = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) { cf[j] = to_f<T>(cv[j]); cn += cf[j] * cf[j]; }
  cn = quarter_sum(cn);
  const float cinv = 1.f / fmaxf(sqrtf(cn), 1e-12f);  // F.normalize: x / max(||x||, eps)

  for (int m0 = qtr; m0 < nm; m0 += 4 * UNR) {
    if (m0 != qtr) request(m0);
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int m = m0 + 4 * u;
      if (m >= nm) break;
      float dot = 0.f, mn = 0.f;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float f = to_f<T>(mv[u][j]);
        mn += f * f;
        dot += (cf[j] * cinv) * f;
      }
      dot = quarter_sum(dot);
      mn = quarter_sum(mn);
      if (sub == 0) sims[wave][m] = dot / fmaxf(sqrtf(mn), 1e-12f);
    }
  }
  }
  __syncthreads();
  if (!valid) return;

  // stable ascending rank of every margin (lane m owns margin m)
  int32_t* keep = p.keep_rows ? p.keep_rows + (int64_t)y * p.keep_sh : nullptr;
  int32_t* kvl = p.kv_rows ? p.kv_rows + (int64_t)y * p.kv_sh : nullptr;
  bool kept = false;
  int row = 0, idx = 0;
  if (lane < nm) {
    const float mine = sims[wave][lane];
    int rank = 0;
    for (int i = 0; i < nm; ++i) {
      const float o = sims[wave][i];
      rank += (o < mine || (o == mine && i < lane)) ? 1 : 0;
    }
    idx = lane < p.centre ? lane : lane + 1;
    row = row_of(token_of(idx));
    kept = rank < p.n_keep;
    if (kept) {
      if (keep) keep[p.n_groups + gidx * p.n_keep + rank] = row;
    } else if (p.drop_rows) {
      p.drop_rows[(int64_t)y * p.drop_sh + (int64_t)gidx * (nm - p.n_keep) + (rank - p.n_keep)] = row;
    }
  }
  if (lane == 0 && keep) keep[gidx] = row_of(ctok);
  if (kvl) {
    // key side: the softmax does not depend on the order of the keys, so the group's centre and kept margins stay
    // together, in ascending token order -- the K/V tiles of the attention kernel then gather rows that are neighbours
    // in memory instead of striding through all G centres first (coreset_select.py:116-124 packs [centres | margins])
    const uint64_t km = __ballot(kept);  // by margin number
    const uint64_t lowm = (1ull << p.centre) - 1;
    const uint64_t ki = (km & lowm) | ((km >> p.centre) << (p.centre + 1)) | (1ull << p.centre);  // by in-window index
    int32_t* dst = kvl + (int64_t)gidx * (1 + p.n_keep);
    if (kept) dst[__popcll(ki & ((1ull << idx) - 1))] = row;
    if (lane == 0) dst[__popcll(ki & lowm)] = row_of(ctok);
  }
  // the tail (text tokens) is appended once per head slot by group 0
  if (gidx == 0) {
    const int base = p.n_groups * (1 + p.n_keep);
    for (int i = lane; i < p.n_tail; i += 64) {
      const int r = row_of(p.tail_first + i);
      if (keep) keep[base + i] = r;
      if (kvl) kvl[base + i] = r;
    }
  }
}

}  // namespace

extern "C" int vorta_coreset_select(const vorta_coreset_args* a, void* hip_stream) {
  if (!a || a->struct_size != sizeof(vorta_coreset_args)) return VORTA_EINVAL;
  if (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16) return VORTA_EUNSUPPORTED;
  if (a->head_dim != 128) return VORTA_EUNSUPPORTED;
  if (a->n_heads < 0) return VORTA_EINVAL;
  if (a->n_heads == 0) return VORTA_OK;
  if (!a->x.ptr || (!a->keep_rows && !a->keep_rows_kv)) return VORTA_EINVAL;
  if (((uintptr_t)a->x.ptr & 15) || (a->x.stride_s % 8) || (a->x.stride_h % 8) || a->x.stride_s < 128) return VORTA_EINVAL;
  CParams p{};
  p.g = 1;
  p.n_groups = 1;
  for (int i = 0; i < 3; ++i) {
    if (a->latent[i] <= 0 || a->group[i] <= 0) return VORTA_EINVAL;
    // the reference crops partial windows (coreset_select.py:40) but its callers require an exact fit
    // (hunyuan.py:269-272): cropped tokens would never be written by the fused unpool, so refuse them
    if (a->latent[i] % a->group[i]) return VORTA_EINVAL;
    p.lat[i] = a->latent[i]; p.grp[i] = a->group[i]; p.ng[i] = a->latent[i] / a->group[i];
    p.g *= a->group[i]; p.n_groups *= p.ng[i];
  }
  if (p.g > MAX_G || p.g < 2) return VORTA_EUNSUPPORTED;
  if (a->n_keep < 0 || a->n_keep > p.g - 1 || a->n_tail < 0) return VORTA_EINVAL;
  p.centre = (p.grp[0] / 2) * p.grp[1] * p.grp[2] + (p.grp[1] / 2) * p.grp[2] + p.grp[2] / 2;
  p.n_keep = a->n_keep;
  p.x = (const char*)a->x.ptr; p.x_sh = a->x.stride_h * 2; p.x_ss = a->x.stride_s * 2;
  p.head_list = a->head_list; p.n_heads_dev = a->n_heads_dev; p.n_heads = a->n_heads;
  p.tail_first = a->tail_first; p.n_tail = a->n_tail; p.row_map = a->row_map;
  p.keep_rows = a->keep_rows; p.keep_sh = a->keep_rows_stride_h;
  p.drop_rows = a->drop_rows; p.drop_sh = a->drop_rows_stride_h;
  p.kv_rows = a->keep_rows_kv; p.kv_sh = a->keep_rows_kv_stride_h;
  const int64_t items = (int64_t)p.n_heads * p.n_groups;
  const int64_t blocks = (items + 3) / 4;
  if (blocks > 0x7fffffff) return VORTA_EINVAL;
  hipStream_t st = (hipStream_t)hip_stream;
  if (a->dtype == VORTA_BF16)
    hipLaunchKernelGGL(coreset_select_kernel<__bf16>, dim3((unsigned)blocks), dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL(coreset_select_kernel<_Float16>, dim3((unsigned)blocks), dim3(256), 0, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}
