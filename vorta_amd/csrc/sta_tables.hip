// Sliding-tile geometry as row tables for gfx950: include/vorta_hip.h vorta_sta_build_tables.
//
// The reference expresses the sliding-tile expert as (1) a permutation of Q,K,V into tile-major order
// (tile.py:7-41), (2) a FlexAttention BlockMask over (S+T)^2 built from a mask_mod
// (sliding_attn_flex.py:101-134) and (3) the inverse permutation of the output (tile.py:44-78).  Here the
// same geometry becomes two int32 tables consumed by the gather attention kernel: built once per
// (geometry, t_eff) -- the moment the reference builds its BlockMask (pipeline_hunyuan.py:378-392).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

namespace {

struct SParams {
  int lat[3], tile[3], nt[3], half[3], cnt[3];
  int S, tok, n_tiles, n_kv_video, n_kv, t_eff;
  const int32_t* row_map;
  int32_t* q_rows; int32_t* kv_rows;
};

__device__ __forceinline__ int raster_of(const SParams& p, int tt, int th, int tw, int within) {
  const int c = within % p.tile[2];
  const int b = (within / p.tile[2]) % p.tile[1];
  const int a = within / (p.tile[2] * p.tile[1]);
  return ((tt * p.tile[0] + a) * p.lat[1] + (th * p.tile[1] + b)) * p.lat[2] + (tw * p.tile[2] + c);
}

__global__ __launch_bounds__(256) void sta_q_rows_kernel(const SParams p) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= p.S) return;
  const int ti = i / p.tok, within = i - ti * p.tok;
  const int tw = ti % p.nt[2], th = (ti / p.nt[2]) % p.nt[1], tt = ti / (p.nt[2] * p.nt[1]);
  const int r = raster_of(p, tt, th, tw, within);
  p.q_rows[i] = p.row_map ? p.row_map[r] : r;
}

__global__ __launch_bounds__(256) void sta_kv_rows_kernel(const SParams p) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)p.n_tiles * p.n_kv) return;
  const int qt = (int)(i / p.n_kv);
  const int j = (int)(i - (int64_t)qt * p.n_kv);
  int r;
  if (j < p.n_kv_video) {
    const int q3[3] = {qt / (p.nt[2] * p.nt[1]), (qt / p.nt[2]) % p.nt[1], qt % p.nt[2]};
    const int slot = j / p.tok, within = j - slot * p.tok;
    const int a3[3] = {slot / (p.cnt[2] * p.cnt[1]), (slot / p.cnt[2]) % p.cnt[1], slot % p.cnt[2]};
    int kt[3];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      // torch.clamp(x, min=half, max=n-1-half): min first, then max (sliding_attn_flex.py:118-120)
      const int centre = min(max(q3[d], p.half[d]), p.nt[d] - 1 - p.half[d]);
      kt[d] = max(centre - p.half[d], 0) + a3[d];
    }
    r = raster_of(p, kt[0], kt[1], kt[2], within);
  } else {
    r = p.S + (j - p.n_kv_video);
  }
  p.kv_rows[i] = p.row_map ? p.row_map[r] : r;
}

int fill(const vorta_sta_args* a, SParams& p) {
  if (!a || a->struct_size != sizeof(vorta_sta_args)) return VORTA_EINVAL;
  p.S = 1; p.tok = 1; p.n_tiles = 1;
  int nkt = 1;
  for (int d = 0; d < 3; ++d) {
    if (a->latent[d] <= 0 || a->tile[d] <= 0 || a->window[d] <= 0) return VORTA_EINVAL;
    if (a->latent[d] % a->tile[d]) return VORTA_EINVAL;  // hunyuan.py:264-267 raises ValueError
    p.lat[d] = a->latent[d]; p.tile[d] = a->tile[d]; p.nt[d] = a->latent[d] / a->tile[d];
    p.half[d] = a->window[d] / 2;
    p.cnt[d] = min(p.nt[d], 2 * p.half[d] + 1);
    p.S *= p.lat[d]; p.tok *= p.tile[d]; p.n_tiles *= p.nt[d]; nkt *= p.cnt[d];
  }
  if (a->t_eff < 0) return VORTA_EINVAL;
  p.t_eff = a->t_eff;
  p.n_kv_video = nkt * p.tok;
  p.n_kv = p.n_kv_video + p.t_eff;
  p.row_map = a->row_map; p.q_rows = a->q_rows; p.kv_rows = a->kv_rows;
  return VORTA_OK;
}

__global__ __launch_bounds__(256) void seq_row_map_kernel(int32_t* out, int n, int seg_len, int seg_stride) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = (i / seg_len) * seg_stride + i % seg_len;
}

}  // namespace

extern "C" int vorta_sta_table_sizes(const vorta_sta_args* a, int32_t* n_tiles, int32_t* tok, int32_t* n_kv) {
  SParams p{};
  int rc = fill(a, p);
  if (rc != VORTA_OK) return rc;
  if (n_tiles) *n_tiles = p.n_tiles;
  if (tok) *tok = p.tok;
  if (n_kv) *n_kv = p.n_kv;
  return VORTA_OK;
}

extern "C" int vorta_sta_build_tables(const vorta_sta_args* a, void* hip_stream) {
  SParams p{};
  int rc = fill(a, p);
  if (rc != VORTA_OK) return rc;
  if (!p.q_rows || !p.kv_rows) return VORTA_EINVAL;
  hipStream_t st = (hipStream_t)hip_stream;
  hipLaunchKernelGGL(sta_q_rows_kernel, dim3((p.S + 255) / 256), dim3(256), 0, st, p);
  const int64_t n = (int64_t)p.n_tiles * p.n_kv;
  if ((n + 255) / 256 > 0x7fffffff) return VORTA_EINVAL;
  hipLaunchKernelGGL(sta_kv_rows_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}

extern "C" int vorta_seq_row_map(int32_t* row_map, int32_t n_tokens, int32_t seg_len, int32_t seg_stride_rows,
                                 void* hip_stream) {
  if (!row_map || n_tokens < 0 || seg_len <= 0 || seg_stride_rows < 0) return VORTA_EINVAL;
  if (n_tokens == 0) return VORTA_OK;
  hipLaunchKernelGGL(seq_row_map_kernel, dim3((n_tokens + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream,
                     row_map, n_tokens, seg_len, seg_stride_rows);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}
