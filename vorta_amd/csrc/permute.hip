// Head-permuting row copies for gfx950: include/vorta_hip.h vorta_permute_heads.
//
// The Ulysses exchange needs the heads bound for one peer contiguous on the wire (vorta/ulysses/utils.py:61-91 gets there
// with transpose + .contiguous() per tensor and direction).  Here the send-side staging of q, k and v (strided projection
// views -> head-ordered blocks), the text rows that follow the local head slots, and the receive-side un-permute of the
// output are each ONE launch over up to four tensors.  HBM-bound: one read and one write of every row; a row of D = 128
// elements (256 B, or 128 B for e4m3) is moved by 16 (8) lanes with 16-byte accesses, two rows in flight per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

namespace {

constexpr int MAXT = 4;

struct PParams {
  const char* src[MAXT]; int64_t src_sh[MAXT], src_ss[MAXT];  // bytes
  char* dst[MAXT]; int64_t dst_sh[MAXT], dst_ss[MAXT];
  const int32_t* src_map; const int32_t* dst_map;
  int heads, n_rows, lanes_per_row_log2;
};

__global__ __launch_bounds__(256) void permute_heads_kernel(const PParams p) {
  const int t = blockIdx.y;
  const int lpr = 1 << p.lanes_per_row_log2;          // 16 or 8 lanes per row
  const int rows_per_pass = 256 >> p.lanes_per_row_log2;
  const int sub = threadIdx.x & (lpr - 1);
  const int64_t total = (int64_t)p.heads * p.n_rows;
  const int64_t i0 = ((int64_t)blockIdx.x * 2) * rows_per_pass + (threadIdx.x >> p.lanes_per_row_log2);
  const int64_t i1 = i0 + rows_per_pass;
  u32x4 v0, v1;
  char* d0 = nullptr; char* d1 = nullptr;
  if (i0 < total) {
    const int h = (int)(i0 / p.n_rows);
    const int64_t r = i0 - (int64_t)h * p.n_rows;
    const int hs = p.src_map ? p.src_map[h] : h, hd = p.dst_map ? p.dst_map[h] : h;
    v0 = *(const u32x4*)(p.src[t] + (int64_t)hs * p.src_sh[t] + r * p.src_ss[t] + sub * 16);
    d0 = p.dst[t] + (int64_t)hd * p.dst_sh[t] + r * p.dst_ss[t] + sub * 16;
  }
  if (i1 < total) {
    const int h = (int)(i1 / p.n_rows);
    const int64_t r = i1 - (int64_t)h * p.n_rows;
    const int hs = p.src_map ? p.src_map[h] : h, hd = p.dst_map ? p.dst_map[h] : h;
    v1 = *(const u32x4*)(p.src[t] + (int64_t)hs * p.src_sh[t] + r * p.src_ss[t] + sub * 16);
    d1 = p.dst[t] + (int64_t)hd * p.dst_sh[t] + r * p.dst_ss[t] + sub * 16;
  }
  if (d0) *(u32x4*)d0 = v0;
  if (d1) *(u32x4*)d1 = v1;
}

}  // namespace

extern "C" int vorta_permute_heads(const vorta_permute_args* a, void* hip_stream) {
  if (!a || a->struct_size != sizeof(vorta_permute_args)) return VORTA_EINVAL;
  if (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16 && a->dtype != VORTA_FP8E4M3) return VORTA_EUNSUPPORTED;
  if (a->head_dim != 128) return VORTA_EUNSUPPORTED;
  if (a->n_tensors < 0 || a->n_tensors > MAXT || a->heads < 0 || a->n_rows < 0) return VORTA_EINVAL;
  if (a->n_tensors == 0 || a->heads == 0 || a->n_rows == 0) return VORTA_OK;
  const int es = a->dtype == VORTA_FP8E4M3 ? 1 : 2;
  const int64_t align = 16 / es;  // strides in elements must keep 16-byte alignment
  PParams p{};
  for (int t = 0; t < a->n_tensors; ++t) {
    const vorta_tensor& s = a->src[t];
    const vorta_tensor& d = a->dst[t];
    if (!s.ptr || !d.ptr || ((uintptr_t)s.ptr & 15) || ((uintptr_t)d.ptr & 15)) return VORTA_EINVAL;
    if ((s.stride_s % align) || (s.stride_h % align) || (d.stride_s % align) || (d.stride_h % align)) return VORTA_EINVAL;
    if (s.stride_s < a->head_dim || d.stride_s < a->head_dim) return VORTA_EINVAL;
    p.src[t] = (const char*)s.ptr; p.src_sh[t] = s.stride_h * es; p.src_ss[t] = s.stride_s * es;
    p.dst[t] = (char*)d.ptr; p.dst_sh[t] = d.stride_h * es; p.dst_ss[t] = d.stride_s * es;
  }
  p.src_map = a->src_map; p.dst_map = a->dst_map;
  p.heads = a->heads; p.n_rows = a->n_rows;
  p.lanes_per_row_log2 = es == 2 ? 4 : 3;
  const int64_t total = (int64_t)p.heads * p.n_rows;
  const int64_t per_block = 2 * (256 >> p.lanes_per_row_log2);
  const int64_t blocks = (total + per_block - 1) / per_block;
  if (blocks > 0x7fffffffll) return VORTA_EINVAL;
  hipLaunchKernelGGL(permute_heads_kernel, dim3((unsigned)blocks, (unsigned)a->n_tensors), dim3(256), 0,
                     (hipStream_t)hip_stream, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}
