// Shared pieces of the gather flash-attention kernels (attn_fwd.hip: 32 query rows per wave, two waves per
// SIMD; attn_fwd_w64.hip: 64 query rows per wave, one wave per SIMD).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

namespace vorta_attn {


constexpr int KVB = 64;            // keys per block
constexpr int D = 128;             // head dim
constexpr int ROWB = D * 2;        // bytes per row
constexpr int TILE_BYTES = KVB * ROWB;  // 16 KiB
constexpr int BUF_BYTES = 2 * TILE_BYTES;

struct Params {
  const char* q; const char* k; const char* v; char* o;
  int64_t q_sh, k_sh, v_sh, o_sh;  // head strides in bytes
  int64_t q_ss, k_ss, v_ss, o_ss;  // row strides in bytes
  const int32_t* head_list; const int32_t* n_heads_dev;
  int n_heads;
  int n_q, q_group_len, q_row_offset, q_valid;
  int n_groups, blocks_per_group;
  const int32_t* q_rows; int64_t q_rows_sh;
  int n_kv, kv_row_offset;
  const int32_t* n_kv_dev; const int32_t* q_valid_dev;
  const int32_t* kv_rows; int64_t kv_rows_sh, kv_rows_sg;
  const int32_t* dup_rows; int64_t dup_rows_sh; int n_dup_pos, n_dup;
  float scale_log2;  // scale * log2(e)
  int n_splits, blocks_per_split;
  float* ws_o; float* ws_ml;
  int xcd_remap;
  float defer_log2;  // online-softmax rescale is skipped while the row max grows by <= this (log2 units)
  const int32_t* q_block_table;  // optional [n_groups * blocks_per_group][3] = (group, first position, end position)
  int wg_per_slot;               // workgroups of one head slot = n_groups * blocks_per_group * n_splits
};

// query block qb of a launch -> its group (key list), first position and the end of its positions
__device__ __forceinline__ void q_block_of(const Params& p, int qb, int rows_per_block, int& grp, int& p0, int& pend) {
  if (p.q_block_table) {
    const int32_t* t = p.q_block_table + 3 * qb;
    grp = t[0]; p0 = t[1]; pend = t[2];
  } else {
    grp = qb / p.blocks_per_group;
    p0 = grp * p.q_group_len + (qb - grp * p.blocks_per_group) * rows_per_block;
    pend = min((grp + 1) * p.q_group_len, p.n_q);
  }
}

// Logical workgroup id of physical block b of a launch (or of a fused segment) with n ids.
// XCD-aware order: workgroups whose ids are equal mod 8 share an XCD (round-robin dispatch), so each such class gets a
// contiguous chunk of the logical ids (same head, neighbouring query blocks: one L2 serves the K/V stream instead of
// eight).  With a device-resident head count (n_heads_dev) the grid is sized for every head slot but only the first
// *n_heads_dev are live: the chunks are cut from the LIVE ids, so all eight XCDs share the live work to within one
// workgroup; a dead block keeps its own id, which decodes to a slot >= *n_heads_dev, and leaves through the body's slot
// check (no second exit path: one in the kernel wrappers cost the single-launch e4m3 kernels 36-100 B of scratch).  (Cut from all n ids -- slot-major -- the live third of a uniform
// Hunyuan layer landed on three XCDs: the device-routed fused launch took 201 ms against 68 ms with host counts;
// spreading whole slots instead left 14 live slots of 40 at 2 + 2 + ... + 1 + 1 per XCD, 13 % over the host-count time.)
__device__ __forceinline__ int live_order(const Params& p, int b, int n, bool remap) {
  if (p.n_heads_dev) n = min(n, max(*p.n_heads_dev, 0) * p.wg_per_slot);
  if (!remap || b >= n) return b;
  const int xcd = b & 7, qd = n >> 3, r = n & 7;
  return (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (b >> 3);
}

#ifndef VORTA_MAX_SEGMENTS
#define VORTA_MAX_SEGMENTS VORTA_MAX_FUSED_LAUNCHES /* include/vorta_hip.h */
#endif
constexpr int MAX_SEGMENTS = VORTA_MAX_SEGMENTS;  // launches fused into one grid (vorta_attn_fwd_batch): three experts, the text queries or up to
                                 // two partial full-attention heads of a sequence-parallel rank (ulysses/engine.py split_placement)
struct MultiParams {
  Params seg[MAX_SEGMENTS];
  int start[MAX_SEGMENTS + 1];  // first workgroup of each segment; start[n] = grid size
  int n;
};

// attn_fwd.hip: argument validation + launch geometry (in_esize = bytes per q/k element, v_esize per v element: 0 = the
// same; args->dtype names the 2-byte type, or e4m3 when in_esize = 1)
// (k_esize: bytes per k element when it differs from q's -- the int8-score kernel reads 16-bit q and int8 k; 0 = the same)
int fill_params(const vorta_attn_args* a, Params& p, int& block_rows, int in_esize, int v_esize = 0, int k_esize = 0);
// attn_fwd_mx.hip: 16-bit scores, e4m3 P V (vorta_attn_fwd_fp8 / _batch_fp8 with ext->flags bit1)
int mx_fwd(const vorta_attn_args* a, const vorta_attn_fp8_ext* ext, void* hip_stream);
int mx_fwd_batch(const vorta_attn_args* args, const vorta_attn_fp8_ext* ext, int32_t n, void* hip_stream);

template <typename T> struct MF;
template <> struct MF<__bf16> {
  using v8 = bf16x8; using v4 = bf16x4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v4 tr(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((LDS_AS v4*)p);
  }
};
template <> struct MF<_Float16> {
  using v8 = f16x8; using v4 = f16x4;
  static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ v4 tr(const char* p) {
    typedef __attribute__((ext_vector_type(4))) __fp16 h4;
    h4 r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((LDS_AS h4*)p);
    return *(v4*)&r;
  }
};

// v_permlane32_swap(vdst, src) exchanges lanes 32-63 of vdst with lanes 0-31 of src.  Fed the same value
// twice it returns {low half, low half} and {high half, high half}: combining the two results gives every
// lane the reduction over itself and its partner lane ^ 32 (the two lanes that share one query row).
__device__ __forceinline__ float half_max(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float half_sum(float x) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}


}  // namespace vorta_attn
