// Gather flash-attention forward, 64 query rows per wave (gfx950).
//
// Same math, LDS images and row-table semantics as attn_fwd.hip, different occupancy point: a workgroup is
// 4 waves x 64 query rows (two 32-row sub-tiles per wave), ONE wave per SIMD with the whole 512-register
// file.  Every K / V fragment read from LDS feeds two MFMAs (one per sub-tile), so LDS read traffic per
// FLOP halves (the 32-row kernel keeps the LDS pipe ~70 % busy at full MFMA rate), and the two sub-tiles
// give the wave two independent chains: the softmax (VALU) of one sub-tile is issued between the MFMAs of
// the other.  Order inside a key block:
//     QK(0) | QK(1) + softmax(0) | PV(0) + softmax(1) | PV(1) + staging of the next block
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "attn_common.h"

namespace vorta_attn {

template <typename T>
__global__ __launch_bounds__(256, 1) void attn_fwd_w64_kernel(const Params p) {
  using V8 = typename MF<T>::v8;
  using V4 = typename MF<T>::v4;
  constexpr int NT = 256;
  constexpr int QB = 256;
  constexpr int CH = (KVB * 16) / NT;  // 4
  constexpr int ROWSTEP = NT / 16;     // 16

  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];

  const int nwg = gridDim.x;
  int wg;
  {
    const int b = blockIdx.x, xcd = b & 7, qd = nwg >> 3, r = nwg & 7;
    wg = (xcd < r ? xcd * (qd + 1) : r * (qd + 1) + (xcd - r) * qd) + (b >> 3);
  }
  const int sp = wg % p.n_splits;
  const int rest = wg / p.n_splits;
  const int n_qb = p.n_groups * p.blocks_per_group;
  const int qb = rest % n_qb;
  const int y = rest / n_qb;
  if (p.n_heads_dev && y >= *p.n_heads_dev) return;
  const int head = p.head_list ? p.head_list[y] : y;
  const int grp = qb / p.blocks_per_group;
  const int bi = qb - grp * p.blocks_per_group;
  const int p0 = grp * p.q_group_len + bi * QB;
  const int pend = min((grp + 1) * p.q_group_len, p.n_q);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31;
  const int hh = lane >> 5;

  const int n_kv = p.n_kv_dev ? max(1, min(*p.n_kv_dev, p.n_kv)) : p.n_kv;
  const int q_valid = p.q_valid_dev ? min(*p.q_valid_dev, p.q_valid) : p.q_valid;
  const int nblk_total = (n_kv + KVB - 1) / KVB;
  const int blk0 = sp * p.blocks_per_split;
  const int blk1 = min(blk0 + p.blocks_per_split, nblk_total);

  const int wrow0 = p0 + wave * 64;
  const bool wave_active = wrow0 < pend;  // wave-uniform
  const int32_t* q_rows = p.q_rows ? p.q_rows + (int64_t)y * p.q_rows_sh : nullptr;
  int my_p[2];
  int64_t my_row[2];
  V8 qf[2][8];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    my_p[qs] = wrow0 + 32 * qs + r32;
    const int ld_p = min(my_p[qs], pend - 1);
    my_row[qs] = q_rows ? (int64_t)q_rows[ld_p] : (int64_t)(p.q_row_offset + ld_p);
    const char* qp = p.q + (int64_t)head * p.q_sh + my_row[qs] * p.q_ss + hh * 16;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[qs][ks] = *(const V8*)(qp + ks * 32);
  }

  const int32_t* kv_rows =
      p.kv_rows ? p.kv_rows + (int64_t)y * p.kv_rows_sh + (int64_t)grp * p.kv_rows_sg : nullptr;
  const char* kbase = p.k + (int64_t)head * p.k_sh + (tid & 15) * 16;
  const char* vbase = p.v + (int64_t)head * p.v_sh + (tid & 15) * 16;
  const int lrow0 = tid >> 4;
  const int lcc = tid & 15;
  int k_wr[CH], v_wr[CH];
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int row = lrow0 + i * ROWSTEP;
    k_wr[i] = row * ROWB + ((lcc ^ (row & 15)) << 4);
    v_wr[i] = TILE_BYTES + row * ROWB + ((lcc ^ ((row & 3) << 2)) << 4);
  }
  u32x4 kreg[CH], vreg[CH];
  int64_t nrow[CH];
#define FETCH_ROWS(blk_)                                                          \
  _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) {                             \
    const int pos_ = min((blk_) * KVB + lrow0 + i_ * ROWSTEP, n_kv - 1);          \
    nrow[i_] = kv_rows ? (int64_t)kv_rows[pos_] : (int64_t)(p.kv_row_offset + pos_); \
  }
#define ISSUE_LOADS()                                                             \
  _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) {                             \
    kreg[i_] = *(const u32x4*)(kbase + nrow[i_] * p.k_ss);                        \
    vreg[i_] = *(const u32x4*)(vbase + nrow[i_] * p.v_ss);                        \
  }
#define WRITE_LDS(buf_)                                                           \
  _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) {                             \
    *(u32x4*)(smem + (buf_) * BUF_BYTES + k_wr[i_]) = kreg[i_];                   \
    *(u32x4*)(smem + (buf_) * BUF_BYTES + v_wr[i_]) = vreg[i_];                   \
  }

  int k_rd[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_rd[ks] = r32 * ROWB + (((2 * ks + hh) ^ (r32 & 15)) << 4);
  int v_rd[4];
  {
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      v_rd[dt] = TILE_BYTES + (4 * (g >> 1) + q4) * ROWB + ((dt ^ q4) << 6) + 32 * (g & 1) + 8 * pp;
  }

  f32x16 o[2][4];
#pragma unroll
  for (int qs = 0; qs < 2; ++qs)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) o[qs][dt][i] = 0.f;
  float m_run[2] = {-1e30f, -1e30f}, l_run[2] = {0.f, 0.f};
  const float c = p.scale_log2;

  if (blk0 < blk1) {
    FETCH_ROWS(blk0);
    ISSUE_LOADS();
    if (blk0 + 1 < blk1) { FETCH_ROWS(blk0 + 1); }
    WRITE_LDS(0);
    if (blk0 + 1 < blk1) { ISSUE_LOADS(); }
    if (blk0 + 2 < blk1) { FETCH_ROWS(blk0 + 2); }
    __syncthreads();
  }

  for (int blk = blk0; blk < blk1; ++blk) {
    const int buf = (blk - blk0) & 1;
    const char* sb = smem + buf * BUF_BYTES;
    const int kv0 = blk * KVB;
    const bool tail = kv0 + KVB > n_kv;
    if (wave_active) {
      f32x16 s[2][2];
      V8 pb[2][4];
#pragma unroll
      for (int qs = 0; qs < 2; ++qs) {
#pragma unroll
        for (int i = 0; i < 16; ++i) { s[qs][0][i] = 0.f; s[qs][1][i] = 0.f; }
      }
      // ---- S^T = K . Q^T: every K fragment is read once and feeds both sub-tiles ----
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const V8 k0 = *(const V8*)(sb + k_rd[ks]);
        const V8 k1 = *(const V8*)(sb + k_rd[ks] + 32 * ROWB);
        s[0][0] = MF<T>::mfma(k0, qf[0][ks], s[0][0]);
        s[0][1] = MF<T>::mfma(k1, qf[0][ks], s[0][1]);
        s[1][0] = MF<T>::mfma(k0, qf[1][ks], s[1][0]);
        s[1][1] = MF<T>::mfma(k1, qf[1][ks], s[1][1]);
      }
      if (tail) {
#pragma unroll
        for (int qs = 0; qs < 2; ++qs)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (kv0 + row >= n_kv) s[qs][0][i] = -INFINITY;
            if (kv0 + 32 + row >= n_kv) s[qs][1][i] = -INFINITY;
          }
      }
      auto softmax = [&](int qs) {
        float mx = s[qs][0][0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s[qs][0][i]);
#pragma unroll
        for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s[qs][1][i]);
        mx = half_max(mx);
        const float m_new = fmaxf(m_run[qs], mx);
        if (!__all(m_new == m_run[qs])) {
          const float alpha = __builtin_amdgcn_exp2f((m_run[qs] - m_new) * c);
#pragma unroll
          for (int dt = 0; dt < 4; ++dt)
#pragma unroll
            for (int i = 0; i < 16; ++i) o[qs][dt][i] *= alpha;
          l_run[qs] *= alpha;
          m_run[qs] = m_new;
        }
        const float mc = m_run[qs] * c;
        float lsum = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          s[qs][0][i] = __builtin_amdgcn_exp2f(fmaf(s[qs][0][i], c, -mc));
          s[qs][1][i] = __builtin_amdgcn_exp2f(fmaf(s[qs][1][i], c, -mc));
          lsum += s[qs][0][i] + s[qs][1][i];
        }
        l_run[qs] += lsum;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          pb[qs][0][j] = (T)s[qs][0][j];
          pb[qs][1][j] = (T)s[qs][0][8 + j];
          pb[qs][2][j] = (T)s[qs][1][j];
          pb[qs][3][j] = (T)s[qs][1][8 + j];
        }
      };
      softmax(0);
      softmax(1);
      // ---- O^T += V^T . P^T: every V fragment feeds both sub-tiles ----
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
          const V4 lo = MF<T>::tr(sb + v_rd[dt] + (16 * kg) * ROWB);
          const V4 hi = MF<T>::tr(sb + v_rd[dt] + (16 * kg + 8) * ROWB);
          V8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
          o[0][dt] = MF<T>::mfma(vf, pb[0][kg], o[0][dt]);
          o[1][dt] = MF<T>::mfma(vf, pb[1][kg], o[1][dt]);
        }
      }
    }
    if (blk + 1 < blk1) {
      WRITE_LDS(buf ^ 1);
      if (blk + 2 < blk1) { ISSUE_LOADS(); }
      if (blk + 3 < blk1) { FETCH_ROWS(blk + 3); }
    }
    __syncthreads();
  }
#undef FETCH_ROWS
#undef ISSUE_LOADS
#undef WRITE_LDS

  if (!wave_active) return;
#pragma unroll
  for (int qs = 0; qs < 2; ++qs) {
    const float l_tot = half_sum(l_run[qs]);
    const bool row_ok = my_p[qs] < pend;
    if (p.n_splits > 1) {
      if (row_ok) {
        const int64_t slot = ((int64_t)y * p.n_splits + sp) * p.n_q + my_p[qs];
        float* wo = p.ws_o + slot * D;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            f32x4 v = {o[qs][dt][4 * rg], o[qs][dt][4 * rg + 1], o[qs][dt][4 * rg + 2], o[qs][dt][4 * rg + 3]};
            *(f32x4*)(wo + 32 * dt + 8 * rg + 4 * hh) = v;
          }
        if (hh == 0) {
          p.ws_ml[slot * 2] = m_run[qs];
          p.ws_ml[slot * 2 + 1] = l_tot;
        }
      }
      continue;
    }
    if (!row_ok) continue;
    const float inv = (my_p[qs] < q_valid && l_tot > 0.f) ? 1.f / l_tot : 0.f;
    u32x2 packed[16];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        V4 t;
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (T)(o[qs][dt][4 * rg + j] * inv);
        packed[dt * 4 + rg] = *(u32x2*)&t;
      }
    char* obase = p.o + (int64_t)head * p.o_sh + hh * 8;
    {
      char* op = obase + my_row[qs] * p.o_ss;
#pragma unroll
      for (int i = 0; i < 16; ++i) *(u32x2*)(op + (32 * (i >> 2) + 8 * (i & 3)) * 2) = packed[i];
    }
    if (p.dup_rows && my_p[qs] < p.n_dup_pos) {
      const int32_t* dr = p.dup_rows + (int64_t)y * p.dup_rows_sh + (int64_t)my_p[qs] * p.n_dup;
      for (int d = 0; d < p.n_dup; ++d) {
        char* op = obase + (int64_t)dr[d] * p.o_ss;
#pragma unroll
        for (int i = 0; i < 16; ++i) *(u32x2*)(op + (32 * (i >> 2) + 8 * (i & 3)) * 2) = packed[i];
      }
    }
  }
}

template <typename T>
int launch_w64(const Params& p, hipStream_t st) {
  const int64_t n_qb = (int64_t)p.n_groups * p.blocks_per_group;
  const int64_t total = n_qb * p.n_heads * p.n_splits;
  if (total <= 0) return VORTA_OK;
  if (total > 0x7fffffff) return VORTA_EINVAL;
  hipLaunchKernelGGL((attn_fwd_w64_kernel<T>), dim3((unsigned)total), dim3(256), 0, st, p);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}

template int launch_w64<__bf16>(const Params&, hipStream_t);
template int launch_w64<_Float16>(const Params&, hipStream_t);

}  // namespace vorta_attn
