// Gather flash-attention forward for gfx950 (MI355X, CDNA4).  One kernel serves the dense, coreset and
// sliding-tile experts of VORTA's routed attention (include/vorta_hip.h: vorta_attn_fwd).
//
// Structure (per workgroup = NW waves, each wave owns 32 query rows; keys in blocks of 64):
//   * swapped QK^T: S^T[kv][q] = K . Q^T with v_mfma_f32_32x32x16 -> a lane owns ONE query (lane&31) and
//     16+16 of the 64 scores of the block, so the row max/sum are in-lane plus one half-wave exchange;
//   * the S^T accumulator is used directly as the B operand of the PV product O^T[d][q] = V^T . P^T
//     (rows of the 32x32 accumulator are the k index of the next MFMA; no lane movement);
//   * K tile in LDS with a 16-way XOR swizzle (ds_read_b128 conflict free), V tile row-major with a
//     64-byte-quadrant swizzle read through ds_read_b64_tr_b16 (hardware transpose);
//   * K/V blocks are fetched global -> registers one block ahead (issue early, write to LDS late),
//     LDS double buffered, one barrier per block; every row goes through an optional int32 row table,
//     which is how pool/unpool (coreset) and tile/untile (sliding tile) are fused into the kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

#include "attn_common.h"
#include "attn_fwd_diag.inc"

namespace {
using namespace vorta_attn;

template <typename T, int NW>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_kernel(const Params p) {
  using V8 = typename MF<T>::v8;
  using V4 = typename MF<T>::v4;
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int CH = (KVB * 16) / NT;  // 16-byte chunks of one tile per thread (2 or 4)
  constexpr int ROWSTEP = NT / 16;     // rows between a thread's consecutive chunks

  __shared__ __attribute__((aligned(16))) char smem[2 * BUF_BYTES];

  // ---- work decomposition (XCD-aware: consecutive logical ids share an XCD's L2) ----
  const int wg = live_order(p, blockIdx.x, gridDim.x, p.xcd_remap);
  const int sp = wg % p.n_splits;
  const int rest = wg / p.n_splits;
  const int n_qb = p.n_groups * p.blocks_per_group;
  const int qb = rest % n_qb;
  const int y = rest / n_qb;
  if (p.n_heads_dev && y >= *p.n_heads_dev) return;
  const int head = p.head_list ? p.head_list[y] : y;
  int grp, p0, pend;
  q_block_of(p, qb, QB, grp, p0, pend);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31;
  const int hh = lane >> 5;

  // ---- key block range of this split (n_kv / q_valid may live on the device: no host sync) ----
  const int n_kv = p.n_kv_dev ? max(1, min(*p.n_kv_dev, p.n_kv)) : p.n_kv;
  const int q_valid = p.q_valid_dev ? min(*p.q_valid_dev, p.q_valid) : p.q_valid;
  const int nblk_total = (n_kv + KVB - 1) / KVB;
  const int blk0 = sp * p.blocks_per_split;
  const int blk1 = min(blk0 + p.blocks_per_split, nblk_total);

  // ---- query rows ----
  const int wrow0 = p0 + wave * 32;
  const bool wave_active = wrow0 < pend;  // wave-uniform
  const int my_p = wrow0 + r32;
  const bool row_ok = my_p < pend;
  const int ld_p = min(my_p, pend - 1);
  const int32_t* q_rows = p.q_rows ? p.q_rows + (int64_t)y * p.q_rows_sh : nullptr;
  const int64_t my_row = q_rows ? (int64_t)q_rows[ld_p] : (int64_t)(p.q_row_offset + ld_p);

  V8 qf[8];
  {
    const char* qp = p.q + (int64_t)head * p.q_sh + my_row * p.q_ss + hh * 16;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) qf[ks] = *(const V8*)(qp + ks * 32);
  }

  // ---- loader setup ----
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
  int64_t nrow[CH];  // row ids of the NEXT block to fetch (index prefetch)
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

  // ---- LDS read addresses ----
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

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  float m_run = -1e30f, l_run = 0.f;
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
    V8 pb[4];
    if (wave_active) {
      // ---------------- S^T = K . Q^T ----------------
      f32x16 s0, s1;
#pragma unroll
      for (int i = 0; i < 16; ++i) { s0[i] = 0.f; s1[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 8; ++ks) {
        const V8 k0 = *(const V8*)(sb + k_rd[ks]);
        const V8 k1 = *(const V8*)(sb + k_rd[ks] + 32 * ROWB);
        s0 = MF<T>::mfma(k0, qf[ks], s0);
        s1 = MF<T>::mfma(k1, qf[ks], s1);
      }
      // ---------------- mask the tail of the key list ----------------
      const int kv0 = blk * KVB;
      if (kv0 + KVB > n_kv) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (kv0 + row >= n_kv) s0[i] = -INFINITY;
          if (kv0 + 32 + row >= n_kv) s1[i] = -INFINITY;
        }
      }
      // ---------------- online softmax (one query per lane) ----------------
      float mx = s0[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s0[i]);
#pragma unroll
      for (int i = 0; i < 16; ++i) mx = fmaxf(mx, s1[i]);
      mx = half_max(mx);
      const float m_new = fmaxf(m_run, mx);
      if (!__all(m_new == m_run)) {
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) o[dt][i] *= alpha;
        l_run *= alpha;
        m_run = m_new;
      }
      const float mc = m_run * c;
      float lsum = 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        s0[i] = __builtin_amdgcn_exp2f(fmaf(s0[i], c, -mc));
        s1[i] = __builtin_amdgcn_exp2f(fmaf(s1[i], c, -mc));
        lsum += s0[i] + s1[i];
      }
      l_run += lsum;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        pb[0][j] = (T)s0[j];
        pb[1][j] = (T)s0[8 + j];
        pb[2][j] = (T)s1[j];
        pb[3][j] = (T)s1[8 + j];
      }
    }
    // ---------------- stage the next key block (loads were issued one block ago) ----------------
    if (blk + 1 < blk1) {
      WRITE_LDS(buf ^ 1);
      if (blk + 2 < blk1) { ISSUE_LOADS(); }
      if (blk + 3 < blk1) { FETCH_ROWS(blk + 3); }
    }
    if (wave_active) {
      // ---------------- O^T += V^T . P^T ----------------
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
        for (int kg = 0; kg < 4; ++kg) {
          const V4 lo = MF<T>::tr(sb + v_rd[dt] + (16 * kg) * ROWB);
          const V4 hi = MF<T>::tr(sb + v_rd[dt] + (16 * kg + 8) * ROWB);
          V8 vf;
#pragma unroll
          for (int j = 0; j < 4; ++j) { vf[j] = lo[j]; vf[4 + j] = hi[j]; }
          o[dt] = MF<T>::mfma(vf, pb[kg], o[dt]);
        }
      }
    }
    __syncthreads();
  }

  if (!wave_active) return;
  // ---------------- epilogue ----------------
  const float l_tot = half_sum(l_run);
  if (p.n_splits > 1) {
    // unnormalised partials: ws_o[y][sp][pos][d], ws_ml[y][sp][pos][2]
    if (row_ok) {
      const int64_t slot = ((int64_t)y * p.n_splits + sp) * p.n_q + my_p;
      float* wo = p.ws_o + slot * D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          f32x4 v = {o[dt][4 * rg], o[dt][4 * rg + 1], o[dt][4 * rg + 2], o[dt][4 * rg + 3]};
          *(f32x4*)(wo + 32 * dt + 8 * rg + 4 * hh) = v;
        }
      if (hh == 0) {
        p.ws_ml[slot * 2] = m_run * c;  // exp2 domain
        p.ws_ml[slot * 2 + 1] = l_tot;
      }
    }
    return;
  }
  if (!row_ok) return;
  const float inv = (my_p < q_valid && l_tot > 0.f) ? 1.f / l_tot : 0.f;
  uint2 packed[16];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      V4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (T)(o[dt][4 * rg + j] * inv);
      packed[dt * 4 + rg] = *(uint2*)&t;
    }
  char* obase = p.o + (int64_t)head * p.o_sh + hh * 8;
  auto store_row = [&](int64_t row) {
    char* op = obase + row * p.o_ss;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) *(uint2*)(op + (32 * dt + 8 * rg) * 2) = packed[dt * 4 + rg];
  };
  store_row(my_row);
  if (p.dup_rows && my_p < p.n_dup_pos) {
    const int32_t* dr = p.dup_rows + (int64_t)y * p.dup_rows_sh + (int64_t)my_p * p.n_dup;
    for (int i = 0; i < p.n_dup; ++i) store_row((int64_t)dr[i]);
  }
}

// The default kernel: scores computed one key block ahead (software pipeline across blocks), K/V tiles staged
// global -> LDS by DMA, softmax folded into the score MFMA.  See the notes inside and DESIGN.md (d).
// K/V ring depth: 2 = one step of DMA latency cover (64 KiB LDS).  A ring of 3 (two steps of cover), paired steps on a ring of
// 4 (one request burst + one barrier per two key blocks) and staggered requests all measured within -1.5 ... +0.3 % of it:
// the end-of-step wait is not where this loop stalls (profiles/r04_probe_mfma_shape_energy.txt; the code is in the history
// of round 4).
constexpr int RING = 2;

template <typename T, int NW, bool KVTAB, int NS>
__device__ __forceinline__ void attn_pipe_dma_body(const Params& p, char* __restrict__ smem, const int wg) {
  // NS = depth of the K and of the V tile rings (NS * 32 KiB of LDS): K(j+NS) / V(j+NS-1) are requested at the
  // top of step j, NS-1 steps before the step that reads them
  static_assert(NS == 2, "ring depth");
  TRACE_ENTRY_()
  using V8 = typename MF<T>::v8;
  using V4 = typename MF<T>::v4;
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int CH = (KVB * 16) / NT;  // 16-byte chunks of one tile per thread (2 or 4)
  constexpr int ROWSTEP = NT / 16;     // rows between a thread's consecutive chunks
#ifndef VORTA_KPRE
#define VORTA_KPRE 2
#endif
  // k-steps of K fragments read ahead of the softmax head; the 128-row body with a key table keeps two row ids per lane
  // on top of that and spilled 5 VGPRs at depth 2 (24 B of scratch): one step there
  constexpr int KPRE = (NW == 4 && KVTAB && VORTA_KPRE > 1) ? 1 : VORTA_KPRE;
  const int sp = wg % p.n_splits;
  const int rest = wg / p.n_splits;
  const int n_qb = p.n_groups * p.blocks_per_group;
  const int qb = rest % n_qb;
  const int y = rest / n_qb;
  if (p.n_heads_dev && y >= *p.n_heads_dev) return;
  const int head = p.head_list ? p.head_list[y] : y;
  int grp, p0, pend;
  q_block_of(p, qb, QB, grp, p0, pend);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r32 = lane & 31;
  const int hh = lane >> 5;

  // ---- key block range of this split (n_kv / q_valid may live on the device: no host sync) ----
  const int n_kv = p.n_kv_dev ? max(1, min(*p.n_kv_dev, p.n_kv)) : p.n_kv;
  const int q_valid = p.q_valid_dev ? min(*p.q_valid_dev, p.q_valid) : p.q_valid;
  const int nblk_total = (n_kv + KVB - 1) / KVB;
  const int blk0 = sp * p.blocks_per_split;
  const int blk1 = min(blk0 + p.blocks_per_split, nblk_total);

  // ---- query rows ----
  const int wrow0 = p0 + wave * 32;
  const bool wave_active = wrow0 < pend;  // wave-uniform
  const int my_p = wrow0 + r32;
  const bool row_ok = my_p < pend;
  const int ld_p = min(my_p, pend - 1);
  const int32_t* q_rows = p.q_rows ? p.q_rows + (int64_t)y * p.q_rows_sh : nullptr;
  const int64_t my_row = q_rows ? (int64_t)q_rows[ld_p] : (int64_t)(p.q_row_offset + ld_p);

  // Q is pre-multiplied by scale*log2(e) once (re-rounded to T): the MFMA then delivers scores in the exp2
  // domain, and with the running max folded into the accumulator's initial value (below) the softmax needs no
  // per-element multiply/subtract at all.
  V8 qf[8];
  {
    const char* qp = p.q + (int64_t)head * p.q_sh + my_row * p.q_ss + hh * 16;
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      const V8 raw = *(const V8*)(qp + ks * 32);
#pragma unroll
      for (int i = 0; i < 8; ++i) qf[ks][i] = (T)((float)raw[i] * p.scale_log2);
    }
  }

  // ---- loader setup ----
  const int32_t* kv_rows =
      p.kv_rows ? p.kv_rows + (int64_t)y * p.kv_rows_sh + (int64_t)grp * p.kv_rows_sg : nullptr;
  const char* kbase = p.k + (int64_t)head * p.k_sh + (tid & 15) * 16;
  const char* vbase = p.v + (int64_t)head * p.v_sh + (tid & 15) * 16;
  const int lrow0 = tid >> 4;
  const int lcc = tid & 15;
  // K / V tiles go global -> LDS directly (buffer_load ... lds): no staging registers, no ds_write.  One wave
  // instruction fills 1 KiB = 4 tile rows (16 lanes x 16 B per row); the destination is lane-linear, so the
  // bank swizzles of the tile images are applied on the SOURCE side: the lane that lands in chunk c' of row r
  // fetches chunk c' ^ swz(r) of that row (same involution the fragment reads apply).
  // K(b) is staged one block ahead of V(b); rings: K tiles at [0, NS*16K), V tiles behind them.
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.k + (int64_t)head * p.k_sh), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.v + (int64_t)head * p.v_sh), 0, 0x7fffffff, 0x00020000);
  const int k_ss32 = (int)p.k_ss, v_ss32 = (int)p.v_ss;
  int k_col[CH], v_col[CH];  // source byte offset inside the row for this lane's chunk
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int row = 4 * (CH * wave + i) + (lane >> 4);
    k_col[i] = ((lane & 15) ^ (row & 15)) << 4;
    v_col[i] = ((lane & 15) ^ ((row & 3) << 2)) << 4;
  }
  int rowK[CH], rowV[CH];  // rows of the next K block / next V block to fetch
  int posK[CH];            // key positions behind rowK (the loop's running value)
#define ROWS_OF(dst_, blk_)                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) {                             \
    const int pos_ = min((blk_) * KVB + 4 * (CH * wave + i_) + (lane >> 4), n_kv - 1); \
    if constexpr (KVTAB) dst_[i_] = kv_rows[pos_];                                \
    else dst_[i_] = p.kv_row_offset + pos_;                                       \
  }
#define DMA_K(par_) _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(   \
      k_rsrc, (LDS_AS void*)(smem + (par_) * TILE_BYTES + (CH * wave + i_) * 1024), 16,                            \
      (int)__umul24((unsigned)rowK[i_], (unsigned)k_ss32) + k_col[i_], 0, 0, 0);
#define DMA_V(par_) _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(   \
      v_rsrc, (LDS_AS void*)(smem + (NS + (par_)) * TILE_BYTES + (CH * wave + i_) * 1024), 16,                     \
      (int)__umul24((unsigned)rowV[i_], (unsigned)v_ss32) + v_col[i_], 0, 0, 0);

  // ---- LDS read addresses ----
  int k_rd[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_rd[ks] = r32 * ROWB + (((2 * ks + hh) ^ (r32 & 15)) << 4);
  int v_rd[4];
  {
    const int g = lane >> 4, i16 = lane & 15, q4 = i16 >> 2, pp = i16 & 3;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      v_rd[dt] = NS * TILE_BYTES + (4 * (g >> 1) + q4) * ROWB + ((dt ^ q4) << 6) + 32 * (g & 1) + 8 * pp;
  }

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  // Online softmax in the exp2 domain.  m_run = reference point of this row (a lower bound of its running max,
  // at most defer_log2 below it); the score MFMAs start from minit = -m_run in every accumulator register, so
  // they produce z - m_run directly and P = exp2 of that.  minit only changes in the (rare) rescale branch.
  float m_run = -1e30f, l_run = 0.f;
  const float defer = p.defer_log2;
  f32x16 sA0, sA1, sB0, sB1;  // scores (minus m_run) of the current / next key block (roles swap every block)
  f32x16 minit;
#pragma unroll
  for (int i = 0; i < 16; ++i) minit[i] = 0.f;
  float mx_cur = -1e30f;       // row max of the current block's (offset) scores

  // scores of a block from the K ring slot `par_` into (d0_, d1_); the tail mask is applied by the consumer
#define QK(d0_, d1_, par_)                                                        \
  {                                                                               \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 8; ++ks_) {                         \
      const V8 k0_ = *(const V8*)(smem + (par_) * TILE_BYTES + k_rd[ks_]);        \
      const V8 k1_ = *(const V8*)(smem + (par_) * TILE_BYTES + k_rd[ks_] + 32 * ROWB); \
      d0_ = MF<T>::mfma(k0_, qf[ks_], ks_ == 0 ? minit : d0_);                    \
      d1_ = MF<T>::mfma(k1_, qf[ks_], ks_ == 0 ? minit : d1_);                    \
    }                                                                             \
  }
#define ROW_MAX(dst_, a_, b_)                                                      \
  {                                                                               \
    float mx_ = a_[0];                                                            \
    _Pragma("unroll") for (int i_ = 1; i_ < 16; ++i_) mx_ = fmaxf(mx_, a_[i_]);   \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) mx_ = fmaxf(mx_, b_[i_]);   \
    dst_ = half_max(mx_);                                                         \
  }
  // The loop only asks two things of the NEXT block's row max: "is it above `defer` (> 0)?" and, if so, its value.  Both
  // are answered by a signed-integer max over the float bit patterns (order-preserving for non-negative floats, any
  // negative result reads as "not above"; -inf of masked keys is a negative integer, there are no NaNs): v_max3_i32
  // needs no canonicalising v_max x,x of the MFMA outputs, and two chains halve the dependent latency.
#define ROW_MAX_POS(dst_, a_, b_)                                                  \
  {                                                                               \
    int m0_ = max(max(__float_as_int(a_[0]), __float_as_int(a_[1])), __float_as_int(a_[2])); \
    int m1_ = max(max(__float_as_int(b_[0]), __float_as_int(b_[1])), __float_as_int(b_[2])); \
    _Pragma("unroll") for (int i_ = 3; i_ < 15; i_ += 2) {                        \
      m0_ = max(max(m0_, __float_as_int(a_[i_])), __float_as_int(a_[i_ + 1]));    \
      m1_ = max(max(m1_, __float_as_int(b_[i_])), __float_as_int(b_[i_ + 1]));    \
    }                                                                             \
    m0_ = max(max(m0_, __float_as_int(a_[15])), __float_as_int(b_[15]));          \
    m0_ = max(m0_, m1_);                                                          \
    auto r_ = __builtin_amdgcn_permlane32_swap((unsigned)m0_, (unsigned)m0_, false, false); \
    dst_ = __int_as_float(max((int)r_[0], (int)r_[1]));                           \
  }
  // the same with the fragments of the first KPRE k-steps already in registers (read at the top of the step,
  // their LDS latency hides under the row-max phase)
#define QK_PRE(d0_, d1_, par_)                                                    \
  {                                                                               \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 8; ++ks_) {                         \
      V8 k0_, k1_;                                                                \
      if (ks_ < KPRE) { k0_ = kpre_[ks_][0]; k1_ = kpre_[ks_][1]; }               \
      else {                                                                      \
        k0_ = *(const V8*)(smem + (par_) * TILE_BYTES + k_rd[ks_]);               \
        k1_ = *(const V8*)(smem + (par_) * TILE_BYTES + k_rd[ks_] + 32 * ROWB);   \
      }                                                                           \
      d0_ = MF<T>::mfma(k0_, qf[ks_], ks_ == 0 ? minit : d0_);                    \
      d1_ = MF<T>::mfma(k1_, qf[ks_], ks_ == 0 ? minit : d1_);                    \
    }                                                                             \
  }
  // move the reference point of the row up by g_ (>= 0): everything accumulated so far and the current block's
  // offset scores are brought to the new reference, and the accumulator seed follows.  The empty asm keeps the
  // seed an opaque 16-register value (otherwise the compiler re-materialises the splat before every use).
#define RAISE_REF(g_, c0_, c1_)                                                   \
  {                                                                               \
    const float alpha_ = __builtin_amdgcn_exp2f(-(g_));                           \
    _Pragma("unroll") for (int dt_ = 0; dt_ < 4; ++dt_)                           \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) o[dt_][i_] *= alpha_;     \
    l_run *= alpha_;                                                              \
    m_run += (g_);                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { c0_[i_] -= (g_); c1_[i_] -= (g_); minit[i_] = -m_run; } \
    asm volatile("" : "+v"(minit));                                               \
  }
  // top of step j: K(j+2) -> the slot K(j) left, V(j+1) -> the slot V(j-1) left.  They are read in step j+1, after the
  // barrier that ends step j (one step of latency cover)
  // (the key positions of the loop are running values, one add per step: written as (block index) * 64 + lane term every
  // unrolled step keeps its own hoisted copy of the sum in a register)
#define STAGE_DMA(kfree_, vfree_, j_)                                             \
  DMA_K(kfree_)                                                                   \
  DMA_V(vfree_)                                                                   \
  _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) rowV[i_] = rowK[i_];          \
  if constexpr (NW == 8) {                                                        \
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) {                           \
      posK[i_] += KVB;                                                            \
      const int pos_ = min(posK[i_], n_kv - 1);                                   \
      if constexpr (KVTAB) rowK[i_] = kv_rows[pos_];                              \
      else rowK[i_] = p.kv_row_offset + pos_;                                     \
    }                                                                             \
  } else { /* (four pieces per wave: the running values cost more registers than the hoisted sums) */ \
    ROWS_OF(rowK, (j_) + NS + 1)                                                  \
  }                                                                               \
  __builtin_amdgcn_sched_barrier(0);
  // end of a step: own DMA requests older than the current step have landed, then the workgroup barrier (which
  // also orders every wave's LDS reads of this step before the next step's overwrites)
  TRACE_VARS_()
#ifndef STEP_SYNC
#define STEP_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif

  // Issue-order recipe for the step's basic block (sched_group_barrier: 0x008 MFMA, 0x100 DS read, 0x400 transcendental,
  // 0x002 VALU).  Score phase: per MFMA one K-fragment read, two exp, two plain VALU (the converts and half of the row
  // sum); PV phase: per MFMA the two transposed V reads and four VALU (the rest of the row sum, the next block's row
  // max).  The score phase is the issue-bound one (exp costs two slots), so everything that can wait moves under the PV
  // MFMAs.  With the post-RA scheduler off (build.py) the recipe is what the hardware sees: +3.5 % on the dense launch
  // and +4 % on the fused layer kernel against no recipe; a dozen other groupings measured between -3 % and +2 %.
  // -DVORTA_SCHED=0 disables it.
#ifndef VORTA_SCHED
#define VORTA_SCHED 1
#endif
#if VORTA_SCHED == 1
#define SCHED_RECIPE()                                                            \
  _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                             \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                            \
    __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);                            \
    __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);                            \
  }                                                                               \
  _Pragma("unroll") for (int g_ = 0; g_ < 16; ++g_) {                             \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                            \
    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                            \
    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);                            \
  }
#else
#define SCHED_RECIPE()
#endif
  // one key block.  The active path is ONE basic block after the (rare) mask / rescale branches: the MFMAs of
  // the next block's scores, the exp/convert VALU work of this block, the staging traffic and the PV MFMAs are
  // all visible to the scheduler together.
#define STEP(c0_, c1_, n0_, n1_, kcur_, knext_, vfree_, j_)                       \
  { /* kcur_ = j % NS: slot of K(j) (free) and of V(j); knext_ = (j+1) % NS; vfree_ = (j-1) % NS */ \
    STAGE_DMA(kcur_, vfree_, j_)                                                  \
    STEP_BODY(c0_, c1_, n0_, n1_, kcur_, knext_, j_)                              \
    STEP_SYNC()                                                                   \
  }
#define STEP_BODY(c0_, c1_, n0_, n1_, kcur_, knext_, j_)                          \
  {                                                                               \
    if (wave_active) {                                                            \
      V8 kpre_[KPRE > 0 ? KPRE : 1][2];                                                          \
      _Pragma("unroll") for (int ks_ = 0; ks_ < KPRE; ++ks_) {                    \
        kpre_[ks_][0] = *(const V8*)(smem + (knext_) * TILE_BYTES + k_rd[ks_]);   \
        kpre_[ks_][1] = *(const V8*)(smem + (knext_) * TILE_BYTES + k_rd[ks_] + 32 * ROWB); \
      }                                                                           \
      /* mx_cur (row max of this block's scores) was computed under the previous step's PV MFMAs; only the */ \
      /* last, partial key block has to mask its tail and redo it here                                      */ \
      if ((j_) * KVB + KVB > n_kv) {                                              \
        int h4_ = 4 * hh; /* opaque: else the 16 sums lane term + register row are hoisted into 16 registers */ \
        asm volatile("" : "+v"(h4_));                                             \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                       \
          const int row_ = (i_ & 3) + 8 * (i_ >> 2) + h4_;                        \
          if ((j_) * KVB + row_ >= n_kv) c0_[i_] = -INFINITY;                     \
          if ((j_) * KVB + 32 + row_ >= n_kv) c1_[i_] = -INFINITY;                \
        }                                                                         \
        ROW_MAX(mx_cur, c0_, c1_)                                                 \
      }                                                                           \
      /* deferred rescale: the reference point moves only when some row of the wave outgrew it by more than */ \
      /* `defer` (so P <= 2^defer: exact in the fp32 sums, representable in fp16/bf16); rows that did not grow  */ \
      /* keep theirs (g = 0)                                                                                  */ \
      if (!__all(mx_cur <= defer)) {                                              \
        const float g_ = fmaxf(mx_cur, 0.f);                                      \
        RAISE_REF(g_, c0_, c1_)                                                   \
      }                                                                           \
      V8 pb_[4];                                                                  \
      QK_PRE(n0_, n1_, knext_) /* block j+1 (harmless garbage past the end) */    \
      float lsum_ = 0.f;                                                          \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                         \
        c0_[i_] = __builtin_amdgcn_exp2f(c0_[i_]);                                \
        c1_[i_] = __builtin_amdgcn_exp2f(c1_[i_]);                                \
        lsum_ += c0_[i_] + c1_[i_];                                               \
      }                                                                           \
      l_run += lsum_;                                                             \
      _Pragma("unroll") for (int e_ = 0; e_ < 8; ++e_) {                          \
        pb_[0][e_] = (T)c0_[e_];                                                  \
        pb_[1][e_] = (T)c0_[8 + e_];                                              \
        pb_[2][e_] = (T)c1_[e_];                                                  \
        pb_[3][e_] = (T)c1_[8 + e_];                                              \
      }                                                                           \
      _Pragma("unroll") for (int dt_ = 0; dt_ < 4; ++dt_) {                       \
        _Pragma("unroll") for (int kg_ = 0; kg_ < 4; ++kg_) {                     \
          const V4 lo_ = MF<T>::tr(smem + (kcur_) * TILE_BYTES + v_rd[dt_] + (16 * kg_) * ROWB); \
          const V4 hi_ = MF<T>::tr(smem + (kcur_) * TILE_BYTES + v_rd[dt_] + (16 * kg_ + 8) * ROWB); \
          V8 vf_;                                                                 \
          _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { vf_[e_] = lo_[e_]; vf_[4 + e_] = hi_[e_]; } \
          o[dt_] = MF<T>::mfma(vf_, pb_[kg_], o[dt_]);                            \
        }                                                                         \
      }                                                                           \
      ROW_MAX_POS(mx_cur, n0_, n1_) /* VALU work that overlaps the PV MFMAs above */ \
      SCHED_RECIPE()                                                              \
    }                                                                             \
  }

  if (blk0 < blk1) {
    // prologue: K(0), K(1) and V(0) -> their ring slots; then rowK = rows(2), rowV = rows(1)
    ROWS_OF(rowK, blk0)
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) rowV[i_] = rowK[i_];
    DMA_K(0)
    DMA_V(0)
    ROWS_OF(rowK, blk0 + 1)
    DMA_K(1)
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) rowV[i_] = rowK[i_];
    ROWS_OF(rowK, blk0 + NS)
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) posK[i_] = (blk0 + NS) * KVB + 4 * (CH * wave + i_) + (lane >> 4);
    __syncthreads();
    if (wave_active) {
      QK(sA0, sA1, 0)  // seed 0: plain scores of the first block
      if (blk0 * KVB + KVB > n_kv) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (blk0 * KVB + row >= n_kv) sA0[i] = -INFINITY;
          if (blk0 * KVB + 32 + row >= n_kv) sA1[i] = -INFINITY;
        }
      }
      ROW_MAX(mx_cur, sA0, sA1)
      // the first block fixes the reference point at its true row max (block blk0 always has a valid key);
      // O and l are still zero, so nothing is rescaled (exp2(-max) could overflow for very negative scores)
      m_run = mx_cur;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sA0[i] -= m_run; sA1[i] -= m_run; minit[i] = -m_run; }
      asm volatile("" : "+v"(minit));
      mx_cur = 0.f;
    }
    __syncthreads();  // every wave has read K(0) before iteration 0 overwrites its slot with K(2)
  }
  for (int blk = blk0; blk < blk1; blk += 2) {
    STEP(sA0, sA1, sB0, sB1, 0, 1, 1, blk)
    if (blk + 1 >= blk1) break;
    STEP(sB0, sB1, sA0, sA1, 1, 0, 0, blk + 1)
  }
#undef QK
#undef QK_PRE
#undef ROW_MAX
#undef ROW_MAX_POS
#undef RAISE_REF
#undef STEP
#undef STEP_BODY
#undef STAGE_DMA
#undef STEP_SYNC
#undef ROWS_OF
#undef DMA_K
#undef DMA_V

  TRACE_FLUSH_()
  if (!wave_active) return;
  // ---------------- epilogue ----------------
  const float l_tot = half_sum(l_run);
  if (p.n_splits > 1) {
    // unnormalised partials: ws_o[y][sp][pos][d], ws_ml[y][sp][pos][2]
    if (row_ok) {
      const int64_t slot = ((int64_t)y * p.n_splits + sp) * p.n_q + my_p;
      float* wo = p.ws_o + slot * D;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          f32x4 v = {o[dt][4 * rg], o[dt][4 * rg + 1], o[dt][4 * rg + 2], o[dt][4 * rg + 3]};
          *(f32x4*)(wo + 32 * dt + 8 * rg + 4 * hh) = v;
        }
      if (hh == 0) {
        p.ws_ml[slot * 2] = m_run;  // already in the exp2 domain
        p.ws_ml[slot * 2 + 1] = l_tot;
      }
    }
    return;
  }
  if (!row_ok) return;
  const float inv = (my_p < q_valid && l_tot > 0.f) ? 1.f / l_tot : 0.f;
  uint2 packed[16];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      V4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (T)(o[dt][4 * rg + j] * inv);
      packed[dt * 4 + rg] = *(uint2*)&t;
    }
  char* obase = p.o + (int64_t)head * p.o_sh + hh * 8;
  auto store_row = [&](int64_t row) {
    char* op = obase + row * p.o_ss;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) *(uint2*)(op + (32 * dt + 8 * rg) * 2) = packed[dt * 4 + rg];
  };
  store_row(my_row);
  if (p.dup_rows && my_p < p.n_dup_pos) {
    const int32_t* dr = p.dup_rows + (int64_t)y * p.dup_rows_sh + (int64_t)my_p * p.n_dup;
    for (int i = 0; i < p.n_dup; ++i) store_row((int64_t)dr[i]);
  }
}


template <typename T, int NW, bool KVTAB>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_pipe_kernel(const Params p) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
  constexpr int NS = RING;
  __shared__ __attribute__((aligned(16))) char smem[2 * NS * TILE_BYTES];
  // XCD-aware work order: consecutive logical ids (same head, neighbouring query blocks) share an XCD's L2
  const int wg = live_order(p, blockIdx.x, gridDim.x, p.xcd_remap);
  attn_pipe_dma_body<T, NW, KVTAB, NS>(p, smem, wg);
#endif
}

// Several launches of the pipelined kernel fused into ONE grid (the experts of a routed layer).  Workgroups
// are dispatched in segment order, longest key loops first, so the tail of one expert (a launch holds only a
// few waves of workgroups per CU-set, fewer still under sequence parallelism) is filled by the next expert's
// workgroups instead of idling until a kernel boundary.

template <typename T>
__global__ __launch_bounds__(512, 2) void attn_fwd_multi_kernel(const MultiParams mp) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
  __shared__ __attribute__((aligned(16))) char smem[2 * RING * TILE_BYTES];  // K and V rings
  const int b = blockIdx.x;
  int s = 0;
#pragma unroll
  for (int i = 1; i < MAX_SEGMENTS; ++i) s += (i < mp.n && b >= mp.start[i]) ? 1 : 0;
  const Params& p = mp.seg[s];
  // XCD-aware order INSIDE the segment: workgroups whose ids are equal mod 8 share an XCD (round-robin
  // dispatch), so give each such class a contiguous chunk of the segment's logical ids (same head,
  // neighbouring query blocks -> one L2 serves the K/V stream instead of eight).  Every XCD still gets 1/8 of
  // every segment, which keeps the chip balanced across segments of different cost.
  const int wg = live_order(p, b - mp.start[s], mp.start[s + 1] - mp.start[s], true);
  if (p.kv_rows) attn_pipe_dma_body<T, 8, true, RING>(p, smem, wg);
  else attn_pipe_dma_body<T, 8, false, RING>(p, smem, wg);
#endif
}

// Merge the split-key partials: one wave per (head slot, query position).
template <typename T>
__global__ __launch_bounds__(256) void attn_combine_kernel(const Params p) {
  const int lane = threadIdx.x & 63;
  const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (item >= (int64_t)p.n_heads * p.n_q) return;
  const int y = (int)(item / p.n_q);
  const int pos = (int)(item - (int64_t)y * p.n_q);
  if (p.n_heads_dev && y >= *p.n_heads_dev) return;
  const int head = p.head_list ? p.head_list[y] : y;
  float m = -1e30f;
  for (int s = 0; s < p.n_splits; ++s) m = fmaxf(m, p.ws_ml[(((int64_t)y * p.n_splits + s) * p.n_q + pos) * 2]);
  float acc0 = 0.f, acc1 = 0.f, l = 0.f;
  for (int s = 0; s < p.n_splits; ++s) {
    const int64_t slot = ((int64_t)y * p.n_splits + s) * p.n_q + pos;
    const float w = __builtin_amdgcn_exp2f(p.ws_ml[slot * 2] - m);  // reference points are stored in the exp2 domain
    l += w * p.ws_ml[slot * 2 + 1];
    const float2 v = *(const float2*)(p.ws_o + slot * D + lane * 2);
    acc0 += w * v.x;
    acc1 += w * v.y;
  }
  const int q_valid = p.q_valid_dev ? min(*p.q_valid_dev, p.q_valid) : p.q_valid;
  const float inv = (pos < q_valid && l > 0.f) ? 1.f / l : 0.f;
  const int32_t* q_rows = p.q_rows ? p.q_rows + (int64_t)y * p.q_rows_sh : nullptr;
  const int64_t row = q_rows ? (int64_t)q_rows[pos] : (int64_t)(p.q_row_offset + pos);
  T pair[2] = {(T)(acc0 * inv), (T)(acc1 * inv)};
  char* ob = p.o + (int64_t)head * p.o_sh + lane * 4;
  *(uint32_t*)(ob + row * p.o_ss) = *(uint32_t*)pair;
  if (p.dup_rows && pos < p.n_dup_pos) {
    const int32_t* dr = p.dup_rows + (int64_t)y * p.dup_rows_sh + (int64_t)pos * p.n_dup;
    for (int i = 0; i < p.n_dup; ++i) *(uint32_t*)(ob + (int64_t)dr[i] * p.o_ss) = *(uint32_t*)pair;
  }
}

template <typename T, int NW>
int launch(const Params& p, hipStream_t st, bool pipe) {
  const int64_t n_qb = (int64_t)p.n_groups * p.blocks_per_group;
  const int64_t total = n_qb * p.n_heads * p.n_splits;
  if (total <= 0) return VORTA_OK;
  if (total > 0x7fffffff) return VORTA_EINVAL;
  hipError_t e;
  if (pipe) {
    if (p.kv_rows) hipLaunchKernelGGL((attn_fwd_pipe_kernel<T, NW, true>), dim3((unsigned)total), dim3(NW * 64), 0, st, p);
    else hipLaunchKernelGGL((attn_fwd_pipe_kernel<T, NW, false>), dim3((unsigned)total), dim3(NW * 64), 0, st, p);
    e = hipGetLastError();
    if (e != hipSuccess) return vorta_set_hip_error(e);
  } else {
    hipLaunchKernelGGL((attn_fwd_kernel<T, NW>), dim3((unsigned)total), dim3(NW * 64), 0, st, p);
    e = hipGetLastError();
    if (e != hipSuccess) return vorta_set_hip_error(e);
  }
  if (p.n_splits > 1) {
    const int64_t items = (int64_t)p.n_heads * p.n_q;
    hipLaunchKernelGGL((attn_combine_kernel<T>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, p);
    e = hipGetLastError();
    if (e != hipSuccess) return vorta_set_hip_error(e);
  }
  return VORTA_OK;
}

}  // namespace

// argument validation + launch geometry, shared with the fp8 kernels (attn_fwd_fp8.hip): in_esize = bytes per q/k/v
// element (2: bf16/fp16, 1: e4m3); the output is always 16-bit
int vorta_attn::fill_params(const vorta_attn_args* a, Params& p, int& block_rows, int in_esize, int v_esize, int k_esize) {
  if (v_esize == 0) v_esize = in_esize;
  if (k_esize == 0) k_esize = in_esize;
  if (!a || a->struct_size != sizeof(vorta_attn_args)) return VORTA_EINVAL;
  if (in_esize == 2 ? (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16) : a->dtype != VORTA_FP8E4M3) return VORTA_EUNSUPPORTED;
  if (a->head_dim != D) return VORTA_EUNSUPPORTED;
  if (a->n_heads < 0 || a->n_q < 0 || a->n_kv < 0) return VORTA_EINVAL;
  if (a->n_heads == 0 || a->n_q == 0) { block_rows = 0; p.n_groups = 0; p.blocks_per_group = 0; p.n_heads = 0; return VORTA_OK; }
  if (a->n_kv == 0) return VORTA_EINVAL;  // softmax over nothing is undefined in the reference too
  if (!a->q.ptr || !a->k.ptr || !a->v.ptr || !a->o.ptr) return VORTA_EINVAL;
  const vorta_tensor* ts[4] = {&a->q, &a->k, &a->v, &a->o};
  for (int i = 0; i < 4; ++i) {
    const vorta_tensor* t = ts[i];
    const int al = 16 / (i == 0 ? in_esize : i == 1 ? k_esize : i == 2 ? v_esize : 2);  // elements per 16 bytes
    if (((uintptr_t)t->ptr & 15) || (t->stride_s % al) || (t->stride_h % al) || t->stride_s < D) return VORTA_EINVAL;
  }
  if (a->n_splits < 1 || a->n_splits > 1024) return VORTA_EINVAL;
  if (!(a->scale > 0.f)) return VORTA_EINVAL;  // the online-softmax thresholds assume a positive scale
  if (a->n_splits > 1 && (!a->ws_o || !a->ws_ml)) return VORTA_EINVAL;
  if (a->q_group_len < 0 || a->q_valid < 0) return VORTA_EINVAL;
  if (a->dup_rows && (a->n_dup < 0 || a->n_dup_pos < 0 || a->n_dup_pos > a->n_q)) return VORTA_EINVAL;
  if (a->block_rows != 0 && a->block_rows != 128 && a->block_rows != 256) return VORTA_EINVAL;
  if (a->variant < 0 || a->variant > 2) return VORTA_EINVAL;
  if (a->variant != 1 && (a->k.stride_s * k_esize >= (1 << 24) || a->v.stride_s * v_esize >= (1 << 24))) return VORTA_EUNSUPPORTED;
  // the pipelined kernel addresses K/V rows with 32-bit buffer offsets (24-bit row x 24-bit stride, 2 GiB window per
  // head): checked here for contiguous key ranges; with a kv_rows table the caller guarantees it (vorta_hip.h)
  if (a->variant != 1 && !a->kv_rows) {
    const int64_t last = (int64_t)a->kv_row_offset + a->n_kv;
    const int64_t ks = a->k.stride_s * k_esize, vs = a->v.stride_s * v_esize;
    const int64_t ss = ks > vs ? ks : vs;
    if (last >= (1 << 24) || last * ss > 0x7fffffffll) return VORTA_EUNSUPPORTED;
  }
  p.q = (const char*)a->q.ptr; p.k = (const char*)a->k.ptr; p.v = (const char*)a->v.ptr; p.o = (char*)a->o.ptr;
  p.q_sh = a->q.stride_h * in_esize; p.k_sh = a->k.stride_h * k_esize; p.v_sh = a->v.stride_h * v_esize; p.o_sh = a->o.stride_h * 2;
  p.q_ss = a->q.stride_s * in_esize; p.k_ss = a->k.stride_s * k_esize; p.v_ss = a->v.stride_s * v_esize; p.o_ss = a->o.stride_s * 2;
  p.head_list = a->head_list; p.n_heads_dev = a->n_heads_dev; p.n_heads = a->n_heads;
  p.n_q = a->n_q; p.q_group_len = a->q_group_len > 0 ? a->q_group_len : a->n_q;
  p.q_row_offset = a->q_row_offset; p.q_valid = a->q_valid;
  p.q_rows = a->q_rows; p.q_rows_sh = a->q_rows_stride_h;
  p.n_kv = a->n_kv; p.kv_row_offset = a->kv_row_offset;
  p.n_kv_dev = a->n_kv_dev; p.q_valid_dev = a->q_valid_dev;
  p.kv_rows = a->kv_rows; p.kv_rows_sh = a->kv_rows_stride_h; p.kv_rows_sg = a->kv_rows_stride_g;
  p.dup_rows = a->dup_rows; p.dup_rows_sh = a->dup_rows_stride_h; p.n_dup_pos = a->n_dup_pos; p.n_dup = a->n_dup;
  p.scale_log2 = a->scale * 1.4426950408889634f;
  p.n_splits = a->n_splits; p.ws_o = a->ws_o; p.ws_ml = a->ws_ml;
  p.xcd_remap = (a->reserved & 1) ? 0 : 1;  // experiment knobs: bit0 disables the XCD remap,
  p.defer_log2 = (a->reserved & 4) ? 0.f : 6.f;  // bit2 turns the deferred rescale off (exact running max)
  p.n_groups = (p.n_q + p.q_group_len - 1) / p.q_group_len;
  block_rows = a->block_rows;
  if (block_rows == 0) {
    // 256-row workgroups (8 waves, one per CU) measured ~1.06x the rate of 128-row ones (4 waves, two per CU); a
    // group that does not fill its last workgroup wastes CU time in proportion to the padding, so compare
    //   rate256 * g / roundup(g,256)   with   rate128 * g / roundup(g,128)
    const int g = p.q_group_len;
    const int64_t pad256 = ((g + 255) / 256) * 256, pad128 = ((g + 127) / 128) * 128;
    block_rows = (pad256 * 100 > pad128 * 106) ? 128 : 256;
  }
  p.blocks_per_group = (p.q_group_len + block_rows - 1) / block_rows;
  p.q_block_table = a->q_block_table;
  if (a->q_block_table) {  // one table row per workgroup; the rows are the caller's contract (device memory)
    if (a->n_q_blocks <= 0 || (a->block_rows != 128 && a->block_rows != 256)) return VORTA_EINVAL;
    p.n_groups = a->n_q_blocks;
    p.blocks_per_group = 1;
  }
  const int nblk = (p.n_kv + KVB - 1) / KVB;
  p.blocks_per_split = (nblk + p.n_splits - 1) / p.n_splits;
  p.wg_per_slot = p.n_groups * p.blocks_per_group * p.n_splits;
  return VORTA_OK;
}

namespace {
using namespace vorta_attn;
int fill_params(const vorta_attn_args* a, Params& p, int& block_rows) { return vorta_attn::fill_params(a, p, block_rows, 2); }
}  // namespace

extern "C" int vorta_attn_workspace_bytes(const vorta_attn_args* a, uint64_t* ws_o_bytes, uint64_t* ws_ml_bytes) {
  if (!a || a->struct_size != sizeof(vorta_attn_args) || !ws_o_bytes || !ws_ml_bytes) return VORTA_EINVAL;
  if (a->n_splits <= 1) { *ws_o_bytes = 0; *ws_ml_bytes = 0; return VORTA_OK; }
  const uint64_t slots = (uint64_t)a->n_heads * (uint64_t)a->n_splits * (uint64_t)a->n_q;
  *ws_o_bytes = slots * D * sizeof(float);
  *ws_ml_bytes = slots * 2 * sizeof(float);
  return VORTA_OK;
}

extern "C" int vorta_attn_plan(const vorta_attn_args* a, int32_t* block_rows_out, int64_t* n_workgroups_out,
                               int32_t* kernel_id_out) {
  Params p{};
  int block_rows = 0;
  int rc = vorta_attn::fill_params(a, p, block_rows, (a && a->dtype == VORTA_FP8E4M3) ? 1 : 2);  // the fp8 kernels share the launch geometry
  if (rc != VORTA_OK) return rc;
  if (block_rows_out) *block_rows_out = block_rows;
  if (kernel_id_out) {
    const bool pipe = a->variant != 1;
    *kernel_id_out = (block_rows == 256 ? 8 : 4) * 16 + (pipe ? 1 : 0) + ((pipe && a->kv_rows) ? 2 : 0);
  }
  if (n_workgroups_out) *n_workgroups_out = (int64_t)p.n_groups * p.blocks_per_group * p.n_heads * (p.n_heads ? p.n_splits : 0);
  return VORTA_OK;
}

extern "C" int vorta_attn_fwd_batch(const vorta_attn_args* args, int32_t n, void* hip_stream) {
  if (!args || n < 0 || n > MAX_SEGMENTS) return VORTA_EINVAL;
  MultiParams mp{};
  int64_t total = 0;
  int dtype = -1, m = 0;
  for (int i = 0; i < n; ++i) {
    Params p{};
    int block_rows = 0;
    int rc = fill_params(&args[i], p, block_rows);
    if (rc != VORTA_OK) return rc;
    if (p.n_heads == 0 || p.n_groups == 0) continue;
    // only launches that resolve to the 256-row pipelined kernel can share a grid
    if (block_rows != 256 || args[i].variant == 1) return VORTA_EUNSUPPORTED;
    if (dtype >= 0 && dtype != args[i].dtype) return VORTA_EINVAL;
    dtype = args[i].dtype;
    p.xcd_remap = 0;
    mp.seg[m] = p;
    mp.start[m] = (int)total;
    total += (int64_t)p.n_groups * p.blocks_per_group * p.n_heads * p.n_splits;
    if (total > 0x7fffffff) return VORTA_EINVAL;
    ++m;
  }
  if (m == 0) return VORTA_OK;
  for (int i = m; i <= MAX_SEGMENTS; ++i) mp.start[i] = (int)total;
  mp.n = m;
  hipStream_t st = (hipStream_t)hip_stream;
  if (dtype == VORTA_BF16) hipLaunchKernelGGL((attn_fwd_multi_kernel<__bf16>), dim3((unsigned)total), dim3(512), 0, st, mp);
  else hipLaunchKernelGGL((attn_fwd_multi_kernel<_Float16>), dim3((unsigned)total), dim3(512), 0, st, mp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  for (int i = 0; i < m; ++i) {
    const Params& p = mp.seg[i];
    if (p.n_splits > 1) {
      const int64_t items = (int64_t)p.n_heads * p.n_q;
      if (dtype == VORTA_BF16)
        hipLaunchKernelGGL((attn_combine_kernel<__bf16>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, p);
      else
        hipLaunchKernelGGL((attn_combine_kernel<_Float16>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, p);
      e = hipGetLastError();
      if (e != hipSuccess) return vorta_set_hip_error(e);
    }
  }
  return VORTA_OK;
}

extern "C" int vorta_attn_fwd(const vorta_attn_args* a, void* hip_stream) {
  Params p{};
  int block_rows = 0;
  int rc = fill_params(a, p, block_rows);
  if (rc != VORTA_OK) return rc;
  if (p.n_heads == 0 || p.n_groups == 0) return VORTA_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  // variant 0 (auto) = the software-pipelined kernel with LDS-DMA staging, for both workgroup sizes
  const bool pipe = a->variant != 1;
  if (a->dtype == VORTA_BF16)
    return block_rows == 256 ? launch<__bf16, 8>(p, st, pipe) : launch<__bf16, 4>(p, st, pipe);
  return block_rows == 256 ? launch<_Float16, 8>(p, st, pipe) : launch<_Float16, 4>(p, st, pipe);
}
