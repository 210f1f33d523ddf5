// Gather flash-attention forward with both contractions in fp8 (OCP e4m3) on v_mfma_f32_32x32x64_f8f6f4
// (include/vorta_hip.h: vorta_attn_fwd_fp8).  Same work decomposition, row tables, split-key mode and fused-grid
// form as the 16-bit kernel (attn_fwd.hip); what changes is everything between the tile in LDS and the MFMA:
//
//   * scores  S^T[kv][q] = K8 . Q8^T: one 32x32x64 MFMA covers 64 of the 128 channels (2 per 32-key tile, 4 per
//     64-key block, 64 cycles each = half the MFMA time of the bf16 kernel).  q8/k8 carry the softmax scale and
//     log2(e) (vorta_fp8_quantize_qkv), so the accumulator holds the score in the exp2 domain; it starts from
//     p_bias - m_run, so P' = exp2(acc) = P * 2^p_bias directly (P' <= 2^(p_bias+defer) <= 256 < 448 = e4m3 max);
//   * P' is packed to e4m3 (v_cvt_pk_fp8_f32, RNE) straight from the accumulator registers: a lane owns one query
//     and 32 of the block's 64 keys, which is exactly the B operand of one K=64 MFMA -- no lane movement;
//   * O^T[d][q] += V8^T . P'^T: the A operand (32 channels x 64 keys, the keys in the accumulator's row order) is
//     read from the row-major V tile with ds_read_b64_tr_b8 (8 keys x 16 channels per 16 lanes, hardware transpose);
//   * row sums: one more MFMA per block against a tile of ones gives sum_k P'[k][q] in every accumulator register
//     (the same rounded P' that multiplies V; no per-element adds, no half-wave exchange);
//   * epilogue: o[d] * v_descale[head][d] / rowsum -> bf16 / fp16.
//
// LDS images (rows of 128 bytes): K tile XOR-swizzled in 16-byte chunks, chunk ^= (row >> 1) & 7 (ds_read_b128 of 16
// rows hits 16 distinct 16-byte bank groups); V tile swizzled in 32-byte pairs, pair ^= ((row>>1)&1) | ((row>>3)&1)<<1
// (the 8 rows x 32 bytes one half-wave transposes land in 8 distinct bank groups).  Tiles arrive by LDS-DMA; the
// swizzles are applied on the source address.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

#include "attn_common.h"

namespace {
using namespace vorta_attn;

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(2))) int i32x2;

constexpr int ROWB8 = D;             // bytes per e4m3 row
constexpr int TILE8 = KVB * ROWB8;   // 8 KiB

struct Params8 {
  Params p;
  const float* v_descale; int64_t v_descale_sh;
  float p_bias;  // log2 bias of the packed probabilities
  float thr;     // p_bias + defer: offset scores above this move the reference point
  int lsum_valu; // row sums by VALU adds (experiment) instead of the ones-tile MFMA
};
struct MultiParams8 {
  Params8 seg[MAX_SEGMENTS];
  int start[MAX_SEGMENTS + 1];
  int n;
};

__device__ __forceinline__ int imax3(int a, int b, int c) { return max(max(a, b), c); }

__device__ __forceinline__ f32x16 mfma8(i32x8 a, i32x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);  // cbsz = blgp = 0: e4m3 x e4m3, no block scale
}

template <typename TO> struct OutT;
template <> struct OutT<__bf16> { using v4 = bf16x4; };
template <> struct OutT<_Float16> { using v4 = f16x4; };

// LDS: K ring of 2 tiles, V ring of 3 tiles (8 KiB each), one workgroup barrier per key block.  (Rings of 4 + 4 tiles with
// one barrier per TWO key blocks were built and measured at the same speed -- 11.25 vs 11.29 ms on the dense launch,
// DESIGN.md (f).1 -- and removed in round 3.)
// VORTA_DMA_SPLIT bit0 (default): the waves whose step starts with the matrix part request their tile pieces BEHIND it
// instead of right after the barrier -- they go from the barrier straight into their MFMAs (no address arithmetic, no
// request issue in the head of the segment) and the requests of the two roles no longer queue up together.  Same box,
// dense / table launch at S = 75 600, H = 8: 11.78 / 11.45 ms with every request at the top of the step (and a run-time
// "is this wave a loader" test), 10.85 / 10.63 ms now.  bit1: the other role requests behind its VALU part (slower; so
// is a V request issued a while after the K request).  The loader test has to be a compile-time fact: as a run-time
// branch it puts a control-flow join behind the requests, the compiler waits there for the table path's pending row-id
// load, and the queue order makes that a wait for the tile request before it (13.4 ms on the table launch).
// The 8-wave kernels only (the 4-wave ones have no roles).  The gain is the single-launch kernels' (-7 %: all heads dense,
// one launch per expert); the fused layer kernel had no SGPR-spill reloads in the head of its step and moves by -1 %.
#ifndef VORTA_DMA_SPLIT
#define VORTA_DMA_SPLIT 1
#endif
constexpr int K_SLOTS = 2, V_SLOTS = 3;
constexpr int SMEM8 = (K_SLOTS + V_SLOTS) * TILE8;

template <typename TO, int NW, bool KVTAB, bool LMFMA, bool SWAP = false>
__device__ __forceinline__ void attn8_body(const Params8& pp, char* __restrict__ smem, const int wg) {
  const Params& p = pp.p;
  using O4 = typename OutT<TO>::v4;
  constexpr int QB = NW * 32;
  // every wave is a loader.  (Four loader waves, 2 of a tile's 8 1-KiB pieces each, measured neutral: the barrier waits
  // shrink, the step stays at ~1 700 cycles -- the loop is bound by the SIMD's issue slots, profiles/r02_fp8_loop_trace.txt.)
  constexpr int LW = NW;      // loader waves
  constexpr int CH = 8 / LW;  // 1-KiB DMA pieces (8 tile rows) of one tile per loader wave
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

  // B operand of the score MFMAs: lane (query r32, half hh) holds channels [64 ks + 32 hh, +32) of its row
  i32x8 qf[2];
  {
    const char* qp = p.q + (int64_t)head * p.q_sh + my_row * p.q_ss + hh * 32;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const i32x4 lo = *(const i32x4*)(qp + ks * 64), hi = *(const i32x4*)(qp + ks * 64 + 16);
#pragma unroll
      for (int i = 0; i < 4; ++i) { qf[ks][i] = lo[i]; qf[ks][4 + i] = hi[i]; }
    }
  }

  // ---- loader: one wave instruction moves 1 KiB = 8 tile rows (8 lanes x 16 B per row), lane-linear in LDS ----
  const int32_t* kv_rows =
      p.kv_rows ? p.kv_rows + (int64_t)y * p.kv_rows_sh + (int64_t)grp * p.kv_rows_sg : nullptr;
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.k + (int64_t)head * p.k_sh), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.v + (int64_t)head * p.v_sh), 0, 0x7fffffff, 0x00020000);
  const int k_ss32 = (int)p.k_ss, v_ss32 = (int)p.v_ss;
  constexpr bool loader = true;   // a compile-time fact: no branch, no join around the requests (see VORTA_DMA_SPLIT)
  const int lwave = wave;
  int k_col[CH], v_col[CH];  // source byte offset inside the row for the chunk this lane lands in
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int row = 8 * (CH * lwave + i) + (lane >> 3);
    k_col[i] = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
    v_col[i] = ((lane & 7) ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1)) << 4;
  }
  int rowK[CH], rowV[CH];
#define ROWS_OF(dst_, blk_)                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) {                             \
    const int pos_ = min((blk_) * KVB + 8 * (CH * lwave + i_) + (lane >> 3), n_kv - 1); \
    if constexpr (KVTAB) dst_[i_] = kv_rows[pos_];                                \
    else dst_[i_] = p.kv_row_offset + pos_;                                       \
  }
#define DMA_K(slot_) _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(  \
      k_rsrc, (LDS_AS void*)(smem + (slot_) * TILE8 + (CH * lwave + i_) * 1024), 16,                               \
      (int)__umul24((unsigned)rowK[i_], (unsigned)k_ss32) + k_col[i_], 0, 0, 0);
#define DMA_V(slot_) _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(  \
      v_rsrc, (LDS_AS void*)(smem + (K_SLOTS + (slot_)) * TILE8 + (CH * lwave + i_) * 1024), 16,                   \
      (int)__umul24((unsigned)rowV[i_], (unsigned)v_ss32) + v_col[i_], 0, 0, 0);

  // ---- LDS read addresses ----
  // K fragment (A operand, rows = keys): lane (key r32 [+32], half hh), k-step ks: chunks 4 ks + 2 hh + {0,1}
  int k_rd[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int c = 0; c < 2; ++c) k_rd[ks][c] = r32 * ROWB8 + (((4 * ks + 2 * hh + c) ^ ((r32 >> 1) & 7)) << 4);
  // V^T fragment (A operand, rows = channels): 16-lane group g = (half hh, channel half dsub); lane pq of the group
  // addresses 8 bytes of key row  16 n + 4 hh + (tt & 3) + 8 (tt >> 2),  tt = pq >> 1,  at channel 32 dt + 16 dsub +
  // 8 (pq & 1), and receives channel 32 dt + 16 dsub + pq of the group's 8 key rows = k-slots 8 n .. 8 n + 7
  int v_rd[4];
  {
    const int dsub = (lane >> 4) & 1, pq = lane & 15, tt = pq >> 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      v_rd[dt] = K_SLOTS * TILE8 + (4 * hh + (tt & 3) + 8 * (tt >> 2)) * ROWB8 + ((dt ^ ((tt >> 1) & 3)) << 5) +
                 16 * dsub + 8 * (pq & 1);
  }

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  f32x16 lacc;  // LMFMA: every register = sum of P' over the keys so far (ones-tile MFMA); else lacc[0] = partial sum of this lane
#pragma unroll
  for (int i = 0; i < 16; ++i) lacc[i] = 0.f;
  // the ones tile of the row-sum MFMA in fp4 (e2m1 1.0 = 0b0010): the A operand then takes 4 registers instead of 8
  // (the instruction reads only the first 4 of the 8 it is given; B stays e4m3)
  i32x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = 0x22222222;
  asm volatile("" : "+v"(ones));
  // Online softmax in the exp2 domain.  m_run = reference point of the row (at most `defer` below its running max);
  // the score MFMAs start from minit = p_bias - m_run, so they deliver z - m_run + p_bias and P' = exp2 of that.
  float m_run = -1e30f;
  const float pbias = pp.p_bias, thr = pp.thr;
  f32x16 sA0, sA1, sB0, sB1;
  f32x16 minit;
#pragma unroll
  for (int i = 0; i < 16; ++i) minit[i] = 0.f;
  float mx_cur = -1e30f;  // row max of the current block's offset scores
  i32x8 pb_;              // packed probabilities of the block whose PV product is next (B operand of the PV MFMAs)
#pragma unroll
  for (int i = 0; i < 8; ++i) pb_[i] = 0;

#define KFRAG(dst_, slot_, t_, ks_)                                                \
  {                                                                               \
    const i32x4 lo_ = *(const i32x4*)(smem + (slot_) * TILE8 + (t_) * 32 * ROWB8 + k_rd[ks_][0]); \
    const i32x4 hi_ = *(const i32x4*)(smem + (slot_) * TILE8 + (t_) * 32 * ROWB8 + k_rd[ks_][1]); \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { dst_[e_] = lo_[e_]; dst_[4 + e_] = hi_[e_]; } \
  }
#define KFRAGS0(slot_)                                                            \
  i32x8 kf00_, kf10_;                                                             \
  KFRAG(kf00_, slot_, 0, 0)                                                       \
  KFRAG(kf10_, slot_, 1, 0)
#define KFRAGS1(slot_)                                                            \
  i32x8 kf01_, kf11_;                                                             \
  KFRAG(kf01_, slot_, 0, 1)                                                       \
  KFRAG(kf11_, slot_, 1, 1)
#define QK_FRAGS(d0_, d1_)                                                        \
  {                                                                               \
    d0_ = mfma8(kf00_, qf[0], minit);                                             \
    d1_ = mfma8(kf10_, qf[0], minit);                                             \
    d0_ = mfma8(kf01_, qf[1], d0_);                                               \
    d1_ = mfma8(kf11_, qf[1], d1_);                                               \
  }
#define ONES_MFMA_(a_, b_, c_) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_, b_, c_, 4, 0, 0, 0, 0, 0)
#define PVA_(dt_) vf_[dt_]
#define PVKEEP_()
#define VFRAG(dt_, slot_)                                                         \
  _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) {                              \
    const i32x2 t_ = __builtin_amdgcn_ds_read_tr8_b64_v2i32(                      \
        (LDS_AS i32x2*)(smem + (slot_) * TILE8 + v_rd[dt_] + n_ * 16 * ROWB8));   \
    vf_[dt_][2 * n_] = t_[0]; vf_[dt_][2 * n_ + 1] = t_[1];                       \
  }
  // the fragments of the first two channel tiles are requested ahead of the part (before the rare branches), those of
  // the other two under the first two MFMAs -- 16 live fragment registers instead of 32
#define VFRAGS_HEAD(slot_) i32x8 vf_[4]; VFRAG(0, slot_) VFRAG(1, slot_)
#define PV_FRAGS(slot_)                                                           \
  {                                                                               \
    VFRAG(2, slot_)                                                               \
    VFRAG(3, slot_)                                                               \
    _Pragma("unroll") for (int dt_ = 0; dt_ < 4; ++dt_) o[dt_] = mfma8(PVA_(dt_), pb_, o[dt_]); \
    PVKEEP_()                                                                     \
    if constexpr (LMFMA) lacc = ONES_MFMA_(ones, pb_, lacc);                      \
  }
#define EXP_BLOCK(c0_, c1_)                                                       \
  {                                                                               \
    float lsum_ = 0.f;                                                            \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                           \
      c0_[i_] = __builtin_amdgcn_exp2f(c0_[i_]);                                  \
      c1_[i_] = __builtin_amdgcn_exp2f(c1_[i_]);                                  \
      if constexpr (!LMFMA) lsum_ += c0_[i_] + c1_[i_];                           \
    }                                                                             \
    if constexpr (!LMFMA) lacc[0] += lsum_;                                       \
  }
  // v_cvt_pk_fp8_f32 writes half of its destination and keeps the other half: feed it the stale word of the previous
  // block instead of a zero (the second convert overwrites the rest)
#define PACK_BLOCK(c0_, c1_)                                                      \
  _Pragma("unroll") for (int w_ = 0; w_ < 4; ++w_) {                              \
    pb_[w_] = __builtin_amdgcn_cvt_pk_fp8_f32(c0_[4 * w_], c0_[4 * w_ + 1], pb_[w_], false); \
    pb_[w_] = __builtin_amdgcn_cvt_pk_fp8_f32(c0_[4 * w_ + 2], c0_[4 * w_ + 3], pb_[w_], true); \
    pb_[4 + w_] = __builtin_amdgcn_cvt_pk_fp8_f32(c1_[4 * w_], c1_[4 * w_ + 1], pb_[4 + w_], false); \
    pb_[4 + w_] = __builtin_amdgcn_cvt_pk_fp8_f32(c1_[4 * w_ + 2], c1_[4 * w_ + 3], pb_[4 + w_], true); \
  }
#define ROW_MAX(dst_, a_, b_)                                                      \
  {                                                                               \
    float mx_ = a_[0];                                                            \
    _Pragma("unroll") for (int i_ = 1; i_ < 16; ++i_) mx_ = fmaxf(mx_, a_[i_]);   \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) mx_ = fmaxf(mx_, b_[i_]);   \
    dst_ = half_max(mx_);                                                         \
  }
  // The loop only asks two things of a block's row max: "is it above thr (> 0)?" and, if so, its value.  Both are
  // answered by a SIGNED-INTEGER max over the float bit patterns (order-preserving for the non-negative floats, and
  // any negative result reads as "not above"): v_max3_i32 needs no canonicalising v_max x,x of the MFMA outputs,
  // and two chains halve the dependent latency.  (-inf of masked keys is a negative integer; there are no NaNs.)
#define ROW_MAX_POS(dst_, a_, b_)                                                  \
  {                                                                               \
    int m0_ = imax3(__float_as_int(a_[0]), __float_as_int(a_[1]), __float_as_int(a_[2])); \
    int m1_ = imax3(__float_as_int(b_[0]), __float_as_int(b_[1]), __float_as_int(b_[2])); \
    _Pragma("unroll") for (int i_ = 3; i_ < 15; i_ += 2) {                        \
      m0_ = imax3(m0_, __float_as_int(a_[i_]), __float_as_int(a_[i_ + 1]));       \
      m1_ = imax3(m1_, __float_as_int(b_[i_]), __float_as_int(b_[i_ + 1]));       \
    }                                                                             \
    m0_ = imax3(m0_, __float_as_int(a_[15]), __float_as_int(b_[15]));             \
    m0_ = max(m0_, m1_);                                                          \
    auto r_ = __builtin_amdgcn_permlane32_swap((unsigned)m0_, (unsigned)m0_, false, false); \
    dst_ = __int_as_float(max((int)r_[0], (int)r_[1]));                           \
  }
#define MASK_TAIL(c0_, c1_, jabs_)                                                \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                             \
    const int row_ = (i_ & 3) + 8 * (i_ >> 2) + 4 * hh;                           \
    if ((jabs_) * KVB + row_ >= n_kv) c0_[i_] = -INFINITY;                        \
    if ((jabs_) * KVB + 32 + row_ >= n_kv) c1_[i_] = -INFINITY;                   \
  }
  // requests of a step: K(j+2) into the slot K(j) left, V(j+1) into the slot V(j-2) left; both are read in the next
  // step (V(j+1) by role Y), so the end-of-step wait covers both
#define STAGE_DMA(kw_, vw_, jabs_)                                                \
  if (loader) {                                                                   \
    DMA_K(kw_)                                                                    \
    DMA_V(vw_)                                                                    \
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) rowV[i_] = rowK[i_];        \
    ROWS_OF(rowK, (jabs_) + 3)                                                    \
  }                                                                               \
  __builtin_amdgcn_sched_barrier(0);
#define STEP_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#ifndef VORTA_SCHED8
#define VORTA_SCHED8 1
#endif
  // Issue-order recipe (sched_group_barrier: 0x008 MFMA, 0x100 DS read) of the PV half of the matrix part: the reads of
  // V channel tiles 0,1 first (claimed by a group of their own: without it the scheduler counts them as the groups behind
  // the first MFMAs, allocates ONE register set to tiles 0, 1, 2 in turn, and every PV MFMA waits a full LDS round trip
  // for reads issued just before it), then 5 MFMAs with the fragment reads of the later ones (V channel tiles 2,3;
  // k-step 0 of the next K block) under the earlier ones; the score half follows (its k-step-1 fragments are requested
  // first).  Each PV MFMA then waits with two fragments still in flight (lgkmcnt(4)): +1.5-2 % on the dense launch, +1 %
  // on the Wan-14B step, Hunyuan-129f unchanged (profiles/r03_fp8_loop_experiments.txt).
#if VORTA_SCHED8 == 1
#define SG_(mask_, n_) __builtin_amdgcn_sched_group_barrier(mask_, n_, 0);
#define SCHED_M()                                                                 \
  SG_(0x100, 8)                                                                   \
  SG_(0x008, 1) SG_(0x100, 4)                                                     \
  SG_(0x008, 1) SG_(0x100, 4)                                                     \
  SG_(0x008, 1) SG_(0x100, 2)                                                     \
  SG_(0x008, 1) SG_(0x100, 2)                                                     \
  SG_(0x008, 1)
#else
#define SCHED_M()
#endif
#ifndef VORTA_PRIO8
#define VORTA_PRIO8 1  /* s_setprio around the matrix part (the partner wave is in its VALU part then) */
#endif
#if VORTA_PRIO8 == 1
#define PRIO_HI() __builtin_amdgcn_s_setprio(2);
#define PRIO_LO() __builtin_amdgcn_s_setprio(0);
#else
#define PRIO_HI()
#define PRIO_LO()
#endif
  // The two waves of a SIMD (wave w and w + NW/2 of the workgroup) take turns on its pipes: while one runs its matrix
  // part -- 9 MFMAs back to back: O += V^T P^T and the row sums of one block, the scores of a later one -- the other
  // runs its VALU part -- the 32 exp2 and 16 packs of a block (quarter-rate ops: ~9 cycles of SIMD issue each,
  // tools/probe_overlap.hip) and a row max.  Both parts take ~600 cycles, so the matrix pipe and the VALU are busy
  // together all the time.  (With both waves in the same part at the same time the SIMD's issue is by age: the older
  // wave takes the matrix pipe, the younger waits, and their VALU parts end up exposed one after the other --
  // measured 48 % MFMA utilisation, every VALU instruction removed from the loop came off the run time in full.)
  //   step j, waves < NW/2 :   matrix(j)  then  valu(block j; max of block j+1)
  //   step j, waves >= NW/2:   valu(block j-1; max of block j)  then  matrix(j)
  //   matrix(j) = PV(j-1) + row sums, [mask the tail of block j], [move the reference point for block j], QK(j+1)
  // The matrix part is the same code for both roles; the VALU part is the same code on swapped score buffers.  Both
  // roles see the same reference points (the decision for block j looks at max(j) against the reference after block
  // j-1) and run the same barriers.
  //   kw_/kr_: K slots written (K(j+2)) / read (K(j+1));  vw_/vr_: V slots written (V(j+1)) / read (V(j-1)).
  //   c = scores of block j (offset by the reference point), n = scores of block j+1 (written by the matrix part).
#define VALU_PART(e0_, e1_, m0_, m1_)                                             \
  {                                                                               \
    EXP_BLOCK(e0_, e1_)                                                           \
    PACK_BLOCK(e0_, e1_)                                                          \
    ROW_MAX_POS(mx_cur, m0_, m1_)                                                 \
  }
#define MATRIX_PART(c0_, c1_, n0_, n1_, kr_, vr_, jabs_)                          \
  {                                                                               \
    PRIO_HI()                                                                     \
    VFRAGS_HEAD(vr_)                                                              \
    PV_FRAGS(vr_)                                                                 \
    KFRAGS0(kr_)                                                                  \
    SCHED_M()                                                                     \
    /* last, partial key block: mask its tail, exact row max (once per workgroup) */ \
    if ((jabs_) * KVB + KVB > n_kv) { MASK_TAIL(c0_, c1_, jabs_) ROW_MAX(mx_cur, c0_, c1_) } \
    /* deferred rescale: the reference point moves only when some row of the wave outgrew it by more than `defer`; */ \
    /* O and the row sums follow AFTER block j-1 went in at the old reference                                        */ \
    if (!__all(mx_cur <= thr)) {                                                  \
      const float g_ = fmaxf(mx_cur - pbias, 0.f);                                \
      const float alpha_ = __builtin_amdgcn_exp2f(-g_);                           \
      m_run += g_;                                                                \
      _Pragma("unroll") for (int dt_ = 0; dt_ < 4; ++dt_)                         \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) o[dt_][i_] *= alpha_;   \
      if constexpr (LMFMA) { _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) lacc[i_] *= alpha_; } \
      else lacc[0] *= alpha_;                                                     \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { c0_[i_] -= g_; c1_[i_] -= g_; minit[i_] = pbias - m_run; } \
      asm volatile("" : "+v"(minit));                                             \
    }                                                                             \
    KFRAGS1(kr_)                                                                  \
    QK_FRAGS(n0_, n1_)                                                            \
    PRIO_LO()                                                                     \
  }
  // Diagnostic builds only (suffixed libraries, vorta_amd/build.py; never the product): -DVORTA_FP8_DIAG pulls in the
  // in-loop cycle stamps (-DVORTA_TRACE8=i, tools/trace_fp8.py) and the wrong-result timing ablations (-DVORTA_DIAG_*),
  // which re-define the macros above.
#define TR_(i_)
#define TR_FLUSH_()
#define ROLE_Y_ (NW == 8 && (SWAP ? wave < NW / 2 : wave >= NW / 2))
#ifdef VORTA_FP8_DIAG
#include "attn_fwd_fp8_diag.inc"
#endif
#define STEP_(c0_, c1_, n0_, n1_, kw_, kr_, vw_, vr_, jabs_, sync_)                \
  {                                                                               \
    TR_(0)                                                                        \
    /* VORTA_DMA_SPLIT bit0: role X requests behind its matrix part; bit1: role Y behind its VALU part; */ \
    {                                                                             \
    if (!wave_active || NW != 8 || (role_y ? !(VORTA_DMA_SPLIT & 2) : !(VORTA_DMA_SPLIT & 1))) { STAGE_DMA(kw_, vw_, jabs_) } \
    if (wave_active) {                                                            \
      if (role_y) VALU_PART(n0_, n1_, c0_, c1_)                                   \
      if ((VORTA_DMA_SPLIT & 2) && NW == 8 && role_y) { STAGE_DMA(kw_, vw_, jabs_) } \
      __builtin_amdgcn_sched_barrier(0);                                          \
      TR_(1)                                                                      \
      MATRIX_PART(c0_, c1_, n0_, n1_, kr_, vr_, jabs_)                            \
      __builtin_amdgcn_sched_barrier(0);                                          \
      TR_(2)                                                                      \
      if ((VORTA_DMA_SPLIT & 1) && NW == 8 && !role_y) { STAGE_DMA(kw_, vw_, jabs_) } \
      if (!role_y) VALU_PART(c0_, c1_, n0_, n1_)                                  \
    }                                                                             \
    }                                                                             \
    if constexpr (sync_) { STEP_SYNC() }                                          \
  }
#define STEP(c0_, c1_, n0_, n1_, kw_, kr_, vw_, vr_, jabs_) STEP_(c0_, c1_, n0_, n1_, kw_, kr_, vw_, vr_, jabs_, true)

  const int nsteps = blk1 - blk0;
  // which half of the workgroup starts its steps with the VALU part (wave-uniform).  SWAP = the earlier-dispatched half:
  // measured per kernel -- the table-free body of the fused layer kernel runs 2.5-3.6 % faster that way (dense-only
  // fused launch 11.13 -> 10.86 ms), its table body 0-1 % slower, the single-launch kernels 0.5-1 % slower
  const bool role_y = ROLE_Y_;
  if (nsteps > 0) {
    // ---- prologue: K(0), V(0), K(1); the scores of block 0 fix the reference point ----
    if (loader) {
      ROWS_OF(rowK, blk0)
      _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) rowV[i_] = rowK[i_];
      DMA_K(0)
      DMA_V(0)
      ROWS_OF(rowK, blk0 + 1)
      DMA_K(1)
      _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) rowV[i_] = rowK[i_];
      ROWS_OF(rowK, blk0 + 2)
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave_active) {
      KFRAGS0(0)
      KFRAGS1(0)
      QK_FRAGS(sA0, sA1)  // seed 0: plain scores of the first block
      if (blk0 * KVB + KVB > n_kv) { MASK_TAIL(sA0, sA1, blk0) }
      ROW_MAX(mx_cur, sA0, sA1)
      // the first block fixes the reference point at its true row max (it always holds a valid key)
      m_run = mx_cur;
#pragma unroll
      for (int i = 0; i < 16; ++i) { sA0[i] += pbias - m_run; sA1[i] += pbias - m_run; minit[i] = pbias - m_run; }
      asm volatile("" : "+v"(minit));
    }
    __syncthreads();  // every wave has read K(0) before its slot is overwritten
    {  // step 0: no PV yet -- the scores of block 1, then (first role) the VALU part of block 0
      TR_(0)
      STAGE_DMA(0, 1, blk0)
      if (wave_active) {
        TR_(1)
        KFRAGS0(1)
        KFRAGS1(1)
        QK_FRAGS(sB0, sB1)
        __builtin_amdgcn_sched_barrier(0);
        TR_(2)
        if (!role_y) VALU_PART(sA0, sA1, sB0, sB1)
      }
      STEP_SYNC()
    }
    // K slots cycle with period 2, V slots with period 3, score roles with period 2: unrolled by 6
    for (int jj = 1; jj < nsteps; jj += 6) {
      STEP(sB0, sB1, sA0, sA1, 1, 0, 2, 0, blk0 + jj)
      if (jj + 1 >= nsteps) break;
      STEP(sA0, sA1, sB0, sB1, 0, 1, 0, 1, blk0 + jj + 1)
      if (jj + 2 >= nsteps) break;
      STEP(sB0, sB1, sA0, sA1, 1, 0, 1, 2, blk0 + jj + 2)
      if (jj + 3 >= nsteps) break;
      STEP(sA0, sA1, sB0, sB1, 0, 1, 2, 0, blk0 + jj + 3)
      if (jj + 4 >= nsteps) break;
      STEP(sB0, sB1, sA0, sA1, 1, 0, 0, 1, blk0 + jj + 4)
      if (jj + 5 >= nsteps) break;
      STEP(sA0, sA1, sB0, sB1, 0, 1, 1, 2, blk0 + jj + 5)
    }
    // ---- drain: the second role still owes the VALU part of the last block; then PV of the last block ----
    if (wave_active) {
      if (role_y) {
        if ((nsteps - 1) & 1) { EXP_BLOCK(sB0, sB1) PACK_BLOCK(sB0, sB1) }
        else { EXP_BLOCK(sA0, sA1) PACK_BLOCK(sA0, sA1) }
      }
      const int vs = (nsteps - 1) % V_SLOTS;
      VFRAGS_HEAD(vs)
      PV_FRAGS(vs)
    }
  }
#undef KFRAG
#undef KFRAGS0
#undef KFRAGS1
#undef QK_FRAGS
#undef VFRAG
#undef VFRAGS_HEAD
#undef PV_FRAGS
#undef EXP_BLOCK
#undef PACK_BLOCK
#undef ROW_MAX
#undef ROW_MAX_POS
#undef MASK_TAIL
#undef STEP
#undef VALU_PART
#undef MATRIX_PART
#undef PRIO_HI
#undef PRIO_LO
#undef STAGE_DMA
#undef STEP_SYNC
#undef ROWS_OF
#undef DMA_K
#undef DMA_V
#undef SCHED_M
#undef SG_
  TR_FLUSH_()
#undef TR_
#undef TR_FLUSH_
#undef ROLE_Y_

  if (!wave_active) return;
  // ---------------- epilogue ----------------
  const float l_tot = LMFMA ? lacc[0] : half_sum(lacc[0]);
  if (p.n_splits > 1) {
    // unnormalised partials (both carry the 2^p_bias factor): ws_o[y][sp][pos][d], ws_ml[y][sp][pos][2]
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
        p.ws_ml[slot * 2] = m_run;
        p.ws_ml[slot * 2 + 1] = l_tot;
      }
    }
    return;
  }
  if (!row_ok) return;
  const float inv = (my_p < q_valid && l_tot > 0.f) ? 1.f / l_tot : 0.f;
  const float* vd = pp.v_descale + (int64_t)head * pp.v_descale_sh + 4 * hh;
  uint2 packed[16];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const f32x4 s = *(const f32x4*)(vd + 32 * dt + 8 * rg);
      O4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (TO)(o[dt][4 * rg + j] * (inv * s[j]));
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

template <typename TO, int NW, bool KVTAB, bool LMFMA>
__global__ __launch_bounds__(NW * 64, 2) void attn8_kernel(const Params8 pp) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char smem[SMEM8];
  const int wg = live_order(pp.p, blockIdx.x, gridDim.x, pp.p.xcd_remap);  // XCD-aware order over the live workgroups
  attn8_body<TO, NW, KVTAB, LMFMA>(pp, smem, wg);
#endif
}

template <typename TO>
__global__ __launch_bounds__(512, 2) void attn8_multi_kernel(const MultiParams8 mp) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char smem[SMEM8];
  const int b = blockIdx.x;
  int s = 0;
#pragma unroll
  for (int i = 1; i < MAX_SEGMENTS; ++i) s += (i < mp.n && b >= mp.start[i]) ? 1 : 0;
  const Params8& pp = mp.seg[s];
  const int wg = live_order(pp.p, b - mp.start[s], mp.start[s + 1] - mp.start[s], true);
#ifndef VORTA_MULTI_SWAP
#define VORTA_MULTI_SWAP 1  /* see attn8_body: which half of the workgroup starts its steps with the VALU part; bit0 the body
                               without row tables, bit1 the body with them.  One box, fp8 step in ms, 0 / 1 / 2 / 3:
                               Wan-14B 2 156 / 2 097 / 2 169 / 2 113, Hunyuan-129f 2 285 / 2 300 / 2 300 / 2 328 */
#endif
  if (pp.p.kv_rows) attn8_body<TO, 8, true, true, (VORTA_MULTI_SWAP & 2) != 0>(pp, smem, wg);
  else attn8_body<TO, 8, false, true, (VORTA_MULTI_SWAP & 1) != 0>(pp, smem, wg);
#endif
}

// Merge the split-key partials: one wave per (head slot, query position).
template <typename TO>
__global__ __launch_bounds__(256) void attn8_combine_kernel(const Params8 pp) {
  const Params& p = pp.p;
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
    const float w = __builtin_amdgcn_exp2f(p.ws_ml[slot * 2] - m);
    l += w * p.ws_ml[slot * 2 + 1];
    const float2 v = *(const float2*)(p.ws_o + slot * D + lane * 2);
    acc0 += w * v.x;
    acc1 += w * v.y;
  }
  const int q_valid = p.q_valid_dev ? min(*p.q_valid_dev, p.q_valid) : p.q_valid;
  const float inv = (pos < q_valid && l > 0.f) ? 1.f / l : 0.f;
  const float2 sd = *(const float2*)(pp.v_descale + (int64_t)head * pp.v_descale_sh + lane * 2);
  const int32_t* q_rows = p.q_rows ? p.q_rows + (int64_t)y * p.q_rows_sh : nullptr;
  const int64_t row = q_rows ? (int64_t)q_rows[pos] : (int64_t)(p.q_row_offset + pos);
  TO pair[2] = {(TO)(acc0 * inv * sd.x), (TO)(acc1 * inv * sd.y)};
  char* ob = p.o + (int64_t)head * p.o_sh + lane * 4;
  *(uint32_t*)(ob + row * p.o_ss) = *(uint32_t*)pair;
  if (p.dup_rows && pos < p.n_dup_pos) {
    const int32_t* dr = p.dup_rows + (int64_t)y * p.dup_rows_sh + (int64_t)pos * p.n_dup;
    for (int i = 0; i < p.n_dup; ++i) *(uint32_t*)(ob + (int64_t)dr[i] * p.o_ss) = *(uint32_t*)pair;
  }
}

int fill8(const vorta_attn_args* a, const vorta_attn_fp8_ext* ext, Params8& pp, int& block_rows) {
  if (!ext || ext->struct_size != sizeof(vorta_attn_fp8_ext)) return VORTA_EINVAL;
  if (ext->out_dtype != VORTA_BF16 && ext->out_dtype != VORTA_FP16) return VORTA_EUNSUPPORTED;
  int rc = fill_params(a, pp.p, block_rows, 1);
  if (rc != VORTA_OK) return rc;
  if (pp.p.n_heads == 0 || pp.p.n_groups == 0) return VORTA_OK;
  if (a->variant == 1) return VORTA_EUNSUPPORTED;  // only the pipelined LDS-DMA body exists in fp8
  if (!ext->v_descale || ext->v_descale_stride_h < D) return VORTA_EINVAL;
  const float pb = ext->p_bias != 0.f ? ext->p_bias : 5.f;
  const float df = ext->defer != 0.f ? ext->defer : 3.f;
  if (!(pb >= 0.f) || !(df > 0.f) || pb + df > 8.f) return VORTA_EINVAL;  // P' <= 2^(p_bias+defer) must stay below 448
  pp.v_descale = ext->v_descale;
  pp.v_descale_sh = ext->v_descale_stride_h;
  pp.p_bias = pb;
  pp.thr = pb + df;
  pp.lsum_valu = ext->flags & 1;
  return VORTA_OK;
}

template <typename TO>
int launch8(const Params8& pp, int block_rows, hipStream_t st) {
  const Params& p = pp.p;
  const int64_t total = (int64_t)p.n_groups * p.blocks_per_group * p.n_heads * p.n_splits;
  if (total <= 0) return VORTA_OK;
  if (total > 0x7fffffff) return VORTA_EINVAL;
  const dim3 g((unsigned)total);
#define L8(NW_, TAB_, LM_) hipLaunchKernelGGL((attn8_kernel<TO, NW_, TAB_, LM_>), g, dim3(NW_ * 64), 0, st, pp)
  if (block_rows == 256) {
    if (pp.lsum_valu) { if (p.kv_rows) L8(8, true, false); else L8(8, false, false); }
    else { if (p.kv_rows) L8(8, true, true); else L8(8, false, true); }
  } else {
    if (pp.lsum_valu) { if (p.kv_rows) L8(4, true, false); else L8(4, false, false); }
    else { if (p.kv_rows) L8(4, true, true); else L8(4, false, true); }
  }
#undef L8
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  if (p.n_splits > 1) {
    const int64_t items = (int64_t)p.n_heads * p.n_q;
    hipLaunchKernelGGL((attn8_combine_kernel<TO>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, pp);
    e = hipGetLastError();
    if (e != hipSuccess) return vorta_set_hip_error(e);
  }
  return VORTA_OK;
}

}  // namespace

extern "C" int vorta_attn_fwd_fp8(const vorta_attn_args* a, const vorta_attn_fp8_ext* ext, void* hip_stream) {
  if (ext && ext->struct_size == sizeof(vorta_attn_fp8_ext) && (ext->flags & 2)) return mx_fwd(a, ext, hip_stream);
  Params8 pp{};
  int block_rows = 0;
  int rc = fill8(a, ext, pp, block_rows);
  if (rc != VORTA_OK) return rc;
  if (pp.p.n_heads == 0 || pp.p.n_groups == 0) return VORTA_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  return ext->out_dtype == VORTA_BF16 ? launch8<__bf16>(pp, block_rows, st) : launch8<_Float16>(pp, block_rows, st);
}

extern "C" int vorta_attn_fwd_batch_fp8(const vorta_attn_args* args, const vorta_attn_fp8_ext* ext, int32_t n,
                                        void* hip_stream) {
  if (!args || !ext || n < 0 || n > MAX_SEGMENTS) return VORTA_EINVAL;
  if (ext->struct_size == sizeof(vorta_attn_fp8_ext) && (ext->flags & 2)) return mx_fwd_batch(args, ext, n, hip_stream);
  MultiParams8 mp{};
  int64_t total = 0;
  int m = 0;
  for (int i = 0; i < n; ++i) {
    Params8 pp{};
    int block_rows = 0;
    int rc = fill8(&args[i], ext, pp, block_rows);
    if (rc != VORTA_OK) return rc;
    if (pp.p.n_heads == 0 || pp.p.n_groups == 0) continue;
    if (block_rows != 256 || pp.lsum_valu) return VORTA_EUNSUPPORTED;  // only 256-row launches share a grid
    pp.p.xcd_remap = 0;
    mp.seg[m] = pp;
    mp.start[m] = (int)total;
    total += (int64_t)pp.p.n_groups * pp.p.blocks_per_group * pp.p.n_heads * pp.p.n_splits;
    if (total > 0x7fffffff) return VORTA_EINVAL;
    ++m;
  }
  if (m == 0) return VORTA_OK;
  for (int i = m; i <= MAX_SEGMENTS; ++i) mp.start[i] = (int)total;
  mp.n = m;
  hipStream_t st = (hipStream_t)hip_stream;
  const bool bf = ext->out_dtype == VORTA_BF16;
  if (bf) hipLaunchKernelGGL((attn8_multi_kernel<__bf16>), dim3((unsigned)total), dim3(512), 0, st, mp);
  else hipLaunchKernelGGL((attn8_multi_kernel<_Float16>), dim3((unsigned)total), dim3(512), 0, st, mp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  for (int i = 0; i < m; ++i) {
    const Params8& pp = mp.seg[i];
    if (pp.p.n_splits > 1) {
      const int64_t items = (int64_t)pp.p.n_heads * pp.p.n_q;
      if (bf) hipLaunchKernelGGL((attn8_combine_kernel<__bf16>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, pp);
      else hipLaunchKernelGGL((attn8_combine_kernel<_Float16>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, pp);
      e = hipGetLastError();
      if (e != hipSuccess) return vorta_set_hip_error(e);
    }
  }
  return VORTA_OK;
}
