// Gather flash-attention forward, MIXED precision: scores in 16 bits, P V in e4m3 (include/vorta_hip.h:
// vorta_attn_fwd_fp8 with ext->flags bit1).
//
// Why: the all-e4m3 kernel (attn_fwd_fp8.hip) is off by 3.7 % of the root sum of squares of a score's 128 products, an
// ABSOLUTE logit error that grows with the logits -- 40 dB against the 16-bit kernels where the softmax is flat, 36-38 dB on
// peaked logits, 21-32 dB when a few outlier channels carry them (tools/dbg/qk_format_study.py).  The
// scores are where the mantissa is needed; P (in [0, 1] after the row max) and V (scaled per channel) tolerate e4m3:
// 16-bit q k^T with e4m3 P, V holds 42-69 dB on every input family.
//
// Structure: the 16-bit kernel's operands (attn_fwd.hip: K tile in 16 bits, XOR-swizzled, LDS-DMA staged, softmax folded into the
// score MFMA's seed) in the e4m3 kernel's step (attn_fwd_fp8.hip: the two waves of a SIMD alternate between a matrix part and a
// VALU part -- round 5; until then one dataflow per wave, as attn_fwd.hip) and the e4m3 kernel from the probabilities on: the accumulator starts from p_bias - m_run, P' = exp2(acc) = P 2^p_bias is packed
// to e4m3 straight from the accumulator registers (a lane owns one query and 32 of the block's 64 keys = the B operand of
// ONE K = 64 MFMA), O^T += V8^T P'^T on v_mfma_f32_32x32x64_f8f6f4 with the V tile (rows of 128 bytes) read through
// ds_read_b64_tr_b8, row sums from one more MFMA against a tile of ones, v_descale in the epilogue.  Per wave and 64-key
// block: 16 MFMAs of 32 cycles + 5 of 64 = 832 pipe cycles (1 024 in 16 bits, 576 in e4m3), 33 v_add_f32 fewer than the
// 16-bit loop, a V tile of half the bytes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

#include "attn_common.h"

namespace {
using namespace vorta_attn;

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) int i32x2;

constexpr int ROWB8 = D;             // bytes per e4m3 row
constexpr int TILE8 = KVB * ROWB8;   // 8 KiB
constexpr int NSMX = 2;              // ring depth of the K (16 KiB) and V (8 KiB) tiles
constexpr int SMEM_MX = NSMX * (TILE_BYTES + TILE8);  // 48 KiB

struct ParamsMx {
  Params p;
  const float* v_descale; int64_t v_descale_sh;
  float p_bias;  // log2 bias of the probabilities (P' = 2^(z - reference + p_bias))
  float etrig;   // binades: a key tile whose block exponent exceeds this moves its row's reference point (rare)
};
struct MultiParamsMx {
  ParamsMx seg[MAX_SEGMENTS];
  int start[MAX_SEGMENTS + 1];
  int n;
};

// O^T += V8^T P8^T with block scales on the B operand, as attn_fwd_i8.hip (round 5): scale block s of a column = bytes 16 s ...
// 16 s + 15 of both lanes of the column = one query row x the 32 keys of key tile s; its E8M0 byte is read from the lane of half s
constexpr int SC_ONE = 127;
__device__ __forceinline__ f32x16 mfma8(i32x8 a, i32x8 b, f32x16 c, int sb) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, SC_ONE, 0, sb);  // cbsz = blgp = 0: e4m3 x e4m3
}

template <typename T, int NW, bool KVTAB, int NS>
__device__ __forceinline__ void attn_mx_body(const ParamsMx& pp, char* __restrict__ smem, const int wg) {
  // NS = depth of the K ring (16 KiB tiles, 16-bit rows) and of the V ring (8 KiB tiles, e4m3 rows): K(j+2) / V(j) are requested at
  // the top of step j, one step before the step that reads them
  static_assert(NS == 2, "ring depth");
  const Params& p = pp.p;
  constexpr int VBASE = NS * TILE_BYTES;  // the V ring sits behind the K ring
  using V8 = typename MF<T>::v8;
  using V4 = typename MF<T>::v4;
  constexpr int NT = NW * 64;
  constexpr int QB = NW * 32;
  constexpr int CH = (KVB * 16) / NT;  // 16-byte chunks of one tile per thread (2 or 4)
  constexpr int ROWSTEP = NT / 16;     // rows between a thread's consecutive chunks
#ifndef VORTA_MX_KPRE
#define VORTA_MX_KPRE 1
#endif
  // k-steps of K fragments read ahead of the softmax head: one (two, the 16-bit kernel's depth, spill 8 VGPRs in the fused
  // kernel here -- the e4m3 row-sum accumulator and ones tile take 24 registers the 16-bit loop does not have); the
  // 128-row body with a key table keeps two row ids per lane on top of that: none there
  constexpr int KPRE = (NW == 4 && KVTAB) ? 0 : VORTA_MX_KPRE;
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
  // V tile: rows of 128 bytes, one wave instruction = 1 KiB = 8 tile rows (8 lanes x 16 B per row): 8 pieces per tile
  constexpr int CHV = 8 / NW;
  int k_col[CH], v_col[CHV];  // source byte offset inside the row for this lane's chunk
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int row = 4 * (CH * wave + i) + (lane >> 4);
    k_col[i] = ((lane & 15) ^ (row & 15)) << 4;
  }
#pragma unroll
  for (int i = 0; i < CHV; ++i) {
    const int row = 8 * (CHV * wave + i) + (lane >> 3);
    v_col[i] = ((lane & 7) ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1)) << 4;
  }
  int rowK[CH], rowV[CHV];  // rows of the next K block / next V block to fetch (each in its own piece-to-row map)
#define ROWS_OF(dst_, blk_)                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) {                             \
    const int pos_ = min((blk_) * KVB + 4 * (CH * wave + i_) + (lane >> 4), n_kv - 1); \
    if constexpr (KVTAB) dst_[i_] = kv_rows[pos_];                                \
    else dst_[i_] = p.kv_row_offset + pos_;                                       \
  }
#define ROWS_OF_V(dst_, blk_)                                                     \
  _Pragma("unroll") for (int i_ = 0; i_ < CHV; ++i_) {                            \
    const int pos_ = min((blk_) * KVB + 8 * (CHV * wave + i_) + (lane >> 3), n_kv - 1); \
    if constexpr (KVTAB) dst_[i_] = kv_rows[pos_];                                \
    else dst_[i_] = p.kv_row_offset + pos_;                                       \
  }
#define DMA_K(par_) _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(   \
      k_rsrc, (LDS_AS void*)(smem + (par_) * TILE_BYTES + (CH * wave + i_) * 1024), 16,                            \
      (int)__umul24((unsigned)rowK[i_], (unsigned)k_ss32) + k_col[i_], 0, 0, 0);
#define DMA_V(par_) _Pragma("unroll") for (int i_ = 0; i_ < CHV; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(  \
      v_rsrc, (LDS_AS void*)(smem + VBASE + (par_) * TILE8 + (CHV * wave + i_) * 1024), 16,                        \
      (int)__umul24((unsigned)rowV[i_], (unsigned)v_ss32) + v_col[i_], 0, 0, 0);

  // ---- LDS read addresses ----
  int k_rd[8];
#pragma unroll
  for (int ks = 0; ks < 8; ++ks) k_rd[ks] = r32 * ROWB + (((2 * ks + hh) ^ (r32 & 15)) << 4);
  // V^T fragment (A operand of the e4m3 MFMA, rows = channels): 16-lane group = (half hh, channel half dsub); lane pq of
  // the group addresses 8 bytes of key row 16 n + 4 hh + (tt & 3) + 8 (tt >> 2), tt = pq >> 1, at channel 32 dt + 16 dsub
  // + 8 (pq & 1), and receives channel 32 dt + 16 dsub + pq of the group's 8 key rows (attn_fwd_fp8.hip)
  int v_rd[4];
  {
    const int dsub = (lane >> 4) & 1, pq = lane & 15, tt = pq >> 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      v_rd[dt] = VBASE + (4 * hh + (tt & 3) + 8 * (tt >> 2)) * ROWB8 + ((dt ^ ((tt >> 1) & 3)) << 5) + 16 * dsub + 8 * (pq & 1);
  }

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  // row sums: one more MFMA per block against a tile of ones (fp4 e2m1 1.0 = 0b0010: 4 registers read) puts sum_k P'[k][q]
  // into every register of its result -- the same rounded P' that multiplies V.  The result is a transient tile (the MFMA
  // starts from zero) and ONE register of it is added to the running sum l_run: an accumulator tile held 16 registers for one
  // number per lane, and this loop is at the register budget (round 5: the block exponents need three more)
  float l_run = 0.f;
  i32x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = 0x22222222;
  asm volatile("" : "+v"(ones));
  i32x8 pb_;  // packed probabilities of the current block (B operand of the PV MFMAs)
#pragma unroll
  for (int i = 0; i < 8; ++i) pb_[i] = 0;
  // Online softmax in the exp2 domain.  m_run = reference point of this row (its first block's maximum; it moves again only
  // when a key tile lies more than `etrig` binades above it); the score MFMAs start from minit = p_bias - m_run + CSH in every
  // accumulator register, so they produce c = z - m_run + p_bias + CSH directly.  minit only changes in the (rare) rescale branch.
  // MX-SCALED PROBABILITIES (round 5, as attn_fwd_i8.hip): per query row and key tile of 32 keys, e = rint(max c - CSH - 8); the
  // e4m3 probability is exp2(c) / 2^(CSH + e) -- v_cvt_scalef32_pk_fp8_f32 divides by the power of two of its scale operand
  // (tools/probe_cvt_scale.hip), round to nearest even with e4m3's subnormals -- so the tile's largest lands in [2^7.5, 2^8.5)
  // whatever its distance from the row's maximum, and 2^e goes to the P V MFMA as the B operand's block scale.  The shift CSH
  // keeps the scores of every key that matters positive: the tile maximum is then a signed-integer max over the float bits.
  float m_run = -1e30f;
  constexpr float CSH = 64.f;
  constexpr float EBIAS = 12582912.f + 127.f;   // biased exponent eb = 1.5 2^23 + 127 + e: a float that is an integer, whose
  constexpr float CE = EBIAS - CSH - 8.f;       // low byte is the E8M0 scale byte (eb = max c + CE rounds once, to nearest)
  constexpr float EB_MIN = EBIAS - 100.f;
  const float pbias = pp.p_bias, etrig_b = EBIAS + pp.etrig;
  f32x16 sA0, sA1, sB0, sB1;  // scores (minus m_run) of the current / next key block (roles swap every block)
  f32x16 minit;
#pragma unroll
  for (int i = 0; i < 16; ++i) minit[i] = 0.f;
  float ecur = EBIAS;          // biased block exponent of key tile hh (whose scale this lane supplies) of the current block
  float sc0 = 1.f, sc1 = 1.f;  // 2^(CSH + e) of key tiles 0, 1 of the current block: the conversions' scale operands

  // two floats / 2^floor(log2 scale) -> two e4m3 bytes into the low (hi_ false) or high half of a word
#define CVT_SC(old_, a_, b_, scale_, hi_)                                         \
  ([&]() { typedef __attribute__((ext_vector_type(2))) short s2_; int cvt_old_ = (old_);                            \
           s2_ cvt_r_ = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(*(s2_*)&cvt_old_, (a_), (b_), (scale_), (hi_));    \
           return *(int*)&cvt_r_; }())
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
  // Block exponents of the NEXT block from its (shifted, hence positive where it matters) scores: signed-integer max over
  // each key tile's 16 registers (v_max3_i32 needs no canonicalising v_max x,x of the MFMA outputs; a tile that is negative
  // throughout -- 64 binades under the reference -- reads as "least": its exponent clamps), one half exchange that leaves tile
  // 0's row max in the lanes of half 0 and tile 1's in half 1 (where the MFMA reads the scales), one add that rounds to the
  // biased exponent, a second exchange for both tiles' conversion scales (bits (127 + CSH + e) << 23).
#define TILE_EXP(eh_, s0_, s1_, a_, b_)                                           \
  {                                                                               \
    int m0_ = max(max(__float_as_int(a_[0]), __float_as_int(a_[1])), __float_as_int(a_[2])); \
    int m1_ = max(max(__float_as_int(b_[0]), __float_as_int(b_[1])), __float_as_int(b_[2])); \
    _Pragma("unroll") for (int i_ = 3; i_ < 15; i_ += 2) {                        \
      m0_ = max(max(m0_, __float_as_int(a_[i_])), __float_as_int(a_[i_ + 1]));    \
      m1_ = max(max(m1_, __float_as_int(b_[i_])), __float_as_int(b_[i_ + 1]));    \
    }                                                                             \
    m0_ = max(m0_, __float_as_int(a_[15]));                                       \
    m1_ = max(m1_, __float_as_int(b_[15]));                                       \
    auto r_ = __builtin_amdgcn_permlane32_swap((unsigned)m0_, (unsigned)m1_, false, false); \
    /* (... , 0): a tile that is negative throughout would hand over its LEAST element; with the maximum read as 0 its */ \
    /* elements still convert without overflow, and e >= -72 needs no floor                                            */ \
    const int mt_ = max(max((int)r_[0], (int)r_[1]), 0);                          \
    eh_ = __int_as_float(mt_) + CE;                                               \
    auto e_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(eh_), __float_as_uint(eh_), false, false); \
    s0_ = __uint_as_float((e_[0] + (unsigned)CSH) << 23);                         \
    s1_ = __uint_as_float((e_[1] + (unsigned)CSH) << 23);                         \
  }
  // move the reference point of the row up by g_ (>= 0): everything accumulated so far and the current block's
  // offset scores are brought to the new reference, and the accumulator seed follows.  The empty asm keeps the
  // seed an opaque 16-register value (otherwise the compiler re-materialises the splat before every use).
#define RAISE_L_(a_) l_run = (l_run + lt_[0]) * (a_); lt_[0] = 0.f; /* block j-1 went in at the old reference */
#define RAISE_REF(c0_, c1_)                                                       \
  {                                                                               \
    const float g_ = fmaxf(half_max(ecur) - EBIAS, 0.f); /* whole binades: the larger of the row's two tile exponents */ \
    const float alpha_ = __builtin_amdgcn_exp2f(-g_);                             \
    _Pragma("unroll") for (int dt_ = 0; dt_ < 4; ++dt_)                           \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) o[dt_][i_] *= alpha_;     \
    RAISE_L_(alpha_)                                                              \
    m_run += g_;                                                                  \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { c0_[i_] -= g_; c1_[i_] -= g_; minit[i_] = pbias - m_run + CSH; } \
    asm volatile("" : "+v"(minit));                                               \
    ecur = fmaxf(ecur - g_, EB_MIN);                                              \
    auto e2_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(ecur), __float_as_uint(ecur), false, false); \
    sc0 = __uint_as_float((e2_[0] + (unsigned)CSH) << 23);                        \
    sc1 = __uint_as_float((e2_[1] + (unsigned)CSH) << 23);                        \
  }
  // end of a step: the wave's requests have landed, then the workgroup barrier (which also orders every wave's LDS reads of this
  // step before the next step's overwrites)
#define STEP_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // WAVE ROLES (round 5; the e4m3 / int8-score kernels' step, attn_fwd_fp8.hip): the two waves of a SIMD take turns between a
  // MATRIX part -- O += V^T P^T and the row sums of block j-1 (5 MFMAs of 64 cycles), then the scores of block j+1 (16 MFMAs of
  // 32) -- and a VALU part -- exp2 + conversion of a block, the block exponents of the next.  One dataflow per wave (rounds 3-5:
  // the 16-bit kernel's step) had both waves of a SIMD issue-bound in the score phase together and pipe-bound in the P V phase
  // together: 61 % busy pipes at 1.98 GHz, a third of the wave cycles at a wait; roles: 67.6 % at 1.87 GHz, +4.5 %, the same bits
  // (tools/measure/r5_pmc_8bit.sh, profiles/r05_fp8pv_roles.txt).
  //   step j, waves < NW/2 :   matrix(j)  then  valu(convert block j; exponents of block j+1)
  //   step j, waves >= NW/2:   valu(convert block j-1; exponents of block j)  then  matrix(j)
  //   matrix(j) = PV(j-1) + row sums, [mask the tail of block j], [move the reference point for block j], QK(j+1)
  // Requests of step j (top of the step): K(j+2) -> the slot K(j) left, V(j) -> the slot V(j-2) left (P V lags the scores by two
  // blocks), rings of two.
#define STAGE_DMA(kfree_, vfree_, j_)                                             \
  DMA_K(kfree_)                                                                   \
  DMA_V(vfree_)                                                                   \
  ROWS_OF_V(rowV, (j_) + 1)                                                       \
  ROWS_OF(rowK, (j_) + 3)                                                         \
  __builtin_amdgcn_sched_barrier(0);
  int sbp = SC_ONE;  // scale bytes (this lane's key tile) of the block whose bytes are in pb_
#define VFRAG(dt_, slot_)                                                         \
  _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) {                              \
    const i32x2 t_ = __builtin_amdgcn_ds_read_tr8_b64_v2i32(                      \
        (LDS_AS i32x2*)(smem + (slot_) * TILE8 + v_rd[dt_] + n_ * 16 * ROWB8));   \
    vf_[dt_][2 * n_] = t_[0]; vf_[dt_][2 * n_ + 1] = t_[1];                       \
  }
#define KF_(ks_, t_, slot_) (*(const V8*)(smem + (slot_) * TILE_BYTES + k_rd[ks_] + (t_) * 32 * ROWB))
#define PV_PART(vslot_, mid_)                                                     \
  i32x8 vf_[4];                                                                   \
  VFRAG(0, vslot_) VFRAG(1, vslot_) VFRAG(2, vslot_) VFRAG(3, vslot_)             \
  f32x16 lt_ = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(                   \
      ones, pb_, f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, 4, 0, 0, SC_ONE, 0, sbp); \
  asm("" : "+v"(lt_), "+v"(vf_[0])); /* the row-sum MFMA before the first that needs a fragment */ \
  mid_                                                                            \
  o[0] = mfma8(vf_[0], pb_, o[0], sbp);                                           \
  o[1] = mfma8(vf_[1], pb_, o[1], sbp);                                           \
  o[2] = mfma8(vf_[2], pb_, o[2], sbp);                                           \
  o[3] = mfma8(vf_[3], pb_, o[3], sbp);
#define SG_(mask_, n_) __builtin_amdgcn_sched_group_barrier(mask_, n_, 0);
  // P V half: the reads of V channel tiles 0, 1, then 5 MFMAs (the row sum first) with the later reads under the earlier ones:
  // V tiles 2, 3 and the first two k-steps of the next K block
#define SCHED_M()                                                                 \
  SG_(0x100, 8)                                                                   \
  SG_(0x008, 1) SG_(0x100, 4)                                                     \
  SG_(0x008, 1) SG_(0x100, 4)                                                     \
  SG_(0x008, 1) SG_(0x100, 2)                                                     \
  SG_(0x008, 1) SG_(0x100, 2)                                                     \
  SG_(0x008, 1)
  // score half: per MFMA one fragment read, two k-steps ahead
#define SCHED_S()                                                                 \
  _Pragma("unroll") for (int g_ = 0; g_ < 12; ++g_) { SG_(0x008, 1) SG_(0x100, 1) } \
  SG_(0x008, 4)
#define VALU_PART(e0_, e1_, m0_, m1_)                                             \
  {                                                                               \
    sbp = __float_as_int(ecur); /* its low byte: the E8M0 scale 127 + e of this lane's key tile */ \
    const float cs0_ = sc0, cs1_ = sc1;                                           \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                           \
      e0_[i_] = __builtin_amdgcn_exp2f(e0_[i_]);                                  \
      e1_[i_] = __builtin_amdgcn_exp2f(e1_[i_]);                                  \
    }                                                                             \
    _Pragma("unroll") for (int w_ = 0; w_ < 4; ++w_) {                            \
      pb_[w_] = CVT_SC(pb_[w_], e0_[4 * w_], e0_[4 * w_ + 1], cs0_, false);        \
      pb_[w_] = CVT_SC(pb_[w_], e0_[4 * w_ + 2], e0_[4 * w_ + 3], cs0_, true);     \
      pb_[4 + w_] = CVT_SC(pb_[4 + w_], e1_[4 * w_], e1_[4 * w_ + 1], cs1_, false); \
      pb_[4 + w_] = CVT_SC(pb_[4 + w_], e1_[4 * w_ + 2], e1_[4 * w_ + 3], cs1_, true); \
    }                                                                             \
    TILE_EXP(ecur, sc0, sc1, m0_, m1_)                                            \
  }
  // (the matrix part at a raised priority: without it the roles gain nothing -- 37.7 against 35.9 ms; levels 1, 2, 3 equal)
#define MX_PRIO_HI() __builtin_amdgcn_s_setprio(2);
#define MX_PRIO_LO() __builtin_amdgcn_s_setprio(0);
#define MATRIX_PART(c0_, c1_, n0_, n1_, knext_, j_)                               \
  {                                                                               \
    MX_PRIO_HI()                                                                  \
    V8 ka_[KPRE > 0 ? KPRE : 1][2];                                               \
    PV_PART(knext_, _Pragma("unroll") for (int ks_ = 0; ks_ < KPRE; ++ks_) { ka_[ks_][0] = KF_(ks_, 0, knext_); ka_[ks_][1] = KF_(ks_, 1, knext_); }) \
    SCHED_M()                                                                     \
    asm volatile("" : "+v"(lt_), "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3])); \
    /* last, partial key block: mask its tail (once per workgroup; the exponents were taken over the clamped rows too) */ \
    if ((j_) * KVB + KVB > n_kv) {                                                \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                         \
        const int row_ = (i_ & 3) + 8 * (i_ >> 2) + 4 * hh;                       \
        if ((j_) * KVB + row_ >= n_kv) c0_[i_] = -INFINITY;                       \
        if ((j_) * KVB + 32 + row_ >= n_kv) c1_[i_] = -INFINITY;                  \
      }                                                                           \
    }                                                                             \
    if (!__all(ecur <= etrig_b)) { RAISE_REF(c0_, c1_) }                          \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 8; ++ks_) {                         \
      V8 k0_, k1_;                                                                \
      if (ks_ < KPRE) { k0_ = ka_[ks_][0]; k1_ = ka_[ks_][1]; }                   \
      else { k0_ = KF_(ks_, 0, knext_); k1_ = KF_(ks_, 1, knext_); }              \
      n0_ = MF<T>::mfma(k0_, qf[ks_], ks_ == 0 ? minit : n0_);                    \
      n1_ = MF<T>::mfma(k1_, qf[ks_], ks_ == 0 ? minit : n1_);                    \
    }                                                                             \
    SCHED_S()                                                                     \
    l_run += lt_[0];                                                              \
    MX_PRIO_LO()                                                                  \
  }
  // kcur_ = j % 2: slot K(j) leaves (-> K(j+2)) and V(j) takes; knext_ = (j+1) % 2: slot of K(j+1) and of V(j-1)
#define STEP(c0_, c1_, n0_, n1_, kcur_, knext_, j_)                               \
  {                                                                               \
    STAGE_DMA(kcur_, kcur_, j_) /* (spread over the VALU part, as attn_fwd_i8.hip does, the request registers live across the */ \
    if (wave_active) {          /*  matrix part: 6 VGPRs spilled, 3 % slower than here: profiles/r05_fp8pv_roles.txt)        */ \
      if (role_y) VALU_PART(n0_, n1_, c0_, c1_)                                   \
      __builtin_amdgcn_sched_barrier(0);                                          \
      MATRIX_PART(c0_, c1_, n0_, n1_, knext_, j_)                                 \
      __builtin_amdgcn_sched_barrier(0);                                          \
      if (!role_y) VALU_PART(c0_, c1_, n0_, n1_)                                  \
    }                                                                             \
    STEP_SYNC()                                                                   \
  }
  // wave-uniform: which half starts its steps with the VALU part -- the later-dispatched one (the other way round: 41.2 against
  // 35.6 ms, profiles/r05_fp8pv_roles.txt)
  const bool role_y = NW == 8 && wave >= NW / 2;
  const int nsteps = blk1 - blk0;
  if (nsteps > 0) {
    // prologue: K(0), K(1); the scores of block 0 fix the reference point
    ROWS_OF(rowK, blk0)
    DMA_K(0)
    ROWS_OF(rowK, blk0 + 1)
    DMA_K(1)
    ROWS_OF_V(rowV, blk0)
    ROWS_OF(rowK, blk0 + 2)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
      float mx0;
      ROW_MAX(mx0, sA0, sA1)
      m_run = mx0;  // the first block fixes the reference point at its true row max (block blk0 always has a valid key)
      const float sh = pbias - m_run + CSH;
      float l0 = sA0[0], l1 = sA1[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) { l0 = fmaxf(l0, sA0[i]); l1 = fmaxf(l1, sA1[i]); }
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(l0), __float_as_uint(l1), false, false);
      const float lt = fmaxf(fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1])) + sh, 0.f);  // half 0: tile 0, half 1: tile 1
      ecur = lt + CE;
      auto e = __builtin_amdgcn_permlane32_swap(__float_as_uint(ecur), __float_as_uint(ecur), false, false);
      sc0 = __uint_as_float((e[0] + (unsigned)CSH) << 23);
      sc1 = __uint_as_float((e[1] + (unsigned)CSH) << 23);
#pragma unroll
      for (int i = 0; i < 16; ++i) { sA0[i] += sh; sA1[i] += sh; minit[i] = sh; }
      asm volatile("" : "+v"(minit));
    }
    __syncthreads();  // every wave has read K(0) before step 0 overwrites its slot with K(2)
    {  // step 0: no P V yet -- the scores of block 1, then (first role) the VALU part of block 0
      STAGE_DMA(0, 0, blk0)
      if (wave_active) {
        QK(sB0, sB1, 1)
        __builtin_amdgcn_sched_barrier(0);
        if (!role_y) VALU_PART(sA0, sA1, sB0, sB1)
      }
      STEP_SYNC()
    }
    for (int jj = 1; jj < nsteps; jj += 2) {
      STEP(sB0, sB1, sA0, sA1, 1, 0, blk0 + jj)
      if (jj + 1 >= nsteps) break;
      STEP(sA0, sA1, sB0, sB1, 0, 1, blk0 + jj + 1)
    }
    // drain: the second role still owes the conversion of the last block; then P V of the last block
    if (wave_active) {
      if (role_y) {
        if ((nsteps - 1) & 1) VALU_PART(sB0, sB1, sA0, sA1)
        else VALU_PART(sA0, sA1, sB0, sB1)
      }
      PV_PART((nsteps - 1) & 1, )
      l_run += lt_[0];
    }
  }
#undef VFRAG
#undef KF_
#undef PV_PART
#undef SCHED_M
#undef SCHED_S
#undef SG_
#undef VALU_PART
#undef MATRIX_PART
#undef QK
#undef ROW_MAX
#undef TILE_EXP
#undef CVT_SC
#undef RAISE_REF
#undef STEP
#undef STAGE_DMA
#undef STEP_SYNC
#undef ROWS_OF
#undef ROWS_OF_V
#undef DMA_K
#undef DMA_V

  if (!wave_active) return;
  // ---------------- epilogue ----------------
  const float l_tot = l_run;
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
  const float* vd = pp.v_descale + (int64_t)head * pp.v_descale_sh + 4 * hh;
  uint2 packed[16];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const f32x4 s = *(const f32x4*)(vd + 32 * dt + 8 * rg);
      V4 t;
#pragma unroll
      for (int j = 0; j < 4; ++j) t[j] = (T)(o[dt][4 * rg + j] * (inv * s[j]));
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
__global__ __launch_bounds__(NW * 64, 2) void attn_mx_kernel(const ParamsMx pp) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
  __shared__ __attribute__((aligned(16))) char smem[SMEM_MX];
  const int wg = live_order(pp.p, blockIdx.x, gridDim.x, pp.p.xcd_remap);  // XCD-aware order over the live workgroups
  attn_mx_body<T, NW, KVTAB, NSMX>(pp, smem, wg);
#endif
}

// Several launches fused into ONE grid (the experts of a routed layer), as attn_fwd_multi_kernel
template <typename T>
__global__ __launch_bounds__(512, 2) void attn_mx_multi_kernel(const MultiParamsMx mp) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char smem[SMEM_MX];
  const int b = blockIdx.x;
  int s = 0;
#pragma unroll
  for (int i = 1; i < MAX_SEGMENTS; ++i) s += (i < mp.n && b >= mp.start[i]) ? 1 : 0;
  const ParamsMx& pp = mp.seg[s];
  const int wg = live_order(pp.p, b - mp.start[s], mp.start[s + 1] - mp.start[s], true);
  if (pp.p.kv_rows) attn_mx_body<T, 8, true, NSMX>(pp, smem, wg);
  else attn_mx_body<T, 8, false, NSMX>(pp, smem, wg);
#endif
}

// Merge the split-key partials: one wave per (head slot, query position); both partial sums carry the 2^p_bias factor
template <typename T>
__global__ __launch_bounds__(256) void attn_mx_combine_kernel(const ParamsMx pp) {
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
  T pair[2] = {(T)(acc0 * inv * sd.x), (T)(acc1 * inv * sd.y)};
  char* ob = p.o + (int64_t)head * p.o_sh + lane * 4;
  *(uint32_t*)(ob + row * p.o_ss) = *(uint32_t*)pair;
  if (p.dup_rows && pos < p.n_dup_pos) {
    const int32_t* dr = p.dup_rows + (int64_t)y * p.dup_rows_sh + (int64_t)pos * p.n_dup;
    for (int i = 0; i < p.n_dup; ++i) *(uint32_t*)(ob + (int64_t)dr[i] * p.o_ss) = *(uint32_t*)pair;
  }
}

int fill_mx(const vorta_attn_args* a, const vorta_attn_fp8_ext* ext, ParamsMx& pp, int& block_rows) {
  if (!ext || ext->struct_size != sizeof(vorta_attn_fp8_ext)) return VORTA_EINVAL;
  if (!a || (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16) || ext->out_dtype != a->dtype) return VORTA_EUNSUPPORTED;
  int rc = fill_params(a, pp.p, block_rows, 2, 1);  // q, k, o in 16 bits (strides in elements); v in e4m3 (bytes)
  if (rc != VORTA_OK) return rc;
  if (pp.p.n_heads == 0 || pp.p.n_groups == 0) return VORTA_OK;
  if (a->variant == 1) return VORTA_EUNSUPPORTED;  // only the pipelined LDS-DMA body exists
  if (!ext->v_descale || ext->v_descale_stride_h < D) return VORTA_EINVAL;
  // p_bias: P' = 2^(score - reference + p_bias) (any value: the block scales carry the range); defer: binades a key tile may lie
  // above its row's reference point before the reference moves (fp32 accumulators: P' <= 2^(defer + 9))
  const float pb = ext->p_bias != 0.f ? ext->p_bias : 5.f;
  const float df = ext->defer != 0.f ? ext->defer : 24.f;
  if (!(pb >= 0.f) || pb > 16.f || !(df >= 0.f) || df > 40.f) return VORTA_EINVAL;
  if (ext->flags & 1) return VORTA_EUNSUPPORTED;  // (the VALU row sum is an experiment of the all-e4m3 kernel)
  pp.v_descale = ext->v_descale;
  pp.v_descale_sh = ext->v_descale_stride_h;
  pp.p_bias = pb;
  pp.etrig = df;
  return VORTA_OK;
}

template <typename T>
int launch_mx(const ParamsMx& pp, int block_rows, hipStream_t st) {
  const Params& p = pp.p;
  const int64_t total = (int64_t)p.n_groups * p.blocks_per_group * p.n_heads * p.n_splits;
  if (total <= 0) return VORTA_OK;
  if (total > 0x7fffffff) return VORTA_EINVAL;
  const dim3 g((unsigned)total);
#define LMX(NW_, TAB_) hipLaunchKernelGGL((attn_mx_kernel<T, NW_, TAB_>), g, dim3(NW_ * 64), 0, st, pp)
  if (block_rows == 256) { if (p.kv_rows) LMX(8, true); else LMX(8, false); }
  else { if (p.kv_rows) LMX(4, true); else LMX(4, false); }
#undef LMX
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  if (p.n_splits > 1) {
    const int64_t items = (int64_t)p.n_heads * p.n_q;
    hipLaunchKernelGGL((attn_mx_combine_kernel<T>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, pp);
    e = hipGetLastError();
    if (e != hipSuccess) return vorta_set_hip_error(e);
  }
  return VORTA_OK;
}

}  // namespace

int vorta_attn::mx_fwd(const vorta_attn_args* a, const vorta_attn_fp8_ext* ext, void* hip_stream) {
  ParamsMx pp{};
  int block_rows = 0;
  int rc = fill_mx(a, ext, pp, block_rows);
  if (rc != VORTA_OK) return rc;
  if (pp.p.n_heads == 0 || pp.p.n_groups == 0) return VORTA_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  return a->dtype == VORTA_BF16 ? launch_mx<__bf16>(pp, block_rows, st) : launch_mx<_Float16>(pp, block_rows, st);
}

int vorta_attn::mx_fwd_batch(const vorta_attn_args* args, const vorta_attn_fp8_ext* ext, int32_t n, void* hip_stream) {
  if (!args || !ext || n < 0 || n > MAX_SEGMENTS) return VORTA_EINVAL;
  MultiParamsMx mp{};
  int64_t total = 0;
  int m = 0;
  int dtype = -1;
  for (int i = 0; i < n; ++i) {
    ParamsMx pp{};
    int block_rows = 0;
    int rc = fill_mx(&args[i], ext, pp, block_rows);
    if (rc != VORTA_OK) return rc;
    if (pp.p.n_heads == 0 || pp.p.n_groups == 0) continue;
    if (block_rows != 256) return VORTA_EUNSUPPORTED;  // only 256-row launches share a grid
    if (dtype >= 0 && dtype != args[i].dtype) return VORTA_EINVAL;
    dtype = args[i].dtype;
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
  const bool bf = dtype == VORTA_BF16;
  if (bf) hipLaunchKernelGGL((attn_mx_multi_kernel<__bf16>), dim3((unsigned)total), dim3(512), 0, st, mp);
  else hipLaunchKernelGGL((attn_mx_multi_kernel<_Float16>), dim3((unsigned)total), dim3(512), 0, st, mp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  for (int i = 0; i < m; ++i) {
    const ParamsMx& pp = mp.seg[i];
    if (pp.p.n_splits > 1) {
      const int64_t items = (int64_t)pp.p.n_heads * pp.p.n_q;
      if (bf) hipLaunchKernelGGL((attn_mx_combine_kernel<__bf16>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, pp);
      else hipLaunchKernelGGL((attn_mx_combine_kernel<_Float16>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, pp);
      e = hipGetLastError();
      if (e != hipSuccess) return vorta_set_hip_error(e);
    }
  }
  return VORTA_OK;
}
