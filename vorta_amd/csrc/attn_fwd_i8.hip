// Gather flash-attention forward with INT8 SCORES and e4m3 P V (include/vorta_hip.h: vorta_attn_fwd_i8, ABI 6).
//
// Why: the score is where an 8-bit format costs.  e4m3 keeps 3 mantissa bits wherever a value sits, so a logit is off by
// 3.7 % of the root sum of squares of its 128 products (attn_fwd_fp8.hip: 40 dB against the 16-bit kernels only where the
// softmax is flat); int8 keeps 7 bits next to the operand's maximum and runs at the same MFMA rate (v_mfma_i32_32x32x32_i8:
// the cycles of the 16-bit 32x32x16 form at twice the K).  With both operands centred -- the term the query centre leaves,
// b[key] = cq . (k - ck), is exact float32 work of the quantiser (i8_quant.hip) and enters the accumulator as its initial
// value -- and their channel ranges balanced, it holds >= 40 dB on every input family of tests/_fp8_inputs.py.
//
// What an int32 accumulator must NOT cost is VALU work per score (a first version with a scale per key row -- convert,
// multiply, multiply-add, exp2, pack: five instructions per score -- ran at the mixed kernel's 1.2 x of bf16, not at the
// e4m3 kernel's 1.7 x).  Here a score costs TWO instructions between the score MFMA and the P V MFMA:
//   * the accumulator starts from 0x4B400000 + seed[key] (the integer whose float reading is 1.5 2^23 + seed), so after the
//     MFMAs its bits READ AS A FLOAT are 1.5 2^23 + (q8 . k8 + seed), exactly: no v_cvt_f32_i32;
//   * one v_fma_f32 with the wave's unit (8 scale log2e sq sk) and an offset that folds the magic constant, the reference
//     point, p_bias and the e4m3 exponent bias gives y = 8 log2 P' + 56;
//   * one v_cvt_pk_u8_f32 writes rint(y) as the e4m3 BYTE of P' (exponent field = integer part of log2 P', mantissa = linear
//     interpolation of its fraction: +-3 % of 2^x, ~0.5-0.8 dB of output PSNR) straight into the B operand of the P V MFMA:
//     no v_exp_f32, no v_cvt_pk_fp8_f32.
// Scales are uniform where the MFMA needs them uniform: one key scale per head (sk), one query scale per WAVE (sq, taken by
// the wave itself over its 32 rows -- queries are read once per workgroup).  The per-key seeds depend on the wave's sq:
// every wave turns the block's 64 float biases (LDS-DMA'd with the K tile, one step further ahead) into integers with three
// VALU instructions, parks them in a private 256-byte LDS slot and reads them back as the accumulators' initial values.
//
// Structure otherwise: the mixed kernel (attn_fwd_mx.hip) -- K tile int8 rows of 128 bytes (8 KiB, the e4m3 kernels' image),
// scores one key block ahead, V tile e4m3 through ds_read_b64_tr_b8, O^T += V8^T P'^T on v_mfma_f32_32x32x64_f8f6f4, row sums
// from the ones-tile MFMA, deferred rescale, v_descale in the epilogue.  Per wave and 64-key block: 8 MFMAs of 32 cycles + 5
// of 64 = 576 pipe cycles (1 024 in 16 bits, 832 mixed, 576 all-e4m3).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

#include "attn_common.h"

namespace {
using namespace vorta_attn;

typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(2))) int i32x2;
typedef __attribute__((ext_vector_type(4))) int i32x4;
typedef __attribute__((ext_vector_type(16))) int i32x16;

constexpr int ROWB8 = D;             // bytes per e4m3 row
constexpr int TILE8 = KVB * ROWB8;   // 8 KiB
constexpr int NSI8 = 2;              // ring depth of the K and V tiles (8 KiB each) and of the key-bias tiles (256 B)
constexpr int SC_BYTES = KVB * 4;    // one float per key of a block
constexpr int SEED_BYTES = 8 * 2 * SC_BYTES;  // per wave two slots of 64 int32 seeds
constexpr int SMEM_I8 = NSI8 * (2 * TILE8 + SC_BYTES) + SEED_BYTES;  // 36.5 KiB
constexpr int MAGIC_I = 0x4B400000;  // float bits of 1.5 * 2^23 = 12 582 912
constexpr float MAGIC_F = 12582912.f;
constexpr float SEED_LIMIT = 2000000.f;  // |q8 . k8| <= 2 064 512; the sum must stay below 2^22

struct ParamsI8 {
  Params p;
  const float* k_bias; int64_t k_bias_sh;
  const float* q_prep; int64_t q_prep_sh;
  const float* k_head_scale;
  const float* v_descale; int64_t v_descale_sh;
  float p_bias;  // log2 bias of the packed probabilities
  float thr;     // p_bias + defer: offset scores above this move the reference point
};
struct MultiParamsI8 {
  ParamsI8 seg[MAX_SEGMENTS];
  int start[MAX_SEGMENTS + 1];
  int n;
};

__device__ __forceinline__ i32x16 mfma_i8(i32x4 a, i32x4 b, i32x16 c) {
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma8(i32x8 a, i32x8 b, f32x16 c) {
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);  // cbsz = blgp = 0: e4m3 x e4m3, no block scale
}

template <typename T, int NW, bool KVTAB, int NS>
__device__ __forceinline__ void attn_i8_body(const ParamsI8& pp, char* __restrict__ smem, const int wg) {
  // NS = depth of the K / V rings: K(j+NS) / V(j+NS-1) are requested at the top of step j, NS-1 steps before the step that
  // reads them; the bias tile of K(j+NS+1) is requested with K(j+NS) (its seeds are made one step before its scores).
  // LDS: K ring [0, NS*8K), V ring behind it, bias ring behind that, the waves' seed slots last.
  static_assert(NS == 2, "ring depth");
  const Params& p = pp.p;
  constexpr int VBASE = NS * TILE8;
  constexpr int BBASE = 2 * NS * TILE8;
  constexpr int SEEDBASE = BBASE + NS * SC_BYTES;
  using V8 = typename MF<T>::v8;
  using V4 = typename MF<T>::v4;
  constexpr int QB = NW * 32;
  constexpr int CH = 8 / NW;  // 1-KiB DMA pieces (8 tile rows of 128 bytes) of one tile per wave
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

  // Q -> int8 here: qt = (q - cq) s with the head's q_prep, the abs-max over the WAVE's 32 rows (rows past the end of the
  // group repeat its last row: they change nothing), q8 = rint(qt 127 / amax).  B operand of v_mfma_i32_32x32x32_i8, k-step
  // ks: byte j of lane (r32, hh) = channel 32 ks + 16 hh + j.  m8 = 8 (amax / 127) scale log2(e) sk: what one integer
  // score unit is worth in the byte domain y = 8 log2 P' + 56; inv_q = 127 / amax turns a key's float bias into its seed.
  i32x4 qf[4];
  float m8, inv_q;
  {
#pragma clang fp contract(off)
    const char* qp = p.q + (int64_t)head * p.q_sh + my_row * p.q_ss + hh * 32;
    const float* cq = pp.q_prep + (int64_t)head * pp.q_prep_sh + hh * 16;
    const float* sm = cq + D;
    float qt[64];
    float am = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const V8 lo = *(const V8*)(qp + ks * 64);
      const V8 hi = *(const V8*)(qp + ks * 64 + 16);
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        qt[16 * ks + i] = ((float)lo[i] - cq[32 * ks + i]) * sm[32 * ks + i];
        qt[16 * ks + 8 + i] = ((float)hi[i] - cq[32 * ks + 8 + i]) * sm[32 * ks + 8 + i];
        am = fmaxf(am, fmaxf(fabsf(qt[16 * ks + i]), fabsf(qt[16 * ks + 8 + i])));
      }
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) am = fmaxf(am, __shfl_xor(am, m, 64));  // max is order-free
    inv_q = am > 0.f ? 127.f / am : 0.f;
    const float sq = am > 0.f ? am * (1.f / 127.f) : 1.f;
    m8 = 8.f * ((sq * p.scale_log2) * pp.k_head_scale[head]);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          int v = (int)__builtin_rintf(qt[16 * ks + 4 * w + b] * inv_q);
          v = max(-127, min(127, v));
          word |= ((uint32_t)v & 0xffu) << (8 * b);
        }
        qf[ks][w] = (int)word;
      }
  }

  // ---- loader setup ----
  const int32_t* kv_rows =
      p.kv_rows ? p.kv_rows + (int64_t)y * p.kv_rows_sh + (int64_t)grp * p.kv_rows_sg : nullptr;
  // K / V tiles go global -> LDS directly (buffer_load ... lds).  One wave instruction fills 1 KiB = 8 tile rows (8 lanes x
  // 16 B per row); the destination is lane-linear, so the bank swizzles of the tile images are applied on the SOURCE side.
  const __amdgpu_buffer_rsrc_t k_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.k + (int64_t)head * p.k_sh), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t v_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.v + (int64_t)head * p.v_sh), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(pp.k_bias + (int64_t)head * pp.k_bias_sh), 0, 0x7fffffff, 0x00020000);
  const int k_ss32 = (int)p.k_ss, v_ss32 = (int)p.v_ss;
  int k_col[CH], v_col[CH];  // source byte offset inside the row for the chunk this lane lands in
#pragma unroll
  for (int i = 0; i < CH; ++i) {
    const int row = 8 * (CH * wave + i) + (lane >> 3);
    k_col[i] = ((lane & 7) ^ ((row >> 1) & 7)) << 4;
    v_col[i] = ((lane & 7) ^ ((((row >> 1) & 1) | (((row >> 3) & 1) << 1)) << 1)) << 4;
  }
  int rowK[CH], rowV[CH];  // rows of the next K block / next V block to fetch
  int rowB = 0;            // wave 0: row of key `lane` of the next bias tile
#define ROWS_OF(dst_, blk_)                                                       \
  _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) {                             \
    const int pos_ = min((blk_) * KVB + 8 * (CH * wave + i_) + (lane >> 3), n_kv - 1); \
    if constexpr (KVTAB) dst_[i_] = kv_rows[pos_];                                \
    else dst_[i_] = p.kv_row_offset + pos_;                                       \
  }
#define ROW_OF_B(blk_)                                                            \
  {                                                                               \
    const int pos_ = min((blk_) * KVB + lane, n_kv - 1);                          \
    if constexpr (KVTAB) rowB = kv_rows[pos_];                                    \
    else rowB = p.kv_row_offset + pos_;                                           \
  }
#define DMA_K(par_) _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(     \
      k_rsrc, (LDS_AS void*)(smem + (par_) * TILE8 + (CH * wave + i_) * 1024), 16,                                  \
      (int)__umul24((unsigned)rowK[i_], (unsigned)k_ss32) + k_col[i_], 0, 0, 0);
#define DMA_B(par_)                                                                                                 \
  if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (LDS_AS void*)(smem + BBASE + (par_) * SC_BYTES), 4, \
                                                          rowB * 4, 0, 0, 0);
#define DMA_V(par_) _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(     \
      v_rsrc, (LDS_AS void*)(smem + VBASE + (par_) * TILE8 + (CH * wave + i_) * 1024), 16,                          \
      (int)__umul24((unsigned)rowV[i_], (unsigned)v_ss32) + v_col[i_], 0, 0, 0);

  // ---- LDS read addresses ----
  // K fragment (A operand, rows = keys): lane (r32, hh) reads 16 bytes of key row r32 (+ 32 for the second tile) at channel
  // 32 ks + 16 hh = chunk 2 ks + hh of the row, swizzled with (row >> 1) & 7 (the same for row and row + 32)
  int k_rd[4];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks) k_rd[ks] = r32 * ROWB8 + (((2 * ks + hh) ^ ((r32 >> 1) & 7)) << 4);
  // V^T fragment (A operand of the e4m3 MFMA, rows = channels): as attn_fwd_mx.hip / attn_fwd_fp8.hip
  int v_rd[4];
  {
    const int dsub = (lane >> 4) & 1, pq = lane & 15, tt = pq >> 1;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
      v_rd[dt] = VBASE + (4 * hh + (tt & 3) + 8 * (tt >> 2)) * ROWB8 + ((dt ^ ((tt >> 1) & 3)) << 5) + 16 * dsub + 8 * (pq & 1);
  }
  // seeds of the keys this lane's accumulator registers hold: register i of tile t <-> key 32 t + 8 (i >> 2) + 4 hh + (i & 3):
  // read j (0..3) of tile t = 16 bytes = keys 32 t + 8 j + 4 hh + {0..3} (every lane of a half reads the same address)
  const int seed_base = SEEDBASE + wave * 2 * SC_BYTES;
  const int seed_rd = seed_base + 16 * hh;
  const int seed_wr = seed_base + 4 * lane;
  const int bias_rd = BBASE + 4 * lane;

  f32x16 o[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt)
#pragma unroll
    for (int i = 0; i < 16; ++i) o[dt][i] = 0.f;
  // row sums: one more MFMA per block against a tile of ones (fp4 e2m1 1.0 = 0b0010: 4 registers read) puts sum_k P'[k][q]
  // into every register of lacc -- the same P' that multiplies V
  f32x16 lacc;
#pragma unroll
  for (int i = 0; i < 16; ++i) lacc[i] = 0.f;
  i32x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = 0x22222222;
  asm volatile("" : "+v"(ones));
  i32x8 pb_;  // bytes of the probabilities of the current block (B operand of the PV MFMAs)
#pragma unroll
  for (int i = 0; i < 8; ++i) pb_[i] = 0;
  // Online softmax in the BYTE domain y = 8 x + 56, x = log2 P' = z - m_run + p_bias.  m_run8 = 8 x the reference point of
  // this row (a lower bound of its running max, at most thr - p_bias below it); y0, y1 hold y of the current block.
  float m_run8 = -1e30f;
  const float ybias = 8.f * pp.p_bias + 56.f, ythr = 8.f * pp.thr + 56.f;
  f32x16 y0, y1;          // byte-domain scores of the current key block (keys 0-31, 32-63 of the block)
  float off8 = 0.f;       // ybias - m_run8 - MAGIC_F * m8: the addend of the conversion
  float mx_cur = -1e30f;  // row max of the current block's y

  // seeds of a block from its bias tile (slot bslot_) into this wave's seed slot sslot_: three VALU instructions per wave
#define MAKE_SEEDS(bslot_, sslot_)                                                \
  {                                                                               \
    const float b_ = *(const float*)(smem + bias_rd + (bslot_) * SC_BYTES);       \
    const float s_ = __builtin_amdgcn_fmed3f(b_ * inv_q, -SEED_LIMIT, SEED_LIMIT); \
    *(int*)(smem + seed_wr + (sslot_) * SC_BYTES) = MAGIC_I + (int)__builtin_rintf(s_); \
  }
  // raw scores of a block (K ring slot par_) on top of its seeds (seed slot sslot_): int32 bits = float 1.5 2^23 + score
#define QK(d0_, d1_, par_, sslot_)                                                \
  {                                                                               \
    _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                            \
      const i32x4 s0_ = *(const i32x4*)(smem + seed_rd + (sslot_) * SC_BYTES + 32 * j_);        \
      const i32x4 s1_ = *(const i32x4*)(smem + seed_rd + (sslot_) * SC_BYTES + 128 + 32 * j_);  \
      _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) { d0_[4 * j_ + e_] = s0_[e_]; d1_[4 * j_ + e_] = s1_[e_]; } \
    }                                                                             \
    _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) {                         \
      const i32x4 k0_ = *(const i32x4*)(smem + (par_) * TILE8 + k_rd[ks_]);       \
      const i32x4 k1_ = *(const i32x4*)(smem + (par_) * TILE8 + k_rd[ks_] + 32 * ROWB8); \
      d0_ = mfma_i8(k0_, qf[ks_], d0_);                                           \
      d1_ = mfma_i8(k1_, qf[ks_], d1_);                                           \
    }                                                                             \
  }
  // int32 bits read as floats -> byte-domain scores: y = (1.5 2^23 + score) m8 + off_  (one fused multiply-add per score)
#define TO_Y(yd0_, yd1_, a0_, a1_, off_)                                          \
  _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                             \
    yd0_[i_] = __builtin_fmaf(__int_as_float(a0_[i_]), m8, off_);                 \
    yd1_[i_] = __builtin_fmaf(__int_as_float(a1_[i_]), m8, off_);                 \
  }
#define ROW_MAX(dst_, a_, b_)                                                      \
  {                                                                               \
    float mx_ = a_[0];                                                            \
    _Pragma("unroll") for (int i_ = 1; i_ < 16; ++i_) mx_ = fmaxf(mx_, a_[i_]);   \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) mx_ = fmaxf(mx_, b_[i_]);   \
    dst_ = half_max(mx_);                                                         \
  }
  // The loop only asks two things of the NEXT block's row max: "is it above `ythr` (> 0)?" and, if so, its value.  Both
  // are answered by a signed-integer max over the float bit patterns (order-preserving for non-negative floats, any
  // negative result reads as "not above"; there are no NaNs).
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
  // move the reference point of the row up by g8_ (>= 0, byte-domain units = 8 x log2): everything accumulated so far and
  // the current block's y are brought to the new reference
#define RAISE_REF(g8_)                                                            \
  {                                                                               \
    const float alpha_ = __builtin_amdgcn_exp2f(-0.125f * (g8_));                 \
    _Pragma("unroll") for (int dt_ = 0; dt_ < 4; ++dt_)                           \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) o[dt_][i_] *= alpha_;     \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) lacc[i_] *= alpha_;         \
    m_run8 += (g8_);                                                              \
    off8 -= (g8_);                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) { y0[i_] -= (g8_); y1[i_] -= (g8_); } \
  }
  // top of step j: K(j+NS) -> the slot K(j) left, the bias tile of K(j+NS+1) -> the slot the bias of K(j+NS-1) left (its seeds
  // were made in step j-1), V(j+NS-1) -> the slot V(j-1) left
#define STAGE_DMA(kfree_, vfree_, j_)                                             \
  DMA_K(kfree_)                                                                   \
  DMA_B(vfree_)                                                                   \
  DMA_V(vfree_)                                                                   \
  ROWS_OF(rowV, (j_) + NS)                                                        \
  ROWS_OF(rowK, (j_) + NS + 1)                                                    \
  ROW_OF_B((j_) + NS + 2)                                                         \
  __builtin_amdgcn_sched_barrier(0);
#define STEP_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");

  // Issue-order recipe for the step's basic block (sched_group_barrier: 0x008 MFMA, 0x100 DS read, 0x200 DS write, 0x002
  // VALU); -DVORTA_I8_SCHED=0 leaves the order to the compiler.
#ifndef VORTA_I8_SCHED
#define VORTA_I8_SCHED 1
#endif
#ifndef VORTA_I8_SC_VALU
#define VORTA_I8_SC_VALU 4
#endif
#ifndef VORTA_I8_PV_VALU
#define VORTA_I8_PV_VALU 14
#endif
#if VORTA_I8_SCHED == 1
  // Every fragment read is issued well ahead of the MFMA that consumes it (left to itself the compiler reads each K
  // fragment into ONE register set right before its MFMA: eight exposed LDS round trips per step).  DS reads in program
  // order: 8 seed reads, 8 K fragments, 16 transposed V reads, the bias read.  Score phase: the seeds and the first four
  // K fragments up front; then per MFMA one more read (the remaining K fragments, then the V fragments of channel tiles 0 and
  // 1) and an eighth of the 32 byte conversions.  P V phase: per MFMA the V reads of the tile two ahead and a quarter of the
  // multiply-adds / row max / seed arithmetic; the bias read rides with the first, the seed write follows the VALU work.
#define SG_(mask_, n_) __builtin_amdgcn_sched_group_barrier(mask_, n_, 0);
#define SCHED_RECIPE()                                                            \
  SG_(0x100, 12)                                                                  \
  SG_(0x008, 1) SG_(0x100, 1) SG_(0x002, VORTA_I8_SC_VALU)                        \
  SG_(0x008, 1) SG_(0x100, 1) SG_(0x002, VORTA_I8_SC_VALU)                        \
  SG_(0x008, 1) SG_(0x100, 1) SG_(0x002, VORTA_I8_SC_VALU)                        \
  SG_(0x008, 1) SG_(0x100, 1) SG_(0x002, VORTA_I8_SC_VALU)                        \
  SG_(0x008, 1) SG_(0x100, 2) SG_(0x002, VORTA_I8_SC_VALU)                        \
  SG_(0x008, 1) SG_(0x100, 2) SG_(0x002, VORTA_I8_SC_VALU)                        \
  SG_(0x008, 1) SG_(0x100, 2) SG_(0x002, VORTA_I8_SC_VALU)                        \
  SG_(0x008, 1) SG_(0x100, 2) SG_(0x002, VORTA_I8_SC_VALU)                        \
  SG_(0x008, 1) SG_(0x100, 4) SG_(0x002, VORTA_I8_PV_VALU)                        \
  SG_(0x008, 1) SG_(0x100, 4) SG_(0x002, VORTA_I8_PV_VALU)                        \
  SG_(0x008, 1) SG_(0x100, 1) SG_(0x002, VORTA_I8_PV_VALU)                        \
  SG_(0x008, 1) SG_(0x002, VORTA_I8_PV_VALU)                                      \
  SG_(0x200, 1)                                                                   \
  SG_(0x008, 1)
#else
#define SCHED_RECIPE()
#endif
  // one key block.  The active path is ONE basic block after the (rare) mask / rescale branches.
#define STEP(kcur_, knext_, vfree_, j_)                                           \
  { /* kcur_ = j % NS: slot of K(j) (free), of V(j) and of the seeds of block j+2; knext_ = (j+1) % NS: K(j+1), its seeds and */ \
    /* the bias tile of K(j+2); vfree_ = (j-1) % NS */                            \
    STAGE_DMA(kcur_, vfree_, j_)                                                  \
    if (wave_active) {                                                            \
      /* mx_cur (row max of this block's y) was computed under the previous step's PV MFMAs; only the last, partial key */ \
      /* block has to mask its tail and redo it here                                                                   */ \
      if ((j_) * KVB + KVB > n_kv) {                                              \
        _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                       \
          const int row_ = (i_ & 3) + 8 * (i_ >> 2) + 4 * hh;                     \
          if ((j_) * KVB + row_ >= n_kv) y0[i_] = -INFINITY;                      \
          if ((j_) * KVB + 32 + row_ >= n_kv) y1[i_] = -INFINITY;                 \
        }                                                                         \
        ROW_MAX(mx_cur, y0, y1)                                                   \
      }                                                                           \
      /* deferred rescale: the reference point moves only when some row of the wave outgrew it by more than */ \
      /* `thr - p_bias` (so P' <= 2^thr: inside e4m3's range); rows that did not grow keep theirs (g = 0)      */ \
      if (!__all(mx_cur <= ythr)) {                                               \
        const float g8_ = fmaxf(mx_cur - ybias, 0.f);                             \
        RAISE_REF(g8_)                                                            \
      }                                                                           \
      i32x16 n0_, n1_;                                                            \
      QK(n0_, n1_, knext_, knext_) /* block j+1 (harmless garbage past the end) */ \
      /* P' bytes straight from y: rint, saturating at 0 (-inf of masked keys -> 0); y <= 8 thr + 56 = 120 < 0x7E */ \
      _Pragma("unroll") for (int w_ = 0; w_ < 4; ++w_)                            \
        _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_) {                        \
          pb_[w_] = __builtin_amdgcn_cvt_pk_u8_f32(y0[4 * w_ + b_], b_, pb_[w_]); \
          pb_[4 + w_] = __builtin_amdgcn_cvt_pk_u8_f32(y1[4 * w_ + b_], b_, pb_[4 + w_]); \
        }                                                                         \
      _Pragma("unroll") for (int dt_ = 0; dt_ < 4; ++dt_) {                       \
        i32x8 vf_;                                                                \
        _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) {                        \
          const i32x2 t_ = __builtin_amdgcn_ds_read_tr8_b64_v2i32(                \
              (LDS_AS i32x2*)(smem + (kcur_) * TILE8 + v_rd[dt_] + n_ * 16 * ROWB8)); \
          vf_[2 * n_] = t_[0]; vf_[2 * n_ + 1] = t_[1];                           \
        }                                                                         \
        o[dt_] = mfma8(vf_, pb_, o[dt_]);                                         \
      }                                                                           \
      lacc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ones, pb_, lacc, 4, 0, 0, 0, 0, 0); \
      /* block j+1: bits -> byte-domain floats and their row max; block j+2: seeds.  VALU work under the PV MFMAs above */ \
      TO_Y(y0, y1, n0_, n1_, off8)                                                \
      ROW_MAX_POS(mx_cur, y0, y1)                                                 \
      MAKE_SEEDS(kcur_, kcur_) /* block j+2: its bias tile and its seed slot have the parity of j */ \
      SCHED_RECIPE()                                                              \
    }                                                                             \
    STEP_SYNC()                                                                   \
  }

  if (blk0 < blk1) {
    // prologue: K(0..NS-1), the bias tiles of K(0), K(1) (and of K(2) once K(0)'s seeds exist), V(0..NS-2)
    ROWS_OF(rowK, blk0)
    ROW_OF_B(blk0)
    ROWS_OF(rowV, blk0)
    DMA_K(0)
    DMA_B(0)
    DMA_V(0)
    ROWS_OF(rowK, blk0 + 1)
    ROW_OF_B(blk0 + 1)
    DMA_K(1)
    DMA_B(1)
    ROWS_OF(rowV, blk0 + NS - 1)
    ROWS_OF(rowK, blk0 + NS)
    ROW_OF_B(blk0 + NS)
    __syncthreads();
    if (wave_active) {
      MAKE_SEEDS(0, 0)
      MAKE_SEEDS(1, 1)
    }
    __syncthreads();  // every wave has read the bias tiles of K(0), K(1): the bias of K(2) may land in slot 0
    DMA_B(0)
    ROW_OF_B(blk0 + NS + 1)
    if (wave_active) {
      i32x16 n0, n1;
      QK(n0, n1, 0, 0)
      const float base = -MAGIC_F * m8;
      TO_Y(y0, y1, n0, n1, base)  // 8 x the plain exp2-domain scores of the first block
      if (blk0 * KVB + KVB > n_kv) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
          const int row = (i & 3) + 8 * (i >> 2) + 4 * hh;
          if (blk0 * KVB + row >= n_kv) y0[i] = -INFINITY;
          if (blk0 * KVB + 32 + row >= n_kv) y1[i] = -INFINITY;
        }
      }
      ROW_MAX(mx_cur, y0, y1)
      // the first block fixes the reference point at its true row max (block blk0 always has a valid key);
      // O and l are still zero, so nothing is rescaled
      m_run8 = mx_cur;
      const float shift = ybias - m_run8;
      off8 = __builtin_fmaf(-MAGIC_F, m8, shift);
#pragma unroll
      for (int i = 0; i < 16; ++i) { y0[i] += shift; y1[i] += shift; }
      mx_cur = ybias;
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");  // K(0) read by every wave; bias of K(2) landed
  }
  for (int blk = blk0; blk < blk1; blk += 2) {
    STEP(0, 1, 1, blk)
    if (blk + 1 >= blk1) break;
    STEP(1, 0, 0, blk + 1)
  }
#undef QK
#undef TO_Y
#undef MAKE_SEEDS
#undef ROW_MAX
#undef ROW_MAX_POS
#undef RAISE_REF
#undef STEP
#undef STAGE_DMA
#undef STEP_SYNC
#undef ROWS_OF
#undef ROW_OF_B
#undef DMA_K
#undef DMA_B
#undef DMA_V

  if (!wave_active) return;
  // ---------------- epilogue ----------------
  const float l_tot = lacc[0];  // every register holds the row's sum
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
        p.ws_ml[slot * 2] = 0.125f * m_run8;  // the reference point in the exp2 domain
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
__global__ __launch_bounds__(NW * 64, 2) void attn_i8_kernel(const ParamsI8 pp) {
#if defined(__HIP_DEVICE_COMPILE__)  // the host pass only needs the launch stub
  __shared__ __attribute__((aligned(16))) char smem[SMEM_I8];
  const int wg = live_order(pp.p, blockIdx.x, gridDim.x, pp.p.xcd_remap);  // XCD-aware order over the live workgroups
  attn_i8_body<T, NW, KVTAB, NSI8>(pp, smem, wg);
#endif
}

// Several launches fused into ONE grid (the experts of a routed layer), as attn_fwd_multi_kernel
template <typename T>
__global__ __launch_bounds__(512, 2) void attn_i8_multi_kernel(const MultiParamsI8 mp) {
#if defined(__HIP_DEVICE_COMPILE__)
  __shared__ __attribute__((aligned(16))) char smem[SMEM_I8];
  const int b = blockIdx.x;
  int s = 0;
#pragma unroll
  for (int i = 1; i < MAX_SEGMENTS; ++i) s += (i < mp.n && b >= mp.start[i]) ? 1 : 0;
  const ParamsI8& pp = mp.seg[s];
  const int wg = live_order(pp.p, b - mp.start[s], mp.start[s + 1] - mp.start[s], true);
  if (pp.p.kv_rows) attn_i8_body<T, 8, true, NSI8>(pp, smem, wg);
  else attn_i8_body<T, 8, false, NSI8>(pp, smem, wg);
#endif
}

// Merge the split-key partials: one wave per (head slot, query position); both partial sums carry the 2^p_bias factor
template <typename T>
__global__ __launch_bounds__(256) void attn_i8_combine_kernel(const ParamsI8 pp) {
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

int fill_i8(const vorta_attn_args* a, const vorta_attn_i8_ext* ext, ParamsI8& pp, int& block_rows) {
  if (!ext || ext->struct_size != sizeof(vorta_attn_i8_ext)) return VORTA_EINVAL;
  if (!a || (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16)) return VORTA_EUNSUPPORTED;
  int rc = fill_params(a, pp.p, block_rows, 2, 1, 1);  // q, o in 16 bits (strides in elements); k int8, v e4m3 (bytes)
  if (rc != VORTA_OK) return rc;
  if (pp.p.n_heads == 0 || pp.p.n_groups == 0) return VORTA_OK;
  if (a->variant == 1) return VORTA_EUNSUPPORTED;  // only the pipelined LDS-DMA body exists
  if (!ext->v_descale || ext->v_descale_stride_h < D || !ext->q_prep || ext->q_prep_stride_h < 2 * D || !ext->k_bias ||
      !ext->k_head_scale)
    return VORTA_EINVAL;
  if (ext->flags != 0) return VORTA_EUNSUPPORTED;
  const float pb = ext->p_bias != 0.f ? ext->p_bias : 5.f;
  const float df = ext->defer != 0.f ? ext->defer : 3.f;
  if (!(pb >= 0.f) || !(df > 0.f) || pb + df > 8.f) return VORTA_EINVAL;  // P' <= 2^(p_bias+defer) must stay below 448
  pp.k_bias = ext->k_bias; pp.k_bias_sh = ext->k_bias_stride_h;
  pp.q_prep = ext->q_prep; pp.q_prep_sh = ext->q_prep_stride_h;
  pp.k_head_scale = ext->k_head_scale;
  pp.v_descale = ext->v_descale;
  pp.v_descale_sh = ext->v_descale_stride_h;
  pp.p_bias = pb;
  pp.thr = pb + df;
  return VORTA_OK;
}

template <typename T>
int launch_i8(const ParamsI8& pp, int block_rows, hipStream_t st) {
  const Params& p = pp.p;
  const int64_t total = (int64_t)p.n_groups * p.blocks_per_group * p.n_heads * p.n_splits;
  if (total <= 0) return VORTA_OK;
  if (total > 0x7fffffff) return VORTA_EINVAL;
  const dim3 g((unsigned)total);
#define LI8(NW_, TAB_) hipLaunchKernelGGL((attn_i8_kernel<T, NW_, TAB_>), g, dim3(NW_ * 64), 0, st, pp)
  if (block_rows == 256) { if (p.kv_rows) LI8(8, true); else LI8(8, false); }
  else { if (p.kv_rows) LI8(4, true); else LI8(4, false); }
#undef LI8
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  if (p.n_splits > 1) {
    const int64_t items = (int64_t)p.n_heads * p.n_q;
    hipLaunchKernelGGL((attn_i8_combine_kernel<T>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, pp);
    e = hipGetLastError();
    if (e != hipSuccess) return vorta_set_hip_error(e);
  }
  return VORTA_OK;
}

}  // namespace

extern "C" int vorta_attn_fwd_i8(const vorta_attn_args* a, const vorta_attn_i8_ext* ext, void* hip_stream) {
  ParamsI8 pp{};
  int block_rows = 0;
  int rc = fill_i8(a, ext, pp, block_rows);
  if (rc != VORTA_OK) return rc;
  if (pp.p.n_heads == 0 || pp.p.n_groups == 0) return VORTA_OK;
  hipStream_t st = (hipStream_t)hip_stream;
  return a->dtype == VORTA_BF16 ? launch_i8<__bf16>(pp, block_rows, st) : launch_i8<_Float16>(pp, block_rows, st);
}

extern "C" int vorta_attn_fwd_batch_i8(const vorta_attn_args* args, const vorta_attn_i8_ext* ext, int32_t n, void* hip_stream) {
  if (!args || !ext || n < 0 || n > MAX_SEGMENTS) return VORTA_EINVAL;
  MultiParamsI8 mp{};
  int64_t total = 0;
  int m = 0;
  int dtype = -1;
  for (int i = 0; i < n; ++i) {
    ParamsI8 pp{};
    int block_rows = 0;
    int rc = fill_i8(&args[i], ext, pp, block_rows);
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
  if (bf) hipLaunchKernelGGL((attn_i8_multi_kernel<__bf16>), dim3((unsigned)total), dim3(512), 0, st, mp);
  else hipLaunchKernelGGL((attn_i8_multi_kernel<_Float16>), dim3((unsigned)total), dim3(512), 0, st, mp);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  for (int i = 0; i < m; ++i) {
    const ParamsI8& pp = mp.seg[i];
    if (pp.p.n_splits > 1) {
      const int64_t items = (int64_t)pp.p.n_heads * pp.p.n_q;
      if (bf) hipLaunchKernelGGL((attn_i8_combine_kernel<__bf16>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, pp);
      else hipLaunchKernelGGL((attn_i8_combine_kernel<_Float16>), dim3((unsigned)((items + 3) / 4)), dim3(256), 0, st, pp);
      e = hipGetLastError();
      if (e != hipSuccess) return vorta_set_hip_error(e);
    }
  }
  return VORTA_OK;
}
