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
// e4m3 kernel's 1.7 x).  Here a score costs TWO instructions between the score MFMA and the P V MFMA (round 6: 1.75 -- the
// conversions two at a time, VORTA_I8_PKNORM below):
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
// Structure otherwise: K tile int8 rows of 128 bytes (8 KiB, the e4m3 kernels' image), V tile e4m3 through ds_read_b64_tr_b8,
// O^T += V8^T P'^T on v_mfma_f32_32x32x64_f8f6f4, row sums from the ones-tile MFMA, deferred rescale, v_descale in the
// epilogue; the loop is the e4m3 kernel's (attn_fwd_fp8.hip): the two waves of a SIMD take turns between a matrix part (13
// MFMAs: per wave and 64-key block 8 of 32 cycles + 5 of 64 = 576 pipe cycles; 1 024 in 16 bits, 832 mixed, 576 all-e4m3)
// and a VALU part.  What bounds it is the SIMD's vector issue: ~92 VALU instructions of ~5.4 cycles and 13 MFMA issues of 8
// per wave and block, two waves = ~1 200 cycles of issue beside 1 152 of matrix pipe, plus the tile requests (~90 cycles
// of the issuing wave each, wherever they stand) and the barrier: tools/trace_i8.py measures ~1 600 cycles per block.
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
constexpr int NSI8 = 2;              // (template tag of the body; the rings are the constants below)
// ONE wait + workgroup barrier per TWO key blocks (round 5).  The wait and the barrier are ~10 % of a step (tools/trace_i8.py:
// ~42 + 100-180 cycles of ~1 700: the waves of a workgroup reach the barrier apart, all wait for the last).  With every tile
// requested TWO steps ahead of its first read and rings of 4 (K, V, key bias) no slot is rewritten inside the pair of steps in
// which another wave may still read it, so the odd steps end without a barrier; slots cycle with period 4 = the unrolling.
// Against round 4's rings of 2 / 3 / 3 with a barrier per key block, same box: Hunyuan-129f fused layer 39.4-40.1 against
// 40.6-41.7 ms (+3 %), Wan-14B-81f 27.5-27.9 against 27.6-28.1 (+0.4 %): profiles/r05_i8_loop_experiments.txt.
constexpr int K_SLOTS_I8 = 4, V_SLOTS_I8 = 4, B_SLOTS_I8 = 4;  // K / V tile rings (8 KiB tiles), key-bias ring (256-B tiles)
constexpr int SC_BYTES = KVB * 4;    // one float per key of a block
constexpr int SEED_BYTES = 8 * 2 * SC_BYTES;  // per wave two slots of 64 int32 seeds
constexpr int SMEM_I8 = (K_SLOTS_I8 + V_SLOTS_I8) * TILE8 + B_SLOTS_I8 * SC_BYTES + SEED_BYTES;  // 69 KiB
constexpr int MAGIC_I = 0x4B400000;  // float bits of 1.5 * 2^23 = 12 582 912
constexpr float MAGIC_F = 12582912.f;
constexpr float SEED_LIMIT = 2000000.f;  // |q8 . k8| <= 2 064 512; the sum must stay below 2^22
// Round 6 (the instruction COUNT of the loop): the 32 byte conversions of a block as 16 v_cvt_pknorm_u16_f32
// (TWO scores per instruction: unorm16(x) = round(clamp(x, 0, 1) * 65535), with 1 / 65535 folded into the multiply-add that makes
// y) + 8 v_perm_b32 (the low bytes of two such registers) = 24 instead of 32 v_cvt_pk_u8_f32.  tools/probe_cvt_pknorm.hip: what it
// rounds to and what it costs; profiles/r06_i8_pknorm.txt: the same-box A/B.
// Rounding: v_cvt_pknorm_u16_f32 gives rint(y) off the ties and rounds ties UP where v_cvt_pk_u8_f32 rounds them to even (0 of
// 133 120 grid values differ off the ties); the folded constant moves 2 of 3 397 bytes by one unit (double rounding next to a tie).
// Same issue cost per instruction (5.1-5.3 cycles, one wave).  Same box, alternating: Wan-14B-81f fused layer 26.35 -> 25.93 ms
// (+1.6 %), Hunyuan-129f 38.10 -> 37.61 (+1.3 %); -DVORTA_I8_PKNORM=0 (variant builds) = the round-5 conversions.
#ifndef VORTA_I8_PKNORM
#define VORTA_I8_PKNORM 1
#endif
constexpr float K65 = 1.f / 65535.f;
typedef __attribute__((ext_vector_type(2))) unsigned short u16x2_t;

struct ParamsI8 {
  Params p;
  const float* k_bias; int64_t k_bias_sh;
  const float* q_prep; int64_t q_prep_sh;
  const float* k_head_scale;
  const float* v_descale; int64_t v_descale_sh;
  float p_bias;  // log2 bias of the probabilities (P' = 2^(z - reference + p_bias))
  float etrig;   // binades: a lane whose block exponent exceeds this moves its row's reference point (rare: the range is the scale's)
};
struct MultiParamsI8 {
  ParamsI8 seg[MAX_SEGMENTS];
  int start[MAX_SEGMENTS + 1];
  int n;
};

__device__ __forceinline__ i32x16 mfma_i8(i32x4 a, i32x4 b, i32x16 c) {
  return __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c, 0, 0, 0);
}
// O^T += V8^T P8^T with block scales on the B operand (v_mfma_scale_f32_32x32x64_f8f6f4, 8-bit operands; tools/probe_mx_scale.hip,
// profiles/r05_probe_mx_scale.txt): scale block s (0, 1) of a column = bytes 16 s ... 16 s + 15 of BOTH lanes of that column, and
// its E8M0 byte (2^(sb - 127), byte 0 of `sb`) is read from the lane of half s.  Here bytes 0-15 of a lane are its 16 keys of
// key tile 0 and bytes 16-31 those of key tile 1: a scale block = one query row x the 32 consecutive keys of one key tile.
// Same cycles as the unscaled form (profiles/r02_probe_fp8_layouts.txt).
constexpr int SC_ONE = 127;  // E8M0 of 2^0: the V operand's scale
#ifdef VORTA_I8_DIAG_NOSCALE
#define ROWSUM_MFMA(a_, b_, c_, s_) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_, b_, c_, 4, 0, 0, 0, 0, 0)
#else
#define ROWSUM_MFMA(a_, b_, c_, s_) __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a_, b_, c_, 4, 0, 0, SC_ONE, 0, s_)  /* A: e2m1 ones */
#endif
__device__ __forceinline__ f32x16 mfma8(i32x8 a, i32x8 b, f32x16 c, int sb) {
#ifdef VORTA_I8_DIAG_NOSCALE  /* timing only (wrong results): what the scale operands cost the MFMA's issue */
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, c, 0, 0, 0, SC_ONE, 0, sb);  // cbsz = blgp = 0: e4m3 x e4m3
#endif
}

template <typename T, int NW, bool KVTAB, int NS>
__device__ __forceinline__ void attn_i8_body(const ParamsI8& pp, char* __restrict__ smem, const int wg) {
  // LDS: K ring (2 tiles) at 0, V ring (3 tiles) behind it, the bias ring (3 tiles of 64 floats) behind that, the waves'
  // seed slots (two of 64 int32 per wave) last.  One workgroup barrier per key block.
  static_assert(NS == 2, "body tag");
  const Params& p = pp.p;
  constexpr int VBASE = K_SLOTS_I8 * TILE8;
  constexpr int BBASE = (K_SLOTS_I8 + V_SLOTS_I8) * TILE8;
  constexpr int SEEDBASE = BBASE + B_SLOTS_I8 * SC_BYTES;
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
    // A wave whose 32 rows all sit on the head's centre (abs-max 0, or below 2^-12: such rows carry no score of their own) keeps
    // the term the centre leaves: q8 = 0 and unit scale, so its scores are u (0 + rint(k_bias)) = softmax of cq . (k - ck), not a
    // uniform row
    const bool flat = !(am >= 0x1p-12f);
    inv_q = flat ? 1.f : 127.f / am;
    const float sq = flat ? 1.f : am * (1.f / 127.f);
    const float q_mul = flat ? 0.f : inv_q;
    m8 = 8.f * ((sq * p.scale_log2) * pp.k_head_scale[head]);
    // both are the same in every lane of the wave: keep them in scalar registers (the loop runs at the VGPR budget)
    m8 = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(m8)));
    inv_q = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(inv_q)));
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        uint32_t word = 0;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          int v = (int)__builtin_rintf(qt[16 * ks + 4 * w + b] * q_mul);
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
  int rowM[CH];            // ... and of the block between them (K runs two blocks ahead of V)
#define ROWS_SHIFT() _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) { rowV[i_] = rowM[i_]; rowM[i_] = rowK[i_]; }
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
  // in the loop the key positions are running values (one add per step); written as (block index) * 64 + ... the six
  // unrolled steps keep six hoisted position registers each
  int posK[CH], posB = 0;
#define ROWS_NEXT()                                                               \
  {                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) {                           \
      posK[i_] += KVB;                                                            \
      const int pos_ = min(posK[i_], n_kv - 1);                                   \
      if constexpr (KVTAB) rowK[i_] = kv_rows[(unsigned)pos_];                    \
      else rowK[i_] = p.kv_row_offset + pos_;                                     \
    }                                                                             \
    if (wave == 0) { /* only wave 0 requests the bias piece: the other waves skip its row arithmetic and table read */ \
      posB += KVB;                                                                \
      const int posb_ = min(posB, n_kv - 1);                                      \
      if constexpr (KVTAB) rowB = kv_rows[(unsigned)posb_];                       \
      else rowB = p.kv_row_offset + posb_;                                        \
    }                                                                             \
  }
#define DMA_K(slot_) _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(    \
      k_rsrc, (LDS_AS void*)(smem + (slot_) * TILE8 + (CH * wave + i_) * 1024), 16,                                 \
      (int)__umul24((unsigned)rowK[i_], (unsigned)k_ss32) + k_col[i_], 0, 0, 0);
#define DMA_B(slot_)                                                                                                \
  if (wave == 0) __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (LDS_AS void*)(smem + BBASE + (slot_) * SC_BYTES), 4, \
                                                          rowB * 4, 0, 0, 0);
#define DMA_V(slot_) _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) __builtin_amdgcn_raw_ptr_buffer_load_lds(    \
      v_rsrc, (LDS_AS void*)(smem + VBASE + (slot_) * TILE8 + (CH * wave + i_) * 1024), 16,                         \
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
  // -- the same P' that multiplies V -- into every register of its result.  The result is a TRANSIENT tile (the MFMA starts
  // from zero) and ONE register of it is added to the running sum l_run: an accumulator tile held 16 registers for one number
  // per lane.  (Reading V fragments ahead of the matrix part with the registers this frees -- two channel tiles at the tail of
  // the VALU part -- measured equal: profiles/r05_i8_loop_experiments.txt.)
  float l_run = 0.f;
  i32x8 ones;
#pragma unroll
  for (int i = 0; i < 8; ++i) ones[i] = 0x22222222;
  asm volatile("" : "+v"(ones));
  // bytes of the probabilities (B operand of the PV MFMAs): two buffers, the matrix part of step j reads the one of block j-1
  // while its score half fills the other with block j
  i32x8 pbA_, pbB_;
#pragma unroll
  for (int i = 0; i < 8; ++i) { pbA_[i] = 0; pbB_[i] = 0; }
  int scA_ = SC_ONE, scB_ = SC_ONE;  // ... and their block scales (E8M0 byte of this lane's 32 probabilities)
  // Online softmax in the BYTE domain y = 8 x + 56, x = log2 P' = z - m_run + p_bias.  m_run8 = 8 x the reference point of
  // this row (its first block's maximum; it moves again only when a block lies more than `etrig` binades above it).
  // MX-SCALED PROBABILITIES (round 5): the e4m3 byte of a probability is rint(y - 8 e) with e = the block exponent of its
  // (query row, key tile of 32 keys), rint((max y of those 32 - YMID) / 8): whatever the distance of a tile from the row's
  // reference point, its largest probability sits in the top binade of e4m3 (bytes 116 ... 124 of 126) and 2^e goes to the MFMA
  // as the B operand's block scale -- no probability is flushed because its ROW has a larger one elsewhere.  (One exponent range for the whole
  // row, as before round 5: everything below 2^-14 ... 2^-11 of the row's maximum was zero or a mis-decoded subnormal -- the
  // byte domain is logarithmic, e4m3's subnormals are linear -- 0.14-0.21 relative error on heavy-tailed inputs, all of it
  // from the probabilities' range: profiles/r05_mx_probabilities.txt.)
  float m_run8 = -1e30f;
  // The exponent lives as a BIASED float eb = 1.5 2^23 + 127 + e: in that binade a float is an integer, so ONE multiply-add
  // (max y / 8 + CE) rounds to it, and its bit pattern's low byte IS the E8M0 scale byte 127 + e -- the register goes to the MFMA
  // as it stands.  e has no upper clamp: a tile far above the reference point (heavy-tailed scores reach thousands of binades)
  // trips the trigger below and the reference point moves under it BEFORE its bytes are used.
  constexpr float YMID = 120.f;                     // a tile's largest byte-domain value lands in [YMID - 4, YMID + 4]
  constexpr float EBIAS = MAGIC_F + 127.f;          // eb - EBIAS = e
  constexpr float CE = EBIAS - 0.125f * YMID;       // an integer: eb = rint(max y / 8 + CE), exactly one rounding
  constexpr float EB_MIN = EBIAS - 100.f;           // e >= -100: 2^-100 of the row's reference is nothing, and a finite scale
  const float ybias = 8.f * pp.p_bias + 56.f, etrig_b = EBIAS + pp.etrig;
  f32x16 y0, y1;          // byte-domain scores (minus 8 e) of one key block (keys 0-31, 32-63 of the block)
  i32x16 n0, n1;          // raw accumulators of the block after it (written by the matrix part)
  float off8 = 0.f;       // ybias - m_run8 - MAGIC_F * m8: the addend of the conversion
#if VORTA_I8_PKNORM
  constexpr float YK_ = K65;           // the conversions take y / 65535 ...
  const float m8n = m8 * K65;          // ... made by the same multiply-add with both constants divided
  float off8n = 0.f;
#else
  constexpr float YK_ = 1.f;
  const float m8n = m8;
#define off8n off8
#endif
  float offp0 = 0.f, offp1 = 0.f;  // the two tiles' conversion offsets of the block whose scores are in n0, n1
  float ecur = EBIAS;     // biased block exponent of key tile hh (the one whose scale this lane supplies) for the block in y0, y1

  // the seed of a key = rint(bias / sq) as the integer the accumulator starts from, 0x4B400000 + seed: ONE multiply-add -- in the
  // binade of 1.5 2^23 a float is an integer, so bias * inv_q + 1.5 2^23 rounds (once, to nearest even) to the float whose BITS
  // are that integer -- and an integer clamp of the bits (monotonic there; a product beyond the binade clamps too)
#define SEED_BITS(b_) min(max(__float_as_int(__builtin_fmaf((b_), inv_q, MAGIC_F)), MAGIC_I - (int)SEED_LIMIT), MAGIC_I + (int)SEED_LIMIT)
  // seeds of a block from its bias tile (slot bslot_) into this wave's seed slot sslot_: two VALU instructions per wave
#define MAKE_SEEDS(bslot_, sslot_)                                                \
  {                                                                               \
    const float b_ = *(const float*)(smem + bias_rd + (bslot_) * SC_BYTES);       \
    *(int*)(smem + seed_wr + (sslot_) * SC_BYTES) = SEED_BITS(b_);                \
  }
  // the seeds of key tile t_ (0, 1) of a block (seed slot sslot_) as an accumulator's initial value: int32 bits = float
  // 1.5 2^23 + seed
#define SEEDS_IN(d_, sslot_, t_)                                                  \
  _Pragma("unroll") for (int j_ = 0; j_ < 4; ++j_) {                              \
    const i32x4 s_ = *(const i32x4*)(smem + seed_rd + (sslot_) * SC_BYTES + 128 * (t_) + 32 * j_); \
    _Pragma("unroll") for (int e_ = 0; e_ < 4; ++e_) d_[4 * j_ + e_] = s_[e_];    \
  }
  // K fragments of key tile t_ (0, 1) of ring slot slot_
#define KFRAGS(dst_, slot_, t_)                                                   \
  _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_)                             \
    dst_[ks_] = *(const i32x4*)(smem + (slot_) * TILE8 + (t_) * 32 * ROWB8 + k_rd[ks_]);
#define QK_TILE(d_, kf_) _Pragma("unroll") for (int ks_ = 0; ks_ < 4; ++ks_) d_ = mfma_i8(kf_[ks_], qf[ks_], d_);
  // int32 bits read as floats -> byte-domain scores: y = (1.5 2^23 + score) m8 + off_  (one fused multiply-add per score)
#define TO_Y(yd0_, yd1_, a0_, a1_, off_)                                          \
  {                                                                               \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) yd0_[i_] = __builtin_fmaf(__int_as_float(a0_[i_]), m8, off_); \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) yd1_[i_] = __builtin_fmaf(__int_as_float(a1_[i_]), m8, off_); \
  }
  // P' bytes straight from y - 8 e: rint, saturating at 0 (-inf of masked keys -> 0); the lane's largest is below 126.5
#if VORTA_I8_PKNORM
  // y0 / y1 hold (y - 8 e) / 65535: two per v_cvt_pknorm_u16_f32, the four low bytes of two results by one v_perm_b32
#define PKN_(a_, b_) __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pknorm_u16(a_, b_))
#define PACK_Y(pb_)                                                               \
  _Pragma("unroll") for (int w_ = 0; w_ < 4; ++w_) {                              \
    pb_[w_] = (int)__builtin_amdgcn_perm(PKN_(y0[4 * w_ + 2], y0[4 * w_ + 3]), PKN_(y0[4 * w_], y0[4 * w_ + 1]), 0x06040200u); \
    pb_[4 + w_] = (int)__builtin_amdgcn_perm(PKN_(y1[4 * w_ + 2], y1[4 * w_ + 3]), PKN_(y1[4 * w_], y1[4 * w_ + 1]), 0x06040200u); \
  }
#else
#define PACK_Y(pb_)                                                               \
  _Pragma("unroll") for (int w_ = 0; w_ < 4; ++w_)                                \
    _Pragma("unroll") for (int b_ = 0; b_ < 4; ++b_) {                            \
      pb_[w_] = __builtin_amdgcn_cvt_pk_u8_f32(y0[4 * w_ + b_], b_, pb_[w_]);     \
      pb_[4 + w_] = __builtin_amdgcn_cvt_pk_u8_f32(y1[4 * w_ + b_], b_, pb_[4 + w_]); \
    }
#endif
#define ROW_MAX(dst_, a_, b_)                                                      \
  {                                                                               \
    float mx_ = a_[0];                                                            \
    _Pragma("unroll") for (int i_ = 1; i_ < 16; ++i_) mx_ = fmaxf(mx_, a_[i_]);   \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) mx_ = fmaxf(mx_, b_[i_]);   \
    dst_ = half_max(mx_);                                                         \
  }
  // The block exponents from the RAW accumulators (their bits read as floats are positive and ordered like the integers):
  // signed-integer max over each key tile's 16 registers (v_max3_i32), ONE half exchange that leaves tile 0's row max in the
  // lanes of half 0 and tile 1's in half 1 -- where the MFMA reads the two scales -- one multiply-add to the byte domain, one
  // to the biased exponent; a second exchange hands every lane both exponents for the offsets of its conversions.
#define TILE_MAXES(a_, b_)                                                        \
    int m0_ = max(max(a_[0], a_[1]), a_[2]);                                      \
    int m1_ = max(max(b_[0], b_[1]), b_[2]);                                      \
    _Pragma("unroll") for (int i_ = 3; i_ < 15; i_ += 2) {                        \
      m0_ = max(max(m0_, a_[i_]), a_[i_ + 1]);                                    \
      m1_ = max(max(m1_, b_[i_]), b_[i_ + 1]);                                    \
    }                                                                             \
    m0_ = max(m0_, a_[15]);                                                       \
    m1_ = max(m1_, b_[15]);
#define TILE_EXP(eh_, off0_, off1_, a_, b_)                                       \
  {                                                                               \
    TILE_MAXES(a_, b_)                                                            \
    auto r_ = __builtin_amdgcn_permlane32_swap((unsigned)m0_, (unsigned)m1_, false, false); \
    const int mt_ = max((int)r_[0], (int)r_[1]); /* half 0: tile 0 over both halves' keys; half 1: tile 1 */ \
    const float ymx_ = __builtin_fmaf(__int_as_float(mt_), m8, off8);             \
    eh_ = fmaxf(__builtin_fmaf(ymx_, 0.125f, CE), EB_MIN);                        \
    const float ef_ = eh_ - EBIAS;                                                \
    auto e_ = __builtin_amdgcn_permlane32_swap(__float_as_uint(ef_), __float_as_uint(ef_), false, false); \
    off0_ = __builtin_fmaf(__uint_as_float(e_[0]), -8.f * YK_, off8n);            \
    off1_ = __builtin_fmaf(__uint_as_float(e_[1]), -8.f * YK_, off8n);            \
  }
  // (the empty asm makes the lane term a value of THIS step: otherwise the 16 sums lane term + register row are hoisted out
  // of the loop into 16 registers for a branch taken once per workgroup)
#define MASK_TAIL(jabs_)                                                          \
  {                                                                               \
    int h4_ = 4 * hh;                                                             \
    asm volatile("" : "+v"(h4_));                                                 \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) {                           \
      const int row_ = (i_ & 3) + 8 * (i_ >> 2) + h4_;                            \
      if ((jabs_) * KVB + row_ >= n_kv) y0[i_] = -INFINITY;                       \
      if ((jabs_) * KVB + 32 + row_ >= n_kv) y1[i_] = -INFINITY;                  \
    }                                                                             \
  }
#define VFRAG_TO(dst_, dt_, slot_)                                                \
  _Pragma("unroll") for (int n_ = 0; n_ < 4; ++n_) {                              \
    const i32x2 t_ = __builtin_amdgcn_ds_read_tr8_b64_v2i32(                      \
        (LDS_AS i32x2*)(smem + (slot_) * TILE8 + v_rd[dt_] + n_ * 16 * ROWB8));   \
    dst_[2 * n_] = t_[0]; dst_[2 * n_ + 1] = t_[1];                               \
  }
#define VFRAG(dt_, slot_) VFRAG_TO(vf_[dt_], dt_, slot_)
  // requests of step j: K(j+3), V(j+1) and the bias tile of K(j+4), each into the slot whose tile was last read two steps ago and
  // each read two steps later at the earliest (the slot arithmetic is at the loop below)
#define STAGE_DMA(kw_, vw_, bw_, jabs_)                                           \
  DMA_K(kw_)                                                                      \
  DMA_V(vw_)                                                                      \
  DMA_B(bw_)                                                                      \
  ROWS_SHIFT()                                                                    \
  ROWS_NEXT() /* the rows of the next K block and of the next bias tile */        \
  __builtin_amdgcn_sched_barrier(0);
#define STEP_SYNC() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#define STEP_NOSYNC() asm volatile("" ::: "memory");  /* (paired steps: the odd step ends without wait or barrier) */

  // The two waves of a SIMD (wave w and w + NW/2 of the workgroup) take turns on its pipes, as in attn_fwd_fp8.hip: while
  // one runs its matrix part -- 13 MFMAs back to back: O += V^T P'^T and the row sums of one block (5 x 64 cycles), the raw
  // scores of a later one (8 x 32) -- the other runs its VALU part -- the 32 byte conversions of a block, the 32
  // multiply-adds and the row max of the next, the seeds of a later one (~85 instructions).  (With both waves in the same
  // part at the same time -- the first build of this kernel -- the MFMA pipe was 55 % busy at 2.0 GHz and removing ALL
  // MFMAs shortened the step by only 23 %: profiles/r04_i8_ablation.txt.)
  //   step j, role X (waves < NW/2) :  matrix(j)  then  valu: y / max of block j+1, seeds of block j+2
  //   step j, role Y (waves >= NW/2):  valu: y / max of block j, seeds of block j+1   then  matrix(j)
  //   matrix(j) = PV(j-1) + row sums, [mask the tail of block j], [move the reference point for block j], QK(j+1) with the
  //               byte conversions of block j in its MFMA gaps
  // Both roles see the same reference points and run the same barriers.  kr_ / vr_: K / V slots read (K(j+1), V(j-1));
  // sr_: seed slot of block j+1;  (bs_, ss_): bias and seed slot of the block whose seeds the VALU part makes.
  // the VALU part issues, first thing, the LDS read whose round trip would otherwise be exposed at its end: the bias tile
  // entry of its seed arithmetic.  (Reading the V fragments of channel tiles 0, 1 of the NEXT matrix part here as well --
  // 16 registers carried across the barrier by role X -- spills ~40 registers at this budget; the matrix part hides most
  // of that round trip under the row-sum MFMA instead, which needs no fragment.)
  // The wave's tile requests are spread over the VALU part (qa_, qb_, qc_: the K piece, the V piece, the bias piece + the
  // next rows): back to back at the top of the step they cost the wave ~100 cycles each -- every wave of the workgroup
  // asks at the same moment and the address unit takes 16 pieces per step, 16 cycles apiece -- with VALU work between
  // them the queue has drained when the next one comes (tools/trace_i8.py: 176-208 cycles per step for the requests).
#define VALU_FMA()                                                                \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) y0[i_] = __builtin_fmaf(__int_as_float(n0[i_]), m8n, offp0); \
    _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) y1[i_] = __builtin_fmaf(__int_as_float(n1[i_]), m8n, offp1);
#define VALU_PART(bs_, ss_, qa_, qb_, qc_)                                        \
  {                                                                               \
    const float sb_ = *(const float*)(smem + bias_rd + (bs_) * SC_BYTES);         \
    qa_                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                            \
    TILE_EXP(ecur, offp0, offp1, n0, n1)                                          \
    __builtin_amdgcn_sched_barrier(0);                                            \
    qb_                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                            \
    VALU_FMA()                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                            \
    qc_                                                                           \
    __builtin_amdgcn_sched_barrier(0);                                            \
    *(int*)(smem + seed_wr + (ss_) * SC_BYTES) = SEED_BITS(sb_);                  \
  }
#define ROWS_UPDATE()                                                             \
  ROWS_SHIFT()                                                                    \
  ROWS_NEXT()
#define REQ_TAIL(bw_)                                                             \
  DMA_B(bw_)                                                                      \
  ROWS_UPDATE()
#ifndef VORTA_I8_SCHED
#define VORTA_I8_SCHED 1
#endif
  // an empty, NON-volatile asm over two values: a data dependence and nothing else (reads stay free to move) -- orders the MFMAs
  // that produce / consume them where the recipe's greedy pick would not
#define TIE_(a_, b_) asm("" : "+v"(a_), "+v"(b_));
#if VORTA_I8_SCHED == 1
  // Issue-order recipe of the PV half of the matrix part (sched_group_barrier: 0x008 MFMA, 0x100 DS read, 0x002 VALU): the
  // reads of V channel tiles 0, 1, then 5 MFMAs -- the row-sum MFMA FIRST: it needs no fragment and covers 64 cycles of the
  // tiles' round trip -- with the later reads under the earlier ones: V tiles 2, 3, the seeds and the K fragments of key tile
  // 0, so that the score half starts with its operands in registers; and under each of the first four MFMAs eight of the 32
  // byte conversions of block j (round 5: they sat under the score MFMAs, four per 32-cycle gap, where the partner wave's VALU
  // part found the SIMD's vector issue three quarters taken; under the 64-cycle P V MFMAs they take a third of it, and the
  // score half leaves the issue to the partner: +2 %, profiles/r05_i8_loop_experiments.txt 9; with the multiply-adds here as
  // well the P V half outgrows its MFMAs: -2 ... -3.5 %)
#define SG_(mask_, n_) __builtin_amdgcn_sched_group_barrier(mask_, n_, 0);
#if VORTA_I8_PKNORM
#define NCVT_ 6 /* 16 v_cvt_pknorm_u16_f32 + 8 v_perm_b32 */
#else
#define NCVT_ 8 /* 32 v_cvt_pk_u8_f32 */
#endif
#define SCHED_M()                                                                 \
  SG_(0x100, 8)                                                                   \
  SG_(0x008, 1) SG_(0x100, 4) SG_(0x002, NCVT_)                                   \
  SG_(0x008, 1) SG_(0x100, 4) SG_(0x002, NCVT_)                                   \
  SG_(0x008, 1) SG_(0x100, 4) SG_(0x002, NCVT_)                                   \
  SG_(0x008, 1) SG_(0x100, 4) SG_(0x002, NCVT_)                                   \
  SG_(0x008, 1)
  // score half (a basic block of its own behind the rare branches): the eight reads of key tile 1 under the four MFMAs of
  // tile 0 (whose operands are in registers: the wave goes from the branch straight into an MFMA), then tile 1
#define SCHED_S()                                                                 \
  SG_(0x008, 1) SG_(0x100, 2) SG_(0x008, 1) SG_(0x100, 2) SG_(0x008, 1) SG_(0x100, 2) SG_(0x008, 1) SG_(0x100, 2) \
  SG_(0x008, 4)
#else
#define SCHED_M()
#define SCHED_S()
#endif
#ifndef VORTA_I8_PRIO
#define VORTA_I8_PRIO 1  /* s_setprio around the matrix part (the partner wave is in its VALU part then) */
#endif
#if VORTA_I8_PRIO == 1
#define PRIO_HI() __builtin_amdgcn_s_setprio(2);
#define PRIO_LO() __builtin_amdgcn_s_setprio(0);
#else
#define PRIO_HI()
#define PRIO_LO()
#endif
  // (tile requests INSIDE the matrix part, each behind an MFMA, measured 1.3 % slower than in the VALU part: round 4)
#define ZERO16_ f32x16{0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}
#define PV_PART(vr_, pb_, sc_, mid_)                                              \
  i32x8 vf_[4];                                                                   \
  VFRAG(0, vr_) VFRAG(1, vr_) VFRAG(2, vr_) VFRAG(3, vr_)                         \
  f32x16 lt_ = ROWSUM_MFMA(ones, pb_, ZERO16_, sc_);                              \
  TIE_(lt_, vf_[0]) /* the row-sum MFMA before the first that needs a fragment */ \
  mid_ /* the score half's first reads */                                         \
  o[0] = mfma8(vf_[0], pb_, o[0], sc_);                                           \
  TIE_(o[0], vf_[1])                                                              \
  o[1] = mfma8(vf_[1], pb_, o[1], sc_);                                           \
  o[2] = mfma8(vf_[2], pb_, o[2], sc_);                                           \
  TIE_(o[2], vf_[3])                                                              \
  o[3] = mfma8(vf_[3], pb_, o[3], sc_);
  // move the reference point of a row up by g_ >= 0 WHOLE binades (the larger of its two tiles' block exponents): everything
  // accumulated so far follows; the current block's bytes do not change -- y - 8 e is what it was -- only its exponents do
#define RAISE_REF()                                                               \
  {                                                                               \
    const float g_ = fmaxf(half_max(ecur) - EBIAS, 0.f);                          \
    const float alpha_ = __builtin_amdgcn_exp2f(-g_);                             \
    _Pragma("unroll") for (int dt_ = 0; dt_ < 4; ++dt_)                           \
      _Pragma("unroll") for (int i_ = 0; i_ < 16; ++i_) o[dt_][i_] *= alpha_;     \
    l_run = (l_run + lt_[0]) * alpha_; /* block j-1 went in at the old reference */ \
    lt_[0] = 0.f;                                                                 \
    m_run8 = __builtin_fmaf(g_, 8.f, m_run8);                                     \
    off8 = __builtin_fmaf(g_, -8.f, off8);                                        \
    RAISE_OFF8N_(g_)                                                              \
    ecur = fmaxf(ecur - g_, EB_MIN);                                              \
  }
#if VORTA_I8_PKNORM
#define RAISE_OFF8N_(g_) off8n = __builtin_fmaf(g_, -8.f * K65, off8n);
#else
#define RAISE_OFF8N_(g_)
#endif
#define MATRIX_PART(kr_, vr_, sr_, pbr_, scr_, pbw_, scw_, jabs_)                 \
  {                                                                               \
    PRIO_HI()                                                                     \
    i32x4 kfa_[4], kfb_[4];                                                       \
    PV_PART(vr_, pbr_, scr_, PACK_Y(pbw_) SEEDS_IN(n0, sr_, 0) KFRAGS(kfa_, kr_, 0)) /* + the bytes of block j */ \
    SCHED_M()                                                                     \
    /* the five MFMAs stay one run AHEAD of the rare branches: left alone, the compiler sinks one that nothing orders below the */ \
    /* tail-mask branch, out of the issue recipe (an empty statement: no instruction, no wait)                              */ \
    asm volatile("" : "+v"(lt_), "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(pbw_)); \
    /* last, partial key block: mask its tail and convert again (once per workgroup; the exponents were taken over the clamped rows too) */ \
    if ((jabs_) * KVB + KVB > n_kv) { MASK_TAIL(jabs_) PACK_Y(pbw_) }             \
    /* the reference point moves only when some tile lies more than `etrig` binades above it; O and the row sums */ \
    /* follow AFTER block j-1 went in at the old reference                                                              */ \
    if (!__all(ecur <= etrig_b)) { RAISE_REF() }                                  \
    scw_ = __float_as_int(ecur); /* its low byte is the E8M0 scale 127 + e */     \
    SEEDS_IN(n1, sr_, 1) /* key tile 1: its seeds and fragments fly under the four MFMAs of tile 0 */ \
    KFRAGS(kfb_, kr_, 1)                                                          \
    QK_TILE(n0, kfa_)                                                             \
    TIE_(n0, n1) /* tile 0 (operands in registers) before tile 1 (operands in flight) */ \
    QK_TILE(n1, kfb_)                                                             \
    SCHED_S()                                                                     \
    l_run += lt_[0]; /* (the row-sum MFMA finished long ago) */                   \
    PRIO_LO()                                                                     \
  }
  // Diagnostic builds only (suffixed libraries: vorta_amd/build.py refuses extra flags for the product): -DVORTA_I8_DIAG pulls
  // in the in-loop cycle stamps (-DVORTA_TRACE_I8=i, tools/trace_i8.py) and the wrong-result timing ablations
  // (-DVORTA_I8_DIAG_*), which re-define the macros above.
// which half of the workgroup starts its steps with the VALU part (the e4m3 kernel measured 0-3.6 % between the two, by body;
// here the later-dispatched half: 26.5 against 26.8-27.2 ms, profiles/r04_i8_ablation.txt part 6)
#define ROLE_Y_ (NW == 8 && wave >= NW / 2)
#define TR_(i_)
#define TR_FLUSH_()
#ifdef VORTA_I8_DIAG
#include "attn_fwd_i8_diag.inc"
#endif
  // both roles request their tile pieces inside their VALU part (role X behind its matrix part: it goes from the barrier
  // straight into its MFMAs); the 4-wave kernels (no roles) and waves past the query rows request at the top of the step
#define STEP(kw_, kr_, vw_, vr_, bw_, sr_, bsx_, ssx_, bsy_, ssy_, pbr_, scr_, pbw_, scw_, jabs_, sync_) \
  {                                                                               \
    TR_(0)                                                                        \
    if (!wave_active || NW != 8) { STAGE_DMA(kw_, vw_, bw_, jabs_) }              \
    TR_(1)                                                                        \
    if (wave_active) {                                                            \
      if (NW == 8 && role_y) { VALU_PART(bsy_, ssy_, DMA_K(kw_), DMA_V(vw_), REQ_TAIL(bw_)) } \
      __builtin_amdgcn_sched_barrier(0);                                          \
      TR_(2)                                                                      \
      MATRIX_PART(kr_, vr_, sr_, pbr_, scr_, pbw_, scw_, jabs_)                   \
      __builtin_amdgcn_sched_barrier(0);                                          \
      TR_(3)                                                                      \
      TR_(4)                                                                      \
      if (NW == 8 && !role_y) { VALU_PART(bsx_, ssx_, DMA_K(kw_), DMA_V(vw_), REQ_TAIL(bw_)) } \
      if (NW != 8) VALU_PART(bsx_, ssx_, , , )                                    \
    }                                                                             \
    sync_()                                                                       \
  }

  const int nsteps = blk1 - blk0;
  const bool role_y = ROLE_Y_;  // wave-uniform
  if (nsteps > 0) {
    // ---- prologue: K(0), K(1), K(2), V(0) and the bias tiles of K(0) ... K(3); the scores of block 0 fix the reference point ----
    ROWS_OF(rowK, blk0)
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) rowV[i_] = rowK[i_];
    ROW_OF_B(blk0)
    DMA_K(0)
    DMA_V(0)
    DMA_B(0)
    ROWS_OF(rowK, blk0 + 1)
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) rowV[i_] = rowK[i_];  // rows of block 1: V(1) goes out in step 0
    ROW_OF_B(blk0 + 1)
    DMA_K(1)
    DMA_B(1)
    ROWS_OF(rowK, blk0 + 2)
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) rowM[i_] = rowK[i_];  // rows of block 2: V(2) in step 1
    ROW_OF_B(blk0 + 2)
    DMA_K(2)
    DMA_B(2)
    ROW_OF_B(blk0 + 3)
    DMA_B(3)
    ROWS_OF(rowK, blk0 + 3)  // K(3) in step 0
    ROW_OF_B(blk0 + 4)
    _Pragma("unroll") for (int i_ = 0; i_ < CH; ++i_) posK[i_] = (blk0 + 3) * KVB + 8 * (CH * wave + i_) + (lane >> 3);
    posB = (blk0 + 4) * KVB + lane;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (wave_active) {
      MAKE_SEEDS(0, 0)
      MAKE_SEEDS(1, 1)
      i32x4 kfa[4], kfb[4];
      SEEDS_IN(n0, 0, 0)
      SEEDS_IN(n1, 0, 1)
      KFRAGS(kfa, 0, 0)
      KFRAGS(kfb, 0, 1)
      QK_TILE(n0, kfa)
      QK_TILE(n1, kfb)
      const float base = -MAGIC_F * m8;
      TO_Y(y0, y1, n0, n1, base)  // 8 x the plain exp2-domain scores of the first block
      if (blk0 * KVB + KVB > n_kv) { MASK_TAIL(blk0) }
      float mx0;
      ROW_MAX(mx0, y0, y1)
      // the first block fixes the reference point at its true row max (block blk0 always has a valid key);
      // O and l are still zero, so nothing is rescaled
      m_run8 = mx0;
      const float shift = ybias - m_run8;
      off8 = __builtin_fmaf(-MAGIC_F, m8, shift);
#if VORTA_I8_PKNORM
      off8n = off8 * K65;
#endif
      // ... and its two key tiles take their block exponents from their own (masked) values
      float l0 = y0[0], l1 = y1[0];
#pragma unroll
      for (int i = 1; i < 16; ++i) { l0 = fmaxf(l0, y0[i]); l1 = fmaxf(l1, y1[i]); }
      auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(l0), __float_as_uint(l1), false, false);
      const float lt = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));  // half 0: tile 0, half 1: tile 1
      ecur = fmaxf(__builtin_fmaf(lt + shift, 0.125f, CE), EB_MIN);
      const float ef = ecur - EBIAS;
      auto e = __builtin_amdgcn_permlane32_swap(__float_as_uint(ef), __float_as_uint(ef), false, false);
      const float sh0 = __builtin_fmaf(__uint_as_float(e[0]), -8.f, shift), sh1 = __builtin_fmaf(__uint_as_float(e[1]), -8.f, shift);
#pragma unroll
      for (int i = 0; i < 16; ++i) { y0[i] = (y0[i] + sh0) * YK_; y1[i] = (y1[i] + sh1) * YK_; }
    }
    __syncthreads();  // every wave has read K(0) and the bias tile of K(0) before their slots are overwritten
    {  // step 0: no PV yet -- the scores of block 1, then (role X) the VALU part of block 0
      STAGE_DMA(3, 1, 0, blk0) /* K(3), V(1), the bias tile of K(4) */
      if (wave_active) {
        i32x4 kfa[4], kfb[4];
        SEEDS_IN(n0, 1, 0)
        SEEDS_IN(n1, 1, 1)
        KFRAGS(kfa, 1, 0)
        KFRAGS(kfb, 1, 1)
        QK_TILE(n0, kfa)
        QK_TILE(n1, kfb)
        PACK_Y(pbA_)  // the bytes of block 0
        scA_ = __float_as_int(ecur);
        __builtin_amdgcn_sched_barrier(0);
        if (!role_y) VALU_PART(2, 0, , , )
        else { MAKE_SEEDS(2, 0) }  /* (role Y makes its seeds two blocks ahead too, see below) */
      }
      STEP_SYNC()
    }
    // Every ring has period 4 (seeds and byte buffers 2): unrolled by 4.  Step j: requests K(j+3) -> slot (j+3) % 4, V(j+1) ->
    // (j+1) % 4, bias(j+4) -> j % 4 -- each read two steps later at the earliest, so a barrier (end of every EVEN step; step 0
    // ends with one) lies between request and read, and each into a slot last read two steps ago, before the previous barrier;
    // reads K(j+1) from (j+1) % 4, V(j-1) from (j-1) % 4, the seeds of block j+1 from (j+1) % 2; BOTH roles make the seeds of block
    // j+2 (bias slot (j+2) % 4, seed slot j % 2): role Y, whose VALU part comes first, works one block further ahead than it needs
    // so that no wave reads a bias tile in the pair of steps in which it is rewritten.
    for (int jj = 1; jj < nsteps; jj += 4) {
      STEP(0, 2, 2, 0, 1, 0, 3, 1, 3, 1, pbA_, scA_, pbB_, scB_, blk0 + jj, STEP_NOSYNC)
      if (jj + 1 >= nsteps) break;
      STEP(1, 3, 3, 1, 2, 1, 0, 0, 0, 0, pbB_, scB_, pbA_, scA_, blk0 + jj + 1, STEP_SYNC)
      if (jj + 2 >= nsteps) break;
      STEP(2, 0, 0, 2, 3, 0, 1, 1, 1, 1, pbA_, scA_, pbB_, scB_, blk0 + jj + 2, STEP_NOSYNC)
      if (jj + 3 >= nsteps) break;
      STEP(3, 1, 1, 3, 0, 1, 2, 0, 2, 0, pbB_, scB_, pbA_, scA_, blk0 + jj + 3, STEP_SYNC)
    }
    // a loop left behind an odd step has not waited for that step's requests and has no barrier behind it: the drain reads V of
    // the last block (requested a step earlier), and the tiles still in flight must land before the workgroup's LDS is released
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // ---- drain: PV of the last block ----
    if (wave_active) {
      const int vs = (nsteps - 1) % V_SLOTS_I8;
      if ((nsteps - 1) & 1) { pbA_ = pbB_; scA_ = scB_; }  // the bytes of the last block
      PV_PART(vs, pbA_, scA_, )
      l_run += lt_[0];
    }
  }
#undef MAKE_SEEDS
#undef SEED_BITS
#undef SEEDS_IN
#undef KFRAGS
#undef QK_TILE
#undef TO_Y
#undef PACK_Y
#undef ROW_MAX
#undef TILE_EXP
#undef MASK_TAIL
#undef VFRAG
#undef VFRAG_TO
#undef STAGE_DMA
#undef STEP_SYNC
#undef STEP_NOSYNC
#undef ROWS_SHIFT
#undef VALU_PART
#undef REQ_TAIL
#undef ROWS_UPDATE
#undef SCHED_M
#undef SCHED_S
#undef TIE_
#undef PRIO_HI
#undef PRIO_LO
#undef PV_PART
#undef RAISE_REF
#undef RAISE_OFF8N_
#undef PKN_
#undef NCVT_
#if !VORTA_I8_PKNORM
#undef off8n
#endif
#undef MATRIX_PART
#undef ROLE_Y_
#undef STEP
  TR_FLUSH_()
#undef TR_
#undef TR_FLUSH_
#undef ROWS_OF
#undef ROWS_NEXT
#undef ROW_OF_B
#undef DMA_K
#undef DMA_B
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
  // p_bias: P' = 2^(score - reference + p_bias) (any value: the block scales carry the range); defer: binades a lane's block
  // may lie above its row's reference point before the reference moves (fp32 accumulators: P' <= 2^(defer + 9))
  const float pb = ext->p_bias != 0.f ? ext->p_bias : 5.f;
  const float df = ext->defer != 0.f ? ext->defer : 24.f;
  if (!(pb >= 0.f) || pb > 16.f || !(df >= 0.f) || df > 64.f) return VORTA_EINVAL;
  pp.k_bias = ext->k_bias; pp.k_bias_sh = ext->k_bias_stride_h;
  pp.q_prep = ext->q_prep; pp.q_prep_sh = ext->q_prep_stride_h;
  pp.k_head_scale = ext->k_head_scale;
  pp.v_descale = ext->v_descale;
  pp.v_descale_sh = ext->v_descale_stride_h;
  pp.p_bias = pb;
  pp.etrig = df;
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
