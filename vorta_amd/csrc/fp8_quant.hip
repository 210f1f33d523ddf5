// vorta_fp8_quantize_qkv (include/vorta_hip.h): post-RoPE q,k,v (bf16 / fp16) -> e4m3 copies for the fp8 attention
// kernels, with the softmax scale and log2(e) folded into the q/k multipliers.  HBM-bound: the abs-max pass reads
// every element once (2 B), the convert pass reads it again and writes 1 B.  No host round trip.
//
// Key centring (flags bit1): softmax over the keys of a head does not change when one vector is subtracted from all of
// them (q . (k - c) = q . k - q . c, a per-query constant), but the e4m3 error of q8 . k8 grows with |k|, so a common
// component of the keys -- a per-channel mean, usual in trained attention layers -- costs precision for nothing
// (tools/dbg/fp8_kbias.py: a mean of 3 / 8 standard deviations costs 6 / 12 dB of output PSNR; centring gives all of it
// back).  ANY vector c works, so c[h][d] is the mean of <= ~1024 evenly spaced key TOKENS of the head (the same tokens in
// either row layout below): one small launch, fixed summation order (deterministic), no extra pass over K.
// The sample is summed in SAMPLE_CHUNKS equal ranges of the head's video tokens plus one slot for its tail (text) tokens,
// one workgroup each, combined in slot order: a sequence shard that holds whole chunks computes exactly the partials the
// whole sequence would, so P ranks (P | SAMPLE_CHUNKS) that each hold Sv/P tokens of every head -- the send side of the
// Ulysses exchange -- add their partial tables (flags bit3, disjoint slots: the sum is exact) and convert their shards
// (flags bit4) to the bytes one call over the assembled sequence writes: q and k cross the links as e4m3.
//
// Row layouts: (heads, n_tokens, D) views (seg_len = 0), or ONE row array of n_tokens rows in which row r belongs to head
// (r / seg_len) % heads (seg_len > 0: the Ulysses receive buffer, ulysses/engine.py -- each rank head keeps its own
// scales and centre although the head views of that buffer overlap); from row tail_first on only the first tail_len
// rows of a segment hold data (the text rows behind each head slot), the rest is skipped; slot_first / slot_count
// restrict a call to some head slots (the slot group whose exchange has landed).
//
// Abs-max of q and k: the multipliers of q and k only BALANCE the two operand ranges (qmul * kmul = c0 whatever they are;
// both maxima land near sqrt(c0 amax_q amax_k), a factor ~250 inside e4m3's range for unit-variance data), so their abs-max
// is taken over the same ~1024 sampled tokens as the centre instead of a full pass over q and k: a true maximum a few
// times the sampled one moves nothing but that headroom (values are clamped to +-448 before the conversion in any case).
// v is scaled per channel to a target just under the format's maximum, so ITS abs-max is exact (one pass over v).
//
// Workspace (floats): amax_q[H] | amax_k[H] | amax_v[H][D] | qmul[H] | kmul[H] | vmul[H][D] | kmean[H][D] |
//                     ksum[H][MEAN_BLOCKS][D] | kcnt[H][MEAN_BLOCKS] | kmax[H][MEAN_BLOCKS][D] | kmin[H][MEAN_BLOCKS][D] |
//                     qamax[H][MEAN_BLOCKS]
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "vorta_hip.h"
#include "common.h"

namespace {

constexpr int D = 128;
constexpr float V_TARGET = 240.f;  // amax of a v channel maps here (e4m3 max 448; relative precision is range-independent)
constexpr float E4M3_MAX = 448.f;
constexpr int MEAN_SAMPLES = 1024;  // key rows per head that define the centre (approximately: see fp8_kmean_kernel)
constexpr int SAMPLE_CHUNKS = 8;    // equal ranges of a head's video tokens whose samples are summed apart
constexpr int MEAN_BLOCKS = SAMPLE_CHUNKS + 1;  // + the tail (text) tokens: workgroups per head, partials combined in slot order

struct QParams {
  const char* x[3]; int64_t x_sh[3], x_ss[3];  // inputs (bytes)
  char* y[3]; int64_t y_sh[3], y_ss[3];        // outputs (bytes)
  int heads, n_tokens, rows_per_block;
  int seg_len, chunks_per_seg;                 // segmented row layout (seg_len > 0)
  int tail_first, tail_len;                    // segments from row tail_first on hold tail_len rows of data each
  int mean_stride, mean_cand;                  // tokens of the centre: s = i * mean_stride, i < mean_cand (per head)
  int64_t video_tokens;                        // tokens of a head before its tail (text) tokens: what the sample's chunks cut
  int64_t total_tokens;                        // tokens of a head's whole sequence (sample stride)
  int64_t token_offset;                        // seg_len == 0: the views hold tokens [token_offset, token_offset + n_tokens)
  const int32_t* src_map;                      // seg_len == 0, convert: output head h <- input head src_map[h]
  float c0;  // qk_scale * log2(e)
  float* ws; float* v_descale;
  int v_per_head, center_k;
  int slot_first, slot_count;                  // segmented layout: only head slots [slot_first, slot_first + slot_count)
  int n_which;                                 // 3: q,k,v; 2: q,k only (v converted elsewhere, vorta_fp8_v_convert)
};

template <typename T> __device__ __forceinline__ float to_f(T v) { return (float)v; }

__device__ __forceinline__ float* kmean_of(const QParams& p) { return p.ws + 2 * (2 * p.heads + p.heads * D); }
__device__ __forceinline__ float* ksum_of(const QParams& p) { return kmean_of(p) + p.heads * D; }
__device__ __forceinline__ float* kcnt_of(const QParams& p) { return ksum_of(p) + p.heads * MEAN_BLOCKS * D; }
__device__ __forceinline__ float* kmaxp_of(const QParams& p) { return kcnt_of(p) + p.heads * MEAN_BLOCKS; }
__device__ __forceinline__ float* kminp_of(const QParams& p) { return kmaxp_of(p) + p.heads * MEAN_BLOCKS * D; }
__device__ __forceinline__ float* qamaxp_of(const QParams& p) { return kminp_of(p) + p.heads * MEAN_BLOCKS * D; }
// centre of channel d of head h from the partial sums: the same expression wherever it is needed (bit-identical)
__device__ __forceinline__ float kcenter(const QParams& p, int h, int d) {
  const float* ks = ksum_of(p) + (int64_t)h * MEAN_BLOCKS * D + d;
  const float* kc = kcnt_of(p) + h * MEAN_BLOCKS;
  float a = 0.f, n = 0.f;
#pragma unroll
  for (int b = 0; b < MEAN_BLOCKS; ++b) { a += ks[b * D]; n += kc[b]; }
  return n > 0.f ? a / n : 0.f;
}

// the rows a block works on: [r0, r1) of head `head`, stored under physical head index `hphys`
__device__ __forceinline__ void block_rows(const QParams& p, int& r0, int& r1, int& head, int& hphys) {
  if (p.seg_len > 0) {
    // blockIdx.x = (segment group, slot inside the range, row chunk): segment = group * heads + slot
    const int per = p.slot_count * p.chunks_per_seg;
    const int sg = blockIdx.x / per, rem = blockIdx.x - sg * per;
    const int si = rem / p.chunks_per_seg, c = rem - si * p.chunks_per_seg;
    const int seg = sg * p.heads + p.slot_first + si;
    const int s0 = seg * p.seg_len;
    r0 = s0 + c * p.rows_per_block;
    const int seg_rows = s0 >= p.tail_first ? p.tail_len : p.seg_len;  // tail_first is a multiple of seg_len
    r1 = min(min(r0 + p.rows_per_block, s0 + seg_rows), p.n_tokens);
    head = seg % p.heads;
    hphys = 0;
  } else {
    head = hphys = blockIdx.y;
    r0 = blockIdx.x * p.rows_per_block;
    r1 = min(r0 + p.rows_per_block, p.n_tokens);
  }
}

// grid (heads, MEAN_BLOCKS), 1024 threads: the head's SAMPLE -- its tokens s = i * mean_stride (i < mean_cand), the same
// tokens whether the head is a (S,D) view, a sequence shard of one, or scattered over the segments of the Ulysses receive
// layout (so a head gets the same centre and multipliers, hence the same e4m3 bytes, on one GPU and on a rank of P).
// Workgroup (h, b) owns the sampled tokens of chunk b (b < SAMPLE_CHUNKS: video tokens [b Sv / C, (b+1) Sv / C); b = C: the
// tail tokens) and writes their partials -- sums of the key rows (the centre), per-channel max / min of them (abs-max of k
// minus ANY centre follows exactly), abs-max of the q rows -- unless the call's views do not hold that chunk (a shard:
// another call, or another rank, owns it).  64 row lanes x 16 channel groups, four rows in flight per lane, lanes by
// position INSIDE the chunk; fixed reduction order here and in `kcenter` (deterministic).
template <typename T>
__global__ __launch_bounds__(1024) void fp8_sample_kernel(const QParams p) {
  typedef __attribute__((ext_vector_type(8))) T T8;
  constexpr int RL = 64;
  const int h = p.slot_first + blockIdx.x;
  const int t = threadIdx.x, cc = t & 15, rl = t >> 4;
  const char* base = p.x[1] + (p.seg_len > 0 ? 0 : (int64_t)h * p.x_sh[1]) + cc * 16;
  const char* qbase = p.x[0] + (p.seg_len > 0 ? 0 : (int64_t)h * p.x_sh[0]) + cc * 16;
  float s[8], kmx[8], kmn[8];
  float qm = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) { s[i] = 0.f; kmx[i] = -3.0e38f; kmn[i] = 3.0e38f; }
  int cnt = 0;
  // sample indices of this chunk: tokens [t0, t1)
  const int b = blockIdx.y;
  const int64_t t0 = b < SAMPLE_CHUNKS ? (int64_t)b * p.video_tokens / SAMPLE_CHUNKS : p.video_tokens;
  const int64_t t1 = b < SAMPLE_CHUNKS ? (int64_t)(b + 1) * p.video_tokens / SAMPLE_CHUNKS : p.total_tokens;
  // a chunk that is not wholly inside this call's views belongs to another call (its slot is left alone)
  if (p.seg_len <= 0 && (t0 < p.token_offset || t1 > p.token_offset + p.n_tokens)) return;
  const int i_lo = (int)((t0 + p.mean_stride - 1) / p.mean_stride);
  const int i_hi = (int)min((int64_t)p.mean_cand, (t1 + p.mean_stride - 1) / p.mean_stride);
  // physical row of sample i of head h
  auto row_of = [&](int i) -> int64_t {
    const int64_t tok = (int64_t)i * p.mean_stride;
    if (p.seg_len <= 0) return tok - p.token_offset;
    if (tok < p.video_tokens) return ((tok / p.seg_len) * p.heads + h) * p.seg_len + tok % p.seg_len;
    return (int64_t)p.tail_first + (int64_t)h * p.seg_len + (tok - p.video_tokens);
  };
  for (int i0 = i_lo + rl; i0 < i_hi; i0 += 4 * RL) {
    T8 v[4], w[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      ok[u] = i0 + u * RL < i_hi;
      if (ok[u]) {
        const int64_t r = row_of(i0 + u * RL);
        v[u] = *(const T8*)(base + r * p.x_ss[1]);
        w[u] = *(const T8*)(qbase + r * p.x_ss[0]);
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ok[u]) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float kf = to_f(v[u][j]);
          s[j] += kf;
          kmx[j] = fmaxf(kmx[j], kf);
          kmn[j] = fminf(kmn[j], kf);
          qm = fmaxf(qm, fabsf(to_f(w[u][j])));
        }
        ++cnt;
      }
  }
  __shared__ float red[RL][D + 1];
  __shared__ int cred[RL];
  __shared__ float qred[RL];
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cc * 8 + i] = s[i];
  if (cc == 0) cred[rl] = cnt;
  // q: over the 16 channel groups of a row lane, then over the row lanes
#pragma unroll
  for (int off = 8; off > 0; off >>= 1) qm = fmaxf(qm, __shfl_xor(qm, off));
  if (cc == 0) qred[rl] = qm;
  __syncthreads();
  const int64_t part = (int64_t)h * MEAN_BLOCKS + blockIdx.y;
  if (t < D) {
    float a = 0.f;
    int n = 0;
    for (int j = 0; j < RL; ++j) { a += red[j][t]; n += cred[j]; }
    ksum_of(p)[part * D + t] = a;
    if (t == 0) {
      kcnt_of(p)[part] = (float)n;
      float q = 0.f;
      for (int j = 0; j < RL; ++j) q = fmaxf(q, qred[j]);
      qamaxp_of(p)[part] = q;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cc * 8 + i] = kmx[i];
  __syncthreads();
  if (t < D) {
    float a = -3.0e38f;
    for (int j = 0; j < RL; ++j) a = fmaxf(a, red[j][t]);
    kmaxp_of(p)[part * D + t] = i_hi > i_lo ? a : 0.f;  // an empty chunk writes zeros: partial tables of several ranks ADD
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cc * 8 + i] = kmn[i];
  __syncthreads();
  if (t < D) {
    float a = 3.0e38f;
    for (int j = 0; j < RL; ++j) a = fminf(a, red[j][t]);
    kminp_of(p)[part * D + t] = i_hi > i_lo ? a : 0.f;
  }
}

// grid (row chunks, heads or 1): per-(head, channel) abs-max of v (exact: v is scaled to just under the format's maximum)
template <typename T>
__global__ __launch_bounds__(256) void fp8_absmax_kernel(const QParams p) {
  typedef __attribute__((ext_vector_type(8))) T T8;
  constexpr int which = 2;
  int r0, r1, h, hphys;
  block_rows(p, r0, r1, h, hphys);
  if (r0 >= r1) return;
  const int t = threadIdx.x, cc = t & 15, rl = t >> 4;
  const char* base = p.x[which] + (int64_t)hphys * p.x_sh[which] + cc * 16;
  float m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = 0.f;
  for (int r = r0 + rl; r < r1; r += 16) {
    const T8 v = *(const T8*)(base + (int64_t)r * p.x_ss[which]);
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], fabsf(to_f(v[i])));
  }
  __shared__ float red[16][D + 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cc * 8 + i] = m[i];
  __syncthreads();
  if (t < D) {
    float cm = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) cm = fmaxf(cm, red[j][t]);
    atomicMax((unsigned*)p.ws + 2 * p.heads + h * D + t, __float_as_uint(cm));
  }
}

// grid (heads), 128 threads: centre, abs-max of q and of k minus its centre from the sample partials, multipliers
__global__ __launch_bounds__(128) void fp8_scales_kernel(const QParams p) {
  const int h = p.slot_first + blockIdx.x, d = threadIdx.x, H = p.heads;
  const float c = p.center_k ? kcenter(p, h, d) : 0.f;
  kmean_of(p)[h * D + d] = c;
  // per channel: max over the sampled keys of |k - c| = max(kmax - c, c - kmin) (fl(x - c) is monotone in x: exactly what
  // a pass over the same rows computing |k - c| would give)
  float kmx = -3.0e38f, kmn = 3.0e38f;
#pragma unroll
  for (int b = 0; b < MEAN_BLOCKS; ++b) {
    if (kcnt_of(p)[h * MEAN_BLOCKS + b] <= 0.f) continue;  // no sampled token in this chunk
    kmx = fmaxf(kmx, kmaxp_of(p)[((int64_t)h * MEAN_BLOCKS + b) * D + d]);
    kmn = fminf(kmn, kminp_of(p)[((int64_t)h * MEAN_BLOCKS + b) * D + d]);
  }
  float ak = kmx >= kmn ? fmaxf(kmx - c, c - kmn) : 0.f;  // (no sample at all: 0)
  __shared__ float red[D];
  red[d] = ak;
  __syncthreads();
  for (int s = 64; s > 0; s >>= 1) {
    if (d < s) red[d] = fmaxf(red[d], red[d + s]);
    __syncthreads();
  }
  const float mk = red[0];
  __syncthreads();
  float* qmul = p.ws + 2 * H + H * D;
  float* kmul = qmul + H;
  float* vmul = kmul + H;
  if (d == 0) {
    float mq = 0.f;
#pragma unroll
    for (int b = 0; b < MEAN_BLOCKS; ++b) mq = fmaxf(mq, qamaxp_of(p)[h * MEAN_BLOCKS + b]);
    p.ws[h] = mq;
    p.ws[H + h] = mk;
    // q8 . k8 = c0 * q . k with both operand maxima at sqrt(c0 * mq * mk)
    float t = 1.f;
    if (mq > 0.f && mk > 0.f) t = sqrtf(mk / (p.c0 * mq));
    qmul[h] = p.c0 * t;
    kmul[h] = 1.f / t;
  }
  if (p.n_which < 3) return;  // v: vorta_fp8_v_convert wrote (or will write) v8 and v_descale
  float mv = p.ws[2 * H + h * D + d];
  if (p.v_per_head) {
    red[d] = mv;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
      if (d < s) red[d] = fmaxf(red[d], red[d + s]);
      __syncthreads();
    }
    mv = red[0];
  }
  vmul[h * D + d] = mv > 0.f ? V_TARGET / mv : 0.f;
  p.v_descale[h * D + d] = mv / V_TARGET;
}

__device__ __forceinline__ float clamp448(float x) { return __builtin_amdgcn_fmed3f(x, -E4M3_MAX, E4M3_MAX); }

// grid (row chunks, heads or 1, 3): thread = 16 channels of a row (32 B in, 16 B out); 8 threads per row, 32 rows per pass
template <typename T>
__global__ __launch_bounds__(256) void fp8_convert_kernel(const QParams p) {
  typedef __attribute__((ext_vector_type(8))) T T8;
  const int which = blockIdx.z, H = p.heads;
  int r0, r1, h, hphys;
  block_rows(p, r0, r1, h, hphys);
  if (r0 >= r1) return;
  const int t = threadIdx.x, cc = t & 7, rl = t >> 3;
  const int hout = hphys;  // output head; with a source map the input head (and its scales) is src_map[hout]
  if (p.src_map && p.seg_len <= 0) {
    const int hs = p.src_map[hout];
    if (hs < 0) return;  // output head not written by this call
    h = hphys = hs;
  }
  const float* qmul = p.ws + 2 * H + H * D;
  float mul[16], sub[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) sub[i] = 0.f;
  if (which == 2) {
    const float* vm = qmul + 2 * H + h * D + cc * 16;
#pragma unroll
    for (int i = 0; i < 16; ++i) mul[i] = vm[i];
  } else {
    const float s = qmul[which * H + h];
#pragma unroll
    for (int i = 0; i < 16; ++i) mul[i] = s;
    if (which == 1 && p.center_k) {
      const float* km = kmean_of(p) + h * D + cc * 16;  // written by the scales kernel (= kcenter)
#pragma unroll
      for (int i = 0; i < 16; ++i) sub[i] = km[i];
    }
  }
  const char* src = p.x[which] + (int64_t)hphys * p.x_sh[which] + cc * 32;
  char* dst = p.y[which] + (int64_t)hout * p.y_sh[which] + cc * 16;
  // UNR row passes in flight per thread (one pass in flight: 4.9 TB/s over q, k, v of Hunyuan-129f)
  constexpr int UNR = 4;
  for (int rb = r0 + rl; rb < r1; rb += 32 * UNR) {
    T8 a[UNR], b[UNR];
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = rb + 32 * u;
      if (r < r1) {
        a[u] = *(const T8*)(src + (int64_t)r * p.x_ss[which]);
        b[u] = *(const T8*)(src + (int64_t)r * p.x_ss[which] + 16);
      }
    }
#pragma unroll
    for (int u = 0; u < UNR; ++u) {
      const int r = rb + 32 * u;
      if (r >= r1) break;
      float f[16];
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        f[i] = clamp448((to_f(a[u][i]) - sub[i]) * mul[i]);
        f[8 + i] = clamp448((to_f(b[u][i]) - sub[8 + i]) * mul[8 + i]);
      }
      u32x4 o;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * w], f[4 * w + 1], 0, false);
        o[w] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[4 * w + 2], f[4 * w + 3], lo, true);
      }
      *(u32x4*)(dst + (int64_t)r * p.y_ss[which]) = o;
    }
  }
}

}  // namespace

extern "C" int vorta_fp8_quant_ws_floats(int32_t heads, int32_t head_dim) {
  if (heads <= 0 || head_dim != D) return VORTA_EINVAL;
  return 2 * (2 * heads + heads * head_dim) + heads * head_dim + heads * MEAN_BLOCKS * (head_dim + 1) +
         2 * heads * MEAN_BLOCKS * head_dim + heads * MEAN_BLOCKS;
}

extern "C" int vorta_fp8_quant_ws_partials(int32_t heads, int32_t head_dim, int64_t* first_float, int64_t* n_floats) {
  if (heads <= 0 || head_dim != D || !first_float || !n_floats) return VORTA_EINVAL;
  *first_float = 2 * (2 * (int64_t)heads + (int64_t)heads * D) + (int64_t)heads * D;  // behind kmean
  *n_floats = (int64_t)heads * MEAN_BLOCKS * (D + 1) + 2 * (int64_t)heads * MEAN_BLOCKS * D + (int64_t)heads * MEAN_BLOCKS;
  return VORTA_OK;
}

extern "C" int vorta_fp8_quantize_qkv(const vorta_fp8_quant_args* a, void* hip_stream) {
  if (!a || a->struct_size != sizeof(vorta_fp8_quant_args)) return VORTA_EINVAL;
  if (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16) return VORTA_EUNSUPPORTED;
  if (a->head_dim != D) return VORTA_EUNSUPPORTED;
  if (a->heads < 0 || a->n_tokens < 0 || a->seg_len < 0 || a->tail_first < 0 || a->tail_len < 0) return VORTA_EINVAL;
  if (a->seg_len > 0 && (a->tail_first > 0 || a->tail_len > 0) && (a->tail_first % a->seg_len || a->tail_len > a->seg_len))
    return VORTA_EINVAL;
  if (a->slot_first < 0 || a->slot_count < 0 || (int64_t)a->slot_first + a->slot_count > a->heads) return VORTA_EINVAL;
  if ((a->slot_first || a->slot_count) && a->seg_len <= 0) return VORTA_EINVAL;  // a slot range needs the segmented layout
  if (a->heads == 0 || a->n_tokens == 0) return VORTA_OK;
  const bool stats_only = a->flags & 8, no_stats = a->flags & 16;
  if (stats_only && no_stats) return VORTA_EINVAL;
  if (a->video_tokens < 0 || a->token_offset < 0 || a->total_tokens < 0) return VORTA_EINVAL;
  if (a->seg_len > 0 && (a->video_tokens || a->token_offset || a->total_tokens || a->src_map || stats_only || no_stats))
    return VORTA_EINVAL;  // shards, source maps and an explicit video length belong to the (heads, tokens, D) views
  if ((a->token_offset || a->total_tokens) && !(stats_only || no_stats)) return VORTA_EINVAL;  // a shard alone has no scales
  if (a->total_tokens && (int64_t)a->token_offset + a->n_tokens > a->total_tokens) return VORTA_EINVAL;
  // the sample and v's abs-max of a plain call are taken per INPUT head: a source map needs the two-phase form or q, k only
  if (a->src_map && !no_stats) return VORTA_EINVAL;
  const int n_which = (a->flags & 4) ? 2 : 3;
  if ((stats_only || no_stats) && n_which == 3) return VORTA_EUNSUPPORTED;  // v has its own pair (vorta_fp8_v_absmax / _convert)
  if (!a->ws || (n_which == 3 && !a->v_descale) || !(a->qk_scale > 0.f)) return VORTA_EINVAL;
  const vorta_tensor* in[3] = {&a->q, &a->k, &a->v};
  const vorta_tensor* out[3] = {&a->q8, &a->k8, &a->v8};
  QParams p{};
  for (int i = 0; i < n_which; ++i) {
    if (!in[i]->ptr || (!stats_only && !out[i]->ptr)) return VORTA_EINVAL;
    if (((uintptr_t)in[i]->ptr & 15) || (in[i]->stride_s % 8) || (in[i]->stride_h % 8) || in[i]->stride_s < D) return VORTA_EINVAL;
    p.x[i] = (const char*)in[i]->ptr; p.x_sh[i] = in[i]->stride_h * 2; p.x_ss[i] = in[i]->stride_s * 2;
    if (stats_only) continue;
    if (((uintptr_t)out[i]->ptr & 15) || (out[i]->stride_s % 16) || (out[i]->stride_h % 16) || out[i]->stride_s < D) return VORTA_EINVAL;
    p.y[i] = (char*)out[i]->ptr; p.y_sh[i] = out[i]->stride_h; p.y_ss[i] = out[i]->stride_s;
  }
  p.src_map = a->src_map;
  p.token_offset = a->token_offset;
  p.heads = a->heads; p.n_tokens = a->n_tokens;
  p.c0 = a->qk_scale * 1.4426950408889634f;
  p.ws = a->ws; p.v_descale = a->v_descale;
  p.v_per_head = a->flags & 1;
  p.center_k = (a->flags >> 1) & 1;
  p.n_which = n_which;
  p.seg_len = a->seg_len;
  // the tail rule applies when either field is set: tail_first > 0 with tail_len = 0 is an EMPTY tail region (Wan: no text)
  const bool has_tail = a->seg_len > 0 && (a->tail_first > 0 || a->tail_len > 0);
  p.tail_first = has_tail ? a->tail_first : 0x7fffffff;
  p.tail_len = has_tail ? a->tail_len : 0;
  p.slot_first = a->slot_count > 0 ? a->slot_first : 0;
  p.slot_count = a->slot_count > 0 ? a->slot_count : a->heads;
  hipStream_t st = (hipStream_t)hip_stream;
  const int H = a->heads, h0 = p.slot_first, hn = p.slot_count;
  // abs-max slots of v for the heads this call converts (q and k: written by the scales kernel from the sample)
  hipError_t e = hipSuccess;
  if (n_which == 3) e = hipMemsetAsync(a->ws + 2 * H + (size_t)h0 * D, 0, sizeof(float) * hn * D, st);
  if (e != hipSuccess) return vorta_set_hip_error(e);
  // enough workgroups to fill the chip several times over, few enough that the atomics stay cheap
  p.rows_per_block = 1024;
  dim3 grid;
  if (p.seg_len > 0) {
    p.chunks_per_seg = (p.seg_len + p.rows_per_block - 1) / p.rows_per_block;
    const int64_t n_seg = ((int64_t)a->n_tokens + p.seg_len - 1) / p.seg_len;
    const int64_t n_sg = (n_seg + H - 1) / H;  // segment groups: one segment per head slot each
    if (n_sg * hn * p.chunks_per_seg > 0x7fffffffll) return VORTA_EINVAL;
    grid = dim3((unsigned)(n_sg * hn * p.chunks_per_seg), 1, 1);
  } else {
    p.chunks_per_seg = 1;
    grid = dim3((unsigned)((a->n_tokens + p.rows_per_block - 1) / p.rows_per_block), (unsigned)H, 1);
  }
  // centre: ~MEAN_SAMPLES evenly spaced tokens of the head (an odd stride: one that divides the segment length would
  // sample the same offsets of every segment)
  int64_t head_tokens = a->total_tokens ? a->total_tokens : a->n_tokens;
  p.video_tokens = a->video_tokens ? a->video_tokens : head_tokens;
  if (p.video_tokens > head_tokens) return VORTA_EINVAL;
  if (p.seg_len > 0) {
    const int64_t data_rows = has_tail ? (int64_t)a->tail_first : (int64_t)a->n_tokens;
    p.video_tokens = (data_rows / p.seg_len / H) * p.seg_len;
    head_tokens = p.video_tokens + p.tail_len;
  }
  p.total_tokens = head_tokens;
  if (stats_only && p.seg_len <= 0 && (a->token_offset || a->total_tokens)) {
    // a statistics call over a shard owns the sample chunks that lie wholly inside it; a chunk cut by a shard edge would be
    // summed by nobody (empty sample -> centre 0, unbalanced multipliers, no error): both edges must be chunk boundaries
    auto boundary = [&](int64_t t) {
      if (t == head_tokens) return true;
      for (int b = 0; b <= SAMPLE_CHUNKS; ++b)
        if ((int64_t)b * p.video_tokens / SAMPLE_CHUNKS == t) return true;
      return false;
    };
    if (!boundary(a->token_offset) || !boundary((int64_t)a->token_offset + a->n_tokens)) return VORTA_EINVAL;
  }
  p.mean_stride = (int)(head_tokens / MEAN_SAMPLES);
  if (p.mean_stride < 1) p.mean_stride = 1;
  p.mean_stride |= 1;
  p.mean_cand = (int)((head_tokens + p.mean_stride - 1) / p.mean_stride);
  const bool bf = a->dtype == VORTA_BF16;
  if (!no_stats) {
    if (bf) hipLaunchKernelGGL((fp8_sample_kernel<__bf16>), dim3((unsigned)hn, MEAN_BLOCKS), dim3(1024), 0, st, p);
    else hipLaunchKernelGGL((fp8_sample_kernel<_Float16>), dim3((unsigned)hn, MEAN_BLOCKS), dim3(1024), 0, st, p);
    e = hipGetLastError();
    if (e != hipSuccess) return vorta_set_hip_error(e);
  }
  if (stats_only) return VORTA_OK;
  if (n_which == 3) {
    if (bf) hipLaunchKernelGGL((fp8_absmax_kernel<__bf16>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((fp8_absmax_kernel<_Float16>), grid, dim3(256), 0, st, p);
    e = hipGetLastError();
    if (e != hipSuccess) return vorta_set_hip_error(e);
  }
  hipLaunchKernelGGL(fp8_scales_kernel, dim3((unsigned)hn), dim3(128), 0, st, p);
  e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  grid.z = (unsigned)n_which;
  if (bf) hipLaunchKernelGGL((fp8_convert_kernel<__bf16>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((fp8_convert_kernel<_Float16>), grid, dim3(256), 0, st, p);
  e = hipGetLastError();
  if (e != hipSuccess) return vorta_set_hip_error(e);
  return VORTA_OK;
}

// ---- V on its own (include/vorta_hip.h vorta_fp8_v_absmax / vorta_fp8_v_convert): the sender side of the Ulysses exchange
// converts its sequence shard of V with the scales of the WHOLE sequence (abs-max all-reduced over the ranks in between the
// two calls), so V travels as e4m3 and lands ready for the attention kernels.
namespace {

struct VParams {
  const char* x; int64_t x_sh, x_ss;  // bytes
  char* y; int64_t y_sh, y_ss;
  int heads, n_tokens, rows_per_block, per_head;
  const int32_t* src_map;
  float* amax; float* v_descale;
};

// grid (row chunks, heads): amax[h][d] = max(amax[h][d], max_rows |v[h][row][d]|)
template <typename T>
__global__ __launch_bounds__(256) void fp8_v_absmax_kernel(const VParams p) {
  typedef __attribute__((ext_vector_type(8))) T T8;
  const int h = blockIdx.y;
  const int r0 = blockIdx.x * p.rows_per_block, r1 = min(r0 + p.rows_per_block, p.n_tokens);
  const int t = threadIdx.x, cc = t & 15, rl = t >> 4;
  const char* base = p.x + (int64_t)h * p.x_sh + cc * 16;
  float m[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) m[i] = 0.f;
  for (int r = r0 + rl; r < r1; r += 16) {
    const T8 v = *(const T8*)(base + (int64_t)r * p.x_ss);
#pragma unroll
    for (int i = 0; i < 8; ++i) m[i] = fmaxf(m[i], fabsf(to_f(v[i])));
  }
  __shared__ float red[16][D + 1];
#pragma unroll
  for (int i = 0; i < 8; ++i) red[rl][cc * 8 + i] = m[i];
  __syncthreads();
  if (t < D) {
    float cm = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) cm = fmaxf(cm, red[j][t]);
    atomicMax((unsigned*)p.amax + h * D + t, __float_as_uint(cm));
  }
}

// grid (row chunks, heads): v8[h] = e4m3(v[src] * 240 / amax[src]) with src = src_map ? src_map[h] : h; the first row
// chunk of a head also writes v_descale[h] (the same expressions as fp8_scales_kernel / fp8_convert_kernel)
template <typename T>
__global__ __launch_bounds__(256) void fp8_v_convert_kernel(const VParams p) {
  typedef __attribute__((ext_vector_type(8))) T T8;
  const int h = blockIdx.y, hs = p.src_map ? p.src_map[h] : h;
  const int r0 = blockIdx.x * p.rows_per_block, r1 = min(r0 + p.rows_per_block, p.n_tokens);
  const int t = threadIdx.x, cc = t & 7, rl = t >> 3;
  float mul[16];
  float ph = 0.f;
  if (p.per_head) {
    for (int d = 0; d < D; ++d) ph = fmaxf(ph, p.amax[hs * D + d]);
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const float mv = p.per_head ? ph : p.amax[hs * D + cc * 16 + i];
    mul[i] = mv > 0.f ? V_TARGET / mv : 0.f;
  }
  if (p.v_descale && blockIdx.x == 0 && t < D) {
    const float mv = p.per_head ? ph : p.amax[hs * D + t];
    p.v_descale[h * D + t] = mv / V_TARGET;
  }
  const char* src = p.x + (int64_t)hs * p.x_sh + cc * 32;
  char* dst = p.y + (int64_t)h * p.y_sh + cc * 16;
  for (int r = r0 + rl; r < r1; r += 32) {
    const T8 a = *(const T8*)(src + (int64_t)r * p.x_ss);
    const T8 b = *(const T8*)(src + (int64_t)r * p.x_ss + 16);
    float f[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f[i] = clamp448(to_f(a[i]) * mul[i]);
      f[8 + i] = clamp448(to_f(b[i]) * mul[8 + i]);
    }
    u32x4 o;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      int lo = __builtin_amdgcn_cvt_pk_fp8_f32(f[4 * w], f[4 * w + 1], 0, false);
      o[w] = (uint32_t)__builtin_amdgcn_cvt_pk_fp8_f32(f[4 * w + 2], f[4 * w + 3], lo, true);
    }
    *(u32x4*)(dst + (int64_t)r * p.y_ss) = o;
  }
}

int v_params(const vorta_fp8_v_args* a, bool convert, VParams& p) {
  if (!a || a->struct_size != sizeof(vorta_fp8_v_args)) return VORTA_EINVAL;
  if (a->dtype != VORTA_BF16 && a->dtype != VORTA_FP16) return VORTA_EUNSUPPORTED;
  if (a->head_dim != D) return VORTA_EUNSUPPORTED;
  if (a->heads < 0 || a->n_tokens < 0) return VORTA_EINVAL;
  if (a->heads == 0 || a->n_tokens == 0) return 1;
  if (!a->amax || !a->v.ptr || ((uintptr_t)a->v.ptr & 15) || (a->v.stride_s % 8) || (a->v.stride_h % 8) || a->v.stride_s < D)
    return VORTA_EINVAL;
  p.x = (const char*)a->v.ptr; p.x_sh = a->v.stride_h * 2; p.x_ss = a->v.stride_s * 2;
  if (convert) {
    if (!a->v8.ptr || ((uintptr_t)a->v8.ptr & 15) || (a->v8.stride_s % 16) || (a->v8.stride_h % 16) || a->v8.stride_s < D)
      return VORTA_EINVAL;
    p.y = (char*)a->v8.ptr; p.y_sh = a->v8.stride_h; p.y_ss = a->v8.stride_s;
  }
  p.heads = a->heads; p.n_tokens = a->n_tokens;
  p.rows_per_block = 1024;
  p.per_head = a->flags & 1;
  p.src_map = convert ? a->src_map : nullptr;
  p.amax = a->amax; p.v_descale = convert ? a->v_descale : nullptr;
  return VORTA_OK;
}

}  // namespace

extern "C" int vorta_fp8_v_absmax(const vorta_fp8_v_args* a, void* hip_stream) {
  VParams p{};
  const int rc = v_params(a, false, p);
  if (rc) return rc < 0 ? rc : VORTA_OK;
  const dim3 grid((unsigned)((p.n_tokens + p.rows_per_block - 1) / p.rows_per_block), (unsigned)p.heads);
  hipStream_t st = (hipStream_t)hip_stream;
  if (a->dtype == VORTA_BF16) hipLaunchKernelGGL((fp8_v_absmax_kernel<__bf16>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((fp8_v_absmax_kernel<_Float16>), grid, dim3(256), 0, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}

extern "C" int vorta_fp8_v_convert(const vorta_fp8_v_args* a, void* hip_stream) {
  VParams p{};
  const int rc = v_params(a, true, p);
  if (rc) return rc < 0 ? rc : VORTA_OK;
  const dim3 grid((unsigned)((p.n_tokens + p.rows_per_block - 1) / p.rows_per_block), (unsigned)p.heads);
  hipStream_t st = (hipStream_t)hip_stream;
  if (a->dtype == VORTA_BF16) hipLaunchKernelGGL((fp8_v_convert_kernel<__bf16>), grid, dim3(256), 0, st, p);
  else hipLaunchKernelGGL((fp8_v_convert_kernel<_Float16>), grid, dim3(256), 0, st, p);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? VORTA_OK : vorta_set_hip_error(e);
}
