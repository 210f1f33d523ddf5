/*
 * vorta_hip.h -- C ABI of libvorta_hip.so: the MI355X (gfx950) implementation of VORTA's routed
 * sparse-attention denoising hot path.
 *
 * The reference (wenhao728/VORTA) is pure Python and has no FFI of its own: the boundary it exposes is
 * the diffusers attention-processor protocol (vorta/attention/__init__.py:1-16).  Each entry point below
 * names the reference code it replaces (paths relative to the reference root).  The Python host side
 * (vorta_amd/) binds these with ctypes and mirrors the reference's processor classes one to one; see
 * INTEGRATION.md for the stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (VORTA_OK) or a negative VORTA_E* code; nothing throws across the ABI;
 *   - the caller owns every buffer (inputs, outputs, index tables, workspaces); the library allocates
 *     nothing and keeps no state; all pointers are DEVICE pointers unless a field says "host";
 *   - work is enqueued on the given hipStream_t (passed as void*); no call synchronises the device;
 *   - a "head slot" y in [0,n_heads) addresses head  head_list ? head_list[y] : y  of the (H,S,D) tensors;
 *     a batch is folded into the head axis by the caller (head = b*H + h with stride_h the same);
 *   - rows of D contiguous elements; strides in ELEMENTS; 16-byte aligned rows.
 */
#ifndef VORTA_HIP_H
#define VORTA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VORTA_OK 0
#define VORTA_EINVAL (-1)       /* bad argument (null pointer, size, alignment, struct_size)          */
#define VORTA_EUNSUPPORTED (-2) /* valid request this build does not implement (head_dim, dtype)      */
#define VORTA_ELAUNCH (-3)      /* the HIP runtime refused the launch (see vorta_last_hip_error)      */

#define VORTA_ABI_VERSION 8 /* 2: adds the fp8 entry points (vorta_fp8_*, vorta_attn_fwd_fp8*); 3: adds vorta_permute_heads;
                                4: vorta_fp8_quant_args gains slot_first / slot_count and flags bit2, adds vorta_fp8_v_absmax /
                                vorta_fp8_v_convert; every earlier call means what it meant
                                5: vorta_fp8_quant_args gains video_tokens / token_offset / total_tokens / src_map and flags
                                bit3 / bit4 (sequence shards), adds vorta_fp8_quant_ws_partials
                                6: adds the int8-score entry points (vorta_i8_quantize_k, vorta_attn_fwd_i8, vorta_attn_fwd_batch_i8);
                                7: the int8-score kernel writes its probabilities with one power-of-two scale per query row
                                   and 32 keys (vorta_attn_i8_ext.defer becomes the reference-point trigger in binades, default 24);
                                8: adds vorta_i8_tail_flags and vorta_split_heads (per-head choice between int8 and 16-bit scores) */

typedef enum vorta_dtype {
  VORTA_BF16 = 0,
  VORTA_FP16 = 1,
  VORTA_FP32 = 2,    /* vorta_route_scores only */
  VORTA_FP8E4M3 = 3, /* OCP e4m3fn bytes: q/k/v of vorta_attn_fwd_fp8, output of vorta_fp8_quantize_qkv */
  VORTA_INT8 = 4     /* two's-complement bytes: k8 of vorta_i8_quantize_k / vorta_attn_fwd_i8 (vorta_permute_heads: any 1-byte rows) */
} vorta_dtype;

/* One (H,S,D) operand: element (h,s,d) lives at ptr + h*stride_h + s*stride_s + d. */
typedef struct vorta_tensor {
  void* ptr;
  int64_t stride_h;
  int64_t stride_s;
} vorta_tensor;

/*
 * vorta_attn_fwd -- softmax(Q K^T * scale) V for a list of heads, with row indirection.
 *
 * One generic gather flash-attention kernel serves all three experts:
 *   dense   (hunyuan.py:136-189 `_step_attention`, wan.py:103-149 `_attn`): no tables, n_kv = valid
 *           keys (Hunyuan L = attention_mask.sum(), hunyuan.py:169), q_valid = L so padded text rows
 *           come out exactly zero (hunyuan.py:176);
 *   coreset (hunyuan.py:410-457, wan.py:243-270 + coreset_select.py:68-185): q_rows / kv_rows are the
 *           keep-lists written by vorta_coreset_select, dup_rows its drop-lists: pool-gather and
 *           unpool-scatter are fused into the loads and the epilogue, no pooled copy exists;
 *   sliding tile (sliding_attn_flex.py:72-211 + tile.py:7-78): q_rows is the tile-major permutation,
 *           kv_rows the per-q-tile key list from vorta_sta_build_tables (q_group_len = tokens/tile);
 *           tile/untile are address arithmetic, no BlockMask and no permuted copy exists.
 * Each head writes straight into its slice of the final (H,S,D) output, which replaces the per-expert
 * head gathers and the boolean index-put of `_get_routed_qkv` / `_combine_attn_outputs`
 * (hunyuan.py:612-661, wan.py:388-437).
 *
 * Query side: positions p in [0,n_q) are cut into groups of q_group_len (0 = one group); a workgroup
 * never straddles a group.  Position p reads/writes row  q_rows ? q_rows[y*q_rows_stride_h + p]
 * : q_row_offset + p.  Positions p >= q_valid are written as zeros.
 * Key side: group g attends positions j in [0,n_kv) -> row  kv_rows ? kv_rows[y*kv_rows_stride_h +
 * g*kv_rows_stride_g + j] : kv_row_offset + j.
 * Duplicates: if dup_rows, every position p < n_dup_pos additionally writes its output row to rows
 * dup_rows[y*dup_rows_stride_h + p*n_dup + i], i < n_dup (coreset: centre -> dropped margins).
 * Split keys: n_splits > 1 cuts the key range into n_splits chunks computed by separate workgroups into
 * ws_o / ws_ml (float, sizes from vorta_attn_workspace_bytes) and merged by a second kernel; used for the
 * few text-query rows of the sliding expert that attend every key.
 */
typedef struct vorta_attn_args {
  uint32_t struct_size; /* = sizeof(vorta_attn_args) */
  int32_t dtype;        /* vorta_dtype */
  int32_t head_dim;     /* 128 */
  int32_t n_heads;      /* head slots in this launch (upper bound when n_heads_dev is set) */
  vorta_tensor q, k, v, o;
  const int32_t* head_list;   /* [n_heads] or NULL */
  const int32_t* n_heads_dev; /* optional device count: slots >= *n_heads_dev exit at once (sync-free routing) */
  int32_t n_q, q_group_len, q_row_offset, q_valid;
  const int32_t* q_rows;
  int64_t q_rows_stride_h;
  int32_t n_kv, kv_row_offset;
  const int32_t* kv_rows;
  int64_t kv_rows_stride_h, kv_rows_stride_g;
  const int32_t* dup_rows;
  int64_t dup_rows_stride_h;
  int32_t n_dup_pos, n_dup;
  float scale;        /* softmax scale, 1/sqrt(D) in the reference */
  int32_t block_rows; /* query rows per workgroup: 0 = auto, 128 or 256 */
  int32_t n_splits;   /* >= 1 */
  float* ws_o;        /* [n_heads][n_splits][n_q][D]   when n_splits > 1 */
  float* ws_ml;       /* [n_heads][n_splits][n_q][2]   when n_splits > 1 */
  /* optional device-resident lengths (e.g. L = attention_mask.sum(), hunyuan.py:169, without the host sync):
   * effective n_kv = clamp(*n_kv_dev, 1, n_kv), effective q_valid = min(*q_valid_dev, q_valid) */
  const int32_t* n_kv_dev;
  const int32_t* q_valid_dev;
  /* variant 2 addresses K/V rows with 32-bit offsets: row < 2^24, row_stride_bytes < 2^24 and
   * row * row_stride_bytes < 2^31 for every key row of a head (checked for contiguous ranges, VORTA_EUNSUPPORTED;
   * guaranteed by the caller for kv_rows tables).  Variant 1 has no such limit. */
  int32_t variant; /* kernel body: 0 = auto = 2; 1 = plain (attn_fwd_kernel<T,NW>), 2 = software-pipelined, scores one
                      key block ahead, K/V tiles by LDS-DMA, softmax folded into the score MFMA
                      (attn_fwd_pipe_kernel<T,NW,KVTAB>) */
  int32_t reserved;
  /* Optional query groups of DIFFERENT lengths (ABI 2): one row per workgroup, device int32 [n_q_blocks][3] =
   * (group g, first position, end position) with end - first <= block_rows (which must then be given, 128 or 256).
   * q_group_len is ignored; group g still reads the key list kv_rows + g*kv_rows_stride_g.  Used by the sliding-tile
   * expert: query tiles whose clamped windows coincide (sliding_attn_flex.py:118-120 clamps the window centre, so the
   * two outermost tiles of a dimension see the same keys) are merged into one group, and the group is cut into full
   * workgroups instead of every 792-token tile ending in a 24-row one. */
  const int32_t* q_block_table;
  int32_t n_q_blocks;
  int32_t reserved2;
} vorta_attn_args;

int vorta_attn_fwd(const vorta_attn_args* args, void* hip_stream);
#define VORTA_MAX_FUSED_LAUNCHES 6 /* (was 4 before round 4; more -> VORTA_EINVAL) */
/* Up to VORTA_MAX_FUSED_LAUNCHES launches fused into ONE grid (the experts of a routed layer, hunyuan.py:564-591, the text
 * queries of the sliding-tile expert, and -- under sequence parallelism -- up to two full-attention heads that compute
 * only a range of their queries on this rank: q_rows = a slice of the row map, n_heads = 1): workgroups are
 * dispatched in argument order -- pass the longest key loops first -- so one expert's tail is filled by the next
 * expert instead of idling until a kernel boundary.  Every entry must resolve to the 256-row pipelined kernel
 * (VORTA_EUNSUPPORTED otherwise: launch those separately); split-key entries get their merge kernels after. */
int vorta_attn_fwd_batch(const vorta_attn_args* args, int32_t n, void* hip_stream);
/* the launch shape vorta_attn_fwd would use: query rows per workgroup (128 -> kernel attn_fwd_kernel<T,4>,
 * 256 -> attn_fwd_kernel<T,8>) and the number of workgroups; pure host computation */
int vorta_attn_plan(const vorta_attn_args* args, int32_t* block_rows, int64_t* n_workgroups, int32_t* kernel_id);
/* kernel_id = waves*16 + (pipelined ? 1 : 0) + (pipelined with kv table ? 2 : 0):
 *   attn_fwd_kernel<T,NW> (plain) or attn_fwd_pipe_kernel<T,NW,KVTAB> */
/* bytes of ws_o and ws_ml for a given launch (0,0 when n_splits <= 1) */
int vorta_attn_workspace_bytes(const vorta_attn_args* args, uint64_t* ws_o_bytes, uint64_t* ws_ml_bytes);

/*
 * fp8 (e4m3) path -- BASELINE.json configs[4] "fp8 MFMA QK^T/PV path".  The reference has no fp8 code: this path serves
 * the same three experts (wan.py:243-294, hunyuan.py:410-507) with both contractions on
 * v_mfma_f32_32x32x64_f8f6f4 (2x the bf16 MFMA rate).  Two steps:
 *
 * vorta_fp8_quantize_qkv -- one pass over post-RoPE q,k,v (16-bit, (H,S,D) views) that writes e4m3 copies:
 *     q8 = e4m3( q * qmul[h] ),  k8 = e4m3( k * kmul[h] ),  v8 = e4m3( v * vmul[h][d] )
 *   qmul[h] * kmul[h] = qk_scale * log2(e): the softmax scale and the exp2 conversion are folded into the operands,
 *   so q8 . k8 is the score in the exp2 domain and the kernel's softmax needs no multiply.  The split between q and k
 *   only BALANCES their ranges: t = sqrt(amax_k / (c0 * amax_q)), qmul = c0 * t, kmul = 1 / t, which puts both maxima near
 *   sqrt(c0 * amax_q * amax_k) -- a factor of ~250 inside e4m3's range (448) for unit-variance data, and the format's
 *   relative precision (2^-4) does not depend on where in the normal range a value sits -- so amax_q / amax_k are taken
 *   over a SAMPLE of ~1024 evenly spaced tokens of the head (the tokens that define the key centre, below) instead of a
 *   pass over q and k; values are clamped to +-448 before the conversion in any case.  v is scaled per head and channel
 *   to amax -> 240 with amax over EVERY token (one pass over v: its target sits just under the format's maximum);
 *   v_descale[h][d] = amax_v[h][d] / 240 is applied to the output row in the attention epilogue.
 *   Enqueues 4 launches (sample, abs-max of v, scales, convert); nothing is read back to the host.
 *
 * vorta_attn_fwd_fp8 / _batch_fp8 -- vorta_attn_fwd / _batch with q,k,v = e4m3 (args->dtype = VORTA_FP8E4M3, strides
 *   in BYTES = elements, rows 16-byte aligned); `scale` is ignored (folded, above); the probabilities are re-packed
 *   to e4m3 as P * 2^p_bias with a running reference point at most `defer` below the row max (p_bias + defer <= 8, so
 *   P * 2^p_bias <= 256 < 448); row sums come from one more MFMA against a ones tile (the same rounded P that
 *   multiplies V); the output (ext->out_dtype, bf16 / fp16) is o[d] * v_descale[head][d] / rowsum.
 *   Every other field of vorta_attn_args means what it means for vorta_attn_fwd (row tables, groups, duplicates,
 *   split keys, device-resident lengths).
 *   ext->flags bit1 selects the mixed-precision kernel (csrc/attn_fwd_mx.hip): 16-bit q k^T, e4m3 P V -- see the field.  Since
 *   ABI 7 that kernel writes its probabilities with ONE POWER-OF-TWO SCALE PER QUERY ROW AND 32 KEYS (the block scale of the
 *   P V MFMA's B operand), as vorta_attn_fwd_i8 does: per tile e = rint(max(max c + 64, 0) - 72) for c = score - reference +
 *   p_bias, probability = e4m3(2^(c - e)) 2^e (round to nearest even, subnormals kept); the reference point is the row's first
 *   block's maximum and moves only when a tile lies more than `defer` binades above it (default 24, 0 ... 40).  Nothing is
 *   flushed for lying far below the row's maximum; the all-e4m3 kernel (flags bit1 clear) keeps the single range above.
 */
typedef struct vorta_fp8_quant_args {
  uint32_t struct_size;
  int32_t dtype;            /* input dtype: VORTA_BF16 / VORTA_FP16 */
  int32_t head_dim, heads;  /* 128, H */
  int32_t n_tokens;         /* rows of every head to convert */
  float qk_scale;           /* softmax scale (1/sqrt(D) in the reference) */
  vorta_tensor q, k, v;     /* inputs, (H,S,D) views, strides in elements */
  vorta_tensor q8, k8, v8;  /* outputs, e4m3, strides in bytes; rows 16-byte aligned */
  float* v_descale;         /* [heads][head_dim] out */
  float* ws;                /* workspace, vorta_fp8_quant_ws_floats(heads, head_dim) floats: abs-max slots, multipliers, centres */
  int32_t flags;            /* bit0: v scaled per head instead of per (head, channel);
                               bit1: centre the keys -- k8 = e4m3((k - c[h]) * kmul[h]) with c[h][:] the mean of ~1024 evenly
                               spaced key rows of the head (softmax does not change when one vector is subtracted from
                               every key; the e4m3 error of q8 . k8 shrinks with |k|) */
  int32_t seg_len;          /* 0: q,k,v,q8,k8,v8 are (heads, n_tokens, D) views.  > 0: they are row arrays of n_tokens rows
                               (stride_h unused) in which row r belongs to head (r / seg_len) % heads -- the Ulysses
                               receive layout (vorta_seq_row_map); every head gets its own scales and centre */
  int32_t tail_first;       /* seg_len > 0 and either field non-zero: from row tail_first (a multiple of seg_len) on, only */
  int32_t tail_len;         /* the first tail_len rows of a segment hold data (text rows); the others are skipped
                               (tail_len = 0: the tail region is empty).  A head's tokens are its rows of the segments
                               before tail_first, in order, then its tail rows: the centre of flags bit1 samples the same
                               TOKENS in both layouts, so a head converts to the same bytes on one GPU and on a rank of P */
  int32_t slot_first;       /* seg_len > 0: convert only the rows of head slots [slot_first, slot_first + slot_count) -- the */
  int32_t slot_count;       /* slot group whose exchange has landed while the next one is in flight (0, 0 = every slot).
                               Scales and centres are per head, so converting slot group by slot group gives the bytes
                               one call over all slots gives */
  int32_t video_tokens;     /* seg_len == 0: tokens [0, video_tokens) of a head's sequence are its video tokens, the rest its
                               tail (text) tokens; 0 = all of them.  The sample of flags bit1 and of the q/k abs-max is summed
                               in 8 equal ranges of the video tokens + the tail, so pass the same value wherever the same
                               heads are converted (the segmented layout derives it from tail_first) */
  int32_t token_offset;     /* seg_len == 0, sequence shards (flags bit3 / bit4): the views hold tokens [token_offset, */
  int32_t total_tokens;     /* token_offset + n_tokens) of heads whose whole sequence has total_tokens tokens (0: the views
                               are the whole sequence) */
  const int32_t* src_map;   /* seg_len == 0, optional: q8 / k8 / v8 head h <- head src_map[h] of q / k / v, with that head's
                               scales and centre (the heads in destination order for the exchange); `heads` entries, a
                               negative one = that output head is not written */
} vorta_fp8_quant_args;
/* flags bit2: q and k only -- v, v8 and v_descale are not touched (v arrived as e4m3: vorta_fp8_v_convert on the sender)
 * flags bit3: statistics only -- the sample partials of the tokens this call holds go to their slots of `ws`, nothing is
 *             converted.  Zero the partial region first (vorta_fp8_quant_ws_partials), call once per piece of the sequence
 *             (a shard must hold whole eighths of the video tokens: token_offset and token_offset + n_tokens each equal to
 *             floor(b video_tokens / 8) for some b, or to total_tokens -- anything else is VORTA_EINVAL, a cut chunk would be
 *             summed by nobody; the tail tokens in a call of their own or behind the last eighth), ADD the regions of all
 *             ranks (disjoint slots: exact);
 * flags bit4: no statistics -- multipliers and centres from the partials already in `ws`, then the conversion of the tokens
 *             this call holds.  bit3 calls + all-reduce + bit4 calls write the bytes ONE plain call over the assembled
 *             sequence writes: the send side of the Ulysses exchange moves q and k as e4m3 (vorta/ulysses/utils.py:61-91
 *             moves them in 16 bits). */
int vorta_fp8_quant_ws_partials(int32_t heads, int32_t head_dim, int64_t* first_float, int64_t* n_floats);

int vorta_fp8_quant_ws_floats(int32_t heads, int32_t head_dim);
int vorta_fp8_quantize_qkv(const vorta_fp8_quant_args* args, void* hip_stream);

/*
 * V on its own, for the sender side of the Ulysses exchange (vorta/ulysses/utils.py:61-91 moves 16-bit q, k, v): each rank
 * takes the per-(head, channel) abs-max of ITS sequence shard of v (vorta_fp8_v_absmax: amax[h][d] is raised atomically,
 * the caller zeroes it first; several calls accumulate -- video rows, then the replicated text rows), the ranks
 * all-reduce amax with MAX (heads x 128 floats), and vorta_fp8_v_convert writes the e4m3 shard with the scales of the
 * whole sequence, heads already in destination order (v8 head h <- v head src_map[h]): v crosses the links at half the
 * bytes and lands as the attention kernels read it.  The bytes and v_descale equal what vorta_fp8_quantize_qkv produces
 * from the assembled 16-bit sequence (abs-max over shards = abs-max over the sequence).
 */
typedef struct vorta_fp8_v_args {
  uint32_t struct_size;
  int32_t dtype;            /* input dtype: VORTA_BF16 / VORTA_FP16 */
  int32_t head_dim, heads;  /* 128; heads of v (absmax) / of v8 (convert) */
  int32_t n_tokens;
  int32_t flags;            /* bit0: one scale per head instead of per (head, channel) */
  vorta_tensor v;           /* (H, n_tokens, D) view, strides in elements */
  vorta_tensor v8;          /* convert: (heads, n_tokens, D) e4m3 out, strides in bytes */
  const int32_t* src_map;   /* convert: [heads] source head of every destination head, or NULL (identity) */
  float* amax;              /* [H][D], indexed by the SOURCE head: absmax raises it, convert reads it */
  float* v_descale;         /* convert: optional out [heads][D], indexed by the DESTINATION head: amax / 240 */
} vorta_fp8_v_args;

int vorta_fp8_v_absmax(const vorta_fp8_v_args* args, void* hip_stream);
int vorta_fp8_v_convert(const vorta_fp8_v_args* args, void* hip_stream);

typedef struct vorta_attn_fp8_ext {
  uint32_t struct_size;
  int32_t out_dtype;           /* VORTA_BF16 / VORTA_FP16: element type of args->o */
  const float* v_descale;      /* [..][head_dim], indexed by the head id (not the slot) */
  int64_t v_descale_stride_h;  /* floats between heads (head_dim) */
  float p_bias;                /* log2 bias of the e4m3 probabilities; 0 = default (5) */
  float defer;                 /* deferred-rescale threshold in log2 units; 0 = default (3; mixed kernel: 24); p_bias + defer <= 8
                                  (mixed kernel: 0 <= defer <= 40, p_bias <= 16) */
  int32_t flags;               /* bit0: row sums by VALU adds of the unrounded P instead of the ones-tile MFMA;
                                  bit1 (ABI 4): MIXED precision -- q, k (and o) are 16-bit tensors of type args->dtype =
                                  out_dtype with strides in elements, the scores run on the 16-bit MFMA with args->scale
                                  folded as in vorta_attn_fwd; only v is e4m3 (strides in bytes, vorta_fp8_v_absmax /
                                  vorta_fp8_v_convert or the v half of vorta_fp8_quantize_qkv) and only P is packed to
                                  e4m3.  The score is where an 8-bit mantissa costs: the all-e4m3 path holds 40 dB against
                                  the 16-bit kernels only where the softmax is flat, this one 42-70 dB on every input
                                  family tried, at 1.2 x the 16-bit rate instead of 1.7 x */
  int32_t reserved;
} vorta_attn_fp8_ext;

int vorta_attn_fwd_fp8(const vorta_attn_args* args, const vorta_attn_fp8_ext* ext, void* hip_stream);
int vorta_attn_fwd_batch_fp8(const vorta_attn_args* args, const vorta_attn_fp8_ext* ext, int32_t n, void* hip_stream);

/*
 * Int8 scores (ABI 6; BASELINE.json configs[4], no reference counterpart: the reference computes the three experts of
 * wan.py:243-294 / hunyuan.py:410-507 in the dtype of q, k, v).  precision "i8pv": q k^T on v_mfma_i32_32x32x32_i8 (the
 * e4m3 MFMA rate) with 7 bits next to the operand's maximum -- the all-e4m3 path loses 40 dB on peaked or outlier-carried
 * logits because e4m3 has 3 mantissa bits wherever a value sits -- and P V in e4m3 as in the mixed kernel.
 * With cq, ck the per-head means of q and k (from ~1024 sampled tokens) and s a per-channel balance vector,
 *     q . k  =  (q - cq) . (k - ck)  +  cq . (k - ck)  +  (a term that does not depend on the key: softmax does not see it)
 *            =  [(q - cq) s] . [(k - ck) / s]  +  b[key]
 * the first product runs in int8 (both operands centred: no common component eats the 8 bits; channel ranges balanced, which
 * is what a FIXED-point format needs when a few qk-norm channels carry the logits), b[key] is computed in float32 and enters
 * the int32 accumulator as its initial value.
 *
 * vorta_i8_quantize_k -- per head h, from ~1024 evenly spaced tokens (the same tokens in the (H,S,D) view and in the segmented
 *   Ulysses receive layout): ck[d] = mean k, cq[d] = mean q, s[d] = clamp((var k[d] / var q[d])^(1/4), 1/8, 8).  Then over every
 *   row of the head:   kt = (k - ck) / s      amax[h] = max |kt|  (exact, over all rows)      sk[h] = amax[h] / 127
 *                      k8[row][d] = rint(kt 127 / amax[h])          k_bias[row] = (cq . (k - ck)) 127 / amax[h]
 *   Outputs: k8 (int8 rows of head_dim bytes), k_bias (one float per row: b[key] in units of sk[h]), q_prep (heads x 2 x
 *   head_dim floats: cq then s -- what the attention kernel applies to its query rows before quantising them itself) and
 *   k_head_scale = sk (heads floats).
 */
typedef struct vorta_i8_quant_args {
  uint32_t struct_size;
  int32_t dtype;            /* input dtype: VORTA_BF16 / VORTA_FP16 */
  int32_t head_dim, heads;  /* 128, H */
  int32_t n_tokens;         /* rows of every head (seg_len == 0) or of the row array (seg_len > 0) */
  vorta_tensor q, k;        /* inputs, strides in elements; q is only sampled (centre and balance statistics) */
  vorta_tensor k8;          /* out: int8, strides in bytes, rows 16-byte aligned; same head / row geometry as k */
  float* k_bias;            /* out: k_bias[h * k_bias_stride_h + row]  (seg_len > 0: k_bias[row]) */
  int64_t k_bias_stride_h;
  float* q_prep;            /* out [heads][2][head_dim]: cq | s */
  float* k_head_scale;      /* out [heads]: sk */
  float* ws;                /* workspace, 2 * heads * head_dim + heads floats: ck | 1 / s | amax */
  int32_t flags;            /* bit0: no balancing (s = 1); bit1: no centring (ck = cq = 0, k_bias = 0) */
  int32_t seg_len;          /* as vorta_fp8_quant_args: 0 = (heads, n_tokens, D) views; > 0 = one row array, row r belongs to */
  int32_t tail_first;       /* head (r / seg_len) % heads, tail (text) rows from tail_first on, tail_len per segment */
  int32_t tail_len;
  int32_t slot_first;       /* seg_len > 0: only head slots [slot_first, slot_first + slot_count) (0, 0 = all) */
  int32_t slot_count;
} vorta_i8_quant_args;

int vorta_i8_quantize_k(const vorta_i8_quant_args* args, void* hip_stream);

/*
 * ABI 8 -- per-head choice between the int8-score kernel and the mixed-precision one ("auto8" of the Python host): int8 scores
 * with ONE key scale per head resolve the bulk of a heavy-tailed head's keys to 0 / +-1 (Student-t(3): abs-max ~ 250 sigma over
 * 10^7 samples; relative error 0.10-0.19 where every other input family tried stays under 0.07).
 * vorta_i8_tail_flags: flags[h] = 1 when the root mean square of head h's int8 keys (k8 of vorta_i8_quantize_k, (heads,
 *   n_tokens, head_dim) view, strides in bytes) over ~1024 evenly spaced tokens is below `min_rms` counts, else 0; row_map (NULL:
 *   token = row) gives the row of each token inside a head's view (the Ulysses receive layout).  Exact integer arithmetic: the
 *   flag is reproducible, and the same under sequence parallelism.
 * vorta_split_heads: list0 / list1 = the heads of head_list[0 .. n) (NULL: 0 .. n-1; n = min(*n_heads_dev, n_heads) when
 *   n_heads_dev is given) whose flag is 0 / 1, order kept; counts[0], counts[1] = their lengths.  The two lists (room for
 *   n_heads entries each) and the counts (device) are what vorta_attn_args.head_list / n_heads_dev of the two launches take:
 *   a launch over an empty list exits in its first instruction.  No host synchronisation anywhere.
 */
int vorta_i8_tail_flags(const vorta_tensor* k8, int32_t heads, int32_t n_tokens, const int32_t* row_map, float min_rms, int32_t* flags,
                        void* hip_stream);
int vorta_split_heads(const int32_t* head_list, const int32_t* n_heads_dev, int32_t n_heads, const int32_t* flags,
                      int32_t* list0, int32_t* list1, int32_t* counts, void* hip_stream);

/*
 * vorta_attn_fwd_i8 -- the gather flash-attention of vorta_attn_fwd with int8 scores and e4m3 P V.
 *   args->q: 16-bit (args->dtype), strides in elements.  Every WAVE takes its 32 query rows, qt = (q - cq) s with the head's
 *            q_prep, the abs-max over the 32 rows (sq = amax / 127) and rounds them to int8 itself -- queries are read once
 *            per workgroup; a wave whose abs-max is below 2^-12 (every row on the head's centre) takes q8 = 0, sq = 1: its scores are
 *            the bias term alone;
 *   args->k: the int8 rows of vorta_i8_quantize_k, strides in BYTES; k_bias its per-row floats (indexed like k rows:
 *            head * k_bias_stride_h + row, through kv_rows when given); k_head_scale[head] = sk;
 *   args->v: e4m3 (vorta_fp8_v_absmax / vorta_fp8_v_convert), strides in bytes, with v_descale as in vorta_attn_fp8_ext;
 *   args->o: 16-bit (args->dtype).
 *   score of (query i of wave w, key j) in the exp2 domain = u (q8[i] . k8[j] + rint(k_bias[j] / sq[w])), u = scale log2(e)
 *   sq[w] sk[head] (the int32 accumulator starts from the rounded bias; |bias| is clamped to 2 000 000 units).
 *   Probabilities: P' = 2^(score - reference + p_bias) against the row's reference point (its first block's maximum; it moves
 *   only when a later block lies more than `defer` binades above it), written to e4m3 by ONE conversion instead of exp2 +
 *   round, with ONE POWER-OF-TWO SCALE PER QUERY ROW AND 32 KEYS (ABI 7): for y = 8 log2 P' + 56 and e = max(rint((max y over
 *   the 32 consecutive keys of one half of the 64-key block - 120) / 8), -100), the byte is rint(y - 8 e) (v_cvt_pk_u8_f32, saturating at 0:
 *   the e4m3 number whose exponent field is the integer part of log2 P' - e and whose mantissa is the LINEAR interpolation of
 *   its fraction, +-3 % of 2^x) and 2^e is the block scale of the P V MFMA's B operand (v_mfma_scale_f32_32x32x64_f8f6f4).
 *   The range of a probability is the scale's, not e4m3's: nothing is flushed for lying far below the ROW's maximum (up to ABI
 *   6 everything below 2^-14 ... 2^-11
 *   of it was).  Tables, groups, duplicates, split keys and the fused grid as vorta_attn_fwd_fp8.
 */
typedef struct vorta_attn_i8_ext {
  uint32_t struct_size;
  int32_t flags;               /* reserved, 0 */
  const float* k_bias;
  int64_t k_bias_stride_h;     /* floats between heads of k_bias */
  const float* q_prep;         /* [..][2][head_dim], indexed by the head id */
  int64_t q_prep_stride_h;     /* floats between heads (2 * head_dim) */
  const float* k_head_scale;   /* [..], indexed by the head id */
  const float* v_descale;      /* [..][head_dim], indexed by the head id */
  int64_t v_descale_stride_h;
  float p_bias, defer;         /* 0 = defaults: p_bias 5 (0 ... 16); defer 24 binades (0 ... 64: P' <= 2^(defer + 9) in fp32) */
} vorta_attn_i8_ext;

int vorta_attn_fwd_i8(const vorta_attn_args* args, const vorta_attn_i8_ext* ext, void* hip_stream);
int vorta_attn_fwd_batch_i8(const vorta_attn_args* args, const vorta_attn_i8_ext* ext, int32_t n, void* hip_stream);

/*
 * vorta_coreset_select -- coreset_select.py:68-124 (ranking part) + :159-166 (scatter destinations).
 *
 * For every head slot and every window group of g = gw[0]*gw[1]*gw[2] tokens of the (t,h,w) latent:
 * cosine similarity (F.normalize eps 1e-12, fp32 arithmetic) between the centre token and each of the
 * g-1 margin tokens, ascending stable rank; the n_keep least similar margins are kept.
 * Writes   keep_rows[y][G*(1+n_keep) (+ n_tail)] : token ids of the packed sequence
 *                    [G centres | G x n_keep kept margins (group-major, least similar first) | tail]
 *          drop_rows[y][G][g-1-n_keep]           : token ids of the dropped margins of each group
 * (tail = n_tail consecutive ids starting at tail_first: the text tokens appended by hunyuan.py:441-444).
 * If row_map is given, every emitted id i is replaced by row_map[i] (physical row of token i: the
 * zero-copy Ulysses layout), while x is still read at row_map[i].
 */
typedef struct vorta_coreset_args {
  uint32_t struct_size;
  int32_t dtype, head_dim, n_heads;
  vorta_tensor x; /* (H,S,D): Q or K */
  const int32_t* head_list;
  const int32_t* n_heads_dev;
  int32_t latent[3], group[3];
  int32_t n_keep; /* kept margins per group = int(g*(1-rate)) - 1 */
  int32_t tail_first, n_tail;
  const int32_t* row_map; /* optional [t*h*w + ...] */
  int32_t* keep_rows;
  int64_t keep_rows_stride_h;
  int32_t* drop_rows; /* may be NULL (K side needs no drop list) */
  int64_t drop_rows_stride_h;
  /* ABI 4: optional second keep list for the KEY side, same length as keep_rows: per group its centre and kept margins
   * together, in ascending token order, groups in order, then the tail -- the same SET of rows as keep_rows in an order
   * that keeps the gathered K/V rows of a tile close in memory (the softmax does not depend on key order; the query
   * side must keep the packed order: positions < G are the centres the duplicate list refers to).  keep_rows may be NULL
   * when only this list is wanted. */
  int32_t* keep_rows_kv;
  int64_t keep_rows_kv_stride_h;
} vorta_coreset_args;

int vorta_coreset_select(const vorta_coreset_args* args, void* hip_stream);

/*
 * vorta_sta_build_tables -- sliding_attn_flex.py:72-134 (mask_mod) + tile.py:7-78 (tile/untile) as tables.
 *
 *   q_rows [S]                 tile-major position -> raster token id (tile.py:26-29, sp = 1)
 *   kv_rows[n_tiles][n_kv]     keys visible to each q tile: the clamped window of tiles
 *                              (sliding_attn_flex.py:118-127) in tile-major order, then the t_eff valid
 *                              text tokens S..S+t_eff-1 (:112);  n_kv = prod(min(n_d, ...)) * tok + t_eff
 * n_kv is returned through *n_kv_out (host).  row_map as in vorta_coreset_select.
 */
typedef struct vorta_sta_args {
  uint32_t struct_size;
  int32_t latent[3], tile[3], window[3];
  int32_t t_eff;
  const int32_t* row_map;
  int32_t* q_rows;
  int32_t* kv_rows;
} vorta_sta_args;

int vorta_sta_table_sizes(const vorta_sta_args* args, int32_t* n_tiles, int32_t* tok_per_tile, int32_t* n_kv);
int vorta_sta_build_tables(const vorta_sta_args* args, void* hip_stream);

/*
 * vorta_router_route -- vorta/patch/router.py:33-43 (Router.forward) + the top-1 / tau rule of
 * `_get_routed_qkv` (hunyuan.py:620-624, wan.py:396-400), without the host sync of torch.nonzero.
 *
 *   scores[b][h][e] = softmax_e( W[3h+e,:] . silu(temb[b,:]) + bias[3h+e] )           (dtype of temb)
 *   expert_of_head[h] = argmax_e scores[0][h][e]  (first max), 0 if that score < tau   (batch item 0)
 *   head_lists[e][..] / head_counts[e] : ascending heads of expert e (device; feed n_heads_dev)
 */
typedef struct vorta_router_args {
  uint32_t struct_size;
  int32_t dtype;
  int32_t batch, embed_dim, heads, n_experts; /* n_experts = 3 */
  const void* temb;   /* [batch][embed_dim] */
  const void* weight; /* [heads*n_experts][embed_dim] */
  const void* bias;   /* [heads*n_experts] */
  float tau;
  void* scores;            /* [batch][heads][n_experts], may be NULL */
  int32_t* expert_of_head; /* [heads] */
  int32_t* head_lists;     /* [n_experts][heads] */
  int32_t* head_counts;    /* [n_experts] */
  float* ws_logits;        /* workspace [batch][heads][n_experts] floats (caller-owned) */
} vorta_router_args;

int vorta_router_route(const vorta_router_args* args, void* hip_stream);
/* Same dispatch rule applied to scores that already exist (the `routing_score` argument of the processors,
 * hunyuan.py:528): uses temb/weight/bias = NULL, `scores` as INPUT [batch][heads][n_experts], no workspace. */
int vorta_route_scores(const vorta_router_args* args, void* hip_stream);
/* Route plan: the router input is the pure timestep embedding (modeling_hunyuan.py:627-628,
 * modeling_wan.py:215), the same for every block, so all layers' routes are known at the start of a denoising
 * step.  One call (2 launches) for n_layers routers sharing `temb`: every array gains a leading [n_layers]
 * dimension -- weight [L][heads*n_experts][embed_dim], bias [L][heads*n_experts], scores [L][batch][heads][n_experts],
 * expert_of_head [L][heads], head_lists [L][n_experts][heads], head_counts [L][n_experts], ws_logits
 * [L][batch][heads][n_experts].  The reference runs one Router module per block inside the block's forward
 * (modeling_hunyuan.py:491,555; modeling_wan.py:127). */
int vorta_route_plan(const vorta_router_args* args, int32_t n_layers, void* hip_stream);

/*
 * vorta_qk_norm_rope -- the producer step right before the attention boundary, in place:
 * RMSNorm of q or k (hunyuan.py:62-73: per head over D with attn.norm_q / norm_k [diffusers RMSNorm];
 * wan.py:85-89: across all H*D channels of a token, before the head split) followed by the rotary embedding
 * (hunyuan.py:75-104 via diffusers apply_rotary_emb(use_real, unbind_dim=-1); wan.py:34-37 complex product --
 * the same rotation of interleaved pairs).  One read and one write of the tensor instead of ~10 torch passes.
 *   tokens [token_offset, token_offset + n_tokens) of every head are normalised; the first rope_tokens of them
 *   are rotated with cos/sin[token][D] (fp32; NULL = no rotation: text tokens, hunyuan.py:90-102).
 */
typedef struct vorta_norm_rope_args {
  uint32_t struct_size;
  int32_t dtype, head_dim, heads;
  vorta_tensor x;       /* (H,S,D) view, updated in place */
  const void* weight;   /* [D] or, if across_heads, [H*D]; dtype of x; NULL = no scale */
  const float* cos;     /* [n_tokens][D] or NULL */
  const float* sin;
  int32_t n_tokens, token_offset, rope_tokens;
  float eps;
  int32_t across_heads; /* 0: mean over D per head (Hunyuan); 1: mean over H*D per token (Wan) */
} vorta_norm_rope_args;

int vorta_qk_norm_rope(const vorta_norm_rope_args* args, void* hip_stream);

/*
 * vorta_mix_experts -- the score-weighted sum of the training-time forward:
 *   out[h][row][:] = sum_e scores[h][e] * x[e][h][row][:]        (fp32 accumulation, one rounding)
 * `_combine_attn_outputs`, hunyuan.py:509-513 == wan.py:296-300 (stack + multiply + sum over the expert axis).
 * Forward only: the library has no backward kernels.
 */
typedef struct vorta_mix_args {
  uint32_t struct_size;
  int32_t dtype, head_dim, heads, n_experts; /* n_experts = 3 */
  int32_t n_rows;
  vorta_tensor x[3];   /* (H, n_rows, D) views: outputs of expert 0 / 1 / 2 for ALL heads */
  vorta_tensor out;    /* (H, n_rows, D) view; may alias one of x */
  const void* scores;  /* [heads][n_experts], dtype of x: routing scores of batch item 0 */
} vorta_mix_args;

int vorta_mix_experts(const vorta_mix_args* args, void* hip_stream);

/*
 * vorta_seq_row_map -- physical row of every token for the zero-copy Ulysses layout.
 * After all_to_all_single of a sequence-sharded (H, S/P, D) tensor (vorta/ulysses/utils.py:61-91) rank r
 * holds P chunks of (H/P, S/P, D); the reference re-packs them into (H/P, S, D) with two
 * transpose+contiguous passes (:84-89).  Instead the kernels read the received buffer in place:
 * token s of local head hl is row  (s / Sl) * (Hl*Sl) + hl*Sl + s % Sl  ->  row_map[s] holds the part
 * that does not depend on the head, the head term is stride_h = Sl rows.
 */
int vorta_seq_row_map(int32_t* row_map, int32_t n_tokens, int32_t seg_len, int32_t seg_stride_rows, void* hip_stream);

/*
 * vorta_permute_heads -- the staging passes of the Ulysses exchange in one launch (vorta/ulysses/utils.py:61-91 does
 * them as transpose + .contiguous() per tensor and direction; A13 in SURVEY.md §8a).  For up to four (heads, n_rows, D)
 * views at once:
 *     dst[t][dst_map ? dst_map[h] : h][r][:] = src[t][src_map ? src_map[h] : h][r][:]      h < heads, r < n_rows
 * Send side: q, k, v projection views -> head-ordered contiguous blocks (src_map = head order), and the replicated text
 * rows behind each local head slot; receive side: the head-ordered output back into the (rows, H*D) result (dst_map).
 * The maps are device arrays of `heads` entries; a map must not repeat a head on the destination side, and no source may
 * overlap a destination (the copy is not an in-place permutation).
 */
typedef struct vorta_permute_args {
  uint32_t struct_size;
  int32_t dtype, head_dim, heads, n_rows, n_tensors; /* dtype: VORTA_BF16 / VORTA_FP16 (2-byte) or VORTA_FP8E4M3 */
  vorta_tensor src[4];
  vorta_tensor dst[4];
  const int32_t* src_map; /* [heads] or NULL */
  const int32_t* dst_map; /* [heads] or NULL */
} vorta_permute_args;

int vorta_permute_heads(const vorta_permute_args* args, void* hip_stream);

/* Introspection */
int vorta_abi_version(void);
const char* vorta_build_info(void); /* static string: arch, compiler */
int vorta_last_hip_error(void);     /* last hipError_t seen by a failed launch in this thread */
int vorta_sizeof(int which);        /* 0 tensor, 1 attn_args, 2 coreset_args, 3 sta_args, 4 router_args, 5 norm_rope_args,
                                       6 mix_args, 7 fp8_quant_args, 8 attn_fp8_ext, 9 permute_args, 10 fp8_v_args,
                                       11 i8_quant_args, 12 attn_i8_ext */

#ifdef __cplusplus
}
#endif
#endif /* VORTA_HIP_H */
