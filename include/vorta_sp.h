/*
 * vorta_sp.h -- C ABI of libvorta_sp.so: the Ulysses sequence-parallel exchange of VORTA's attention path on RCCL
 * (point-to-point xGMI: every GPU sends a distinct 1/P slice straight to each peer, all 7 links at once).
 *
 * What it replaces (paths relative to the reference root; SURVEY.md section 8(b) B-c):
 *   vorta_sp_init / vorta_sp_destroy   SequenceParallelState.setup_sp_group / cleanup  (vorta/ulysses/parallel_states.py:31-52,55-72)
 *   vorta_sp_a2a_seq2head              all_to_all_4D(x, scatter_idx=1, gather_idx=2)   (vorta/ulysses/utils.py:15-57,123-124):
 *                                      (B,H,S/P,D) -> (B,H/P,S,D): rank r receives the contiguous head block [r H/P, (r+1) H/P),
 *                                      the sequence is the rank-major concatenation of the shards
 *   vorta_sp_a2a_head2seq              all_to_all_4D(x, scatter_idx=2, gather_idx=1)   (utils.py:59-91): the exact inverse
 *   vorta_sp_allgather_heads           all_gather(x, dim=1)                            (utils.py:135-146,161-162): rank-ordered concat
 *
 * The reference wraps its collective in two transpose + .contiguous() passes per side and a torch.cuda.synchronize()
 * (utils.py:42-56,68-89).  Here there is NO pack / unpack pass and no synchronisation: every (head, peer) slice is already
 * contiguous on both sides (S/P x D elements), so the exchange is one ncclGroup of sends and receives that land in place --
 * B * H/P operations per peer and direction -- enqueued on the caller's stream.  `vorta_sp_plan_*` returns that operation list
 * (host arithmetic only, no GPU, no communicator): the same list the collective issues, so the index maps can be checked
 * against the reference's semantics for any P on a machine with no GPU at all (tests/test_sp_abi.py does, against the golden
 * maps G9 of SURVEY.md section 8(c)).
 *
 * The Python host of this repository keeps the reference's own interface, torch.distributed ("nccl" = RCCL), with a zero-copy
 * receive layout that the attention kernels read in place (vorta_amd/ulysses/engine.py); this library is the same exchange for
 * a caller that binds the C ABI and has no torch.
 *
 * Conventions as vorta_hip.h: 0 or a negative VORTA_E* code, caller-owned device buffers, explicit stream, nothing throws.
 * One communicator per process and GPU (one process per GPU); not thread-safe per communicator.
 */
#ifndef VORTA_SP_H
#define VORTA_SP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef VORTA_OK
#define VORTA_OK 0
#define VORTA_EINVAL (-1)
#define VORTA_EUNSUPPORTED (-2)
#define VORTA_ELAUNCH (-3)
#endif
#define VORTA_ECOMM (-4) /* RCCL refused (see vorta_sp_last_error) */

#define VORTA_SP_ABI_VERSION 1
#define VORTA_SP_UNIQUE_ID_BYTES 128

typedef struct vorta_sp_comm vorta_sp_comm; /* opaque: an RCCL communicator + this rank's place in it */

/* one point-to-point operation of an exchange: `bytes` contiguous bytes at byte offset `offset` of the send (x) or receive (y)
 * buffer, to / from rank `peer` (peer == own rank: a local copy).  The k-th send of rank a to rank b matches the k-th receive
 * of rank b from rank a. */
typedef struct vorta_sp_op {
  int32_t peer;
  int32_t is_send; /* 1: read from x, 0: write to y */
  int64_t offset;
  int64_t bytes;
} vorta_sp_op;

int vorta_sp_abi_version(void);
const char* vorta_sp_last_error(void); /* text of the last RCCL / HIP failure on this thread's communicator calls ("" if none) */

/* rank 0 makes the 128-byte id (ncclGetUniqueId) and ships it to the other ranks by any means (a file, a socket, MPI) */
int vorta_sp_unique_id(void* id_out);
/* collective over the P ranks: communicator on the CURRENT HIP device */
int vorta_sp_init(vorta_sp_comm** comm_out, int32_t rank, int32_t P, const void* unique_id);
int vorta_sp_destroy(vorta_sp_comm* comm);
int vorta_sp_rank(const vorta_sp_comm* comm);
int vorta_sp_size(const vorta_sp_comm* comm);

/* The operation lists (host only).  Returns the number of operations (>= 0) or a negative code; writes at most `max_ops`.
 * seq2head: x (B,H,Sl,D) -> y (B,H/P,P*Sl,D); head2seq: x (B,H/P,P*Sl,D) -> y (B,H,Sl,D); elem_bytes = bytes per element. */
int64_t vorta_sp_plan_seq2head(int32_t rank, int32_t P, int32_t B, int32_t H, int32_t Sl, int32_t D, int32_t elem_bytes,
                               vorta_sp_op* ops, int64_t max_ops);
int64_t vorta_sp_plan_head2seq(int32_t rank, int32_t P, int32_t B, int32_t H, int32_t Sl, int32_t D, int32_t elem_bytes,
                               vorta_sp_op* ops, int64_t max_ops);

/* The exchanges: contiguous device tensors, `dtype` a vorta_dtype (vorta_hip.h: 0 bf16, 1 fp16, 2 fp32, 3 e4m3, 4 int8); H % P == 0.
 * Enqueued on `hip_stream`; x must stay untouched and y unread until the stream has passed the call. */
int vorta_sp_a2a_seq2head(vorta_sp_comm* comm, const void* x, void* y, int32_t B, int32_t H, int32_t Sl, int32_t D, int32_t dtype,
                          void* hip_stream);
int vorta_sp_a2a_head2seq(vorta_sp_comm* comm, const void* x, void* y, int32_t B, int32_t H, int32_t Sl, int32_t D, int32_t dtype,
                          void* hip_stream);
/* x (B,Hl,T,D) of every rank -> y (B,P*Hl,T,D), rank-ordered along the head axis (the text tokens' outputs, hunyuan.py:187) */
int vorta_sp_allgather_heads(vorta_sp_comm* comm, const void* x, void* y, int32_t B, int32_t Hl, int32_t T, int32_t D, int32_t dtype,
                             void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* VORTA_SP_H */
