// libvorta_sp.so: the Ulysses exchange on RCCL behind a C ABI (include/vorta_sp.h).  Host code only: the exchange needs no
// kernel -- every (head, peer) slice is contiguous on both sides, so it is one ncclGroup of sends and receives that land in
// place (vorta/ulysses/utils.py:15-93 wraps its all_to_all_single in two transpose + contiguous passes per side and a device
// synchronisation; none of that is here).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include <string>
#include <vector>

#include "vorta_sp.h"

static_assert(sizeof(ncclUniqueId) == VORTA_SP_UNIQUE_ID_BYTES, "the id travels as 128 bytes");

struct vorta_sp_comm {
  ncclComm_t comm;
  int rank, size;
};

namespace {
thread_local std::string g_err;

int fail_nccl(ncclResult_t r, const char* what) {
  g_err = std::string(what) + ": " + ncclGetErrorString(r);
  return VORTA_ECOMM;
}
int fail_hip(hipError_t e, const char* what) {
  g_err = std::string(what) + ": " + hipGetErrorString(e);
  return VORTA_ELAUNCH;
}
int elem_bytes_of(int dtype) {
  switch (dtype) {
    case 0: case 1: return 2;  // bf16, fp16
    case 2: return 4;          // fp32
    case 3: case 4: return 1;  // e4m3, int8
    default: return 0;
  }
}
bool bad_shape(int P, int B, int H, int Sl, int D) { return P < 1 || B < 1 || H < 1 || Sl < 1 || D < 1 || H % P != 0; }

// seq2head = true:  x (B,H,Sl,D) -> y (B,Hl,P*Sl,D): the head block of peer j goes to j, peer j's tokens land as segment j
// seq2head = false: x (B,Hl,P*Sl,D) -> y (B,H,Sl,D): the inverse
int64_t plan(bool seq2head, int rank, int P, int B, int H, int Sl, int D, int eb, vorta_sp_op* ops, int64_t max_ops) {
  if (bad_shape(P, B, H, Sl, D) || eb < 1 || rank < 0 || rank >= P || (max_ops > 0 && !ops)) return VORTA_EINVAL;
  const int Hl = H / P;
  const int64_t slice = (int64_t)Sl * D * eb;
  int64_t n = 0;
  auto put = [&](int peer, int is_send, int64_t off) {
    if (n < max_ops) ops[n] = vorta_sp_op{peer, is_send, off, slice};
    ++n;
  };
  for (int j = 0; j < P; ++j)
    for (int b = 0; b < B; ++b)
      for (int hl = 0; hl < Hl; ++hl) {
        const int64_t full = ((int64_t)b * H + (int64_t)j * Hl + hl) * slice;           // (b, head j Hl + hl) of the (B,H,Sl,D) side
        const int64_t seg = (((int64_t)b * Hl + hl) * P + j) * slice;                   // (b, hl, segment j) of the (B,Hl,P Sl,D) side
        put(j, 1, seq2head ? full : seg);
        put(j, 0, seq2head ? seg : full);
      }
  return n;
}

int exchange(vorta_sp_comm* c, bool seq2head, const void* x, void* y, int B, int H, int Sl, int D, int dtype, void* hip_stream) {
  if (!c || !x || !y) return VORTA_EINVAL;
  const int eb = elem_bytes_of(dtype);
  if (!eb) return VORTA_EUNSUPPORTED;
  if (bad_shape(c->size, B, H, Sl, D)) return VORTA_EINVAL;
  const int64_t n = plan(seq2head, c->rank, c->size, B, H, Sl, D, eb, nullptr, 0);
  if (n < 0) return (int)n;
  std::vector<vorta_sp_op> ops((size_t)n);
  plan(seq2head, c->rank, c->size, B, H, Sl, D, eb, ops.data(), n);
  hipStream_t st = (hipStream_t)hip_stream;
  const char* xs = (const char*)x;
  char* ys = (char*)y;
  // own slices: the k-th send to self is the k-th receive from self -- a device copy on the same stream
  std::vector<const vorta_sp_op*> self_s, self_r;
  for (const auto& o : ops)
    if (o.peer == c->rank) (o.is_send ? self_s : self_r).push_back(&o);
  for (size_t k = 0; k < self_s.size(); ++k) {
    const hipError_t e = hipMemcpyAsync(ys + self_r[k]->offset, xs + self_s[k]->offset, (size_t)self_s[k]->bytes, hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return fail_hip(e, "hipMemcpyAsync (own slices)");
  }
  if (c->size == 1) return VORTA_OK;
  ncclResult_t r = ncclGroupStart();
  if (r != ncclSuccess) return fail_nccl(r, "ncclGroupStart");
  for (const auto& o : ops) {
    if (o.peer == c->rank) continue;
    r = o.is_send ? ncclSend(xs + o.offset, (size_t)o.bytes, ncclChar, o.peer, c->comm, st)
                  : ncclRecv(ys + o.offset, (size_t)o.bytes, ncclChar, o.peer, c->comm, st);
    if (r != ncclSuccess) {
      ncclGroupEnd();
      return fail_nccl(r, o.is_send ? "ncclSend" : "ncclRecv");
    }
  }
  r = ncclGroupEnd();
  return r == ncclSuccess ? VORTA_OK : fail_nccl(r, "ncclGroupEnd");
}
}  // namespace

extern "C" int vorta_sp_abi_version(void) { return VORTA_SP_ABI_VERSION; }
extern "C" const char* vorta_sp_last_error(void) { return g_err.c_str(); }

extern "C" int vorta_sp_unique_id(void* id_out) {
  if (!id_out) return VORTA_EINVAL;
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return fail_nccl(r, "ncclGetUniqueId");
  memcpy(id_out, &id, sizeof(id));
  return VORTA_OK;
}

extern "C" int vorta_sp_init(vorta_sp_comm** comm_out, int32_t rank, int32_t P, const void* unique_id) {
  if (!comm_out || !unique_id || P < 1 || rank < 0 || rank >= P) return VORTA_EINVAL;
  ncclUniqueId id;
  memcpy(&id, unique_id, sizeof(id));
  ncclComm_t comm;
  const ncclResult_t r = ncclCommInitRank(&comm, P, id, rank);
  if (r != ncclSuccess) return fail_nccl(r, "ncclCommInitRank");
  *comm_out = new vorta_sp_comm{comm, rank, P};
  return VORTA_OK;
}

extern "C" int vorta_sp_destroy(vorta_sp_comm* c) {
  if (!c) return VORTA_EINVAL;
  const ncclResult_t r = ncclCommDestroy(c->comm);
  delete c;
  return r == ncclSuccess ? VORTA_OK : fail_nccl(r, "ncclCommDestroy");
}

extern "C" int vorta_sp_rank(const vorta_sp_comm* c) { return c ? c->rank : VORTA_EINVAL; }
extern "C" int vorta_sp_size(const vorta_sp_comm* c) { return c ? c->size : VORTA_EINVAL; }

extern "C" int64_t vorta_sp_plan_seq2head(int32_t rank, int32_t P, int32_t B, int32_t H, int32_t Sl, int32_t D, int32_t elem_bytes,
                                          vorta_sp_op* ops, int64_t max_ops) {
  return plan(true, rank, P, B, H, Sl, D, elem_bytes, ops, max_ops);
}
extern "C" int64_t vorta_sp_plan_head2seq(int32_t rank, int32_t P, int32_t B, int32_t H, int32_t Sl, int32_t D, int32_t elem_bytes,
                                          vorta_sp_op* ops, int64_t max_ops) {
  return plan(false, rank, P, B, H, Sl, D, elem_bytes, ops, max_ops);
}

extern "C" int vorta_sp_a2a_seq2head(vorta_sp_comm* c, const void* x, void* y, int32_t B, int32_t H, int32_t Sl, int32_t D, int32_t dtype,
                                     void* hip_stream) {
  return exchange(c, true, x, y, B, H, Sl, D, dtype, hip_stream);
}
extern "C" int vorta_sp_a2a_head2seq(vorta_sp_comm* c, const void* x, void* y, int32_t B, int32_t H, int32_t Sl, int32_t D, int32_t dtype,
                                     void* hip_stream) {
  return exchange(c, false, x, y, B, H, Sl, D, dtype, hip_stream);
}

extern "C" int vorta_sp_allgather_heads(vorta_sp_comm* c, const void* x, void* y, int32_t B, int32_t Hl, int32_t T, int32_t D,
                                        int32_t dtype, void* hip_stream) {
  if (!c || !x || !y || B < 1 || Hl < 1 || T < 1 || D < 1) return VORTA_EINVAL;
  const int eb = elem_bytes_of(dtype);
  if (!eb) return VORTA_EUNSUPPORTED;
  hipStream_t st = (hipStream_t)hip_stream;
  const size_t part = (size_t)Hl * T * D * eb;  // one batch item of one rank
  if (c->size == 1) {
    const hipError_t e = hipMemcpyAsync(y, x, part * B, hipMemcpyDeviceToDevice, st);
    return e == hipSuccess ? VORTA_OK : fail_hip(e, "hipMemcpyAsync (all-gather of one rank)");
  }
  ncclResult_t r = ncclGroupStart();
  if (r != ncclSuccess) return fail_nccl(r, "ncclGroupStart");
  for (int b = 0; b < B; ++b) {  // y[b] = (P Hl, T, D): the rank-ordered concatenation of every rank's x[b]
    r = ncclAllGather((const char*)x + b * part, (char*)y + (size_t)b * c->size * part, part, ncclChar, c->comm, st);
    if (r != ncclSuccess) {
      ncclGroupEnd();
      return fail_nccl(r, "ncclAllGather");
    }
  }
  r = ncclGroupEnd();
  return r == ncclSuccess ? VORTA_OK : fail_nccl(r, "ncclGroupEnd");
}
