// Gradient all-reduce of the data-parallel step as a C ABI over RCCL (include/mmfn_comm.h): the collective that replaces
// torch DDP's NCCL bucket reduction (run_steps/phase2_train_net.py:227,269).  A separate shared library so that
// libmmfn_hip.so (the kernels) has no RCCL dependency: single-GPU users never load it.
//
// The communicator is passed opaquely (void* = ncclComm_t).  Launches are enqueued on the caller's HIP stream, allocate
// nothing and do not synchronise - RCCL collectives on a user stream can be captured into a hipGraph, which torch's
// ProcessGroup collectives cannot (they run on RCCL-owned streams), so a data-parallel step built on this entry is ONE
// graph instead of five graphs cut at the bucket boundaries.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <string.h>

#include "../../include/mmfn_comm.h"

static_assert(NCCL_UNIQUE_ID_BYTES == MMFN_COMM_ID_BYTES, "unique-id size");

extern "C" int mmfn_comm_abi_version(void) { return 1; }

extern "C" int mmfn_comm_unique_id(void* out_id) {
  if (!out_id) return -1;
  ncclUniqueId id;
  const ncclResult_t r = ncclGetUniqueId(&id);
  if (r != ncclSuccess) return (int)r;
  memcpy(out_id, &id, sizeof(id));
  return 0;
}

extern "C" int mmfn_comm_init(void** comm, const void* id_bytes, int nranks, int rank) {
  if (!comm || !id_bytes || nranks < 1 || rank < 0 || rank >= nranks) return -1;
  ncclUniqueId id;
  memcpy(&id, id_bytes, sizeof(id));
  ncclComm_t c = nullptr;
  const ncclResult_t r = ncclCommInitRank(&c, nranks, id, rank);   // uses the calling thread's current HIP device
  if (r != ncclSuccess) return (int)r;
  *comm = (void*)c;
  return 0;
}

extern "C" int mmfn_comm_destroy(void* comm) {
  if (!comm) return 0;
  return (int)ncclCommDestroy((ncclComm_t)comm);
}

extern "C" int mmfn_comm_ranks(void* comm, int* nranks, int* rank) {
  if (!comm || !nranks || !rank) return -1;
  ncclResult_t r = ncclCommCount((ncclComm_t)comm, nranks);
  if (r != ncclSuccess) return (int)r;
  return (int)ncclCommUserRank((ncclComm_t)comm, rank);
}

extern "C" int mmfn_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream) {
  if (!comm || (!buf && n > 0) || n < 0) return -1;
  if (n == 0) return 0;
  return (int)ncclAllReduce(buf, buf, (size_t)n, ncclFloat, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
}

extern "C" int mmfn_allreduce_sum_bf16(void* comm, void* buf, int64_t n, void* stream) {
  if (!comm || (!buf && n > 0) || n < 0) return -1;
  if (n == 0) return 0;
  return (int)ncclAllReduce(buf, buf, (size_t)n, ncclBfloat16, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
}

extern "C" int mmfn_broadcast_bytes(void* comm, void* buf, int64_t nbytes, int root, void* stream) {
  if (!comm || (!buf && nbytes > 0) || nbytes < 0) return -1;
  if (nbytes == 0) return 0;
  return (int)ncclBroadcast(buf, buf, (size_t)nbytes, ncclUint8, root, (ncclComm_t)comm, (hipStream_t)stream);
}
