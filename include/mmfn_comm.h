/* C ABI of libmmfn_comm.so: the data-parallel gradient exchange of the MMFN training step over RCCL (xGMI).
 *
 * Replaces, under the model class, what torch DDP's NCCL reducer does for the reference
 * (run_steps/phase2_train_net.py:227 `DDP(model, ..., find_unused_parameters=True)`, :269 backward hooks):
 * sum-all-reduce of gradient buckets, plus the initial parameter / buffer broadcast.  SURVEY.md section 8(b)
 * lists `mmfn_allreduce_*` with the RCCL communicator passed opaquely: that is `void* comm` (= ncclComm_t) here.
 *
 * Every collective is enqueued on the caller's HIP stream; nothing is allocated, nothing synchronises, and the calls
 * are hipGraph-capturable.  Return 0 = success; a positive value is the ncclResult_t, -1 = bad argument.
 * The library is separate from libmmfn_hip.so so that the kernels carry no RCCL dependency. */
#ifndef MMFN_COMM_H
#define MMFN_COMM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MMFN_COMM_ID_BYTES 128 /* sizeof(ncclUniqueId) */

int mmfn_comm_abi_version(void);
/* rank 0: create the rendezvous id (128 bytes) that every rank passes to mmfn_comm_init; the host program carries it to
 * the other ranks over whatever channel it has (mmfn_amd.comm uses the torch.distributed store the launcher set up). */
int mmfn_comm_unique_id(void* out_id);
/* ncclCommInitRank on the calling thread's current HIP device (one process per GPU). */
int mmfn_comm_init(void** comm, const void* id_bytes, int nranks, int rank);
/* Destroy the communicator.  Every hipGraph that captured a collective of it must have been destroyed first (RCCL reaches
 * into the communicator when such a graph dies); mmfn_amd.comm.RcclComm.destroy() enforces the order. */
int mmfn_comm_destroy(void* comm);
int mmfn_comm_ranks(void* comm, int* nranks, int* rank);
/* In-place sum all-reduce of n floats (a gradient bucket: a contiguous range of the flat gradient buffer); the 1/ranks
 * average is folded into the AdamW launch (mmfn_adamw_groups_f32 grad_scale), not applied here. */
int mmfn_allreduce_sum_f32(void* comm, float* buf, int64_t n, void* stream);
/* The same over n bf16 elements: the bf16 training mode (BASELINE configs[2]) sends its gradient buckets as bf16 - 210 MB instead of
 * 419 MB per step over xGMI, whose ring all-reduce is bound by one ~150 GB/s link - after a cast of the
 * fp32 bucket (mmfn_cast_f32_to_bf16) and casts the sum back into the fp32 gradient buffer (mmfn_cast_bf16_to_f32); master
 * weights, moments and the AdamW arithmetic stay fp32. */
int mmfn_allreduce_sum_bf16(void* comm, void* buf, int64_t n, void* stream);
/* In-place broadcast of nbytes from `root` (initial parameters, BatchNorm buffers, optimizer state after a resume). */
int mmfn_broadcast_bytes(void* comm, void* buf, int64_t nbytes, int root, void* stream);

#ifdef __cplusplus
}
#endif
#endif
