// nb_comm.h -- RCCL communicator behind nb_comm_init / nb_step_distributed (internal).
//
// The reference is single-GPU.  Across the GPUs of one node every process holds all bodies,
// advances its own slice and the new positions are exchanged by ONE in-place ncclAllGather per
// step, enqueued on the simulation's stream between the kernels of that step: the host never
// waits inside a step.  librccl is loaded on first use (dlopen by soname, so a process that already
// holds an RCCL -- PyTorch ships one -- shares that copy); single-GPU users never load it.
#pragma once

#include "nb_common.h"

namespace nb {

struct Comm;  // opaque

constexpr size_t COMM_ID_BYTES = 128;  // sizeof(ncclUniqueId)

// Fills id[COMM_ID_BYTES] (ncclGetUniqueId): one rank calls it and hands the bytes to the others.
int comm_unique_id(void *id);
// Joins the communicator of `world` ranks as `rank`, on the current device (collective call).
int comm_create(Comm **out, int rank, int world, const void *id);
void comm_destroy(Comm *c);
int comm_rank(const Comm *c);
int comm_world(const Comm *c);
// In-place all-gather of doubles on `stream`: rank k contributes buf[k * count, (k + 1) * count).
int comm_all_gather_f64(Comm *c, double *buf, size_t count, hipStream_t stream);
// In-place maximum over the ranks of `count` doubles.
int comm_all_reduce_max_f64(Comm *c, double *buf, size_t count, hipStream_t stream);
// In-place sum over the ranks of `count` doubles.
int comm_all_reduce_sum_f64(Comm *c, double *buf, size_t count, hipStream_t stream);

}  // namespace nb
