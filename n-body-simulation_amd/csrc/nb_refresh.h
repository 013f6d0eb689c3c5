// nb_refresh.h -- re-deriving the exchange order on the device (internal).
//
// Bodies are held sorted by their position along a Hilbert curve (nb_common.h) when they were set; they wander, and
// the direct-sum kernel (whose run-against-wave box tests rely on neighbouring slots being
// neighbours in space) slows down as they do.  The reference keeps bodies in creation order and
// has nothing of the kind.  The renewal is a bounding box, a radix sort of (curve index, slot)
// pairs and one gather per per-body array, all enqueued on the simulation's stream: the host does
// not wait, and a partitioned run can do it too (every rank holds all positions, so every rank
// derives the same order).
#pragma once

#include "nb_common.h"

namespace nb {

struct RefreshBuffers {
    int n = 0;
    uint64_t *keys[2] = {nullptr, nullptr};
    uint32_t *idx[2] = {nullptr, nullptr};
    uint32_t *hist = nullptr, *totals = nullptr, *cut = nullptr;
    double *part = nullptr;  // per-block minima and maxima
    double *box = nullptr;   // lo[3], hi[3]
    void *tmp = nullptr;     // one array's worth of bytes (24 n): gathers go through it
    double *origin_host = nullptr;  // pinned: the middle of the box of the last renewal, written by the device
    hipEvent_t box_done = nullptr;  // recorded behind the kernel that writes origin_host
    bool origin_set = false;        // a renewal was enqueued since the bodies were last set
};

int refresh_alloc(RefreshBuffers &r, int n);
void refresh_free(RefreshBuffers &r);
// New order from the positions (dims = 2: records {x, y}; 3: {x, y, z}): afterwards r.idx[0][k] is
// the slot whose body moves to slot k.  Equal codes keep their present order.
int refresh_sort(hipStream_t s, RefreshBuffers &r, const double *pos, int dims, int n);
// array[k] <- array[r.idx[0][k]] for n elements of elem_bytes (1, 4, 8, 16 or 24) bytes.
int refresh_gather(hipStream_t s, RefreshBuffers &r, void *array, size_t elem_bytes, int n);

}  // namespace nb
