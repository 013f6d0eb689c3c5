// nb_bh_internal.h -- what the translation units of the 2-D Barnes-Hut pipeline share beyond
// nb_bh.h (internal):
//   nb_bh.hip        bounding box, keys, traversal + integration, buffers
//   nb_bh_sort.hip   radix sort, scans, the list of a rank's targets
//   nb_bh_tree.hip   level-by-level build, leaf pass, centres of mass, dense pyramid, cell exchange
#pragma once

#include "nb_bh.h"
#include "nb_dev.h"

namespace nb {

// Dense pyramid: when the keys stop at a level L <= PYR_MAX_LEVEL (a closed level, or the depth cap
// of the GPU-program tree), every cell of levels 0..L has a slot of its own in the node arrays,
// cell c of level l at pyr_offset(l) + c (c = the cell's quadrant path, 2 bits a level), so the four
// children of a cell are the consecutive slots pyr_offset(l + 1) + 4 c .. + 3 and nothing has to be
// numbered or searched for.
constexpr int PYR_MAX_LEVEL = 10;
// Systems whose 2 n node slots hold the cells of one more level anyway get the pyramid down to
// level 11 (a collision run whose ejected bodies stretch the box reaches it): BhBuffers::pyr_max.
#ifndef NB_PYR_DEEP_LEVEL
#define NB_PYR_DEEP_LEVEL 11
#endif
constexpr int PYR_DEEP_LEVEL = NB_PYR_DEEP_LEVEL;  // (10 switches the deeper pyramid off: tuning aid)
__host__ __device__ constexpr uint32_t pyr_offset(int l) { return (uint32_t)((((uint64_t)1 << (2 * l)) - 1) / 3); }
constexpr uint32_t PYR_CELLS = pyr_offset(PYR_MAX_LEVEL + 1);        // cells of all levels
constexpr uint32_t PYR_DEEP_CELLS = pyr_offset(PYR_DEEP_LEVEL + 1);

// Keys and indices travelling as one word through the radix sort (k_bbox_final decides, cut[5]).
constexpr int SORT_PACK_LEVELS = 15;        // key digits that leave bits 31..0 of the key word zero
constexpr int SORT_PACK_MAX_N = 1 << 30;    // the index stays below bit 30, the lowest bit a sort pass with work reads
constexpr uint32_t SORT_PACK_MASK = (1u << 30) - 1u;
constexpr int SORT_BINS = 256;  // 8-bit digits
#ifndef NB_SORT_SEGS
#define NB_SORT_SEGS 16
#endif
constexpr int SORT_SEGS = NB_SORT_SEGS;  // segments a digit's row of per-block counts is scanned in

constexpr int BBOX_BLOCKS = 1024;

// Where this step's box goes in the pinned ring (null: nobody asked, b.ring_step < 0).
inline double *box_ring_slot(const BhBuffers &b)
{
    return (b.box_ring && b.ring_step >= 0) ? b.box_ring + BOX_SLOT * (b.ring_step % BOX_RING) : nullptr;
}

// kernels of nb_bh.hip that the cell exchange (nb_bh_tree.hip) launches too
__global__ __launch_bounds__(256) void k_bbox_partial(const double2 *pos, int n, double *part);
__global__ __launch_bounds__(256) void k_bbox_final(const double *part, int nblk, double margin, int bucket_depth,
                                                    double theta, double cth, double *box, uint32_t *cut, int pyr_levels,
                                                    uint32_t *hint_host, double *hint_box, int cells_levels, uint32_t *err,
                                                    int n, int sort_first, double *ring_slot);
__global__ __launch_bounds__(256) void k_bbox_pack(const double *part, int nblk, double *red4);

}  // namespace nb
