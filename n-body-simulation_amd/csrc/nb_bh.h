// nb_bh.h -- Barnes-Hut pipeline of libnbody_hip.so (internal).
//
// Stage by stage it replaces the reference's GPU kernels
//   ResetKernel / ComputeBoundingBoxKernel / ConstructQuadTreeKernel / ComputeForceKernel
//   (src/gpu/BH/barnesHut_kernel.cu:34-444)
// while reproducing the accept/reject decisions of the CPU tree
//   BarnesHut / QuadTree (src/cpu/barnesHut.cpp:23-144, src/cpu/quadtree.cpp:41-237):
// one body per leaf, unbounded depth (31 levels here), rectangular midpoint cells.
#pragma once

#include "nb_common.h"

namespace nb {

constexpr int BH_KEY_LEVELS = 31;  // 2 bits per level in a 62-bit key
constexpr int BH_MAX_DEPTH = 34;   // tree levels the build loops over (31 cell levels, leaves, spare)
constexpr int BH_SCAN_BLOCKS = 1024;

constexpr int BOX_RING = 8;  // steps the pinned ring of boxes holds ...
constexpr int BOX_SLOT = 4;  // ... in slots of {width, height, 1.0 if a cell-exchange tree was cut short, -}

struct BhBuffers {
    bool owned = false;  // the arrays were hipMalloc'ed by bh_alloc (bh_free releases them)
    int n = 0;
    size_t node_cap = 0;
    double *box = nullptr;        // tl.x, tl.y, br.x, br.y (float-rounded values held in doubles)
    double *bbox_part = nullptr;  // per-block partial min/max
    uint64_t *keys[2] = {nullptr, nullptr};
    uint32_t *idx[2] = {nullptr, nullptr};
    int sorted = 0;               // which of the two holds the sorted result (always 0: see bh_sort)
    uint32_t *cut = nullptr;      // [0] first tree level whose cells are never opened (32: none),
                                  // [1] levels the keys are computed to, [2] first sort pass with work,
                                  // [3] 1: this step builds the dense pyramid, 0: the level-by-level tree,
                                  // [4] the level of the tree's leaves (= [1] except in a cell-exchange step)
    uint32_t *hist = nullptr;     // 256 x sort blocks
    uint32_t *hist_totals = nullptr;  // sort_totals_words(): keys per digit and row segment of the current pass
    int sort_blocks = 0;
    double *leaf_sum = nullptr;   // 4 doubles per sorted body: sums over the run of equal keys starting there
    uint32_t *chunk_info = nullptr;  // per 64 sorted bodies: first lane that starts a run (64: none) |
                                     // lane where the run that goes on into the next chunk starts (0xff: none) << 8
    uint4 *node_meta = nullptr;   // first body, body count, first child, child count | cell level << 8
    uint4 *split = nullptr;       // per frontier slot: the three quadrant boundaries and child count
    uint32_t *level_start = nullptr;  // BH_MAX_DEPTH + 2 entries
    uint32_t *bsum = nullptr;     // BH_SCAN_BLOCKS block sums
    double *node_com = nullptr;   // 4 doubles per node: com.x, com.y, mass, and the cell's x-width,
                                  // or for a leaf minus its body's radius (sign bit set)
    float4 *node_rec = nullptr;   // traversal record {com.x, com.y, mass, w}: w = (width/theta)^2 for a
                                  // cell, -(radius + collision_th) for a leaf, internal units
    uint2 *node_child = nullptr;  // x = first child slot << 4 | mask of the child slots in use, y = float
                                  // bits of the prune threshold: (collision_th - cell diagonal)^2, or -1
    uint32_t *tlist = nullptr;    // sorted positions of this rank's targets (multi-GPU), else unused
    uint32_t *tcount = nullptr;   // 2 entries used in turn: number of entries in tlist (groups of 64, see bh_target_list)
    unsigned tl_calls = 0;        // lists built so far
    unsigned long long *counters = nullptr;  // 16 statistics words (NB_DBG_COUNTERS, NB_DBG_VISIT_KINDS)
    uint32_t *wave_cost = nullptr;           // what each 64-body group of the last traversal took, in 10 ns ticks (NB_DBG_WAVE_COST)
    uint32_t *levels_host = nullptr;  // pinned host word: tree levels of the last finished step
    uint32_t *hint_host = nullptr;    // pinned host word: level of the leaves of the last finished step's tree
    double *hint_box = nullptr;       // pinned host: width and height of the last finished step's bounding box
    double *box_ring = nullptr;       // pinned host: the same by step, slot ring_step % BOX_RING (cell exchange: every
    long ring_step = -1;              //   rank reads the box of ONE agreed earlier step); -1: not recorded
    uint32_t *err_host = nullptr;     // pinned host word: set when a cell-exchange step met a tree that closes
                                      // deeper than the level the cells were exchanged at
    int launched = 0;                 // levels the per-level build kernels were launched for
    uint2 *pc_range = nullptr;        // dense pyramid: first and one-past-last sorted body of every cell of the
                                      // keys' last level (both zero: empty cell); null: no pyramid
    double *agg = nullptr;            // cell exchange: one record per cell of the keys' last level (bh_cells_alloc)
    double *red4 = nullptr;           // cell exchange: {-min x, max x, -min y, max y} of the own bodies
    uint32_t *err = nullptr;          // (unused device word kept beside them)
    uint32_t step_hint = 0xffffffffu; // *hint_host as bh_bbox read it for this step's launches
    bool use_pyr = false;             // this step's kernels are launched for the dense-pyramid build ...
    int pyr_levels = -1;              // ... and can do keys of up to this many levels (-1: not launched)
    int pyr_max = 10;                 // the deepest key level the pyramid of these buffers has slots for
    int sort_first = 0;               // the first sort pass this step's launches include
};

int bh_alloc(BhBuffers &b, int n);
void bh_free(BhBuffers &b);
// The buffers laid out inside a caller-provided device arena of `bytes` bytes, nothing allocated;
// with base == nullptr only *used (bytes needed for n bodies) is computed.
int bh_carve(BhBuffers &b, int n, void *base, size_t bytes, size_t *used);

// The per-body arrays of a 2-D simulation (one of its two sets).
struct BodyArrays2 {
    double2 *pos = nullptr, *vel = nullptr, *acc = nullptr;
    double *mass = nullptr, *radius = nullptr;
    double2 *mass_radius = nullptr;
    uint8_t *dyn = nullptr;
    uint32_t *order = nullptr;  // caller index of the body in each slot
};

struct BhStepArgs {
    const double2 *pos;
    const double *mass;
    const double *radius;
    const double2 *mass_radius;  // {mass, radius} per body: the same values, gathered together
    const uint8_t *dyn;
    double2 *vel;
    double2 *acc;
    double2 *pos_out;  // same array as pos (integration is in place)
    int n;
    int t0, t1;        // exchange-order index range of the bodies this rank integrates
    Units units;
    ForceConsts fc;
    IntegrateArgs ia;
    double theta;
    double bbox_margin;
    double collision_th;
    int bucket_depth;  // 0: one body per leaf; k: leaves are the cells of level k (point masses)
    int collect_stats;
    // to.pos != null: the step leaves the bodies in ITS sorted order in the arrays of `to` (the body at
    // sorted position k in slot k: pos, vel, acc as the step computes them, everything else copied from
    // `from`), and the arrays it read keep the old state.  Only for a step of all n bodies (t0 = 0,
    // t1 = n, ia.do_integrate).  The sort's permutation then stays close to the identity, and the leaf
    // pass's and the traversal's gathers through it close to coalesced: a run that keeps its slots loses
    // that within a handful of steps as the bodies mix.
    BodyArrays2 from, to;
};

// Pieces the 3-D pipeline (nb_bh3.hip) shares.
// Stable LSD radix sort of (key, index) pairs by key << 2, ping-pong between the two buffers;
// the result is in buffer 0.  cut[2] = first pass with work (sort_first_pass in nb_dev.h); the
// unsorted pairs must be in buffer cut[2] & 1.  hist: sort_hist_words(n) words, totals: sort_totals_words().
// first_launch: passes below it are not even launched -- only for a caller that chose the key width
// itself and so knows that cut[2] >= first_launch, or that makes the device keep cut[2] at or above
// it (bh_sort); 0 launches all eight (the device decides).
int sort_blocks(int n);
size_t sort_hist_words(int n);
size_t sort_totals_words();
int sort_pairs(hipStream_t s, uint64_t *const keys[2], uint32_t *const idx[2], int n, const uint32_t *cut, uint32_t *hist,
               uint32_t *totals, int first_launch = 0);
// Level-by-level tree over sorted keys whose digits (bits = 2: quadtree, 3: octree) start at bit
// 61.  node_meta: {first body, count, first child, child count | cell level << 8}; split:
// (n + 256) * build_split_bytes(bits) bytes; level_start: BH_MAX_DEPTH + 2 words; bsum:
// BH_SCAN_BLOCKS words; counters: 16 words, zeroed here.
size_t build_split_bytes(int bits);
int level_blocks(int d, int bits, int nodes_per_block);  // grid of a per-level kernel (<= BH_SCAN_BLOCKS)
int launched_levels(const uint32_t *levels_host);
// When the tree is complete: counters[2] = nodes, [3] = levels; *levels_host = levels (pinned host word).
int build_tree(hipStream_t s, int bits, const uint64_t *keys, int n, uint4 *node_meta, void *split, uint32_t *level_start,
               uint32_t *bsum, unsigned long long *counters, int launched, uint32_t *levels_host);
// Sorted positions k whose body sorted_idx[k] lies in [t0, t1), ascending, into tlist (target_list_capacity(n)
// words) in groups of 64 entries: the targets of every tile of the sorted order start a group of their own, and
// the entries a tile's last group has left over are TL_PAD.  The number of entries, a multiple of 64, into
// **count (at most 64 target_list_groups(n, t1 - t0)); the tiles' groups come in any order.  tcount: two words,
// both zero before the first call, used in turn (`calls` counts the calls; *count is the one this call used).
constexpr uint32_t TL_PAD = 0xffffffffu;
size_t target_list_blocks(int n);
size_t target_list_capacity(int n);
int target_list_groups(int n, int ntarget);
int bh_target_list(hipStream_t s, const uint32_t *sorted_idx, int n, int t0, int t1, uint32_t *tlist, uint32_t *tcount,
                   unsigned &calls, const uint32_t **count);
// Second half of the leaf pass: adds to every run of equal keys that crosses 64-body chunk
// boundaries the parts its chunks left in leaf_sum (dims 2: slots of 4 doubles {3 sums, max
// radius}; 3: slots of 6 doubles {4 sums, max radius, -}); chunk_info: see nb_dev.h.
int leaf_finish(hipStream_t s, const uint32_t *chunk_info, int n, double *leaf_sum, int dims);

// Individual stages (also driven one by one by the stage tests through nb_step).
int bh_bbox(hipStream_t s, BhBuffers &b, const BhStepArgs &a);
int bh_prime(hipStream_t s, BhBuffers &b, const BhStepArgs &a);  // after new bodies: learns the key depth (waits)
int bh_keys(hipStream_t s, BhBuffers &b, const BhStepArgs &a);
int bh_sort(hipStream_t s, BhBuffers &b, const BhStepArgs &a);
int bh_build(hipStream_t s, BhBuffers &b, const BhStepArgs &a);
int bh_com(hipStream_t s, BhBuffers &b, const BhStepArgs &a);
int bh_force(hipStream_t s, BhBuffers &b, const BhStepArgs &a);

// Cell exchange between GPUs (see nb_bh.hip): a step in three parts around two reductions.
int bh_cells_alloc(BhBuffers &b);
int cells_max_level();
size_t cells_agg_doubles(int cells_levels);
int closed_level_for(double w0, double h0, double theta, double cth);
int bh_cells_begin(hipStream_t s, BhBuffers &b, const BhStepArgs &a);
int bh_cells_middle(hipStream_t s, BhBuffers &b, const BhStepArgs &a, int cells_levels, int part);
int bh_cells_end(hipStream_t s, BhBuffers &b, const BhStepArgs &a, int cells_levels, int part);

}  // namespace nb
