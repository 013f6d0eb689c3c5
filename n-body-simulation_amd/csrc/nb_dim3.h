// nb_dim3.h -- the three-dimensional paths of libnbody_hip.so (internal).
//
// The reference is two-dimensional; its authors name 3-D and an octree as future work.  These
// paths apply its rules with a third coordinate: the force law and cutoff of
// DirectSum::calculateAcceleration (src/cpu/directSum.cpp:24-72), and for Barnes-Hut an octree
// built the way QuadTree::insert builds its quadtree (src/cpu/quadtree.cpp:41-144: one body per
// leaf, cells split at the midpoint of every axis, z treated like y), centres of mass as
// updateCenterMass (quadtree.cpp:187-228) and the traversal of BarnesHut::calculateForceHelper
// (src/cpu/barnesHut.cpp:50-89: cell x-width / distance < theta).
//
// FP64 state is held as packed triples (3 doubles per body).
#pragma once

#include "nb_common.h"

namespace nb {

// ---- direct sum (nb_direct3.hip; kernels in nb_direct_impl.h) ----
struct Direct3State {
    float4 *posm[2] = {nullptr, nullptr};  // {x, y, z, m} per body, internal units; [cur] is current; each holds
                                           // 2 * direct3_record_capacity(n) float4: behind the records, the
                                           // low parts {x, y, z, -} of the coordinates (what rounding the scaled
                                           // FP64 coordinate to the record's float left over)
    int cur = 0;
    float *wcut = nullptr;  // radius + collision_th
    float *rad = nullptr;   // radius
    const uint8_t *dyn = nullptr;
    const double *pos_cur = nullptr;
    double *pos_next = nullptr;
    const double *radius = nullptr;
    double *vel = nullptr, *acc = nullptr;
    double *part = nullptr;  // jsplit x targets x 3 partial sums
    int jsplit = 1;
    void *meta = nullptr;     // 2 x direct3_meta_bytes(n): the boxes of the source runs, one set per record buffer
    bool meta_valid = false;  // the set of posm[cur] is up to date (direct3_step keeps this; cleared when records are packed)
};

size_t direct3_record_capacity(int n);
size_t direct3_meta_bytes(int n);
int direct3_choose_jsplit(int n, int nt);
// Fills posm[0..cap), wcut[0..cap) and rad[0..n) from the FP64 state (cap = direct3_record_capacity).
int pack_posm3(hipStream_t s, const double *pos, const double *mass, const double *radius, int n, const Units &units,
               double collision_th, float4 *posm, float *wcut, float *rad);
// Records of bodies [i0, i1) only (the ones other ranks advanced, from the gathered positions).
int pack_posm3_range(hipStream_t s, const double *pos, const double *mass, int n, int i0, int i1, const Units &units,
                     float4 *posm);
// One step for targets [t0, t0 + nt): reads posm[cur] and pos_cur, writes the targets' vel, acc,
// pos_next and posm[cur ^ 1].
int direct3_step(hipStream_t s, Direct3State &st, int n, int t0, int nt, const ForceConsts &fc, const IntegrateArgs &ia,
                 double collision_th);

// ---- Barnes-Hut (nb_bh3.hip) ----
constexpr int BH3_KEY_LEVELS = 20;  // 3 bits per level, bits 61..2 of the key word

struct Bh3Buffers {
    int n = 0;
    size_t node_cap = 0;
    double *box = nullptr;        // tl.x, tl.y, br.x, br.y, upper z, lower z (float-rounded values)
    double *bbox_part = nullptr;
    uint64_t *keys[2] = {nullptr, nullptr};
    uint32_t *idx[2] = {nullptr, nullptr};
    uint32_t *cut = nullptr;      // [0] first closed level (32: none), [1] key levels, [2] first sort pass
    uint32_t *hist = nullptr, *hist_totals = nullptr;
    double *leaf_sum = nullptr;   // 6 doubles per sorted body: sum m x, m y, m z, sum m, max radius, -
    uint32_t *chunk_info = nullptr;  // per 64 sorted bodies, see nb_dev.h
    float4 *posm_s = nullptr;     // per sorted body {x, y, z, m}, internal units
    float *rad_s = nullptr;       // per sorted body: radius
    uint4 *node_meta = nullptr;   // first body, body count, first child, child count | cell level << 8
    void *split = nullptr;
    uint32_t *level_start = nullptr;
    uint32_t *bsum = nullptr;
    double *node_com = nullptr;   // 6 doubles per node: com x, y, z, mass, cell x-width (a leaf: minus
                                  // its body's radius, sign bit set), -
    float4 *node_rec = nullptr;   // {com x, y, z, mass}, internal units
    uint4 *node_aux = nullptr;    // x = float bits of (width/theta)^2, or of -(radius + collision_th) for a
                                  // leaf; y = first child << 4 | child count; z = float bits of the prune
                                  // threshold (collision_th - cell diagonal)^2 or -1
    uint32_t *tlist = nullptr;    // sorted positions of this rank's targets (partitioned runs)
    uint32_t *tcount = nullptr;   // 2 entries used in turn (bh_target_list)
    unsigned tl_calls = 0;
    unsigned long long *counters = nullptr;
    uint32_t *wave_cost = nullptr;  // as BhBuffers::wave_cost
    uint32_t *levels_host = nullptr;
    int launched = 0;
};

struct Bh3StepArgs {
    double *pos;  // packed triples; integrated in place
    const double *mass;
    const double *radius;
    const uint8_t *dyn;
    double *vel;
    double *acc;
    int n;
    int t0, t1;  // slots of the bodies this rank integrates
    Units units;
    ForceConsts fc;
    IntegrateArgs ia;
    double theta, bbox_margin, collision_th;
    int collect_stats;
};

int bh3_alloc(Bh3Buffers &b, int n);
void bh3_free(Bh3Buffers &b);
int bh3_bbox(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a);
int bh3_keys(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a);
int bh3_sort(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a);
int bh3_build(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a);
int bh3_com(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a);
int bh3_force(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a);

}  // namespace nb
