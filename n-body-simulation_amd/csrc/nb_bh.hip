// nb_bh.hip -- Barnes-Hut pipeline for gfx950 (see nb_bh.h for what it replaces).
//
// Stages of one step, all on one stream, no host synchronisation in between:
//   1 bbox     per-block min/max by wave shuffles, one finishing block; the box is rounded to
//              float and padded as BarnesHut::computeBoundingBox does (barnesHut.cpp:23-35)
//   2 keys     62-bit quadrant paths by the reference's own FP64 midpoint descent
//              (quadtree.cpp:41-72,107-143), so ties and rounding fall where they fall there
//   3 sort     stable LSD radix sort of (key, index), 8 bits per pass
//   4 build    quadtree of the cells that hold >= 2 bodies in >= 2 quadrants, level by level:
//              children of a node are contiguous, a node's quadrant boundaries come from
//              binary searches in the sorted keys.  Chains of single-child cells of the
//              reference tree are not stored: they share their descendant's centre of mass and
//              a wider cell, so the opening test passes for the chain iff it passes for its
//              smallest cell, with the same force.
//   5 com      bottom-up centre of mass, FP64, children summed in the reference's order
//              TL,TR,BL,BR (quadtree.cpp:187-228)
//   6 force    one wave per 64 consecutive bodies of the sorted order; the wave shares a stack of
//              (sibling group, lane mask) entries in LDS, fetches up to 64 nodes per round trip, and
//              every lane applies its OWN opening test (barnesHut.cpp:70-71), so each body
//              accepts exactly the cells the reference accepts for it.  Integration
//              (barnesHut.cpp:108-123) is fused into the epilogue.
#include <cmath>
#include <type_traits>

#include "nb_bh_internal.h"
#include "nb_traverse.h"

namespace nb {

// ------------------------------------------------------------------ 1 bounding box


__global__ __launch_bounds__(256) void k_bbox_partial(const double2 *pos, int n, double *part)
{
    double mnx = __builtin_inf(), mxx = -__builtin_inf(), mny = __builtin_inf(), mxy = -__builtin_inf();
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        double2 p = pos[i];
        mnx = fmin(mnx, p.x);
        mxx = fmax(mxx, p.x);
        mny = fmin(mny, p.y);
        mxy = fmax(mxy, p.y);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mnx = fmin(mnx, __shfl_xor(mnx, o));
        mxx = fmax(mxx, __shfl_xor(mxx, o));
        mny = fmin(mny, __shfl_xor(mny, o));
        mxy = fmax(mxy, __shfl_xor(mxy, o));
    }
    __shared__ double sh[4][4];
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) {
        sh[w][0] = mnx;
        sh[w][1] = mxx;
        sh[w][2] = mny;
        sh[w][3] = mxy;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            mnx = fmin(mnx, sh[k][0]);
            mxx = fmax(mxx, sh[k][1]);
            mny = fmin(mny, sh[k][2]);
            mxy = fmax(mxy, sh[k][3]);
        }
        double *o = part + 4 * blockIdx.x;
        o[0] = mnx;
        o[1] = mxx;
        o[2] = mny;
        o[3] = mxy;
    }
}

// cells_levels >= 0 (cell exchange between GPUs): `part` is ONE record {-min x, max x, -min y, max y}
// of all ranks' bodies (minima negated so that one max-reduction serves), and the keys get
// cells_levels digits whatever level closes the tree (cut[4]); a tree that closes deeper than that
// cannot be exchanged at this level: *err is set.
// First level whose cells no body ever opens in a tree over a box of width w0 and height h0 (32:
// none): a cell is opened by a body that neither accepts it (dist <= width / theta) nor sees it
// wholly inside its cutoff distance (dist >= collision_th - diagonal); when width / theta <
// collision_th - diagonal no distance does both.  The margins (1e-4 each side) cover the float
// rounding of the two thresholds the traversal compares.
// One level a lane: lane l and lane 32 + l test level l, and the two halves of the wave may be given
// different boxes.  All 64 lanes must be active; every lane returns the answer of its half.
__device__ __forceinline__ uint32_t first_closed_level(double w0, double h0, double theta, double cth)
{
    static_assert(BH_KEY_LEVELS == 31, "one lane per level, 32 levels");
    const int lane = lane_id(), l = lane & 31;
    const double w = ldexp(w0, -l), h = ldexp(h0, -l);
    const double reach = cth * (1.0 - 1.0e-4) - sqrt(w * w + h * h);
    const bool closed = reach > 0.0 && reach * (1.0 - 1.0e-4) > w / theta;
    // cells only grow towards the root: the answer is the level above the deepest one that fails
    const uint64_t both = __builtin_amdgcn_ballot_w64(!closed);
    const uint32_t fails = lane < 32 ? (uint32_t)both : (uint32_t)(both >> 32);
    return fails ? 32u - (uint32_t)__builtin_clz(fails) : 0u;
}


__global__ __launch_bounds__(256) void k_bbox_final(const double *part, int nblk, double margin, int bucket_depth,
                                                    double theta, double cth, double *box, uint32_t *cut, int pyr_levels,
                                                    uint32_t *hint_host, double *hint_box, int cells_levels, uint32_t *err,
                                                    int n, int sort_first, double *ring_slot)
{
    double mnx = __builtin_inf(), mxx = -__builtin_inf(), mny = __builtin_inf(), mxy = -__builtin_inf();
    for (int i = threadIdx.x; i < nblk; i += 256) {
        mnx = fmin(mnx, cells_levels >= 0 ? -part[4 * i + 0] : part[4 * i + 0]);
        mxx = fmax(mxx, part[4 * i + 1]);
        mny = fmin(mny, cells_levels >= 0 ? -part[4 * i + 2] : part[4 * i + 2]);
        mxy = fmax(mxy, part[4 * i + 3]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mnx = fmin(mnx, __shfl_xor(mnx, o));
        mxx = fmax(mxx, __shfl_xor(mxx, o));
        mny = fmin(mny, __shfl_xor(mny, o));
        mxy = fmax(mxy, __shfl_xor(mxy, o));
    }
    __shared__ double sh[4][4];
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0) {
        sh[w][0] = mnx;
        sh[w][1] = mxx;
        sh[w][2] = mny;
        sh[w][3] = mxy;
    }
    __syncthreads();
    if (threadIdx.x >= 64)
        return;
    // The first wave goes on, every lane with the same box: the level tests below run one level a
    // lane (one lane walking down the levels spent 10 us of every step here); lane 0 writes.
    const bool writer = threadIdx.x == 0;
    mnx = sh[0][0];
    mxx = sh[0][1];
    mny = sh[0][2];
    mxy = sh[0][3];
    for (int k = 1; k < 4; ++k) {
        mnx = fmin(mnx, sh[k][0]);
        mxx = fmax(mxx, sh[k][1]);
        mny = fmin(mny, sh[k][2]);
        mxy = fmax(mxy, sh[k][3]);
    }
    // fminf/fmaxf of the reference round every candidate to float (barnesHut.cpp:28-31);
    // rounding is monotone, so rounding the extreme gives the same box.
    if (bucket_depth > 0) {
        // the GPU program reduces the positions through fminf/fmaxf first and pads afterwards
        // (barnesHut_kernel.cu:94-108): two roundings per bound
        mnx = (double)(float)mnx;
        mxx = (double)(float)mxx;
        mny = (double)(float)mny;
        mxy = (double)(float)mxy;
    }
    const double tlx = (double)(float)(mnx - margin), tly = (double)(float)(mxy + margin);
    const double brx = (double)(float)(mxx + margin), bry = (double)(float)(mny - margin);
    // What lies below a cell that no body ever opens is never looked at, so the tree is not
    // built there and the keys stop at that level.
    uint32_t first_closed = 32u, ahead = 32u;
    if (bucket_depth == 0) {
        // the upper half of the wave answers for a box a tenth larger: the host sizes the coming
        // steps' launches by that, so that it has switched before the box gets there (it runs a few
        // steps ahead of the device)
        const double grow = lane_id() < 32 ? 1.0 : 1.1;
        const uint32_t level = first_closed_level(grow * (brx - tlx), grow * (tly - bry), theta, cth);
        first_closed = (uint32_t)__builtin_amdgcn_readlane((int)level, 0);
        ahead = (uint32_t)__builtin_amdgcn_readlane((int)level, 32);
    }
    if (!writer)
        return;
    box[0] = tlx;  // topLeft.x
    box[1] = tly;  // topLeft.y
    box[2] = brx;  // botRight.x
    box[3] = bry;  // botRight.y
    cut[0] = first_closed;
    uint32_t key_levels = bucket_depth > 0 ? (uint32_t)bucket_depth
                                           : (first_closed < (uint32_t)BH_KEY_LEVELS ? first_closed : (uint32_t)BH_KEY_LEVELS);
    uint32_t first_pass = (uint32_t)sort_first_pass(2 * (int)key_levels);
    // The dense-pyramid build runs when the host launched its kernels (pyr_levels >= 0) and
    // the keys stop at a level it covers; otherwise the level-by-level build does the step.
    uint32_t pyramid = (pyr_levels >= 0 && (int)key_levels <= pyr_levels) ? 1u : 0u;
    uint32_t leaf_level = key_levels;  // the level of the tree's leaves
    if (cells_levels >= 0) {
        key_levels = (uint32_t)cells_levels;
        first_pass = (uint32_t)sort_first_pass(2 * cells_levels);
        pyramid = 1u;
        leaf_level = first_closed <= (uint32_t)cells_levels ? first_closed : (uint32_t)cells_levels;
        if (first_closed > (uint32_t)cells_levels && err)
            *err = 1u;
    }
    cut[1] = key_levels;
    cut[3] = pyramid;
    cut[4] = leaf_level;
    // Keys of at most 15 digits leave the low 32 bits of the key word free: the body's index (below 2^30)
    // travels there through the sort instead of in an array of its own (see k_sort_scatter).
    cut[5] = (key_levels <= (uint32_t)SORT_PACK_LEVELS && n <= SORT_PACK_MAX_N) ? 1u : 0u;
    // The host launches the sort's passes from sort_first on (it sizes them by the keys of the
    // last step it knows): cut[2] becomes the first pass that RUNS, cut[7] keeps the first the
    // keys NEED; if that is lower, k_sort_fix finishes the order.
    cut[7] = first_pass;
    cut[2] = first_pass < (uint32_t)sort_first ? (uint32_t)sort_first : first_pass;
    if (hint_host) {
        // host-visible: sizes the next steps' launches (a stale value only costs speed)
        const uint32_t soon = ahead < (uint32_t)BH_KEY_LEVELS ? ahead : (uint32_t)BH_KEY_LEVELS;
        *hint_host = (cells_levels < 0 && bucket_depth == 0 && soon > leaf_level) ? soon : leaf_level;
    }
    if (hint_box) {           // host-visible: width and height of the box
        hint_box[0] = brx - tlx;
        hint_box[1] = tly - bry;
    }
    if (ring_slot) {          // host-visible, by step: the same for the ranks that share this box
        ring_slot[0] = brx - tlx;
        ring_slot[1] = tly - bry;
        // (every rank learns of a tree cut short at the same step, through the same slot)
        ring_slot[2] = (cells_levels >= 0 && first_closed > (uint32_t)cells_levels) ? 1.0 : 0.0;
    }
}

// Cell exchange: this rank's extent as ONE record {-min x, max x, -min y, max y} (one block).
__global__ __launch_bounds__(256) void k_bbox_pack(const double *part, int nblk, double *red4)
{
    double mnx = __builtin_inf(), mxx = -__builtin_inf(), mny = __builtin_inf(), mxy = -__builtin_inf();
    for (int i = threadIdx.x; i < nblk; i += 256) {
        mnx = fmin(mnx, part[4 * i + 0]);
        mxx = fmax(mxx, part[4 * i + 1]);
        mny = fmin(mny, part[4 * i + 2]);
        mxy = fmax(mxy, part[4 * i + 3]);
    }
    __shared__ double sh[256][4];
    sh[threadIdx.x][0] = mnx;
    sh[threadIdx.x][1] = mxx;
    sh[threadIdx.x][2] = mny;
    sh[threadIdx.x][3] = mxy;
    __syncthreads();
    if (threadIdx.x < 4) {
        const int d = threadIdx.x;
        double v = sh[0][d];
        for (int t = 1; t < 256; ++t)
            v = (d & 1) ? fmax(v, sh[t][d]) : fmin(v, sh[t][d]);
        red4[d] = (d & 1) ? v : -v;
    }
}


int bh_bbox(hipStream_t s, BhBuffers &b, const BhStepArgs &a)
{
    int blocks = (a.n + 255) / 256;
    if (blocks > BBOX_BLOCKS)
        blocks = BBOX_BLOCKS;
    hipLaunchKernelGGL(k_bbox_partial, dim3(blocks), dim3(256), 0, s, a.pos, a.n, b.bbox_part);
    // which build this step's kernels are launched for: the dense pyramid when the keys of the last
    // finished step stopped at a level it covers
    const uint32_t hint = (b.pc_range && b.hint_host) ? *(volatile const uint32_t *)b.hint_host : 0xffffffffu;
    b.step_hint = hint;  // (bh_com sizes its grid by the same reading: the device may update the word in between)
    b.use_pyr = hint <= (uint32_t)b.pyr_max;
    // the deepest key level the pyramid kernels launched for this step can do (k_pyr_mid is only
    // launched when the hint is beyond PYR_MAX_LEVEL; deeper keys than that take the fallback)
    b.pyr_levels = !b.use_pyr ? -1 : (hint > (uint32_t)PYR_MAX_LEVEL ? b.pyr_max : PYR_MAX_LEVEL);
    // The sort's passes below the keys' digits would return at once, at a launch each (13 us for the
    // three kernels of an idle pass at N = 5 M): they are not launched.  The hint already is the
    // level a box a tenth larger would need, and the host runs at most three steps ahead; a step
    // whose keys still outgrow the launched passes is finished by k_sort_fix.
    b.sort_first = 0;
    if (hint <= (uint32_t)BH_KEY_LEVELS)
        b.sort_first = sort_first_pass(2 * (int)hint);
    hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, s, b.bbox_part, blocks, a.bbox_margin, a.bucket_depth,
                       a.theta, a.collision_th, b.box, b.cut, b.pyr_levels, b.hint_host, b.hint_box, -1, (uint32_t *)nullptr,
                       a.n, b.sort_first, box_ring_slot(b));
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// ------------------------------------------------------------------ 2 keys

// Digit of level l (0 = the root's split) occupies bits [61-2l, 60-2l]: the position along the cell's
// Hilbert curve of the quadrant 2*(below the y midpoint) + (right of the x midpoint), TL=0, TR=1, BL=2, BR=3.
// Only the first cut[1] digits are computed, the rest of the key is zero: bodies of one cell of
// that level then share a key and end up in one leaf.  The keys go to the buffer the first sort
// pass that has work (cut[2]) reads from (see sort_pairs).
__global__ __launch_bounds__(256) void k_keys(const double2 *pos, int n, const double *box, const uint32_t *cut,
                                              uint64_t *keys0, uint64_t *keys1, uint32_t *idx0, uint32_t *idx1,
                                              uint2 *pc_range)
{
    int i = blockIdx.x * 256 + threadIdx.x;
    const int levels = (int)cut[1];
    if (cut[3] != 0u) {
        // dense-pyramid build: the {first, end} slots of the cells of the keys' last level start
        // out empty (the leaf pass fills those that hold bodies)
        const uint32_t cells = 1u << (2 * levels);
        for (uint32_t j = (uint32_t)i; j < cells; j += gridDim.x * 256u)
            pc_range[j] = make_uint2(0u, 0u);
    }
    if (i >= n)
        return;
    double tlx = box[0], tly = box[1], brx = box[2], bry = box[3];
    const double2 p = pos[i];
    uint64_t k = 0;
    uint32_t curve = 0u;  // orientation of the Hilbert curve inside the current cell
#pragma unroll 1
    for (int l = 0; l < levels; ++l) {
        const double mx = (tlx + brx) / 2.0, my = (tly + bry) / 2.0;
        const bool right = !(mx >= p.x);  // getQuadrant: mid >= x is the left half
        const bool below = !(my <= p.y);  // mid <= y is the top half
        tlx = right ? mx : tlx;
        brx = right ? brx : mx;
        tly = below ? my : tly;
        bry = below ? bry : my;
        // The quadrant (the reference's decision) enters the key as the position of that quadrant along
        // a Hilbert curve through the cell: the cells of every level are still contiguous ranges of the
        // sorted bodies, as the tree needs them, and consecutive cells are always neighbours in space
        // (in Z order a quarter of them are not), so the 64 consecutive bodies of a traversal wave
        // fill a tighter box.  Two tables of 16 two-bit entries, indexed by orientation and quadrant (nb_common.h).
        const uint32_t at = 2u * (4u * curve + (((uint32_t)below << 1) | (uint32_t)right));
        k = (k << 2) | (uint64_t)((HILBERT_DIGIT >> at) & 3u);
        curve = (HILBERT_NEXT >> at) & 3u;
    }
    const bool odd = cut[2] & 1u;
    const uint64_t key = levels > 0 ? k << (2 * (BH_KEY_LEVELS - levels)) : 0ull;
    if (cut[5] != 0u) {
        (odd ? keys1 : keys0)[i] = key | (uint64_t)(uint32_t)i;  // the index rides in the key's free low bits
        return;
    }
    (odd ? keys1 : keys0)[i] = key;
    (odd ? idx1 : idx0)[i] = (uint32_t)i;
}

// After new bodies were set: one bounding-box pass, waited for, so that the first step already
// knows how deep its keys go (and with that, which build to launch).
int bh_prime(hipStream_t s, BhBuffers &b, const BhStepArgs &a)
{
    if (!b.hint_host)
        return NB_OK;
    NB_TRY(bh_bbox(s, b, a));
    NB_HIP(hipStreamSynchronize(s));
    return NB_OK;
}

int bh_keys(hipStream_t s, BhBuffers &b, const BhStepArgs &a)
{
    hipLaunchKernelGGL(k_keys, dim3((a.n + 255) / 256), dim3(256), 0, s, a.pos, a.n, b.box, b.cut, b.keys[0], b.keys[1],
                       b.idx[0], b.idx[1], b.pc_range);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// ------------------------------------------------------------------ 6 traversal + integration
//
// The kernel is traverse_and_integrate<2> (nb_traverse.h, shared with the octree); this is what
// the quadtree pipeline brings to it.  A stack entry is {first child slot << 4 | mask of the child
// slots in use}; the four child slots of a cell are consecutive (dense pyramid: one per quadrant,
// level-by-level build: packed).
#ifndef NB_TR_ENTRIES
#define NB_TR_ENTRIES 16
#endif
#ifndef NB_TR_HIGH
#define NB_TR_HIGH 64
#endif
// Waves per block.  The waves of a block share nothing, and a block's slot on its CU is free again
// only when its slowest wave is done: one wave per block packs the CUs best (traversal at N = 5 M:
// 1.00 ms with four, 0.99 with two, 0.97 with one).
#ifndef NB_TR_WPB
#define NB_TR_WPB 1
#endif

struct ForceArgs {
    const double *box;       // this step's bounding box (the origin of the float coordinates is its middle)
    double inv_len;          // 1 / length unit
    const float4 *node_rec;  // {x, y, mass, (width / theta)^2; for a body: -(r_j + th)}
    const uint2 *node_aux;   // x = first child slot << 4 | mask of the slots in use, y = float bits of the prune threshold
    const double *node_com;  // {x, y, mass, width (for a body: -r_j)} in FP64, for the re-decisions
    const uint32_t *idx;
    const uint32_t *tlist;   // may be null: targets are all sorted bodies
    const uint32_t *tcount;  // device count of tlist entries (when tlist != null)
    const uint8_t *dyn;
    const double *radius;
    double2 *pos;
    double2 *vel;
    double2 *acc;
    // where the step writes: the same arrays, or (to.pos != null) slot k of the arrays in `to` for the
    // body at sorted position k, every other per-body array carried along
    double2 *pos_w, *vel_w, *acc_w;
    BodyArrays2 from, to;
    unsigned long long *counters;
    uint32_t *wave_cost;  // may be null: per group of 64 targets, the time its wave took (10 ns ticks)
    int n;
    ForceConsts fc;
    IntegrateArgs ia;
    double theta;
    double collision_th;
};

template <>
struct Traversal<2> {
    using Args = ForceArgs;
    using Aux = uint2;
    static constexpr int GROUP = 4, ENTRIES = NB_TR_ENTRIES, HIGH = NB_TR_HIGH, DEPTH = 32, WPB = NB_TR_WPB;
    static constexpr bool EXTRA = false;      // the second staged word holds everything a visit needs
    static constexpr int COM = 4, COM_W = 3;  // doubles per node_com record, index of the width
    // the target in the float coordinates of this step's records (ComArgs: relative to the middle
    // of the box); the traversal converts the FP64 position itself, no sorted record array exists
    static __device__ __forceinline__ void load_target(const Args &a, int, uint32_t body, Pt<2> &x, float &ri)
    {
        const double2 pd = a.pos[body];
        x.xy.x = (float)((pd.x - 0.5 * (a.box[0] + a.box[2])) * a.inv_len);
        x.xy.y = (float)((pd.y - 0.5 * (a.box[1] + a.box[3])) * a.inv_len);
        x.z = 0.f;
        ri = (float)(a.radius[body] * a.inv_len);
    }
    static __device__ __forceinline__ bool has(uint32_t entry, int sib) { return (entry >> sib) & 1u; }
    static __device__ __forceinline__ Aux no_aux() { return make_uint2(0u, 0u); }
    static __device__ __forceinline__ uint32_t wbits(const float4 &r, const Aux &) { return __float_as_uint(r.w); }
    static __device__ __forceinline__ float mass(const float4 &r) { return r.z; }
    static __device__ __forceinline__ Pt<2> point(const float4 &r)
    {
        Pt<2> p;
        p.xy = v2f{r.x, r.y};
        p.z = 0.f;
        return p;
    }
    static __device__ __forceinline__ uint32_t child(const Aux &x) { return x.x; }
    static __device__ __forceinline__ uint32_t prune(const Aux &x) { return x.y; }
    static __device__ __forceinline__ uint4 stage(uint64_t um, const Aux &x)
    {
        return make_uint4((uint32_t)um, (uint32_t)(um >> 32), x.x, x.y);
    }
    static __device__ __forceinline__ float stage_extra(const Aux &) { return 0.f; }
    static __device__ __forceinline__ float staged_width(const float4 &r, const uint4 &) { return r.w; }
    static __device__ __forceinline__ float staged_prune(const uint4 &q, const float *, int) { return __uint_as_float(q.w); }
    static __device__ __forceinline__ uint32_t staged_child(const uint4 &q) { return q.z; }
    static __device__ __forceinline__ const double *pos_ptr(const Args &a, uint32_t body) { return (const double *)(a.pos + body); }
    static __device__ __forceinline__ uint32_t dest(const Args &a, uint32_t body, uint32_t k) { return a.to.pos ? k : body; }
    static __device__ __forceinline__ void store_acc(const Args &a, uint32_t dst, const double (&acc)[2])
    {
        a.acc_w[dst] = make_double2(acc[0], acc[1]);
    }
    static __device__ __forceinline__ void load_acc(const Args &a, uint32_t body, double (&acc)[2])
    {
        const double2 t = a.acc[body];
        acc[0] = t.x;
        acc[1] = t.y;
    }
    static __device__ __forceinline__ void integrate(const Args &a, uint32_t body, uint32_t dst, const double (&acc)[2])
    {
        double2 v = a.vel[body];
        double2 p = a.pos[body];
        v.x += acc[0] * a.ia.dt_kick;
        v.y += acc[1] * a.ia.dt_kick;
        p.x += v.x * a.ia.dt;
        p.y += v.y * a.ia.dt;
        a.vel_w[dst] = v;
        a.pos_w[dst] = p;
    }
    // A step that re-orders the bodies: what it did not write itself goes to the new slot too.
    static __device__ __forceinline__ void carry(const Args &a, uint32_t body, uint32_t dst, bool dynamic, bool advanced)
    {
        if (!a.to.pos)
            return;  // uniform
        if (!dynamic)
            a.to.acc[dst] = a.acc[body];
        if (!advanced) {
            a.to.pos[dst] = a.pos[body];
            a.to.vel[dst] = a.vel[body];
        }
        a.to.mass_radius[dst] = a.from.mass_radius[body];
        a.to.mass[dst] = a.from.mass[body];
        a.to.radius[dst] = a.from.radius[body];
        a.to.dyn[dst] = a.from.dyn[body];
        a.to.order[dst] = a.from.order[body];
    }
};
constexpr int TR_WPB = Traversal<2>::WPB;

template <int MODE, bool STATS>
__global__ __launch_bounds__(64 * TR_WPB) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_bh_force(ForceArgs a)
{
    traverse_and_integrate<2, MODE, STATS>(a);
}

int bh_force(hipStream_t s, BhBuffers &b, const BhStepArgs &a)
{
    ForceArgs f;
    f.box = b.box;
    f.inv_len = 1.0 / a.units.len_unit;
    f.node_rec = b.node_rec;
    f.node_aux = b.node_child;
    f.node_com = b.node_com;
    f.radius = a.radius;
    f.theta = a.theta;
    f.collision_th = a.collision_th;
    f.idx = b.idx[b.sorted];
    f.tlist = nullptr;
    f.tcount = nullptr;
    int ntarget = a.n;
    if (a.t0 != 0 || a.t1 != a.n) {
        // this rank integrates only bodies [t0, t1): walk them in the sorted order
        NB_TRY(bh_target_list(s, b.idx[b.sorted], a.n, a.t0, a.t1, b.tlist, b.tcount, b.tl_calls, &f.tcount));
        f.tlist = b.tlist;
        ntarget = a.t1 - a.t0;
    }
    f.dyn = a.dyn;
    f.pos = a.pos_out;
    f.vel = a.vel;
    f.acc = a.acc;
    f.from = a.from;
    f.to = a.to;
    if (f.to.pos && (f.tlist || !a.ia.do_integrate))
        return fail(NB_ERR_ARG, "bh_force: only a step of all the bodies can re-order them");
    f.pos_w = f.to.pos ? f.to.pos : f.pos;
    f.vel_w = f.to.pos ? f.to.vel : f.vel;
    f.acc_w = f.to.pos ? f.to.acc : f.acc;
    f.counters = b.counters;
    f.wave_cost = b.wave_cost;
    f.n = a.n;
    f.fc = a.fc;
    f.ia = a.ia;
    const int waves = f.tlist ? target_list_groups(a.n, ntarget) : (ntarget + 63) / 64;
    dim3 grid((waves + TR_WPB - 1) / TR_WPB);
    if (ntarget <= 0)
        return NB_OK;
    const int mode = a.fc.exact_denom ? FM_EXACT : (a.fc.eps2_negligible ? FM_NOEPS : FM_EPS);
    if (a.collect_stats) {
        if (mode == FM_EXACT)
            hipLaunchKernelGGL((k_bh_force<FM_EXACT, true>), grid, dim3(64 * TR_WPB), 0, s, f);
        else if (mode == FM_EPS)
            hipLaunchKernelGGL((k_bh_force<FM_EPS, true>), grid, dim3(64 * TR_WPB), 0, s, f);
        else
            hipLaunchKernelGGL((k_bh_force<FM_NOEPS, true>), grid, dim3(64 * TR_WPB), 0, s, f);
    } else {
        if (mode == FM_EXACT)
            hipLaunchKernelGGL((k_bh_force<FM_EXACT, false>), grid, dim3(64 * TR_WPB), 0, s, f);
        else if (mode == FM_EPS)
            hipLaunchKernelGGL((k_bh_force<FM_EPS, false>), grid, dim3(64 * TR_WPB), 0, s, f);
        else
            hipLaunchKernelGGL((k_bh_force<FM_NOEPS, false>), grid, dim3(64 * TR_WPB), 0, s, f);
    }
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// ------------------------------------------------------------------ buffers

// Every device array of the pipeline for n bodies, taken from `alloc` (hipMalloc for a handle's
// own buffers, a bump pointer for buffers carved out of a caller-provided arena).
template <typename Alloc>
static int bh_layout(BhBuffers &b, int n, bool with_pyramid, Alloc &&alloc)
{
    b.n = n;
    b.node_cap = 2 * (size_t)n + 8;
    if (with_pyramid && b.node_cap < PYR_CELLS)
        b.node_cap = PYR_CELLS;  // the dense pyramid uses one node slot per cell
    b.pyr_max = (with_pyramid && b.node_cap >= PYR_DEEP_CELLS) ? PYR_DEEP_LEVEL : PYR_MAX_LEVEL;
    b.sort_blocks = sort_blocks(n);
    NB_TRY(alloc(b.box, 4));
    NB_TRY(alloc(b.cut, 8));
    NB_TRY(alloc(b.leaf_sum, 4 * (size_t)n));
    NB_TRY(alloc(b.chunk_info, (size_t)n / 64 + 2));
    NB_TRY(alloc(b.bbox_part, 4 * BBOX_BLOCKS));
    for (int k = 0; k < 2; ++k) {
        NB_TRY(alloc(b.keys[k], (size_t)n));
        NB_TRY(alloc(b.idx[k], (size_t)n));
    }
    NB_TRY(alloc(b.hist, (size_t)SORT_BINS * b.sort_blocks));
    NB_TRY(alloc(b.hist_totals, sort_totals_words()));
    NB_TRY(alloc(b.node_meta, b.node_cap));
    NB_TRY(alloc(b.split, (size_t)n + 256));  // one Split<2> (16 bytes) per frontier slot
    NB_TRY(alloc(b.level_start, BH_MAX_DEPTH + 2));
    NB_TRY(alloc(b.bsum, BH_SCAN_BLOCKS));
    NB_TRY(alloc(b.node_com, 4 * b.node_cap));
    NB_TRY(alloc(b.node_rec, b.node_cap));
    NB_TRY(alloc(b.node_child, b.node_cap));
    NB_TRY(alloc(b.tlist, target_list_capacity(n)));
    NB_TRY(alloc(b.tcount, 4));
    NB_TRY(alloc(b.counters, 16));
    NB_TRY(alloc(b.wave_cost, (size_t)target_list_groups(n, n) + 2));
    if (with_pyramid) {
        // first / one-past-last sorted body of every cell of the deepest pyramid level
        NB_TRY(alloc(b.pc_range, (size_t)1 << (2 * b.pyr_max)));
    }
    return NB_OK;
}

int bh_alloc(BhBuffers &b, int n)
{
    b.owned = true;
    NB_TRY(bh_layout(b, n, true, [](auto *&p, size_t count) { return dev_alloc(p, count); }));
    NB_HIP(hipMemset(b.tcount, 0, 4 * sizeof(uint32_t)));  // bh_target_list's words
    void *h = nullptr;
    if (hipHostMalloc(&h, 64 + sizeof(double) * BOX_SLOT * BOX_RING, hipHostMallocDefault) != hipSuccess)
        return fail(NB_ERR_NOMEM, "hipHostMalloc failed");
    b.box_ring = (double *)((char *)h + 64);
    b.levels_host = (uint32_t *)h;
    *b.levels_host = 0u;
    b.hint_host = b.levels_host + 1;
    *b.hint_host = 0xffffffffu;  // key levels not known yet: bh_prime finds out
    b.hint_box = (double *)(b.levels_host + 4);
    b.err_host = b.levels_host + 8;
    *b.err_host = 0u;
    return NB_OK;
}

// The same arrays laid out one after the other (each on a 256-byte boundary, with the slack
// dev_alloc gives) in `arena`; nothing is allocated.  base == nullptr only measures: *used gets the
// bytes n bodies need.  Such buffers have no host-visible level word: the per-level build kernels
// are launched for every level each step.
int bh_carve(BhBuffers &b, int n, void *base, size_t bytes, size_t *used)
{
    size_t off = 0;
    bool fits = true;
    b = BhBuffers();
    b.owned = false;
    NB_TRY(bh_layout(b, n, false, [&](auto *&p, size_t count) {
        using T = std::remove_reference_t<decltype(*p)>;
        const size_t need = (count * sizeof(T) + 256 + 255) & ~(size_t)255;
        if (base && off + need > bytes)
            fits = false;
        p = (base && fits) ? (T *)((char *)base + off) : nullptr;
        off += need;
        return (int)NB_OK;
    }));
    if (used)
        *used = off;
    if (base && !fits) {
        b = BhBuffers();
        return fail(NB_ERR_OVERFLOW, "bh_carve: the arena is too small for " + std::to_string(n) + " bodies");
    }
    return NB_OK;
}

void bh_free(BhBuffers &b)
{
    if (b.owned) {
        (void)hipFree(b.box);
        (void)hipFree(b.cut);
        (void)hipFree(b.leaf_sum);
        (void)hipFree(b.chunk_info);
        (void)hipFree(b.bbox_part);
        for (int k = 0; k < 2; ++k) {
            (void)hipFree(b.keys[k]);
            (void)hipFree(b.idx[k]);
        }
        (void)hipFree(b.hist);
        (void)hipFree(b.hist_totals);
        (void)hipFree(b.node_meta);
        (void)hipFree(b.split);
        (void)hipFree(b.level_start);
        (void)hipFree(b.bsum);
        (void)hipFree(b.node_com);
        (void)hipFree(b.node_rec);
        (void)hipFree(b.node_child);
        (void)hipFree(b.tlist);
        (void)hipFree(b.tcount);
        (void)hipFree(b.counters);
        (void)hipFree(b.wave_cost);
        (void)hipFree(b.pc_range);
        (void)hipFree(b.agg);
        (void)hipFree(b.red4);
        (void)hipFree(b.err);
        if (b.levels_host)
            (void)hipHostFree(b.levels_host);
    }
    b = BhBuffers();
}

}  // namespace nb
