// nb_bh3.hip -- the Barnes-Hut pipeline of nb_bh.hip for three dimensions (see nb_dim3.h for the
// rules it follows).  Same stages on one stream, no host synchronisation in between:
//   1 bbox   min/max of the three coordinates, rounded to float and padded as
//            BarnesHut::computeBoundingBox pads x and y (barnesHut.cpp:23-35); the z range is
//            widened to the x-width when it is smaller
//   2 keys   octant paths by the FP64 midpoint descent of QuadTree::getQuadrant
//            (quadtree.cpp:41-72) on three axes: digit = 4 (below mid z) + 2 (below mid y) +
//            (right of mid x), 3 bits per level, at most 20 levels
//   3 sort   the radix sort of nb_bh.hip
//   4 build  the level-by-level build of nb_bh.hip with 3-bit digits (up to 8 children)
//   5 com    bottom-up centres of mass, children summed in digit order
//   6 force  one wave per 64 consecutive bodies of the sorted order with a shared stack in LDS, every lane
//            applying its own opening test, fused integration
// As in 2-D, cells that no body can ever open (width / theta + diagonal < collision_th) are not
// built below: the keys stop at that level and its cells become leaves holding several bodies.
#include <type_traits>

#include "nb_bh.h"
#include "nb_dev.h"
#include "nb_dim3.h"
#include "nb_traverse.h"

namespace nb {

// ------------------------------------------------------------------ 1 bounding box

constexpr int BBOX3_BLOCKS = 1024;

struct MinMax3 {
    double lo[3], hi[3];
};

__device__ __forceinline__ void mm_init(MinMax3 &m)
{
    for (int d = 0; d < 3; ++d) {
        m.lo[d] = __builtin_inf();
        m.hi[d] = -__builtin_inf();
    }
}

// Folds the block's values; the result is valid in thread 0.
__device__ __forceinline__ void mm_block_reduce(MinMax3 &m)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            m.lo[d] = fmin(m.lo[d], __shfl_xor(m.lo[d], o));
            m.hi[d] = fmax(m.hi[d], __shfl_xor(m.hi[d], o));
        }
    __shared__ double sh[4][6];
    const int w = threadIdx.x >> 6;
    if (lane_id() == 0)
        for (int d = 0; d < 3; ++d) {
            sh[w][d] = m.lo[d];
            sh[w][3 + d] = m.hi[d];
        }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int k = 1; k < 4; ++k)
            for (int d = 0; d < 3; ++d) {
                m.lo[d] = fmin(m.lo[d], sh[k][d]);
                m.hi[d] = fmax(m.hi[d], sh[k][3 + d]);
            }
}

__global__ __launch_bounds__(256) void k_bbox3_partial(const double *pos, int n, double *part)
{
    MinMax3 m;
    mm_init(m);
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
        for (int d = 0; d < 3; ++d) {
            const double v = pos[3 * (size_t)i + d];
            m.lo[d] = fmin(m.lo[d], v);
            m.hi[d] = fmax(m.hi[d], v);
        }
    mm_block_reduce(m);
    if (threadIdx.x == 0)
        for (int d = 0; d < 3; ++d) {
            part[6 * blockIdx.x + d] = m.lo[d];
            part[6 * blockIdx.x + 3 + d] = m.hi[d];
        }
}

__global__ __launch_bounds__(256) void k_bbox3_final(const double *part, int nblk, double margin, double theta, double cth,
                                                     double *box, uint32_t *cut, int n)
{
    MinMax3 m;
    mm_init(m);
    for (int i = threadIdx.x; i < nblk; i += 256)
        for (int d = 0; d < 3; ++d) {
            m.lo[d] = fmin(m.lo[d], part[6 * i + d]);
            m.hi[d] = fmax(m.hi[d], part[6 * i + 3 + d]);
        }
    mm_block_reduce(m);
    if (threadIdx.x == 0) {
        // every candidate passes through fminf/fmaxf in the reference (barnesHut.cpp:28-31);
        // rounding is monotone, so rounding the extreme gives the same box
        box[0] = (double)(float)(m.lo[0] - margin);  // topLeft.x
        box[1] = (double)(float)(m.hi[1] + margin);  // topLeft.y
        box[2] = (double)(float)(m.hi[0] + margin);  // botRight.x
        box[3] = (double)(float)(m.lo[1] - margin);  // botRight.y
        double zup = (double)(float)(m.hi[2] + margin), zlo = (double)(float)(m.lo[2] - margin);
        // The root is at least as deep in z as it is wide in x (centred on the z range): cells
        // then are no flatter than wide, so a thin disc is not cut into slabs that the opening
        // test, which looks at the x-width only, would all have to accept one by one.
        if (zup - zlo < box[2] - box[0]) {
            const double mid = (zup + zlo) / 2.0, half = (box[2] - box[0]) / 2.0;
            zup = mid + half;
            zlo = mid - half;
        }
        box[4] = zup;  // upper z
        box[5] = zlo;  // lower z
        // First level whose cells no body ever opens: a cell is opened by a body that neither
        // accepts it (dist <= width / theta) nor sees it wholly inside its cutoff distance
        // (dist >= collision_th - diagonal); when width / theta < collision_th - diagonal no
        // distance does both.  The margins cover the float rounding of the two thresholds.
        uint32_t first_closed = 32u;
        const double w0 = box[2] - box[0], h0 = box[1] - box[3], d0 = box[4] - box[5];
        for (int l = BH3_KEY_LEVELS; l >= 0; --l) {
            const double w = ldexp(w0, -l), h = ldexp(h0, -l), dz = ldexp(d0, -l);
            const double reach = cth * (1.0 - 1.0e-4) - sqrt(w * w + h * h + dz * dz);
            if (reach > 0.0 && reach * (1.0 - 1.0e-4) > w / theta)
                first_closed = (uint32_t)l;
            else
                break;  // cells only grow from here up
        }
        cut[0] = first_closed;
        cut[1] = first_closed < (uint32_t)BH3_KEY_LEVELS ? first_closed : (uint32_t)BH3_KEY_LEVELS;
        cut[2] = (uint32_t)sort_first_pass(3 * (int)cut[1]);
        // keys of at most 10 digits leave the low 32 bits of the key word free for the body's index
        // (see k_sort_scatter in nb_bh.hip)
        cut[5] = (cut[1] <= 10u && n <= (1 << 30)) ? 1u : 0u;
    }
}

int bh3_bbox(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a)
{
    int blocks = (a.n + 255) / 256;
    if (blocks > BBOX3_BLOCKS)
        blocks = BBOX3_BLOCKS;
    hipLaunchKernelGGL(k_bbox3_partial, dim3(blocks), dim3(256), 0, s, a.pos, a.n, b.bbox_part);
    hipLaunchKernelGGL(k_bbox3_final, dim3(1), dim3(256), 0, s, b.bbox_part, blocks, a.bbox_margin, a.theta, a.collision_th,
                       b.box, b.cut, a.n);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// ------------------------------------------------------------------ 2 keys

// The digit of level l occupies bits [61-3l, 59-3l] of the key word; only the first cut[1]
// digits are computed.  The pairs go to the buffer the first sort pass with work reads.
__global__ __launch_bounds__(256) void k_keys3(const double *pos, int n, const double *box, const uint32_t *cut,
                                               uint64_t *keys0, uint64_t *keys1, uint32_t *idx0, uint32_t *idx1)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    const int levels = (int)cut[1];
    double tlx = box[0], tly = box[1], brx = box[2], bry = box[3], zup = box[4], zlo = box[5];
    const double px = pos[3 * (size_t)i], py = pos[3 * (size_t)i + 1], pz = pos[3 * (size_t)i + 2];
    // the octant decisions of every level, one bit string per axis
    uint32_t c[3] = {0u, 0u, 0u};
#pragma unroll 1
    for (int l = 0; l < levels; ++l) {
        const double mx = (tlx + brx) / 2.0, my = (tly + bry) / 2.0, mz = (zup + zlo) / 2.0;
        const bool right = !(mx >= px);  // getQuadrant: mid >= x is the left half
        const bool ylow = !(my <= py);   // mid <= y is the top half
        const bool zlow = !(mz <= pz);   // z like y
        tlx = right ? mx : tlx;
        brx = right ? brx : mx;
        tly = ylow ? my : tly;
        bry = ylow ? bry : my;
        zup = zlow ? mz : zup;
        zlo = zlow ? zlo : mz;
        c[0] = (c[0] << 1) | (uint32_t)right;
        c[1] = (c[1] << 1) | (uint32_t)ylow;
        c[2] = (c[2] << 1) | (uint32_t)zlow;
    }
    // The octants enter the key as their positions along a Hilbert curve (as the quadrants do in
    // k_keys: cells stay contiguous ranges of the sorted bodies, consecutive cells are neighbours
    // in space, a traversal wave's bodies fill a tighter box).  Cut at any level the keys are the
    // curve through that level's cells (hilbert3_index, nb_common.h).
    const uint64_t k = hilbert3_index(c[0], c[1], c[2], levels);
    const bool odd = cut[2] & 1u;
    const uint64_t key = levels > 0 ? k << (2 + 3 * (BH3_KEY_LEVELS - levels)) : 0ull;
    if (cut[5] != 0u) {
        (odd ? keys1 : keys0)[i] = key | (uint64_t)(uint32_t)i;  // the index rides in the key's free low bits
        return;
    }
    (odd ? keys1 : keys0)[i] = key;
    (odd ? idx1 : idx0)[i] = (uint32_t)i;
}

int bh3_keys(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a)
{
    hipLaunchKernelGGL(k_keys3, dim3((a.n + 255) / 256), dim3(256), 0, s, a.pos, a.n, b.box, b.cut, b.keys[0], b.keys[1],
                       b.idx[0], b.idx[1]);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// ------------------------------------------------------------------ 3 sort, 4 build

int bh3_sort(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a)
{
    return sort_pairs(s, b.keys, b.idx, a.n, b.cut, b.hist, b.hist_totals);
}

int bh3_build(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a)
{
    b.launched = launched_levels(b.levels_host);
    return build_tree(s, 3, b.keys[0], a.n, b.node_meta, b.split, b.level_start, b.bsum, b.counters, b.launched,
                      b.levels_host);
}

// ------------------------------------------------------------------ 5 centre of mass

struct Com3Args {
    const uint4 *node_meta;
    const uint32_t *level_start;
    const uint32_t *idx;
    const double *pos;
    const double *mass;
    const double *radius;
    const float4 *posm_s;
    const double *box;
    double *node_com;
    float4 *node_rec;
    uint4 *node_aux;
    const uint32_t *cut;
    const double *leaf_sum;
    // float coordinates are relative to the middle of this step's bounding box (see nb_bh.hip)
    double inv_len, inv_mass, theta, cth;
};

// Width, opening threshold and prune threshold of a cell of level lvl.
__device__ __forceinline__ void cell_record3(const Com3Args &a, int lvl, double &width, float &open2, float &prune)
{
    width = ldexp(a.box[2] - a.box[0], -lvl);  // the cell's x-width, as QuadTree::getWidth (quadtree.cpp:234-237)
    const double wt = width * a.inv_len / a.theta;  // width / dist < theta  <=>  dist > width / theta
    open2 = (float)(wt * wt);
    // the centre of mass lies inside the cell, so every point of the cell is within
    // dist(com) + diagonal of a target: below collision_th nothing in the cell can contribute
    const double h = ldexp(a.box[1] - a.box[3], -lvl), dz = ldexp(a.box[4] - a.box[5], -lvl);
    const double reach = a.cth * (1.0 - 1.0e-4) - sqrt(width * width + h * h + dz * dz);
    prune = reach > 0.0 ? (float)((reach * a.inv_len) * (reach * a.inv_len)) : -1.f;
}

__device__ __forceinline__ void store_node3(const Com3Args &a, uint32_t u, const uint4 meta, const double (&c)[3], double mtot,
                                            double width, float open2, float prune)
{
    double2 *o = (double2 *)a.node_com + 3 * (size_t)u;
    o[0] = make_double2(c[0], c[1]);
    o[1] = make_double2(c[2], mtot);
    o[2] = make_double2(width, 0.0);
    const double ox = 0.5 * (a.box[0] + a.box[2]), oy = 0.5 * (a.box[1] + a.box[3]), oz = 0.5 * (a.box[4] + a.box[5]);
    a.node_rec[u] = make_float4((float)((c[0] - ox) * a.inv_len), (float)((c[1] - oy) * a.inv_len),
                                (float)((c[2] - oz) * a.inv_len), (float)(mtot * a.inv_mass));
    a.node_aux[u] = make_uint4(__float_as_uint(open2), (meta.z << 4) | (meta.w & 0xffu), __float_as_uint(prune), 0u);
}

// Sums over every run of bodies with one key (= every leaf with several bodies), left at the
// run's first sorted position as {sum m x, sum m y, sum m z, sum m, max radius}; also writes
// every body's float record in sorted order.  One wave per 64 consecutive sorted bodies, as
// k_leaf_sums in nb_bh.hip.
__global__ __launch_bounds__(256) void k_leaf_sums3(const uint64_t *keys, const uint32_t *idx, int n, const double *pos,
                                                    const double *mass, const double *radius, const double *box,
                                                    double inv_len, double inv_mass, float4 *posm_s, float *rad_s,
                                                    double *leaf_sum, uint32_t *chunk_info)
{
    const double ox = 0.5 * (box[0] + box[2]), oy = 0.5 * (box[1] + box[3]), oz = 0.5 * (box[4] + box[5]);
    const int lane = lane_id();
    const int chunk = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (chunk >= n)
        return;
    const int k = chunk + lane;
    const bool valid = k < n;
    const uint64_t key = valid ? keys[k] : ~0ull;
    const uint64_t prev = (valid && k > 0) ? keys[k - 1] : ~0ull;  // ~0 is no key (its low bits are set)
    double sm = 0.0, sx = 0.0, sy = 0.0, sz = 0.0, rm = 0.0;
    if (valid) {
        const uint32_t o = idx[k];
        const double px = pos[3 * (size_t)o], py = pos[3 * (size_t)o + 1], pz = pos[3 * (size_t)o + 2];
        sm = mass[o];
        sx = px * sm;
        sy = py * sm;
        sz = pz * sm;
        rm = radius[o];
        posm_s[k] = make_float4((float)((px - ox) * inv_len), (float)((py - oy) * inv_len), (float)((pz - oz) * inv_len),
                                (float)(sm * inv_mass));
        rad_s[k] = (float)(rm * inv_len);
    }
    const uint64_t heads = __builtin_amdgcn_ballot_w64(valid && (k == 0 || key != prev));
    // Segmented scan.  The bodies before the chunk's first head continue a run from an earlier
    // chunk; for the scan they are a run of their own that starts at lane 0.
    const int first_head = heads ? (int)__builtin_ctzll(heads) : 64;
    const uint64_t upto = (heads | 1ull) & ((2ull << lane) - 1ull);
    const int start = 63 - __builtin_clzll(upto);  // first lane of this lane's run
    const int dist = lane - start;
    {
        double sums[4] = {sm, sx, sy, sz};
        wave_run_scan<4>(sums, rm, lane, start);
        sm = sums[0];
        sx = sums[1];
        sy = sums[2];
        sz = sums[3];
    }
    // the part of the earlier run goes to the chunk's first slot (no run starts there) for
    // leaf_finish to add to that run
    if (first_head > 0 && lane == first_head - 1) {
        double *o = leaf_sum + 6 * (size_t)chunk;
        o[0] = sx;
        o[1] = sy;
        o[2] = sz;
        o[3] = sm;
        o[4] = rm;
    }
    if (heads == 0ull) {
        if (lane == 0)
            chunk_info[chunk >> 6] = 64u | (LEAF_NO_OWNER << 8);
        return;  // the whole chunk lies inside a run that started earlier
    }
    const bool last_in_chunk = lane == 63 || !valid || ((heads >> (lane + 1)) & 1ull) || k + 1 >= n;
    const bool ends = valid && lane >= first_head && last_in_chunk;
    const int tail = chunk + 64;
    const bool goes_on = tail < n && keys[tail] == lane_value(key, 63);  // uniform
    if (ends && !(lane == 63 && goes_on) && dist > 0) {
        double *o = leaf_sum + 6 * (size_t)(chunk + start);
        o[0] = sx;
        o[1] = sy;
        o[2] = sz;
        o[3] = sm;
        o[4] = rm;
    }
    // the run of the chunk's last body, when it started here and goes on: its part inside the
    // chunk stays at its start, leaf_finish adds the parts of the chunks it crosses
    const int start63 = 63 - __builtin_clzll(heads);
    const bool owner = goes_on;
    if (owner && lane == 63) {
        double *o = leaf_sum + 6 * (size_t)(chunk + start63);
        o[0] = sx;
        o[1] = sy;
        o[2] = sz;
        o[3] = sm;
        o[4] = rm;
    }
    if (lane == 0)
        chunk_info[chunk >> 6] = (uint32_t)first_head | ((owner ? (uint32_t)start63 : LEAF_NO_OWNER) << 8);
}

// Centres of mass, masses and traversal records of the nodes of tree level d (children first).
__device__ __forceinline__ void com3_level(const Com3Args &a, int d)
{
    const uint32_t lbeg = a.level_start[d], lend = a.level_start[d + 1];
    for (uint32_t u = lbeg + blockIdx.x * 256 + threadIdx.x; u < lend; u += gridDim.x * 256) {
        const uint4 meta = a.node_meta[u];
        const uint32_t nchild = meta.w & 0xffu;
        double c[3], mtot, width;
        float open2, prune = -1.f;
        if (nchild == 0) {
            const uint32_t first = meta.x, count = meta.y;
            if (count == 1) {
                const uint32_t o = a.idx[first];
                for (int k = 0; k < 3; ++k)
                    c[k] = a.pos[3 * (size_t)o + k];
                mtot = a.mass[o];
                open2 = -(float)((a.radius[o] + a.cth) * a.inv_len);
                width = -a.radius[o];
            } else {
                const double *ls = a.leaf_sum + 6 * (size_t)first;
                mtot = ls[3];
                for (int k = 0; k < 3; ++k)
                    c[k] = ls[k] / mtot;
                const uint32_t closed_level = a.cut[0];
                if (closed_level <= (uint32_t)BH3_KEY_LEVELS) {
                    // the bodies of a cell no body ever opens: a cell record, no children
                    cell_record3(a, (int)closed_level, width, open2, prune);
                } else {
                    // bodies equal in all key levels: they act as one body
                    open2 = -(float)((ls[4] + a.cth) * a.inv_len);
                    width = -ls[4];
                }
            }
        } else {
            // updateCenterMass: total = sum of child masses, com = sum(child com * child mass) / total,
            // children in ascending digit order
            const uint32_t cf = meta.z;
            double s[3] = {0.0, 0.0, 0.0};
            mtot = 0.0;
            const double2 *cc = (const double2 *)a.node_com;
            for (uint32_t ch = cf; ch < cf + nchild; ++ch) {
                const double2 xy = cc[3 * (size_t)ch], zm = cc[3 * (size_t)ch + 1];
                mtot += zm.y;
                s[0] += xy.x * zm.y;
                s[1] += xy.y * zm.y;
                s[2] += zm.x * zm.y;
            }
            for (int k = 0; k < 3; ++k)
                c[k] = s[k] / mtot;
            cell_record3(a, (int)(meta.w >> 8), width, open2, prune);
        }
        store_node3(a, u, meta, c, mtot, width, open2, prune);
    }
}

__global__ __launch_bounds__(256) void k_com3(Com3Args a, int d, int d_hi)
{
    if (d_hi > d + 1) {
        // several levels by one block: those past the ones the host expects to exist (normally
        // empty), and the few hundred cells of the top levels
        for (int dd = d_hi - 1; dd >= d; --dd) {
            if (a.level_start[dd + 1] > a.level_start[dd]) {  // uniform
                com3_level(a, dd);
                __threadfence_block();
            }
            __syncthreads();
        }
        return;
    }
    com3_level(a, d);
}

int bh3_com(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a)
{
    Com3Args c;
    c.node_meta = b.node_meta;
    c.level_start = b.level_start;
    c.idx = b.idx[0];
    c.pos = a.pos;
    c.mass = a.mass;
    c.radius = a.radius;
    c.posm_s = b.posm_s;
    c.box = b.box;
    c.node_com = b.node_com;
    c.node_rec = b.node_rec;
    c.node_aux = b.node_aux;
    c.cut = b.cut;
    c.leaf_sum = b.leaf_sum;
    c.inv_len = 1.0 / a.units.len_unit;
    c.inv_mass = 1.0 / a.units.mass_unit;
    c.theta = a.theta;
    c.cth = a.collision_th;
    hipLaunchKernelGGL(k_leaf_sums3, dim3((a.n + 255) / 256), dim3(256), 0, s, b.keys[0], b.idx[0], a.n, a.pos, a.mass,
                       a.radius, b.box, c.inv_len, c.inv_mass, b.posm_s, b.rad_s, b.leaf_sum, b.chunk_info);
    NB_TRY(leaf_finish(s, b.chunk_info, a.n, b.leaf_sum, 3));
    if (b.launched < BH_MAX_DEPTH - 1)
        hipLaunchKernelGGL(k_com3, dim3(1), dim3(256), 0, s, c, b.launched, BH_MAX_DEPTH - 1);
    // the top levels hold a few hundred nodes at most (8^d): one block does them in one launch
    constexpr int COM3_TOP_LEVELS = 4;  // levels 0..3: at most 585 cells
    const int top = b.launched < COM3_TOP_LEVELS ? b.launched : COM3_TOP_LEVELS;
    for (int d = b.launched - 1; d >= top; --d)
        hipLaunchKernelGGL(k_com3, dim3(level_blocks(d, 3, 256)), dim3(256), 0, s, c, d, d + 1);
    if (top == 1)
        hipLaunchKernelGGL(k_com3, dim3(1), dim3(256), 0, s, c, 0, 1);
    else if (top > 1)
        hipLaunchKernelGGL(k_com3, dim3(1), dim3(256), 0, s, c, 0, top);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// ------------------------------------------------------------------ 6 traversal + integration
//
// The kernel is traverse_and_integrate<3> (nb_traverse.h, shared with the quadtree); this is what
// the octree pipeline brings to it.  A stack entry is {first child << 4 | child count}: the (up to
// 8) children of a cell are packed, and a round gives each popped entry a group of 8 lanes.  With one
// stack entry per node the 440-entry stack cost occupancy and its rounds shrank to 9 nodes (11.0 ms
// in the traversal at N = 5 M); entries per opened node with a prefix-sum pop: 8.2; lane groups: 7.45.
#ifndef NB_TR3_HIGH
#define NB_TR3_HIGH 64
#endif
#ifndef NB_TR3_ENTRIES
#define NB_TR3_ENTRIES 6
#endif
#ifndef NB_TR3_WPB
#define NB_TR3_WPB 1  // waves per block: they share nothing, one per block packs the CUs best (7.27 -> 6.97 ms)
#endif

struct Force3Args {
    const float4 *posm_s;    // {x, y, z, mass} of the sorted bodies, float coordinates of this step
    const float *rad_s;      // their radii
    const float4 *node_rec;  // {x, y, z, mass}
    const uint4 *node_aux;   // x = float bits of (width / theta)^2 (a body: -(r_j + th)), y = first child << 4 | count, z = prune threshold
    const double *node_com;  // {x, y, z, mass, width (a body: -r_j), -} in FP64, for the re-decisions
    const uint32_t *idx;
    const uint32_t *tlist;   // may be null: targets are all sorted bodies
    const uint32_t *tcount;  // device count of tlist entries (when tlist != null)
    const uint8_t *dyn;
    const double *radius;
    double *pos, *vel, *acc;  // packed triples
    unsigned long long *counters;
    uint32_t *wave_cost;
    int n;
    ForceConsts fc;
    IntegrateArgs ia;
    double theta, collision_th;
};

template <>
struct Traversal<3> {
    using Args = Force3Args;
    using Aux = uint4;
    static constexpr int GROUP = 8, ENTRIES = NB_TR3_ENTRIES, HIGH = NB_TR3_HIGH, DEPTH = 21, WPB = NB_TR3_WPB;
    static constexpr bool EXTRA = true;       // the prune threshold is staged beside the two words
    static constexpr int COM = 6, COM_W = 4;  // doubles per node_com record, index of the width
    static __device__ __forceinline__ void load_target(const Args &a, int k, uint32_t, Pt<3> &x, float &ri)
    {
        x = point(a.posm_s[k]);
        ri = a.rad_s[k];
    }
    static __device__ __forceinline__ bool has(uint32_t entry, int sib) { return (uint32_t)sib < (entry & 15u); }
    static __device__ __forceinline__ Aux no_aux() { return make_uint4(0u, 0u, 0u, 0u); }
    static __device__ __forceinline__ uint32_t wbits(const float4 &, const Aux &x) { return x.x; }
    static __device__ __forceinline__ float mass(const float4 &r) { return r.w; }
    static __device__ __forceinline__ Pt<3> point(const float4 &r)
    {
        Pt<3> p;
        p.xy = v2f{r.x, r.y};
        p.z = r.z;
        return p;
    }
    static __device__ __forceinline__ uint32_t child(const Aux &x) { return x.y; }
    static __device__ __forceinline__ uint32_t prune(const Aux &x) { return x.z; }
    static __device__ __forceinline__ uint4 stage(uint64_t um, const Aux &x)
    {
        return make_uint4((uint32_t)um, (uint32_t)(um >> 32), x.x, x.y);
    }
    static __device__ __forceinline__ float stage_extra(const Aux &x) { return __uint_as_float(x.z); }
    static __device__ __forceinline__ float staged_width(const float4 &, const uint4 &q) { return __uint_as_float(q.z); }
    static __device__ __forceinline__ float staged_prune(const uint4 &, const float *extra, int j) { return extra[j]; }
    static __device__ __forceinline__ uint32_t staged_child(const uint4 &q) { return q.w; }
    static __device__ __forceinline__ const double *pos_ptr(const Args &a, uint32_t body) { return a.pos + 3 * (size_t)body; }
    static __device__ __forceinline__ void store_acc(const Args &a, uint32_t body, const double (&acc)[3])
    {
        for (int d = 0; d < 3; ++d)
            a.acc[3 * (size_t)body + d] = acc[d];
    }
    static __device__ __forceinline__ void load_acc(const Args &a, uint32_t body, double (&acc)[3])
    {
        for (int d = 0; d < 3; ++d)
            acc[d] = a.acc[3 * (size_t)body + d];
    }
    // (the octree step leaves every body in its slot)
    static __device__ __forceinline__ uint32_t dest(const Args &, uint32_t body, uint32_t) { return body; }
    static __device__ __forceinline__ void carry(const Args &, uint32_t, uint32_t, bool, bool) {}
    static __device__ __forceinline__ void integrate(const Args &a, uint32_t body, uint32_t, const double (&acc)[3])
    {
        for (int d = 0; d < 3; ++d) {
            const double v = a.vel[3 * (size_t)body + d] + acc[d] * a.ia.dt_kick;
            a.vel[3 * (size_t)body + d] = v;
            a.pos[3 * (size_t)body + d] = a.pos[3 * (size_t)body + d] + v * a.ia.dt;
        }
    }
};
constexpr int TR3_WPB = Traversal<3>::WPB;

template <int MODE, bool STATS>
__global__ __launch_bounds__(64 * TR3_WPB) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_bh3_force(Force3Args a)
{
    traverse_and_integrate<3, MODE, STATS>(a);
}

int bh3_force(hipStream_t s, Bh3Buffers &b, const Bh3StepArgs &a)
{
    Force3Args f;
    f.posm_s = b.posm_s;
    f.rad_s = b.rad_s;
    f.node_rec = b.node_rec;
    f.node_aux = b.node_aux;
    f.node_com = b.node_com;
    f.idx = b.idx[0];
    f.tlist = nullptr;
    f.tcount = nullptr;
    int ntarget = a.n;
    if (a.t0 != 0 || a.t1 != a.n) {
        // this rank integrates only bodies [t0, t1): walk them in the sorted order
        NB_TRY(bh_target_list(s, b.idx[0], a.n, a.t0, a.t1, b.tlist, b.tcount, b.tl_calls, &f.tcount));
        f.tlist = b.tlist;
        ntarget = a.t1 - a.t0;
    }
    if (ntarget <= 0)
        return NB_OK;
    f.dyn = a.dyn;
    f.radius = a.radius;
    f.pos = a.pos;
    f.vel = a.vel;
    f.acc = a.acc;
    f.counters = b.counters;
    f.wave_cost = b.wave_cost;
    f.n = a.n;
    f.fc = a.fc;
    f.ia = a.ia;
    f.theta = a.theta;
    f.collision_th = a.collision_th;
    const int waves = f.tlist ? target_list_groups(a.n, ntarget) : (ntarget + 63) / 64;
    const dim3 grid((waves + TR3_WPB - 1) / TR3_WPB);
    const int mode = a.fc.exact_denom ? FM_EXACT : (a.fc.eps2_negligible ? FM_NOEPS : FM_EPS);
    if (a.collect_stats) {
        if (mode == FM_EXACT)
            hipLaunchKernelGGL((k_bh3_force<FM_EXACT, true>), grid, dim3(64 * TR3_WPB), 0, s, f);
        else if (mode == FM_EPS)
            hipLaunchKernelGGL((k_bh3_force<FM_EPS, true>), grid, dim3(64 * TR3_WPB), 0, s, f);
        else
            hipLaunchKernelGGL((k_bh3_force<FM_NOEPS, true>), grid, dim3(64 * TR3_WPB), 0, s, f);
    } else {
        if (mode == FM_EXACT)
            hipLaunchKernelGGL((k_bh3_force<FM_EXACT, false>), grid, dim3(64 * TR3_WPB), 0, s, f);
        else if (mode == FM_EPS)
            hipLaunchKernelGGL((k_bh3_force<FM_EPS, false>), grid, dim3(64 * TR3_WPB), 0, s, f);
        else
            hipLaunchKernelGGL((k_bh3_force<FM_NOEPS, false>), grid, dim3(64 * TR3_WPB), 0, s, f);
    }
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// ------------------------------------------------------------------ buffers

int bh3_alloc(Bh3Buffers &b, int n)
{
    b.n = n;
    b.node_cap = 2 * (size_t)n + 8;
    NB_TRY(dev_alloc(b.box, 6));
    NB_TRY(dev_alloc(b.cut, 8));
    NB_TRY(dev_alloc(b.bbox_part, 6 * BBOX3_BLOCKS));
    for (int k = 0; k < 2; ++k) {
        NB_TRY(dev_alloc(b.keys[k], (size_t)n));
        NB_TRY(dev_alloc(b.idx[k], (size_t)n));
    }
    NB_TRY(dev_alloc(b.hist, sort_hist_words(n)));
    NB_TRY(dev_alloc(b.hist_totals, sort_totals_words()));
    NB_TRY(dev_alloc(b.leaf_sum, 6 * (size_t)n));
    NB_TRY(dev_alloc(b.chunk_info, (size_t)n / 64 + 2));
    NB_TRY(dev_alloc(b.posm_s, (size_t)n));
    NB_TRY(dev_alloc(b.rad_s, (size_t)n));
    NB_TRY(dev_alloc(b.node_meta, b.node_cap));
    {
        uint8_t *sp = nullptr;
        NB_TRY(dev_alloc(sp, ((size_t)n + 256) * build_split_bytes(3)));
        b.split = sp;
    }
    NB_TRY(dev_alloc(b.level_start, BH_MAX_DEPTH + 2));
    NB_TRY(dev_alloc(b.bsum, BH_SCAN_BLOCKS));
    NB_TRY(dev_alloc(b.node_com, 6 * b.node_cap));
    NB_TRY(dev_alloc(b.node_rec, b.node_cap));
    NB_TRY(dev_alloc(b.node_aux, b.node_cap));
    NB_TRY(dev_alloc(b.tlist, target_list_capacity(n)));
    NB_TRY(dev_alloc(b.tcount, 4));
    NB_HIP(hipMemset(b.tcount, 0, 4 * sizeof(uint32_t)));
    NB_TRY(dev_alloc(b.counters, 16));
    NB_TRY(dev_alloc(b.wave_cost, (size_t)target_list_groups(n, n) + 2));
    void *h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocDefault) != hipSuccess)
        return fail(NB_ERR_NOMEM, "hipHostMalloc failed");
    b.levels_host = (uint32_t *)h;
    *b.levels_host = 0u;
    return NB_OK;
}

void bh3_free(Bh3Buffers &b)
{
    (void)hipFree(b.box);
    (void)hipFree(b.cut);
    (void)hipFree(b.bbox_part);
    for (int k = 0; k < 2; ++k) {
        (void)hipFree(b.keys[k]);
        (void)hipFree(b.idx[k]);
    }
    (void)hipFree(b.hist);
    (void)hipFree(b.hist_totals);
    (void)hipFree(b.leaf_sum);
    (void)hipFree(b.chunk_info);
    (void)hipFree(b.posm_s);
    (void)hipFree(b.rad_s);
    (void)hipFree(b.node_meta);
    (void)hipFree(b.split);
    (void)hipFree(b.level_start);
    (void)hipFree(b.bsum);
    (void)hipFree(b.node_com);
    (void)hipFree(b.node_rec);
    (void)hipFree(b.node_aux);
    (void)hipFree(b.tlist);
    (void)hipFree(b.tcount);
    (void)hipFree(b.counters);
    (void)hipFree(b.wave_cost);
    if (b.levels_host)
        (void)hipHostFree(b.levels_host);
    b = Bh3Buffers();
}

}  // namespace nb
