// nb_bh_tree.hip -- the tree of the Barnes-Hut pipeline: the level-by-level build, the leaf pass,
// the centres of mass, the dense pyramid that replaces them when the keys stop at a level it
// covers, and the cell exchange between GPUs.  Replaces ConstructQuadTreeKernel and its helpers
// (src/gpu/BH/barnesHut_kernel.cu:149-360); the tree itself is the CPU program's
// (src/cpu/quadtree.cpp:73-144, centres of mass :187-228).
#include <cmath>

#include "nb_bh_internal.h"

namespace nb {

// ------------------------------------------------------------------ 4 tree build

// The empty tree: the root holds every body, no other level has a node yet.  (A block of >= 64 threads.)
__device__ __forceinline__ void build_init(uint4 *node_meta, uint32_t *level_start, int n, unsigned long long *counters)
{
    int t = threadIdx.x;
    if (t < BH_MAX_DEPTH + 2)
        level_start[t] = t == 0 ? 0u : 1u;
    if (t == 0)
        node_meta[0] = make_uint4(0u, (uint32_t)n, 0u, 0u);
    if (t < 16)
        counters[t] = 0ull;
}

// Lanes that search one node's quadrant boundaries together.  The searches of a level with few
// nodes are pure latency (a chain of dependent loads per node, over ranges of millions of
// bodies near the root), so such levels give each node a whole wave or 8 lanes, which probe
// 64 or 8 positions per round instead of one.
__device__ __forceinline__ int search_group(uint32_t m) { return m <= 4096u ? 64 : (m <= 65536u ? 8 : 1); }

// Slots of tree level d are dealt to the grid's blocks in contiguous chunks (multiples
// of 256 where a thread has a node to itself), the same way in the count and in the emit kernel.
__device__ __forceinline__ void level_chunk(const uint32_t *level_start, int d, uint32_t &lbeg, uint32_t &m,
                                            uint32_t &cbeg, uint32_t &cend)
{
    lbeg = level_start[d];
    m = level_start[d + 1] - lbeg;
    uint32_t chunk = (m + gridDim.x - 1) / gridDim.x;
    if (search_group(m) == 1)
        chunk = (chunk + 255u) & ~255u;
    cbeg = blockIdx.x * chunk;
    cend = cbeg + chunk;
    if (cbeg > m)
        cbeg = m;
    if (cend > m)
        cend = m;
}

// What the build keeps per frontier slot: the boundaries between a node's children in the sorted
// order (3 for a quadtree, BITS = 2; 7 for an octree, BITS = 3) and child count | cell level << 8.
template <int BITS>
struct Split {
    uint32_t s[(1 << BITS) - 1];
    uint32_t w;
};

// Child boundaries of a node's body range: the first indices in [first, end) whose digit at
// `shift` is >= 1, >= 2, ... (digits are non-decreasing there).  The binary searches advance in
// lockstep, so their dependent loads overlap.
template <int BITS>
__device__ __forceinline__ void digit_boundaries(const uint64_t *keys, uint32_t first, uint32_t end, int shift,
                                                 uint32_t (&out)[(1 << BITS) - 1])
{
    constexpr int NB = (1 << BITS) - 1;
    constexpr uint32_t DMASK = (1u << BITS) - 1u;
    uint32_t lo[NB], hi[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        lo[b] = first;
        hi[b] = end;
    }
    for (;;) {
        bool open = false;
#pragma unroll
        for (int b = 0; b < NB; ++b)
            open = open || lo[b] < hi[b];
        if (!open)
            break;
        uint32_t mid[NB], dg[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            mid[b] = lo[b] + ((hi[b] - lo[b]) >> 1);
            dg[b] = lo[b] < hi[b] ? (uint32_t)(keys[mid[b]] >> shift) & DMASK : 0u;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
            if (lo[b] < hi[b]) {
                if (dg[b] < (uint32_t)(b + 1))
                    lo[b] = mid[b] + 1;
                else
                    hi[b] = mid[b];
            }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
        out[b] = lo[b];
}

// The same boundaries found by G lanes together (G = 8 or 64, a power of two; the G lanes are
// consecutive lanes of one wave and all call this with the same range): every round each lane
// probes one of G evenly spaced positions of each range, and the first lane whose digit reaches
// the boundary's value narrows that range to the gap before its probe.
template <int G, int BITS>
__device__ __forceinline__ void digit_boundaries_group(const uint64_t *keys, uint32_t first, uint32_t end, int shift,
                                                       uint32_t (&out)[(1 << BITS) - 1])
{
    constexpr int NB = (1 << BITS) - 1;
    constexpr uint32_t DMASK = (1u << BITS) - 1u;
    const int lane = threadIdx.x & 63, g = lane & (G - 1), gshift = lane & ~(G - 1);
    const uint64_t gmask = G == 64 ? ~0ull : ((1ull << G) - 1ull);
    uint32_t lo[NB], hi[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        lo[b] = first;
        hi[b] = end;
    }
    for (;;) {
        bool open = false;
#pragma unroll
        for (int b = 0; b < NB; ++b)
            open = open || lo[b] < hi[b];
        if (!open)
            break;
        uint32_t probe[NB];
        bool reach[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const uint32_t len = hi[b] - lo[b];
            probe[b] = lo[b] + (uint32_t)(((uint64_t)len * (uint32_t)(g + 1)) / (uint32_t)(G + 1));
            reach[b] = len > 0u && ((uint32_t)(keys[len > 0u ? probe[b] : first] >> shift) & DMASK) >= (uint32_t)(b + 1);
        }
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            const uint32_t len = hi[b] - lo[b];
            if (len == 0u)
                continue;
            const uint64_t hit = (__builtin_amdgcn_ballot_w64(reach[b]) >> gshift) & gmask;
            const int t = hit ? __builtin_ctzll(hit) : G;  // first probe at or past the boundary
            const uint32_t before = t == 0 ? lo[b] : lo[b] + (uint32_t)(((uint64_t)len * (uint32_t)t) / (uint32_t)(G + 1)) + 1u;
            const uint32_t at = t == G ? hi[b] : lo[b] + (uint32_t)(((uint64_t)len * (uint32_t)(t + 1)) / (uint32_t)(G + 1));
            lo[b] = before;
            hi[b] = at;
        }
    }
#pragma unroll
    for (int b = 0; b < NB; ++b)
        out[b] = lo[b];
}

// What a node becomes: its child boundaries and child count | cell level << 8 (0 children for
// a leaf: one body, or several bodies with one key).  G lanes call it together for one node.
// Keys hold their digits from bit 61 down, BITS bits per level.
template <int G, int BITS>
__device__ __forceinline__ Split<BITS> classify_node(const uint64_t *keys, uint4 meta)
{
    constexpr int NB = (1 << BITS) - 1;
    Split<BITS> sp;
#pragma unroll
    for (int b = 0; b < NB; ++b)
        sp.s[b] = 0u;
    sp.w = 0u;
    const uint32_t first = meta.x, count = meta.y, last = first + count - 1;
    const uint64_t kf = keys[first], kl = keys[last];
    if (count <= 1 || kf == kl)
        return sp;
    // level of the smallest cell holding the whole range: the first differing digit
    const int lvl = (__clzll((long long)(kf ^ kl)) - 2) / BITS;
    const int shift = 62 - BITS - BITS * lvl;
    const uint32_t end = first + count;
    if (G == 1)
        digit_boundaries<BITS>(keys, first, end, shift, sp.s);
    else
        digit_boundaries_group<G, BITS>(keys, first, end, shift, sp.s);
    uint32_t nchild = (sp.s[0] > first) + (end > sp.s[NB - 1]);
#pragma unroll
    for (int b = 1; b < NB; ++b)
        nchild += sp.s[b] > sp.s[b - 1];
    sp.w = nchild | ((uint32_t)lvl << 8);
    return sp;
}

// Writes a node's final metadata and the ranges of its children starting at slot cf.
template <int BITS>
__device__ __forceinline__ void emit_children(uint4 *node_meta, uint32_t u, const Split<BITS> &sp, uint32_t cf)
{
    constexpr int NB = (1 << BITS) - 1;
    uint4 meta = node_meta[u];
    meta.z = cf;
    meta.w = sp.w;
    node_meta[u] = meta;
    if (sp.w & 0xffu) {
        const uint32_t end = meta.x + meta.y;
        uint32_t c = cf, from = meta.x;
#pragma unroll
        for (int b = 0; b <= NB; ++b) {
            const uint32_t to = b < NB ? sp.s[b] : end;
            if (to > from)
                node_meta[c++] = make_uint4(from, to - from, 0u, 0u);
            from = to;
        }
    }
}

template <int BITS>
__device__ __forceinline__ void build_count(const uint64_t *keys, uint4 *node_meta, Split<BITS> *split,
                                            const uint32_t *level_start, uint32_t *bsum, int d)
{
    uint32_t lbeg, m, cbeg, cend;
    level_chunk(level_start, d, lbeg, m, cbeg, cend);
    int local = 0;
    const int group = search_group(m);  // uniform over the grid
    if (group == 64) {
        for (uint32_t slot = cbeg + (threadIdx.x >> 6); slot < cend; slot += 4) {
            const Split<BITS> sp = classify_node<64, BITS>(keys, node_meta[lbeg + slot]);
            if ((threadIdx.x & 63) == 0) {
                local += (int)(sp.w & 0xffu);
                split[slot] = sp;
            }
        }
    } else if (group == 8) {
        // a wave's eight groups stay together until the last of them is done (they share ballots)
        for (uint32_t base = cbeg; base < cend; base += 32) {
            const uint32_t slot = base + (threadIdx.x >> 3);
            Split<BITS> sp;
            sp.w = 0u;
            if (slot < cend)
                sp = classify_node<8, BITS>(keys, node_meta[lbeg + slot]);
            if (slot < cend && (threadIdx.x & 7) == 0) {
                local += (int)(sp.w & 0xffu);
                split[slot] = sp;
            }
        }
    } else {
        for (uint32_t slot = cbeg + threadIdx.x; slot < cend; slot += 256) {
            const Split<BITS> sp = classify_node<1, BITS>(keys, node_meta[lbeg + slot]);
            local += (int)(sp.w & 0xffu);
            split[slot] = sp;
        }
    }
    const int tot = block_sum_256(local);
    if (threadIdx.x == 0)
        bsum[blockIdx.x] = (uint32_t)tot;
}

template <int BITS>
__global__ __launch_bounds__(256) void k_build_count(const uint64_t *keys, uint4 *node_meta, Split<BITS> *split,
                                                     const uint32_t *level_start, uint32_t *bsum, int d)
{
    build_count<BITS>(keys, node_meta, split, level_start, bsum, d);
}

template <int BITS>
__device__ __forceinline__ void build_emit(uint4 *node_meta, const Split<BITS> *split, uint32_t *level_start,
                                           const uint32_t *bsum, int d)
{
    uint32_t lbeg, m, cbeg, cend;
    level_chunk(level_start, d, lbeg, m, cbeg, cend);
    // children allocated by the blocks before this one, and by the whole level
    int before = 0, all = 0;
    for (int k = threadIdx.x; k < (int)gridDim.x; k += 256) {
        int v = (int)bsum[k];
        all += v;
        if (k < (int)blockIdx.x)
            before += v;
    }
    before = block_sum_256(before);
    all = block_sum_256(all);
    const uint32_t next_beg = level_start[d + 1];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        level_start[d + 2] = next_beg + (uint32_t)all;
    uint32_t running = next_beg + (uint32_t)before;
    for (uint32_t tile = cbeg; tile < cend; tile += 256) {
        const uint32_t slot = tile + threadIdx.x;
        const bool valid = slot < cend;
        Split<BITS> sp;
        sp.w = 0u;
        if (valid)
            sp = split[slot];
        const int nchild = (int)(sp.w & 0xffu);
        int tile_total;
        const int excl = block_excl_scan_256(nchild, tile_total);
        if (valid)
            emit_children<BITS>(node_meta, lbeg + slot, sp, running + (uint32_t)excl);
        running += (uint32_t)tile_total;
    }
}

template <int BITS>
__global__ __launch_bounds__(256) void k_build_emit(uint4 *node_meta, const Split<BITS> *split, uint32_t *level_start,
                                                    const uint32_t *bsum, int d)
{
    build_emit<BITS>(node_meta, split, level_start, bsum, d);
}

// The start of the build in ONE launch of one block: the empty tree, then levels 0 .. top - 1
// (a single block is all these levels -- at most 2^BITS nodes -- get anyway).
template <int BITS>
__global__ __launch_bounds__(256) void k_build_top(const uint64_t *keys, uint4 *node_meta, Split<BITS> *split,
                                                   uint32_t *level_start, uint32_t *bsum, int n,
                                                   unsigned long long *counters, int top)
{
    build_init(node_meta, level_start, n, counters);
    for (int d = 0; d < top; ++d) {
        __threadfence_block();
        __syncthreads();  // level_start[d + 1] and the level's nodes are written
        build_count<BITS>(keys, node_meta, split, level_start, bsum, d);
        __threadfence_block();
        __syncthreads();  // bsum[0] and the splits are written
        build_emit<BITS>(node_meta, split, level_start, bsum, d);
    }
}

// Levels from d0 on, by ONE block: the host launches the per-level kernels only for as many
// levels as the previous steps' trees had (plus a margin) and lets this kernel finish whatever
// is left -- normally nothing, so it returns at once; a tree that grew deeper is still built
// completely, just by fewer threads.
// When the tree is complete it leaves the numbers of nodes and levels in counters[2], [3] and the
// level count in *levels_host (host-visible: it sizes the next steps' launches).
template <int BITS>
__global__ __launch_bounds__(256) void k_build_tail(const uint64_t *keys, uint4 *node_meta, uint32_t *level_start, int d0,
                                                    unsigned long long *counters, uint32_t *levels_host)
{
    for (int d = d0; d < BH_MAX_DEPTH - 2; ++d) {
        const uint32_t lbeg = level_start[d], m = level_start[d + 1] - lbeg;
        if (m == 0)
            break;  // uniform: every thread reads the same words
        uint32_t running = level_start[d + 1];
        for (uint32_t tile = 0; tile < m; tile += 256) {
            const uint32_t slot = tile + threadIdx.x;
            Split<BITS> sp;
            sp.w = 0u;
            if (slot < m)
                sp = classify_node<1, BITS>(keys, node_meta[lbeg + slot]);
            int tile_total;
            const int excl = block_excl_scan_256((int)(sp.w & 0xffu), tile_total);
            if (slot < m)
                emit_children<BITS>(node_meta, lbeg + slot, sp, running + (uint32_t)excl);
            running += (uint32_t)tile_total;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            level_start[d + 2] = running;
        __threadfence_block();
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        int levels = 0;
        for (int d = 0; d < BH_MAX_DEPTH; ++d)
            if (level_start[d + 1] > level_start[d])
                levels = d + 1;
        counters[2] = level_start[levels];
        counters[3] = (unsigned long long)levels;
        if (levels_host)
            *levels_host = (uint32_t)levels;
    }
}

// Tree levels the per-level kernels are launched for: what the last finished step needed (the
// device leaves that number in host-visible memory; reading a stale one only costs speed) plus a
// margin, everything on the first steps.
int launched_levels(const uint32_t *levels_host)
{
    const uint32_t seen = levels_host ? *(volatile const uint32_t *)levels_host : 0u;
    int l = seen == 0u ? BH_MAX_DEPTH - 2 : (int)seen + 1;
    if (l > BH_MAX_DEPTH - 2)
        l = BH_MAX_DEPTH - 2;
    return l;
}

size_t build_split_bytes(int bits) { return bits == 3 ? sizeof(Split<3>) : sizeof(Split<2>); }

// Blocks for a per-level kernel of tree level d: the level holds at most 2^(bits d) nodes, and a
// kernel over a handful of nodes finishes sooner when it does not dispatch a thousand idle blocks.
int level_blocks(int d, int bits, int nodes_per_block)
{
    const int shift = bits * d;
    if (shift >= 20)
        return BH_SCAN_BLOCKS;
    const int nodes = 1 << shift;
    const int blocks = (nodes + nodes_per_block - 1) / nodes_per_block;
    return blocks < BH_SCAN_BLOCKS ? blocks : BH_SCAN_BLOCKS;
}

constexpr int BUILD_TOP_LEVELS = 2;  // level_blocks() gives levels 0 and 1 one block

template <int BITS>
static void launch_build(hipStream_t s, const uint64_t *keys, int n, uint4 *node_meta, void *split, uint32_t *level_start,
                         uint32_t *bsum, unsigned long long *counters, int launched, uint32_t *levels_host)
{
    const int top = launched < BUILD_TOP_LEVELS ? launched : BUILD_TOP_LEVELS;
    hipLaunchKernelGGL(k_build_top<BITS>, dim3(1), dim3(256), 0, s, keys, node_meta, (Split<BITS> *)split, level_start, bsum,
                       n, counters, top);
    for (int d = top; d < launched; ++d) {
        const dim3 grid(level_blocks(d, BITS, 4));  // 4 nodes per block while a wave searches for each node
        hipLaunchKernelGGL(k_build_count<BITS>, grid, dim3(256), 0, s, keys, node_meta, (Split<BITS> *)split,
                           level_start, bsum, d);
        hipLaunchKernelGGL(k_build_emit<BITS>, grid, dim3(256), 0, s, node_meta,
                           (const Split<BITS> *)split, level_start, bsum, d);
    }
    hipLaunchKernelGGL(k_build_tail<BITS>, dim3(1), dim3(256), 0, s, keys, node_meta, level_start, launched, counters,
                       levels_host);
}

int build_tree(hipStream_t s, int bits, const uint64_t *keys, int n, uint4 *node_meta, void *split, uint32_t *level_start,
               uint32_t *bsum, unsigned long long *counters, int launched, uint32_t *levels_host)
{
    if (bits == 3)
        launch_build<3>(s, keys, n, node_meta, split, level_start, bsum, counters, launched, levels_host);
    else
        launch_build<2>(s, keys, n, node_meta, split, level_start, bsum, counters, launched, levels_host);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

int bh_build(hipStream_t s, BhBuffers &b, const BhStepArgs &a)
{
    if (b.use_pyr)
        return NB_OK;  // the dense pyramid is built from the leaf pass's output: see bh_com
    b.launched = launched_levels(b.levels_host);
    return build_tree(s, 2, b.keys[b.sorted], a.n, b.node_meta, b.split, b.level_start, b.bsum, b.counters, b.launched,
                      b.levels_host);
}

// ------------------------------------------------------------------ 5 centre of mass

struct ComArgs {
    const uint4 *node_meta;
    const uint32_t *level_start;
    const uint32_t *idx;
    const double2 *pos;
    const double2 *mass_radius;
    const double *box;
    double *node_com;
    float4 *node_rec;
    uint2 *node_aux;
    const uint32_t *cut;
    const double *leaf_sum;  // per sorted position: sums of the run of equal keys that starts there
    // The float coordinates of a step are taken relative to the middle of that step's bounding
    // box (a.box): every kernel of the step reads the same box, so the records agree, and a system
    // that drifts away from where it was set keeps all 24 bits for its own extent.
    double inv_len, inv_mass, theta, cth;
    int bucket;  // leaves are point masses judged like cells (the GPU program's tree)
};

// Traversal record of a leaf that holds one body: its position relative to the step's origin and
// its mass in internal units, and minus its pair cutoff radius + collision_th (the sign marks a body).
__device__ __forceinline__ float4 body_record(const ComArgs &a, double2 p, double2 mass_radius)
{
    const double ox = 0.5 * (a.box[0] + a.box[2]), oy = 0.5 * (a.box[1] + a.box[3]);
    return make_float4((float)((p.x - ox) * a.inv_len), (float)((p.y - oy) * a.inv_len), (float)(mass_radius.x * a.inv_mass),
                       -(float)((mass_radius.y + a.cth) * a.inv_len));
}

// Traversal record, width and prune threshold of a cell of level lvl with the given centre of mass.
__device__ __forceinline__ void cell_record(const ComArgs &a, double cx, double cy, double mtot, int lvl, double width0,
                                            float4 &rec, double &width, float &prune)
{
    width = ldexp(width0, -lvl);
    const double wt = width * a.inv_len / a.theta;  // width / dist < theta  <=>  dist > width / theta
    const double ox = 0.5 * (a.box[0] + a.box[2]), oy = 0.5 * (a.box[1] + a.box[3]);
    rec = make_float4((float)((cx - ox) * a.inv_len), (float)((cy - oy) * a.inv_len), (float)(mtot * a.inv_mass),
                      (float)(wt * wt));
    // Prune threshold: the cell's centre of mass lies inside the cell, so every point of
    // the cell is within dist(com) + diagonal of a target.  If that is below collision_th
    // (less a margin for float rounding) nothing in the cell can contribute.
    const double height = ldexp(a.box[1] - a.box[3], -lvl);
    const double reach = a.cth * (1.0 - 1.0e-4) - sqrt(width * width + height * height);
    prune = reach > 0.0 ? (float)((reach * a.inv_len) * (reach * a.inv_len)) : -1.f;
}

// Record of a leaf that holds several bodies (all with one key), from their sums.
__device__ __forceinline__ void multi_body_record(const ComArgs &a, double sx, double sy, double mtot, double rmax,
                                                  double width0, double &cx, double &cy, float4 &rec, double &width,
                                                  float &prune)
{
    cx = sx / mtot;
    cy = sy / mtot;
    const uint32_t closed_level = a.cut[0];
    if (closed_level <= (uint32_t)BH_KEY_LEVELS) {
        // the bodies of a cell no body ever opens (see k_bbox_final): a cell record, no children
        cell_record(a, cx, cy, mtot, (int)closed_level, width0, rec, width, prune);
    } else {
        // bodies at one position (equal in all 31 levels): they act as one body
        const double ox = 0.5 * (a.box[0] + a.box[2]), oy = 0.5 * (a.box[1] + a.box[3]);
        rec = make_float4((float)((cx - ox) * a.inv_len), (float)((cy - oy) * a.inv_len), (float)(mtot * a.inv_mass),
                          -(float)((rmax + a.cth) * a.inv_len));
        width = -rmax;
    }
    if (a.bucket) {
        // a cell of zero width: accepted from any distance > 0, then subject to the cell
        // rule 2 r_i + collision_th (ComputeForce, barnesHut_kernel.cu:387-399)
        rec.w = 0.f;
        width = 0.0;
    }
}

__device__ __forceinline__ void store_node(const ComArgs &a, uint32_t u, const uint4 meta, double cx, double cy, double mtot,
                                           double width, float4 rec, float prune)
{
    double2 *o = (double2 *)a.node_com + 2 * (size_t)u;
    o[0] = make_double2(cx, cy);
    o[1] = make_double2(mtot, width);
    a.node_rec[u] = rec;
    // children of a node of the level-by-level tree are the consecutive slots meta.z .. + count - 1
    a.node_aux[u] = make_uint2((meta.z << 4) | ((1u << (meta.w & 0xffu)) - 1u), __float_as_uint(prune));
}

// Sums over every run of bodies with one key (= every leaf with several bodies), left at the
// run's first sorted position as {sum m x, sum m y, sum m, max radius}.  One wave per 64
// consecutive sorted bodies: a segmented scan inside the wave gives the runs that end in its
// chunk; of a run that crosses chunk boundaries every chunk leaves its part (the chunk where it
// starts at the run's first position, the others in their own first slot) and k_leaf_finish adds
// them up.  The order of the additions is fixed by the positions alone (wave_run_scan), so the
// sums are reproducible.
__global__ __launch_bounds__(256) void k_leaf_sums(const uint64_t *keys, const uint32_t *idx, int n, const double2 *pos,
                                                   const double2 *mass_radius, double *leaf_sum, uint32_t *chunk_info,
                                                   const uint32_t *cut, uint2 *pc_range)
{
    const int lane = lane_id();
    const int chunk = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64;
    if (chunk >= n)
        return;
    const int k = chunk + lane;
    const bool valid = k < n;
    const uint64_t key = valid ? keys[k] : ~0ull;
    const uint64_t prev = (valid && k > 0) ? keys[k - 1] : ~0ull;  // ~0 is no key (62 bits)
    double sm = 0.0, sx = 0.0, sy = 0.0, rm = 0.0;
    if (valid) {
        const uint32_t o = idx[k];
        const double2 p = pos[o];
        const double2 mr = mass_radius[o];  // one 16-byte gather for the two constants of a body
        sm = mr.x;
        sx = p.x * sm;
        sy = p.y * sm;
        rm = mr.y;
    }
    const bool head = valid && (k == 0 || key != prev);
    const uint64_t heads = __builtin_amdgcn_ballot_w64(head);
    // the run of the chunk's last body goes on in the next chunk?  (uniform)
    const int tail = chunk + 64;
    const uint64_t tail_key = tail < n ? keys[tail] : ~0ull;
    // the next lane starts a run, or the sorted bodies end
    const bool next_differs = lane == 63 ? tail_key != key : (((heads >> (lane + 1)) & 1ull) != 0ull || k + 1 >= n);
    if (cut[3] != 0u) {
        // Dense-pyramid build: a run of equal keys is the population of one cell of the level the
        // keys stop at; its first and one-past-last sorted positions go to that cell's slot (the
        // slots of empty cells were zeroed before this kernel).
        const int shift = 62 - 2 * (int)cut[1];
        const uint32_t cell = (uint32_t)(key >> shift);
        if (head)
            pc_range[cell].x = (uint32_t)k;
        if (valid && next_differs)
            pc_range[cell].y = (uint32_t)k + 1u;
    }
    // Segmented scan.  The bodies before the chunk's first head continue a run that started in an
    // earlier chunk; for the scan they are a run of their own that starts at lane 0.
    const int first_head = heads ? (int)__builtin_ctzll(heads) : 64;
    const uint64_t upto = (heads | 1ull) & ((2ull << lane) - 1ull);
    const int start = 63 - __builtin_clzll(upto);  // first lane of this lane's run
    const int dist = lane - start;
    {
        double sums[3] = {sm, sx, sy};
        wave_run_scan<3>(sums, rm, lane, start);
        sm = sums[0];
        sx = sums[1];
        sy = sums[2];
    }
    // The part of the earlier run goes to the chunk's first slot (no run starts there, so nothing
    // else uses it), and k_leaf_finish adds it to that run.
    if (first_head > 0 && lane == first_head - 1) {
        double *o = leaf_sum + 4 * (size_t)chunk;
        o[0] = sx;
        o[1] = sy;
        o[2] = sm;
        o[3] = rm;
    }
    if (heads == 0ull) {
        if (lane == 0)
            chunk_info[chunk >> 6] = 64u | (LEAF_NO_OWNER << 8);
        return;  // the whole chunk lies inside a run that started earlier
    }
    // last lane of each run that starts inside the chunk: the next lane starts a run, or the chunk ends
    const bool last_in_chunk = lane == 63 || !valid || ((heads >> (lane + 1)) & 1ull) || k + 1 >= n;
    const bool ends = valid && lane >= first_head && last_in_chunk;
    const bool goes_on = tail < n && tail_key == lane_value(key, 63);
    if (ends && !(lane == 63 && goes_on) && dist > 0) {
        double *o = leaf_sum + 4 * (size_t)(chunk + start);
        o[0] = sx;
        o[1] = sy;
        o[2] = sm;
        o[3] = rm;
    }
    // The run of the chunk's last body, when it started here and goes on: its part inside the
    // chunk is left at its start, and k_leaf_finish adds the parts of the chunks it crosses.
    const int start63 = 63 - __builtin_clzll(heads);
    const bool owner = goes_on;
    if (owner && lane == 63) {
        double *o = leaf_sum + 4 * (size_t)(chunk + start63);
        o[0] = sx;
        o[1] = sy;
        o[2] = sm;
        o[3] = rm;
    }
    if (lane == 0)
        chunk_info[chunk >> 6] = (uint32_t)first_head | ((owner ? (uint32_t)start63 : LEAF_NO_OWNER) << 8);
}

// Finishes the runs that cross chunk boundaries.  A run's parts are: what the chunk where it
// starts left at the run's first position, the first slot of every chunk wholly inside the run (no
// head), and the first slot of the chunk where it ends (the bodies before that chunk's first
// head).  One thread per chunk: the thread of a chunk whose last run goes on adds the parts of the
// next LEAF_SHORT chunks itself, which finishes the runs of a few hundred bodies (leaves of ~130
// bodies are the rule when the tree is closed at level 9: about every second chunk owns a run).
// A run that is longer than that -- the core of a collapsed disc inside one closed cell holds tens
// of thousands of bodies -- is summed by the whole wave, 64 chunks a round.  (Left to one wave
// walking the bodies 64 at a time, such a run took longer than the rest of the pass: 0.09 ->
// 0.49 ms.)  The order of the additions depends on the positions alone.
// A slot of leaf_sum holds STRIDE doubles: NSUM sums, then the largest radius.
constexpr int LEAF_SHORT = 4;

template <int NSUM, int STRIDE>
__device__ __forceinline__ void leaf_finish_chunk(const uint32_t *chunk_info, int nchunks, double *leaf_sum, const int c)
{
    const int lane = lane_id();
    const uint32_t mine = c < nchunks ? chunk_info[c] : (LEAF_NO_OWNER << 8);
    const bool own = (mine >> 8) != LEAF_NO_OWNER;
    double ps[NSUM], pr = 0.0;
#pragma unroll
    for (int v = 0; v < NSUM; ++v)
        ps[v] = 0.0;
    // the next LEAF_SHORT chunks' words in one round of loads, then the parts that belong to the
    // run in another (not one chunk after the other: each step would wait for two loads)
    uint32_t fh[LEAF_SHORT];
#pragma unroll
    for (int it = 0; it < LEAF_SHORT; ++it) {
        const int cc = c + 1 + it;
        // a chunk past the end ends the run like a chunk that starts with a head
        fh[it] = (own && cc < nchunks) ? (chunk_info[cc] & 0xffu) : 0u;
    }
    bool open = own;  // the end of the run is not found yet
#pragma unroll
    for (int it = 0; it < LEAF_SHORT; ++it) {
        if (open && fh[it] > 0u) {
            const double *p = leaf_sum + (size_t)STRIDE * ((size_t)(c + 1 + it) * 64);
#pragma unroll
            for (int v = 0; v < NSUM; ++v)
                ps[v] += p[v];
            pr = fmax(pr, p[NSUM]);
        }
        open = open && fh[it] == 64u;
    }
    if (own && !open) {
        double *t = leaf_sum + (size_t)STRIDE * ((size_t)c * 64 + (mine >> 8));
#pragma unroll
        for (int v = 0; v < NSUM; ++v)
            t[v] += ps[v];
        t[NSUM] = fmax(t[NSUM], pr);
    }
    uint64_t owners = __builtin_amdgcn_ballot_w64(own && open);  // the long runs
    while (owners != 0ull) {
        const int o = (int)__builtin_ctzll(owners);
        owners &= owners - 1ull;
        const int c0 = c - lane + o;
        const uint32_t st = (uint32_t)__builtin_amdgcn_readlane((int)(mine >> 8), o);
#pragma unroll
        for (int v = 0; v < NSUM; ++v)
            ps[v] = 0.0;
        pr = 0.0;
        for (int base = c0 + 1;; base += 64) {
            const int cc = base + lane;
            const uint32_t fh = cc < nchunks ? (chunk_info[cc] & 0xffu) : 0u;
            const uint64_t term = __builtin_amdgcn_ballot_w64(fh < 64u);
            const int e = term ? (int)__builtin_ctzll(term) : 64;
            if (lane <= e && fh > 0u) {
                const double *p = leaf_sum + (size_t)STRIDE * ((size_t)cc * 64);
#pragma unroll
                for (int v = 0; v < NSUM; ++v)
                    ps[v] += p[v];
                pr = fmax(pr, p[NSUM]);
            }
            if (term != 0ull)
                break;
        }
#pragma unroll
        for (int q = 32; q > 0; q >>= 1) {
#pragma unroll
            for (int v = 0; v < NSUM; ++v)
                ps[v] += __shfl_xor(ps[v], q);
            pr = fmax(pr, __shfl_xor(pr, q));
        }
        if (lane == 0) {
            double *t = leaf_sum + (size_t)STRIDE * ((size_t)c0 * 64 + st);
#pragma unroll
            for (int v = 0; v < NSUM; ++v)
                t[v] += ps[v];
            t[NSUM] = fmax(t[NSUM], pr);
        }
    }
}

// Thread c of the grid takes chunk c (whole waves take consecutive chunks).
template <int NSUM, int STRIDE>
__global__ __launch_bounds__(256) void k_leaf_finish(const uint32_t *chunk_info, int nchunks, double *leaf_sum)
{
    leaf_finish_chunk<NSUM, STRIDE>(chunk_info, nchunks, leaf_sum, blockIdx.x * 256 + threadIdx.x);
}

int leaf_finish(hipStream_t s, const uint32_t *chunk_info, int n, double *leaf_sum, int dims)
{
    const int nchunks = (n + 63) / 64;
    const dim3 grid((nchunks + 255) / 256);
    if (dims == 3)
        hipLaunchKernelGGL((k_leaf_finish<4, 6>), grid, dim3(256), 0, s, chunk_info, nchunks, leaf_sum);
    else
        hipLaunchKernelGGL((k_leaf_finish<3, 4>), grid, dim3(256), 0, s, chunk_info, nchunks, leaf_sum);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// Centres of mass, masses and traversal records of the nodes of tree level d (children first:
// the caller goes through the levels bottom-up).
__device__ __forceinline__ void com_level(const ComArgs &a, int d)
{
    const uint32_t lbeg = a.level_start[d], lend = a.level_start[d + 1];
    const double width0 = a.box[2] - a.box[0];  // QuadTree::getWidth of the root (quadtree.cpp:234-237)
    for (uint32_t u = lbeg + blockIdx.x * 256 + threadIdx.x; u < lend; u += gridDim.x * 256) {
        const uint4 meta = a.node_meta[u];
        const uint32_t nchild = meta.w & 0xffu;
        double cx, cy, mtot, width;
        float4 rec;
        float prune = -1.f;
        if (nchild == 0) {
            // a leaf: one body, or the bodies that share a key
            const uint32_t first = meta.x, count = meta.y;
            if (count == 1) {
                const uint32_t o = a.idx[first];
                const double2 p = a.pos[o];
                cx = p.x;
                cy = p.y;
                const double2 mr = a.mass_radius[o];
                mtot = mr.x;
                rec = body_record(a, p, mr);
                width = -mr.y;
                if (a.bucket) {
                    rec.w = 0.f;  // see multi_body_record; the target's own leaf (distance 0) is
                    width = 0.0;  // "opened" and has no children
                }
            } else {
                const double2 *ls = (const double2 *)a.leaf_sum + 2 * (size_t)first;
                const double2 sxy = ls[0], smr = ls[1];
                mtot = smr.x;
                multi_body_record(a, sxy.x, sxy.y, mtot, smr.y, width0, cx, cy, rec, width, prune);
            }
        } else {
            // updateCenterMass: total = sum of child masses, com = sum(child com * child mass) / total,
            // children in the order TL,TR,BL,BR = ascending digit
            const uint32_t cf = meta.z;
            double sx = 0.0, sy = 0.0;
            mtot = 0.0;
            const double2 *cc = (const double2 *)a.node_com;  // per node {x, y}, {mass, width}
            for (uint32_t c = cf; c < cf + nchild; ++c) {
                const double2 cxy = cc[2 * (size_t)c], cmw = cc[2 * (size_t)c + 1];
                mtot += cmw.x;
                sx += cxy.x * cmw.x;
                sy += cxy.y * cmw.x;
            }
            cx = sx / mtot;
            cy = sy / mtot;
            cell_record(a, cx, cy, mtot, (int)(meta.w >> 8), width0, rec, width, prune);
        }
        store_node(a, u, meta, cx, cy, mtot, width, rec, prune);
    }
}

// d_hi == d + 1: level d by the whole grid.  d_hi > d + 1: levels d_hi-1 down to d by ONE block
// in one launch (the levels past the ones the host expects to exist, normally empty, and the
// few hundred cells of the top levels).
__global__ __launch_bounds__(256) void k_com(ComArgs a, int d, int d_hi)
{
    if (d_hi > d + 1) {
        for (int dd = d_hi - 1; dd >= d; --dd) {
            if (a.level_start[dd + 1] > a.level_start[dd]) {  // uniform
                com_level(a, dd);
                __threadfence_block();
            }
            __syncthreads();
        }
        return;
    }
    com_level(a, d);
}

// ---- dense pyramid: tree, centres of mass and traversal records in two launches ----
//
// Slot pyr_offset(l) + c of the node arrays belongs to cell c of level l and holds the record of the
// node that REPRESENTS the cell in the reference's tree as the traversal sees it:
//   a cell of the last level L: its bodies as one leaf (one body: the body; several: the cell,
//       which no body ever opens, or in the GPU-program tree a point mass);
//   a cell with bodies in two or more quadrants: the cell itself, children = its four child slots
//       (the child word carries a mask of the quadrants that hold bodies);
//   a cell whose bodies all lie in one quadrant: a copy of that child's slot -- such a chain of
//       cells shares its centre of mass, and a wider cell of the chain is accepted only if the
//       narrowest is, with the same force;
//   an empty cell: nothing (its parent's mask leaves it out).
// What one cell hands to its parent:
struct PyrCell {
    double cx, cy, m, w;  // node_com: centre of mass, mass, cell x-width (or minus a body's radius)
    float4 rec;           // node_rec
    uint32_t child, prune;  // node_aux: first child slot << 4 | quadrant mask, prune threshold bits
    uint32_t first, count;  // bodies (sorted positions)
};
static_assert(sizeof(PyrCell) == 64, "LDS budget of the pyramid kernels");

__device__ __forceinline__ void pyr_store(const ComArgs &a, uint4 *node_meta, uint32_t slot, const PyrCell &c)
{
    // (node_meta is what nb_debug_read shows: body range, first child slot, child slots)
    const bool inner = c.count != 0u && (c.child & 15u) != 0u;
    node_meta[slot] = make_uint4(c.first, c.count, inner ? c.child >> 4 : 0u, inner ? 4u : 0u);
    if (c.count == 0u)
        return;
    double2 *o = (double2 *)a.node_com + 2 * (size_t)slot;
    o[0] = make_double2(c.cx, c.cy);
    o[1] = make_double2(c.m, c.w);
    a.node_rec[slot] = c.rec;
    a.node_aux[slot] = make_uint2(c.child, c.prune);
}

__device__ __forceinline__ PyrCell pyr_load(const ComArgs &a, const uint4 *node_meta, uint32_t slot)
{
    PyrCell c;
    const uint4 meta = node_meta[slot];
    c.first = meta.x;
    c.count = meta.y;
    if (c.count != 0u) {
        const double2 *o = (const double2 *)a.node_com + 2 * (size_t)slot;
        const double2 xy = o[0], mw = o[1];
        c.cx = xy.x;
        c.cy = xy.y;
        c.m = mw.x;
        c.w = mw.y;
        c.rec = a.node_rec[slot];
        const uint2 aux = a.node_aux[slot];
        c.child = aux.x;
        c.prune = aux.y;
    }
    return c;
}

// Sums {sum m x, sum m y, sum m, max radius} of the bodies [first, end) of one cell from the parts
// the leaf pass left per 64-body chunk: the chunk where the run starts at the run's first position,
// every later chunk in its own first slot.  A thread adds up to LEAF_SEQ later chunks itself
// (leaves of ~130 bodies are the rule); a longer run -- the core of a collapsed disc puts tens of
// thousands of bodies into one cell -- is left to pyr_long_sums, a wave per run.  (With 8 instead of
// 64 the block that owns the core of a collapsed disc, 256 cells of a few thousand bodies each, had
// its four waves walk 64 such runs one after the other: k_pyr_sub 0.10 ms instead of 0.03.)  The
// order of the additions depends on the positions alone.
constexpr uint32_t LEAF_SEQ = 64;

__device__ __forceinline__ bool pyr_short_sums(const double *leaf_sum, uint32_t first, uint32_t end, double (&out)[4])
{
    if (end <= first + 1u)
        return false;
    const uint32_t c0 = first >> 6, cl = (end - 1u) >> 6;
    const double2 *p = (const double2 *)leaf_sum + 2 * (size_t)first;
    const double2 u = p[0], v = p[1];
    out[0] = u.x;
    out[1] = u.y;
    out[2] = v.x;
    out[3] = v.y;
    if (cl - c0 > LEAF_SEQ)
        return true;  // the later chunks are still to be added
    // four chunks' parts in flight at a time, added in chunk order
    for (uint32_t cc = c0 + 1u; cc <= cl; cc += 4u) {
        double2 uu[4], vv[4];
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i) {
            const double2 *q = (const double2 *)leaf_sum + 2 * ((size_t)(cc + i <= cl ? cc + i : cl) << 6);
            uu[i] = q[0];
            vv[i] = q[1];
        }
#pragma unroll
        for (uint32_t i = 0; i < 4u; ++i)
            if (cc + i <= cl) {
                out[0] += uu[i].x;
                out[1] += uu[i].y;
                out[2] += vv[i].x;
                out[3] = fmax(out[3], vv[i].y);
            }
    }
    return false;
}

// The later chunks of one long run, by a whole wave: 256 chunks a round, four loads in flight per lane.
__device__ __forceinline__ void pyr_long_sums(const double *leaf_sum, uint32_t first, uint32_t end, double (&out)[4])
{
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t b0 = (first >> 6) + 1u, b1 = (end - 1u) >> 6;
    double s[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
        s[i][0] = s[i][1] = s[i][2] = s[i][3] = 0.0;
    for (uint32_t base = b0 + lane; base <= b1; base += 256u) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const uint32_t cc = base + 64u * (uint32_t)i;
            if (cc <= b1) {
                const double2 *q = (const double2 *)leaf_sum + 2 * ((size_t)cc << 6);
                const double2 uu = q[0], vv = q[1];
                s[i][0] += uu.x;
                s[i][1] += uu.y;
                s[i][2] += vv.x;
                s[i][3] = fmax(s[i][3], vv.y);
            }
        }
    }
#pragma unroll
    for (int i = 1; i < 4; ++i) {
        s[0][0] += s[i][0];
        s[0][1] += s[i][1];
        s[0][2] += s[i][2];
        s[0][3] = fmax(s[0][3], s[i][3]);
    }
#pragma unroll
    for (int q = 32; q > 0; q >>= 1) {
        s[0][0] += __shfl_xor(s[0][0], q);
        s[0][1] += __shfl_xor(s[0][1], q);
        s[0][2] += __shfl_xor(s[0][2], q);
        s[0][3] = fmax(s[0][3], __shfl_xor(s[0][3], q));
    }
    out[0] = s[0][0];
    out[1] = s[0][1];
    out[2] = s[0][2];
    out[3] = s[0][3];
}

// The leaf of one cell of the last level (the arithmetic of com_level's leaf branch); sums: the
// cell's pyr_leaf_sums when it holds several bodies.
__device__ __forceinline__ PyrCell pyr_leaf(const ComArgs &a, uint32_t first, uint32_t end, int level, double width0,
                                            const double (&sums)[4])
{
    PyrCell c;
    c.first = first;
    c.count = end > first ? end - first : 0u;
    c.child = 0u;
    c.prune = __float_as_uint(-1.f);
    if (c.count == 1u) {
        const uint32_t o = a.idx[first];
        const double2 p = a.pos[o];
        c.cx = p.x;
        c.cy = p.y;
        const double2 mr = a.mass_radius[o];
        c.m = mr.x;
        c.rec = body_record(a, p, mr);
        c.w = -mr.y;
        if (a.bucket) {
            c.rec.w = 0.f;
            c.w = 0.0;
        }
    } else if (c.count > 1u) {
        c.m = sums[2];
        float prune = -1.f;
        multi_body_record(a, sums[0], sums[1], c.m, sums[3], width0, c.cx, c.cy, c.rec, c.w, prune);
        c.prune = __float_as_uint(prune);
    }
    return c;
}

// A cell from its four child cells (ch[0..3] = quadrants TL,TR,BL,BR; child_slot = slot of ch[0]):
// updateCenterMass (quadtree.cpp:187-228) for a cell with bodies in several quadrants.
__device__ __forceinline__ PyrCell pyr_combine(const ComArgs &a, const PyrCell *ch, uint32_t child_slot, int level,
                                               double width0)
{
    uint32_t mask = 0u, count = 0u, first = 0u;
    int only = -1, nch = 0;
#pragma unroll
    for (int q = 3; q >= 0; --q)
        if (ch[q].count != 0u) {
            mask |= 1u << q;
            count += ch[q].count;
            first = ch[q].first;
            only = q;
            nch += 1;
        }
    if (nch == 1)
        return ch[only];
    PyrCell c;
    c.first = first;
    c.count = count;
    c.child = 0u;
    c.prune = __float_as_uint(-1.f);
    if (nch == 0)
        return c;
    double sx = 0.0, sy = 0.0, mtot = 0.0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (ch[q].count != 0u) {
            mtot += ch[q].m;
            sx += ch[q].cx * ch[q].m;
            sy += ch[q].cy * ch[q].m;
        }
    c.cx = sx / mtot;
    c.cy = sy / mtot;
    c.m = mtot;
    float prune;
    cell_record(a, c.cx, c.cy, mtot, level, width0, c.rec, c.w, prune);
    c.prune = __float_as_uint(prune);
    c.child = (child_slot << 4) | mask;
    return c;
}

// The pyramid is cut at level r = pyr_split(L): k_pyr_sub gives every cell of level r a block that
// builds levels r..L under it bottom-up in LDS (at most 4 levels below the block's cell: 341
// cells); k_pyr_top then builds levels r-1..0 in one block, also in LDS.  With L = 11 the cut is at
// level 7, 16384 cells, too many for one block: k_pyr_mid takes levels 6..3 in between, again a
// block per cell of its top level, and k_pyr_top starts from level pyr_top_from(L) = 3 (L = 10: levels
// 5..2 and 2, when the host launched k_pyr_mid for the step).
__host__ __device__ constexpr int pyr_split(int L) { return L > 4 ? L - 4 : 0; }
// (k_pyr_mid also takes a pyramid of PYR_MID_FROM levels when the host launched it: its 4096 cells of
// level 6 otherwise pass through one block's staging area, 32 us against 8 + 6)
constexpr int PYR_MID_FROM = 10;
__host__ __device__ constexpr int pyr_top_from(int L, bool mid = true)
{
    return (mid && L >= PYR_MID_FROM) ? pyr_split(L) - 4 : pyr_split(L);
}
constexpr int PYR_SUB_CELLS = 341;   // 1 + 4 + 16 + 64 + 256
constexpr int PYR_TOP_CELLS = 1365;  // levels 0..5 ...
constexpr int PYR_TOP_STAGE = 512;   // ... and, when r = 6, the level-6 cells in batches of this many
static_assert(pyr_top_from(PYR_MAX_LEVEL, false) <= 6 && pyr_top_from(PYR_DEEP_LEVEL) <= 6 && pyr_top_from(PYR_MID_FROM) >= 0 &&
                  PYR_MID_FROM <= PYR_MAX_LEVEL,
              "k_pyr_top stages at most one level beyond the six it holds; deeper pyramids come with k_pyr_mid");

// The sums of pyr_short_sums / pyr_long_sums for the cell of every thread of a 256-thread block
// (all threads call; range = {first, end} of the thread's cell, {0, 0} for none).  Long runs are
// dealt to the block's four waves: they cluster in a few neighbouring cells, i.e. in one wave's
// threads.
// scratch: LEAF_SUMS_SCRATCH bytes of LDS, 16-byte aligned, free again when the call returns.
constexpr int LEAF_SUMS_SCRATCH = 256 * (int)(sizeof(uint4) + 4 * sizeof(double));
__device__ __forceinline__ void block_leaf_sums(const double *leaf_sum, uint2 range, double (&sums)[4], void *scratch)
{
    uint4 *long_range = (uint4 *)scratch;  // [256] cells whose bodies span many chunks: first, end, owning thread
    double(*long_sum)[4] = (double(*)[4])((char *)scratch + 256 * sizeof(uint4));  // [256][4]
    __shared__ int nlong;
    sums[0] = sums[1] = sums[2] = sums[3] = 0.0;
    if (threadIdx.x == 0)
        nlong = 0;
    __syncthreads();
    if (pyr_short_sums(leaf_sum, range.x, range.y, sums)) {
        const int at = atomicAdd(&nlong, 1);
        long_range[at] = make_uint4(range.x, range.y, threadIdx.x, 0u);
    }
    __syncthreads();
    for (int i = (int)(threadIdx.x >> 6); i < nlong; i += 4) {
        const uint4 lr = long_range[i];
        double extra[4];
        pyr_long_sums(leaf_sum, lr.x, lr.y, extra);
        if ((threadIdx.x & 63u) == 0u) {
            long_sum[lr.z][0] = extra[0];
            long_sum[lr.z][1] = extra[1];
            long_sum[lr.z][2] = extra[2];
            long_sum[lr.z][3] = extra[3];
        }
    }
    __syncthreads();
    const uint32_t c0 = range.x >> 6, cl = range.y > range.x + 1u ? (range.y - 1u) >> 6 : c0;
    if (cl - c0 > LEAF_SEQ) {
        sums[0] += long_sum[threadIdx.x][0];
        sums[1] += long_sum[threadIdx.x][1];
        sums[2] += long_sum[threadIdx.x][2];
        sums[3] = fmax(sums[3], long_sum[threadIdx.x][3]);
    }
    __syncthreads();  // the shared arrays may be used again
}

// ---- cell exchange between GPUs ----
// Every rank sorts only the bodies it integrates and describes each cell of the keys' last level by
// one record of AGG_DOUBLES doubles: {sum m x, sum m y, sum m, x, y, radius, count, -} -- x, y and
// radius are those of the rank's body when it has exactly one in the cell, else 0.  The records of
// all ranks are added up (one all-reduce); where the total count is 1, the sums x, y, radius are
// that one body's values to the bit (the others add zeros), so single bodies keep their exact
// positions and radii, and every rank builds the same pyramid from the totals without ever seeing
// another rank's bodies.
constexpr int AGG_DOUBLES = 8;

__global__ __launch_bounds__(256) void k_agg_fill(ComArgs a, const uint2 *pc_range, double *agg)
{
    __shared__ __attribute__((aligned(16))) char scratch[LEAF_SUMS_SCRATCH];
    const uint32_t ncell = 1u << (2 * (int)a.cut[1]);
    for (uint32_t base = blockIdx.x * 256u; base < ncell; base += gridDim.x * 256u) {  // whole blocks go round
        const uint32_t c = base + threadIdx.x;
        const uint2 range = c < ncell ? pc_range[c] : make_uint2(0u, 0u);
        const uint32_t count = range.y > range.x ? range.y - range.x : 0u;
        double sums[4];
        block_leaf_sums(a.leaf_sum, range, sums, scratch);
        double rec[AGG_DOUBLES] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        if (count == 1u) {
            const uint32_t o = a.idx[range.x];
            const double2 p = a.pos[o];
            const double2 mr = a.mass_radius[o];
            rec[0] = p.x * mr.x;
            rec[1] = p.y * mr.x;
            rec[2] = mr.x;
            rec[3] = p.x;
            rec[4] = p.y;
            rec[5] = mr.y;
        } else if (count > 1u) {
            rec[0] = sums[0];
            rec[1] = sums[1];
            rec[2] = sums[2];
        }
        rec[6] = (double)count;
        if (c < ncell) {
            double2 *o = (double2 *)agg + (size_t)(AGG_DOUBLES / 2) * c;
#pragma unroll
            for (int k = 0; k < AGG_DOUBLES / 2; ++k)
                o[k] = make_double2(rec[2 * k], rec[2 * k + 1]);
        }
    }
}

// The leaf of cell c of the tree's last level L from the exchanged records of its 4^down cells of
// the keys' last level (down = 0 or more: the keys may go deeper than the tree).
__device__ __forceinline__ PyrCell pyr_leaf_from_cells(const ComArgs &a, const double *agg, uint32_t c, int level, int down,
                                                       double width0)
{
    double t[AGG_DOUBLES - 1] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
    const uint32_t per = 1u << (2 * down);
    for (uint32_t j = 0; j < per; ++j) {
        const double2 *r = (const double2 *)agg + (size_t)(AGG_DOUBLES / 2) * ((size_t)c * per + j);
        const double2 r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
        t[0] += r0.x;
        t[1] += r0.y;
        t[2] += r1.x;
        t[3] += r1.y;
        t[4] += r2.x;
        t[5] += r2.y;
        t[6] += r3.x;
    }
    PyrCell leaf;
    leaf.first = 0u;  // (sorted positions mean nothing across ranks)
    leaf.count = (uint32_t)t[6];
    leaf.child = 0u;
    leaf.prune = __float_as_uint(-1.f);
    if (leaf.count == 1u) {
        const double ox = 0.5 * (a.box[0] + a.box[2]), oy = 0.5 * (a.box[1] + a.box[3]);
        leaf.cx = t[3];
        leaf.cy = t[4];
        leaf.m = t[2];
        leaf.rec = make_float4((float)((t[3] - ox) * a.inv_len), (float)((t[4] - oy) * a.inv_len), (float)(t[2] * a.inv_mass),
                               -(float)((t[5] + a.cth) * a.inv_len));
        leaf.w = -t[5];
    } else if (leaf.count > 1u) {
        leaf.m = t[2];
        float prune = -1.f;
        multi_body_record(a, t[0], t[1], t[2], 0.0, width0, leaf.cx, leaf.cy, leaf.rec, leaf.w, prune);
        leaf.prune = __float_as_uint(prune);
    }
    return leaf;
}

// Levels r .. L: one block per cell of level r.  CELLS: the leaves come from the exchanged cell
// records (agg) instead of this GPU's own sorted bodies (pc_range, leaf sums).
#ifndef NB_PYR_SUB_WAVES
#define NB_PYR_SUB_WAVES 5  // waves per SIMD the kernel is compiled for: five blocks per CU instead of four
#endif
template <bool CELLS>
__global__ __launch_bounds__(256, NB_PYR_SUB_WAVES) void k_pyr_sub(ComArgs a, uint4 *node_meta, const uint2 *pc_range, const double *agg)
{
    if (a.cut[3] == 0u)
        return;
    __shared__ PyrCell cells[PYR_SUB_CELLS];  // levels r..L of the block's cell
    const int L = (int)a.cut[4];
    const int r = pyr_split(L);
    const int sub = L - r;
    const uint32_t nroots = 1u << (2 * r), nleaf = 1u << (2 * sub);
    const double width0 = a.box[2] - a.box[0];
    for (uint32_t root = blockIdx.x; root < nroots; root += gridDim.x) {
        if (!CELLS) {
            // A block whose cell holds no body (outliers stretch the box: most cells of a collapsed
            // system's level 6 are empty) only marks its cells as empty.
            bool any = false;
            for (uint32_t t = threadIdx.x; t < nleaf; t += 256) {
                const uint2 range = pc_range[root * nleaf + t];
                any = any || range.y > range.x;
            }
            if (!__syncthreads_or(any)) {
                for (int k = sub; k >= 0; --k) {
                    const uint32_t cnt = 1u << (2 * k);
                    for (uint32_t t = threadIdx.x; t < cnt; t += 256)
                        node_meta[pyr_offset(r + k) + root * cnt + t] = make_uint4(0u, 0u, 0u, 0u);
                }
                continue;
            }
        }
        for (uint32_t t0 = 0; t0 < nleaf; t0 += 256) {
            const uint32_t t = t0 + threadIdx.x;
            const bool mine = t < nleaf;
            const uint32_t c = root * nleaf + t;
            PyrCell leaf;
            if (CELLS) {
                leaf = pyr_leaf_from_cells(a, agg, mine ? c : 0u, L, (int)a.cut[1] - L, width0);
            } else {
                const uint2 range = mine ? pc_range[c] : make_uint2(0u, 0u);
                double sums[4];
                // (the long runs' scratch lies where this round's leaves go once their sums are known)
                static_assert(LEAF_SUMS_SCRATCH <= 256 * (int)sizeof(PyrCell), "the scratch fits under one round of leaves");
                block_leaf_sums(a.leaf_sum, range, sums, (void *)&cells[pyr_offset(sub) + t0]);
                leaf = pyr_leaf(a, range.x, range.y, L, width0, sums);
            }
            if (mine) {
                cells[pyr_offset(sub) + t] = leaf;
                pyr_store(a, node_meta, pyr_offset(L) + c, leaf);
            }
        }
        __syncthreads();
        for (int k = sub - 1; k >= 0; --k) {
            const uint32_t cnt = 1u << (2 * k);
            for (uint32_t t = threadIdx.x; t < cnt; t += 256) {
                const uint32_t c = root * cnt + t;  // the cell's index within level r + k
                // (the children are read where they lie in LDS: a copy in registers would be indexed
                // by the quadrant that holds the bodies, i.e. end up in scratch memory)
                const PyrCell me = pyr_combine(a, &cells[pyr_offset(k + 1) + 4 * t], pyr_offset(r + k + 1) + 4 * c, r + k, width0);
                cells[pyr_offset(k) + t] = me;
                pyr_store(a, node_meta, pyr_offset(r + k) + c, me);
            }
            __syncthreads();
        }
    }
}

// Levels r - 1 .. r - 4 of a pyramid of PYR_MID_FROM levels or more (r = pyr_split(L)): one block per
// cell of level r - 4 reads the 256 cells of level r under it, which k_pyr_sub stored, and builds
// the four levels above them in LDS.  Launched whenever the host expects such keys (k_pyr_top is told);
// a step whose keys stop earlier has nothing for it to do.
__global__ __launch_bounds__(256) void k_pyr_mid(ComArgs a, uint4 *node_meta)
{
    if (a.cut[3] == 0u)
        return;
    const int L = (int)a.cut[4];
    if (L < PYR_MID_FROM)
        return;
    __shared__ PyrCell cells[PYR_SUB_CELLS];  // levels r - 4 .. r of the block's cell
    const int r = pyr_split(L), r2 = pyr_top_from(L);
    const int sub = r - r2;                   // 4
    const uint32_t nroots = 1u << (2 * r2), nleaf = 1u << (2 * sub);
    const double width0 = a.box[2] - a.box[0];
    for (uint32_t root = blockIdx.x; root < nroots; root += gridDim.x) {
        for (uint32_t t = threadIdx.x; t < nleaf; t += 256)
            cells[pyr_offset(sub) + t] = pyr_load(a, node_meta, pyr_offset(r) + root * nleaf + t);
        __syncthreads();
        for (int k = sub - 1; k >= 0; --k) {
            const uint32_t cnt = 1u << (2 * k);
            for (uint32_t t = threadIdx.x; t < cnt; t += 256) {
                const uint32_t c = root * cnt + t;  // the cell's index within level r2 + k
                const PyrCell me = pyr_combine(a, &cells[pyr_offset(k + 1) + 4 * t], pyr_offset(r2 + k + 1) + 4 * c, r2 + k, width0);
                cells[pyr_offset(k) + t] = me;
                pyr_store(a, node_meta, pyr_offset(r2 + k) + c, me);
            }
            __syncthreads();
        }
    }
}

constexpr int COM_TOP_LEVELS = 5;  // levels 0..4: at most 341 cells

// The last launch of the dense-pyramid build, one block.  Normally: levels r - 1 .. 0 of the
// pyramid (in LDS), then the bookkeeping words.  When the keys of this step turned out deeper than
// the pyramid kernels launched for it cover (cut[3] == 0: the device decides, the host only knew
// the previous steps) the same block builds the whole level-by-level tree and its centres of mass
// instead: slow, complete, and rare -- the host learns the new depth from this very step.
__global__ __launch_bounds__(256) void k_pyr_top(ComArgs a, uint4 *node_meta, uint32_t *level_start,
                                                 unsigned long long *counters, const uint64_t *keys, Split<2> *split,
                                                 uint32_t *bsum, int n, uint32_t *levels_host, const uint32_t *chunk_info,
                                                 int mid_launched)
{
    __shared__ PyrCell cells[PYR_TOP_CELLS + PYR_TOP_STAGE];  // levels 0..min(r, 5), then the staging area
    if (a.cut[3] == 0u) {
        // (the pyramid sums the leaf pass's per-chunk parts itself; this build wants them finished)
        const int nchunks = (n + 63) / 64;
        for (int base = 0; base < nchunks; base += 256)
            leaf_finish_chunk<3, 4>(chunk_info, nchunks, const_cast<double *>(a.leaf_sum), base + (int)threadIdx.x);
        build_init(node_meta, level_start, n, counters);
        int levels = 0;
        for (int d = 0; d < BH_MAX_DEPTH - 2; ++d) {
            __threadfence_block();
            __syncthreads();  // level_start[d + 1] and the level's nodes are written
            if (level_start[d + 1] == level_start[d])
                break;  // uniform
            levels = d + 1;
            build_count<2>(keys, node_meta, split, level_start, bsum, d);
            __threadfence_block();
            __syncthreads();  // bsum[0] and the splits are written
            build_emit<2>(node_meta, split, level_start, bsum, d);
        }
        __threadfence_block();
        __syncthreads();
        if (threadIdx.x == 0) {
            counters[2] = level_start[levels];
            counters[3] = (unsigned long long)levels;
            if (levels_host)
                *levels_host = (uint32_t)levels;
        }
        for (int dd = levels - 1; dd >= 0; --dd) {
            com_level(a, dd);
            __threadfence_block();
            __syncthreads();
        }
        return;
    }
    const int L = (int)a.cut[4];
    const int r = pyr_top_from(L, mid_launched != 0);  // the deepest level already stored (by k_pyr_sub or k_pyr_mid)
    const double width0 = a.box[2] - a.box[0];
    int held = r;  // the level held in LDS from which the sweep goes on
    if (r > 5) {
        // level 6 does not fit: it passes through the staging area, 512 cells (128 parents) at a time
        PyrCell *stage = cells + PYR_TOP_CELLS;
        for (uint32_t base = 0; base < (1u << (2 * r)); base += PYR_TOP_STAGE) {
            for (uint32_t c = threadIdx.x; c < PYR_TOP_STAGE; c += 256)
                stage[c] = pyr_load(a, node_meta, pyr_offset(r) + base + c);
            __syncthreads();
            for (uint32_t j = threadIdx.x; j < PYR_TOP_STAGE / 4; j += 256) {
                const uint32_t c = base / 4 + j;  // the parent's index within level r - 1
                const PyrCell me = pyr_combine(a, &stage[4 * j], pyr_offset(r) + 4 * c, r - 1, width0);
                cells[pyr_offset(r - 1) + c] = me;
                pyr_store(a, node_meta, pyr_offset(r - 1) + c, me);
            }
            __syncthreads();
        }
        held = r - 1;
    } else {
        for (uint32_t c = threadIdx.x; c < (1u << (2 * r)); c += 256)
            cells[pyr_offset(r) + c] = pyr_load(a, node_meta, pyr_offset(r) + c);
        __syncthreads();
    }
    for (int l = held - 1; l >= 0; --l) {
        const uint32_t cnt = 1u << (2 * l);
        for (uint32_t c = threadIdx.x; c < cnt; c += 256) {
            const PyrCell me = pyr_combine(a, &cells[pyr_offset(l + 1) + 4 * c], pyr_offset(l + 1) + 4 * c, l, width0);
            cells[pyr_offset(l) + c] = me;
            pyr_store(a, node_meta, pyr_offset(l) + c, me);
        }
        __syncthreads();
    }
    // every slot of levels 0..L is a node slot; the levels of this tree are the cell levels
    if (threadIdx.x < BH_MAX_DEPTH + 2)
        level_start[threadIdx.x] = threadIdx.x == 0 ? 0u : (threadIdx.x == 1 ? 1u : pyr_offset(L + 1));
    if (threadIdx.x < 16)
        counters[threadIdx.x] = threadIdx.x == 2 ? (unsigned long long)pyr_offset(L + 1)
                                                 : (threadIdx.x == 3 ? (unsigned long long)(L + 1) : 0ull);
}

static ComArgs com_args(const BhBuffers &b, const BhStepArgs &a);

int bh_com(hipStream_t s, BhBuffers &b, const BhStepArgs &a)
{
    const ComArgs c = com_args(b, a);
    hipLaunchKernelGGL(k_leaf_sums, dim3((a.n + 255) / 256), dim3(256), 0, s, b.keys[b.sorted], b.idx[b.sorted], a.n, a.pos,
                       a.mass_radius, b.leaf_sum, b.chunk_info, b.cut, b.pc_range);
    if (b.use_pyr) {
        // Two launches build the tree, the centres of mass and the traversal records (k_pyr_top also
        // holds the one-block fallback for keys deeper than these launches cover).
        // one block per cell of the split level of the previous steps' keys (blocks go round if this
        // step's keys are deeper)
        const uint32_t hint = b.step_hint;
        const int deepest = (int)(hint <= (uint32_t)b.pyr_max ? hint : (uint32_t)b.pyr_max);
        const dim3 grid(1u << (2 * pyr_split(deepest)));
        hipLaunchKernelGGL(k_pyr_sub<false>, grid, dim3(256), 0, s, c, b.node_meta, b.pc_range, (const double *)nullptr);
        // (keys of PYR_MID_FROM levels that the hint did not announce still pass through k_pyr_top's staging area)
        const bool mid = b.pyr_levels > PYR_MAX_LEVEL || (hint >= (uint32_t)PYR_MID_FROM && hint <= (uint32_t)b.pyr_max);
        if (mid)
            hipLaunchKernelGGL(k_pyr_mid, dim3(1u << (2 * pyr_top_from(deepest))), dim3(256), 0, s, c, b.node_meta);
        hipLaunchKernelGGL(k_pyr_top, dim3(1), dim3(256), 0, s, c, b.node_meta, b.level_start, b.counters, b.keys[b.sorted],
                           (Split<2> *)b.split, b.bsum, a.n, b.levels_host, b.chunk_info, mid ? 1 : 0);
        NB_HIP(hipGetLastError());
        return NB_OK;
    }
    NB_TRY(leaf_finish(s, b.chunk_info, a.n, b.leaf_sum, 2));
    // levels beyond the ones the build launched per-level kernels for: one block, one launch
    if (b.launched < BH_MAX_DEPTH - 1)
        hipLaunchKernelGGL(k_com, dim3(1), dim3(256), 0, s, c, b.launched, BH_MAX_DEPTH - 1);
    // the top levels hold a few hundred nodes at most (4^d): one block does them in one launch
    const int top = b.launched < COM_TOP_LEVELS ? b.launched : COM_TOP_LEVELS;
    for (int d = b.launched - 1; d >= top; --d)
        hipLaunchKernelGGL(k_com, dim3(level_blocks(d, 2, 256)), dim3(256), 0, s, c, d, d + 1);
    if (top == 1)
        hipLaunchKernelGGL(k_com, dim3(1), dim3(256), 0, s, c, 0, 1);
    else if (top > 1)
        hipLaunchKernelGGL(k_com, dim3(1), dim3(256), 0, s, c, 0, top);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

static ComArgs com_args(const BhBuffers &b, const BhStepArgs &a)
{
    ComArgs c;
    c.node_meta = b.node_meta;
    c.level_start = b.level_start;
    c.idx = b.idx[b.sorted];
    c.pos = a.pos;
    c.mass_radius = a.mass_radius;
    c.box = b.box;
    c.node_com = b.node_com;
    c.node_rec = b.node_rec;
    c.node_aux = b.node_child;
    c.inv_len = 1.0 / a.units.len_unit;
    c.inv_mass = 1.0 / a.units.mass_unit;
    c.theta = a.theta;
    c.cut = b.cut;
    c.leaf_sum = b.leaf_sum;
    c.cth = a.collision_th;
    c.bucket = a.bucket_depth > 0 ? 1 : 0;
    return c;
}

// ------------------------------------------------------------------ cell exchange (multi-GPU)
//
// A step in three parts, with two reductions over the ranks in between (done by the caller: RCCL in
// nb_step_distributed, anything else through nb_cells_buffer).  `a` describes THIS RANK'S bodies
// only: n = their number, the per-body arrays offset to the first of them.
//   begin:   own extent -> b.red4 {-min x, max x, -min y, max y}              [max over the ranks]
//   middle:  box and levels from the reduced extent, keys / sort / leaf pass of the own bodies,
//            one record per cell of the keys' last level -> b.agg             [sum over the ranks]
//   end:     pyramid from the summed records, traversal + integration of the own bodies.
// cells_levels: digits of the keys (host-chosen, the same on every rank; the tree must close at or
// above that level, else *b.err_host is set and the results of the step are not the reference's).

size_t cells_agg_doubles(int cells_levels) { return (size_t)AGG_DOUBLES << (2 * cells_levels); }

int bh_cells_alloc(BhBuffers &b)
{
    if (b.agg)
        return NB_OK;
    if (!b.pc_range || !b.owned)
        return fail(NB_ERR_STATE, "cell exchange needs a handle's own Barnes-Hut buffers");
    NB_TRY(dev_alloc(b.agg, cells_agg_doubles(PYR_MAX_LEVEL)));
    NB_TRY(dev_alloc(b.red4, 4));
    NB_TRY(dev_alloc(b.err, 1));
    return NB_OK;
}

int cells_max_level() { return PYR_MAX_LEVEL; }

// Level at which a tree over a box of width w0 and height h0 closes (k_bbox_final's rule; 32: none).
int closed_level_for(double w0, double h0, double theta, double cth)
{
    int first_closed = 32;
    for (int l = BH_KEY_LEVELS; l >= 0; --l) {
        const double w = std::ldexp(w0, -l), h = std::ldexp(h0, -l);
        const double reach = cth * (1.0 - 1.0e-4) - std::sqrt(w * w + h * h);
        if (reach > 0.0 && reach * (1.0 - 1.0e-4) > w / theta)
            first_closed = l;
        else
            break;
    }
    return first_closed;
}

int bh_cells_begin(hipStream_t s, BhBuffers &b, const BhStepArgs &a)
{
    int blocks = (a.n + 255) / 256;
    if (blocks > BBOX_BLOCKS)
        blocks = BBOX_BLOCKS;
    if (blocks > 0)
        hipLaunchKernelGGL(k_bbox_partial, dim3(blocks), dim3(256), 0, s, a.pos, a.n, b.bbox_part);
    hipLaunchKernelGGL(k_bbox_pack, dim3(1), dim3(256), 0, s, b.bbox_part, blocks, b.red4);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// part 0: box and keys, 1: sort, 2: leaf pass and cell records (three calls, so that the caller can
// time them apart).
int bh_cells_middle(hipStream_t s, BhBuffers &b, const BhStepArgs &a, int cells_levels, int part)
{
    if (cells_levels < 0 || cells_levels > PYR_MAX_LEVEL)
        return fail(NB_ERR_ARG, "bh_cells_middle: bad level");
    if (part == 0) {
        b.use_pyr = true;
        b.pyr_levels = PYR_MAX_LEVEL;
        b.sorted = 0;
        hipLaunchKernelGGL(k_bbox_final, dim3(1), dim3(256), 0, s, b.red4, 1, a.bbox_margin, 0, a.theta, a.collision_th,
                           b.box, b.cut, b.pyr_levels, b.hint_host, b.hint_box, cells_levels, b.err_host, a.n, 0, box_ring_slot(b));
        NB_HIP(hipGetLastError());
        return a.n > 0 ? bh_keys(s, b, a) : (int)NB_OK;
    }
    if (part == 1)
        // the keys have exactly cells_levels digits (k_bbox_final): the passes with work are known here
        return a.n > 0 ? sort_pairs(s, b.keys, b.idx, a.n, b.cut, b.hist, b.hist_totals, sort_first_pass(2 * cells_levels))
                       : (int)NB_OK;
    const ComArgs c = com_args(b, a);
    if (a.n > 0) {
        hipLaunchKernelGGL(k_leaf_sums, dim3((a.n + 255) / 256), dim3(256), 0, s, b.keys[0], b.idx[0], a.n, a.pos,
                           a.mass_radius, b.leaf_sum, b.chunk_info, b.cut, b.pc_range);
    } else {
        // no own bodies: every cell slot reads as empty
        NB_HIP(hipMemsetAsync(b.pc_range, 0, sizeof(uint2) << (2 * cells_levels), s));
    }
    const uint32_t ncell = 1u << (2 * cells_levels);
    const uint32_t blocks = (ncell + 255u) / 256u;
    hipLaunchKernelGGL(k_agg_fill, dim3(blocks < 4096u ? blocks : 4096u), dim3(256), 0, s, c, b.pc_range, b.agg);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

// part 0: the tree from the reduced cell records, 1: traversal and integration.
int bh_cells_end(hipStream_t s, BhBuffers &b, const BhStepArgs &a, int cells_levels, int part)
{
    if (part == 1)
        return bh_force(s, b, a);
    const ComArgs c = com_args(b, a);
    const dim3 grid(1u << (2 * pyr_split(cells_levels)));
    hipLaunchKernelGGL(k_pyr_sub<true>, grid, dim3(256), 0, s, c, b.node_meta, b.pc_range, b.agg);
    hipLaunchKernelGGL(k_pyr_top, dim3(1), dim3(256), 0, s, c, b.node_meta, b.level_start, b.counters, b.keys[0],
                       (Split<2> *)b.split, b.bsum, a.n, b.levels_host, b.chunk_info, 0);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

}  // namespace nb
