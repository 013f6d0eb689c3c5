// nb_traverse.h -- the Barnes-Hut traversal + integration kernel, written once for the quadtree
// (nb_bh.hip) and the octree (nb_bh3.hip).  What the two dimensions do differently -- where a
// target's float coordinates come from, how a node's record and auxiliary words are laid out, how
// many siblings share a stack entry, how the FP64 state is stored -- is in the Traversal<DIM>
// traits each pipeline defines before it includes this file; the walk itself (stack, wave-level
// classification of the popped nodes, the visit loops, FP64 re-decisions, the fused integration) is
// below and exists once.
//
// Rules followed (reference: src/cpu/barnesHut.cpp:50-89 calculateForceHelper, :125-134 isCollide):
// a cell is accepted when width / dist < theta, an accepted cell is dropped inside 2 r_i + th of its
// centre of mass, a body is skipped inside r_i + r_j + th; every lane decides for its own body.
//
// Traversal<DIM> provides:
//   Args                         kernel argument struct (node_rec, node_aux, node_com, idx, tlist, tcount, dyn,
//                                radius, pos, vel, acc, counters, n, fc, ia, theta, collision_th + its own fields)
//   Aux                          type of a node's auxiliary words
//   GROUP, ENTRIES, HIGH, DEPTH  siblings per stack entry, entries popped per round, watermark, tree levels
//   WPB                          waves per block
//   load_target(a, k, body, x, ri)   float coordinates (a Pt<DIM>) and radius of the target at sorted position k
//   has(entry, sib)              whether sibling `sib` of a stack entry exists
//   wbits(rec, aux)              float bits of (width / theta)^2 of a cell; sign bit set for a body: -(r_j + th)
//   mass(rec), point(rec)        record fields
//   child(aux), prune(aux)       child word (first child << 4 | ...; low four bits zero: no children) and the
//                                float bits of the prune threshold (collision_th - cell diagonal)^2, or -1
//   stage / staged accessors     what goes into the second staged word and how it is read back
//   pos_ptr / load_acc / dest / carry / ...   FP64 state layout, and where a step leaves it
#pragma once

#include <type_traits>

#include "nb_common.h"
#include "nb_dev.h"

namespace nb {

// A float comparison whose two sides differ by less than this fraction is repeated in FP64
// on the FP64 positions.  The float distance of two bodies carries the rounding of their
// coordinates (up to 2^-24 of the domain each), which near the cutoff distance is a few
// 1e-6 of the distance; the band is several times that.
constexpr float TR_BAND = 4.0e-5f;

// Scale of the force of one accepted cell or body on the lane's target, m / r^3 with
// r^2 = d^2 + eps2.  FM_EXACT keeps the reference's second softening term in the denominator
// (r^3 + eps3); FM_NOEPS leaves eps out altogether: the host selects it when eps2 is below the
// float resolution of every distance that can contribute (all are beyond the cutoff).
enum { FM_NOEPS = 0, FM_EPS = 1, FM_EXACT = 2 };
template <int MODE>
__device__ __forceinline__ float force_scale(float d2, float nm, float eps2, float eps3)
{
    if (MODE == FM_EXACT) {
        const float r2 = d2 + eps2;
        const float r3 = r2 * __builtin_sqrtf(r2);
        return nm * __builtin_amdgcn_rcpf(r3 + eps3);
    }
    const float rinv = __builtin_amdgcn_rsqf(MODE == FM_EPS ? d2 + eps2 : d2);
    return ((rinv * rinv) * rinv) * nm;  // one dependent chain: as two products the compiler packs them, at a loss
}

// A point or vector of DIM float coordinates: x and y as one two-element vector, so that the
// packed FP32 instructions take them as they come out of a 16-byte record (an aligned register
// pair), z beside them.  (Left to itself the compiler pairs y with z in the 3-D loops and pays
// two register moves per visit to bring them together.)
typedef float v2f __attribute__((ext_vector_type(2)));
template <int DIM>
struct Pt {
    v2f xy;
    float z;  // unused in two dimensions
    __device__ __forceinline__ float operator[](int d) const { return d == 0 ? xy.x : (d == 1 ? xy.y : z); }
};
template <int DIM>
__device__ __forceinline__ Pt<DIM> pt_sub(const Pt<DIM> &a, const Pt<DIM> &b)
{
    Pt<DIM> r;
    r.xy = a.xy - b.xy;
    r.z = DIM == 3 ? a.z - b.z : 0.f;
    return r;
}
template <int DIM>
__device__ __forceinline__ float pt_norm2(const Pt<DIM> &v)
{
    float d2 = __builtin_fmaf(v.xy.y, v.xy.y, v.xy.x * v.xy.x);
    if (DIM == 3)
        d2 = __builtin_fmaf(v.z, v.z, d2);
    return d2;
}
template <int DIM>
__device__ __forceinline__ void pt_fma(Pt<DIM> &acc, const Pt<DIM> &v, float sc)
{
    acc.xy = __builtin_elementwise_fma(v.xy, (v2f)(sc), acc.xy);
    if (DIM == 3)
        acc.z = __builtin_fmaf(v.z, sc, acc.z);
}

__device__ __forceinline__ uint32_t mbcnt64(uint64_t m)
{
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
}

template <int DIM>
struct Traversal;

// A staged 16-byte record read whole.  Where a loop needs only three of its words the compiler narrows the read to a
// ds_read_b96, which the LDS serves in eight cycles per wave against four for a ds_read_b128 -- and the LDS array of
// a CU, shared by the visits of its four SIMDs, is as busy as the vector pipes in this kernel.
static __device__ __forceinline__ uint4 lds_read_whole(const uint4 *p)  // p: a record in LDS, nowhere else
{
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    // (volatile: the compiler may not narrow the read; the pointer is named as one into LDS again, which the cast
    // to a volatile type would otherwise forget)
    const u32x4 v = *(const volatile __attribute__((address_space(3))) u32x4 *)p;
    return make_uint4(v.x, v.y, v.z, v.w);
}

// The stack holds one entry per opened node: {first child slot << 4 | which of its child slots are
// in use, mask of the lanes that opened it}; the siblings share the lane mask.  A round pops up to
// ENTRIES entries and gives each a group of GROUP lanes: lane GROUP e + i takes child slot i of the
// e-th entry from the top (lanes whose slot is not in use sit the round out: no prefix sums).
// Every visited node pushes at most one entry.  Above HIGH entries the wave pops one entry per
// round: a depth-first descent then adds at most GROUP - 1 entries per tree level, on top of what
// the last full round pushed, so STACK - HIGH >= GROUP ENTRIES + (GROUP - 1) DEPTH rules out an
// overflow.
template <int DIM>
struct TraversalSizes {
    using T = Traversal<DIM>;
    static constexpr int BATCH = T::GROUP * T::ENTRIES;  // most nodes a round can hold
    static constexpr int STACK = T::HIGH + BATCH + (T::GROUP - 1) * T::DEPTH + 1;
    static_assert(T::ENTRIES >= 1 && T::GROUP * T::ENTRIES <= 64, "one group of lanes per popped entry");
};

// One wave per 64 consecutive targets of the sorted order, eight waves per SIMD.
template <int DIM, int MODE, bool STATS>
__device__ __forceinline__ void traverse_and_integrate(const typename Traversal<DIM>::Args &a)
{
    using T = Traversal<DIM>;
    using Aux = typename T::Aux;
    constexpr int WPB = T::WPB, GROUP = T::GROUP, ENTRIES = T::ENTRIES, HIGH = T::HIGH;
    constexpr int BATCH = TraversalSizes<DIM>::BATCH, STACK = TraversalSizes<DIM>::STACK;
    __shared__ uint32_t st_node[WPB][STACK];
    __shared__ uint64_t st_mask[WPB][STACK];
    // the batch being visited: per entry {record} and {lane mask, two more words of the node}
    __shared__ __attribute__((aligned(16))) uint4 st_batch[WPB][BATCH][2];
    __shared__ float st_extra[WPB][T::EXTRA ? BATCH : 1];
    const int lane = lane_id(), w = threadIdx.x >> 6;
    uint32_t *snode = st_node[w];
    uint64_t *smask = st_mask[w];
    uint4(*batch)[2] = st_batch[w];
    float *extra = st_extra[w];

    const int ntarget = a.tlist ? (int)*a.tcount : a.n;
    // neighbouring groups of bodies visit mostly the same nodes: give them to one XCD's L2
    const int wave = (int)xcd_chunked_block(blockIdx.x, gridDim.x) * WPB + w;
    const int slot = wave * 64 + lane;
    if (wave * 64 >= ntarget)
        return;  // whole wave idle (no block barriers below)
    const uint64_t t_begin = a.wave_cost ? wall_clock64() : 0ull;
    bool valid = slot < ntarget;
    const int last = valid ? slot : ntarget - 1;
    int k = last;
    if (a.tlist) {
        // a list of targets comes in groups of 64 entries that start with a target; the rest of a group may be padding
        // (the group's first entry is read from memory, not from lane 0: written as a select between a lane's entry
        // and a cross-lane read of the same entry, the compiler folds the read to the padding value the select's
        // condition implies for that lane)
        const uint32_t e = a.tlist[slot], e0 = a.tlist[wave * 64];
        valid = e != TL_PAD;
        k = (int)(valid ? e : e0);
    }
    const uint32_t body = a.idx[k];
    // (the FP64 position and radius of the lane's body are read again where the rare FP64
    // re-decisions and the epilogue need them: held in registers they cost a wave per SIMD in 3-D)
    Pt<DIM> xi;
    float ri;
    T::load_target(a, k, body, xi, ri);
    const bool dynamic = valid && a.dyn[body];
    // BarnesHut::isCollide(Body, Vector): 2 r_i + th > dist drops an accepted cell (barnesHut.cpp:131-134)
    const float thn = 2.f * ri + a.fc.cth;
    const float thn2 = thn * thn;
    const float thn2_band = thn2 * TR_BAND;
    // with a positive cutoff a body fails its own pair test (distance 0); only without one
    // must the zero distance be excluded explicitly
    const bool self_test = !(a.fc.cth > 0.f);
    const float eps2 = a.fc.eps2, eps3 = a.fc.eps3;
    // Bounding box of the wave's targets and their largest cell cutoff: a cell whose centre of
    // mass is farther from the box than that (with a margin far above float rounding) passes
    // the cutoff test for every lane and cannot be pruned.
    // The smallest cutoffs of the wave: a node closer than that to every point of the box lies
    // inside every lane's cutoff distance.
    float g0[DIM], g1[DIM], safe2 = thn2, in2 = thn2, ri_min = ri;
#pragma unroll
    for (int d = 0; d < DIM; ++d)
        g0[d] = g1[d] = xi[d];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            g0[d] = fminf(g0[d], __shfl_xor(g0[d], o));
            g1[d] = fmaxf(g1[d], __shfl_xor(g1[d], o));
        }
        safe2 = fmaxf(safe2, __shfl_xor(safe2, o));
        in2 = fminf(in2, __shfl_xor(in2, o));
        ri_min = fminf(ri_min, __shfl_xor(ri_min, o));
    }
    safe2 *= 1.001f;
    in2 *= 0.999f;

    double sum[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d)
        sum[d] = 0.0;
    unsigned long long n_eval = 0, n_acc = 0, n_recheck = 0, n_visit = 0, n_batch = 0;
    // visits by what the wave-level classification made of the node: opened by all, nothing for
    // anybody, force only (all lanes / under a mask), per-lane opening test, full tests, bodies,
    // per-lane cutoff test only
    unsigned long long n_kind[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    int max_top = 0;

    int top = 0;
    // every lane the wave computes a force for; a node visited for all of them needs no lane mask
    const uint64_t m0 = __builtin_amdgcn_ballot_w64(dynamic);
    if (m0 != 0ull) {
        if (lane == 0) {
            snode[0] = 1u;  // the root as a group of one
            smask[0] = m0;
        }
        top = 1;
    }
    while (top > 0) {
        if (STATS) {
            n_batch += 1;
            max_top = top > max_top ? top : max_top;
        }
        // Pop: the group of lanes e takes the e-th entry from the top, lane i of it sibling i.
        int npop = top < ENTRIES ? top : ENTRIES;
        if (top > HIGH)
            npop = 1;
        const int grp = lane / GROUP, sib = lane % GROUP;
        uint32_t u = 0;
        uint64_t um = 0ull;
        bool have = false;
        if (grp < npop) {
            const uint32_t ew = snode[top - 1 - grp];
            have = T::has(ew, sib);
            u = (ew >> 4) + (uint32_t)sib;
            um = smask[top - 1 - grp];
        }
        const uint64_t cmask = __builtin_amdgcn_ballot_w64(have);  // lanes holding a node
        const int base = top - npop;
        __builtin_amdgcn_wave_barrier();
        float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
        Aux aux = T::no_aux();
        if (have) {
            rec = a.node_rec[u];
            aux = a.node_aux[u];
        }
        top = base;
        // Each popped node is first judged for the wave as a whole, by the distance of its centre
        // of mass to the bounding box of the wave's targets (margins far above float rounding):
        //   opened by every lane   -> its children go onto the stack right here, no visit;
        //   accepted by every lane -> visited by a loop that only adds the force;
        //   far from every lane    -> visited without the cutoff and prune tests;
        // the remaining cells and the bodies take the full tests.  Only cells beyond the largest
        // cutoff distance of the wave take the first two shortcuts, so neither the cutoff nor
        // the pruning rule can apply to them.
        const uint32_t wb = T::wbits(rec, aux);
        const float nw = __uint_as_float(wb);
        float dmin2, dmax2;
        {
            const Pt<DIM> c = T::point(rec);
            const float e0 = fmaxf(fmaxf(g0[0] - c[0], c[0] - g1[0]), 0.f);
            const float f0 = fmaxf(c[0] - g0[0], g1[0] - c[0]);
            dmin2 = e0 * e0;
            dmax2 = f0 * f0;
#pragma unroll
            for (int d = 1; d < DIM; ++d) {
                const float e = fmaxf(fmaxf(g0[d] - c[d], c[d] - g1[d]), 0.f);
                const float f = fmaxf(c[d] - g0[d], g1[d] - c[d]);
                dmin2 = __builtin_fmaf(e, e, dmin2);
                dmax2 = __builtin_fmaf(f, f, dmax2);
            }
        }
        const uint64_t allbodym = __builtin_amdgcn_ballot_w64((int)wb < 0) & cmask;
        // Nodes no lane gets anything from, and that no lane opens (barnesHut.cpp:59,75: contributions
        // inside the cutoff distance are skipped): a body inside r_i + r_j + th of every lane; a cell
        // without children inside 2 r_i + th of every lane (a lane that accepts it drops it, a lane
        // that would open it finds nothing below); a cell that lies wholly inside th of every lane
        // (its prune threshold: accepted it is dropped, and nothing below it can contribute).
        // They are neither visited nor pushed.
        float none2;
        {
            const float tb = ri_min - nw;  // a body: r_i + r_j + th for the lane with the smallest radius
            const float inside = (T::child(aux) & 15u) != 0u ? __uint_as_float(T::prune(aux)) : in2;
            none2 = (int)wb < 0 ? tb * tb : inside;
        }
        const uint64_t nonem = __builtin_amdgcn_ballot_w64(dmax2 < none2 * 0.999f) & cmask;
        const uint64_t bodym = allbodym & ~nonem;
        const uint64_t farm_all = __builtin_amdgcn_ballot_w64(dmin2 > safe2) & cmask & ~allbodym;
        // Cells every lane opens.  Near the cutoff a lane may see such a cell wholly inside its
        // cutoff distance and not open it (the per-lane pruning); opened for that lane as well,
        // everything below contributes nothing to it, which is the same result.
        const uint64_t openm = __builtin_amdgcn_ballot_w64(dmax2 < nw * 0.999f) & cmask & ~(allbodym | nonem);
        const uint64_t sureall = __builtin_amdgcn_ballot_w64(dmin2 > nw * 1.001f) & farm_all;
        const uint64_t farcm = farm_all & ~(openm | sureall);
        // accepted by every lane AND visited for all of the wave's lanes: the plain force loop
        const uint64_t fullm = sureall & __builtin_amdgcn_ballot_w64(um == m0);
        const uint64_t surem = sureall & ~fullm;
        const uint64_t closem = cmask & ~(allbodym | farm_all | openm | nonem);  // cells near some lane's cutoff
        // Of those, the cells whose acceptance is settled for the whole wave -- accepted by every
        // lane, or without children: then width / theta < collision_th (the rule that closes the
        // tree at such a level; a bucket leaf has width 0), so a lane outside 2 r_i + th of the
        // centre of mass accepts the cell and a lane inside gets nothing either way -- need only
        // the per-lane cutoff test (barnesHut.cpp:75).
        const uint64_t dropm = closem & __builtin_amdgcn_ballot_w64((T::child(aux) & 15u) == 0u || dmin2 > nw * 1.001f);
        const uint64_t nearm = closem & ~dropm;
        if (STATS) {
            n_visit += (unsigned long long)__popcll(nonem);
            n_kind[0] += (unsigned long long)__popcll(openm);
            n_kind[1] += (unsigned long long)__popcll(nonem);
            n_kind[2] += (unsigned long long)__popcll(fullm);
            n_kind[3] += (unsigned long long)__popcll(surem);
            n_kind[4] += (unsigned long long)__popcll(farcm);
            n_kind[5] += (unsigned long long)__popcll(nearm);
            n_kind[7] += (unsigned long long)__popcll(dropm);
            n_kind[6] += (unsigned long long)__popcll(bodym);
        }
        if (openm != 0ull) {
            // one entry per opened node that has children
            const uint64_t pushm = openm & __builtin_amdgcn_ballot_w64((T::child(aux) & 15u) != 0u);
            if (__builtin_amdgcn_inverse_ballot_w64(pushm)) {
                const int at = top + (int)mbcnt64(pushm);
                snode[at] = T::child(aux);
                smask[at] = um;
            }
            top += (int)__popcll(pushm);
            if (STATS) {
                n_visit += (unsigned long long)__popcll(openm);
                unsigned long long e = __builtin_amdgcn_inverse_ballot_w64(openm) ? (unsigned long long)__popcll(um) : 0ull;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1)
                    e += __shfl_xor(e, o);
                n_eval += e;
            }
        }
        // Stage the rest in LDS, by kind: every lane then reads entry j with the same address (an
        // LDS broadcast), which keeps the per-node data off the vector ALU's lane-read path, and
        // each kind of node gets a loop without the other kinds' code.
        const int nfull = (int)__popcll(fullm), nsure = nfull + (int)__popcll(surem);
        const int nfar = nsure + (int)__popcll(farcm), ndrop = nfar + (int)__popcll(dropm);
        const int ncell = ndrop + (int)__popcll(nearm);
        const int cstaged = ncell + (int)__popcll(bodym);
        int slot_j = (int)mbcnt64(fullm);
        slot_j = __builtin_amdgcn_inverse_ballot_w64(surem) ? nfull + (int)mbcnt64(surem) : slot_j;
        slot_j = __builtin_amdgcn_inverse_ballot_w64(farcm) ? nsure + (int)mbcnt64(farcm) : slot_j;
        slot_j = __builtin_amdgcn_inverse_ballot_w64(dropm) ? nfar + (int)mbcnt64(dropm) : slot_j;
        slot_j = __builtin_amdgcn_inverse_ballot_w64(nearm) ? ndrop + (int)mbcnt64(nearm) : slot_j;
        slot_j = __builtin_amdgcn_inverse_ballot_w64(bodym) ? ncell + (int)mbcnt64(bodym) : slot_j;
        const bool staged = __builtin_amdgcn_inverse_ballot_w64(cmask & ~(openm | nonem));
        if (staged) {
            batch[slot_j][0] = make_uint4(__float_as_uint(rec.x), __float_as_uint(rec.y), __float_as_uint(rec.z),
                                          __float_as_uint(rec.w));
            batch[slot_j][1] = T::stage(um, aux);
            if (T::EXTRA)
                extra[slot_j] = T::stage_extra(aux);
        }
        __builtin_amdgcn_wave_barrier();
        Pt<DIM> f;
        f.xy = (v2f)(0.f);
        f.z = 0.f;
        // node index of batch entry j (only the rare FP64 re-decisions need it)
        auto node_of = [&](int j) {
            j = __builtin_amdgcn_readfirstlane(j);  // opaque: keeps a running copy of slot_j - j out of the loop
            const uint64_t at = __builtin_amdgcn_ballot_w64(staged && slot_j == j);
            return (uint32_t)__builtin_amdgcn_readlane((int)u, (int)__builtin_ctzll(at));
        };
        // FP64 distance of the lane's body to a node's centre of mass, as the reference computes it
        // (Vector::getDistance, vector.cpp:31-35)
        auto dist64 = [&](const double *nc) {
            const double *pd = T::pos_ptr(a, body);
            double q2 = 0.0;
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                const double q = nc[d] - pd[d];
                q2 += q * q;
            }
            return sqrt(q2);
        };
        auto lanes_of = [&](const uint4 q) {  // lanes a staged node is visited for
            const uint32_t mlo = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.x);
            const uint32_t mhi = (uint32_t)__builtin_amdgcn_readfirstlane((int)q.y);
            return ((uint64_t)mhi << 32) | mlo;
        };
        // Per-lane conditions are kept as 64-bit lane masks in scalar registers (the result of a
        // vector compare IS such a mask), so combining them costs scalar instructions only.
        // ---- cells: accepted when width / dist < theta (barnesHut.cpp:70-71), else opened ----
        auto visit_cell = [&](const int j, auto far_tag) {
            constexpr bool FAR = decltype(far_tag)::value;
            const uint4 r = lds_read_whole(&batch[j][0]), q = lds_read_whole(&batch[j][1]);
            const float4 rf = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            const float cw = T::staged_width(rf, q);
            const uint64_t mj = lanes_of(q);
            const Pt<DIM> dv = pt_sub(T::point(rf), xi);
            const float d2 = pt_norm2(dv);
            const float g = d2 - cw;
            uint64_t accm = __builtin_amdgcn_ballot_w64(g > 0.f);
            uint64_t ambm = __builtin_amdgcn_ballot_w64(__builtin_fabsf(g) <= cw * TR_BAND);
            // an accepted cell counts only outside 2 r_i + th of its centre of mass (barnesHut.cpp:75)
            uint64_t farm = ~0ull;
            if (!FAR) {
                const float g2 = d2 - thn2;
                farm = __builtin_amdgcn_ballot_w64(g2 >= 0.f);
                ambm |= __builtin_amdgcn_ballot_w64(__builtin_fabsf(g2) <= thn2_band);
            }
            if ((ambm & mj) != 0ull) {
                // too close to call in float: decide in FP64 as the reference does
                const double *nc = a.node_com + T::COM * (size_t)node_of(j);
                const double dist = dist64(nc);
                accm = __builtin_amdgcn_ballot_w64(nc[T::COM_W] / dist < a.theta);
                if (!FAR)
                    farm = __builtin_amdgcn_ballot_w64(!(a.radius[body] * 2 + a.collision_th > dist));
                if (STATS)
                    n_recheck += 1;
            }
            const uint64_t macc = mj & accm;
            const uint64_t okm = FAR ? macc : macc & farm;
            // Nothing below a cell that lies wholly inside the cutoff distance can contribute:
            // every body and every centre of mass in it is closer than collision_th.  A lane
            // that sees the cell that way does not open it.
            uint64_t om = mj ^ macc;
            if (!FAR)
                om &= ~__builtin_amdgcn_ballot_w64(d2 < T::staged_prune(q, extra, j));
            if (om != 0ull) {
                const uint32_t packed = (uint32_t)__builtin_amdgcn_readfirstlane((int)T::staged_child(q));
                if ((packed & 15u) != 0u) {
                    if (lane == 0) {
                        snode[top] = packed;
                        smask[top] = om;
                    }
                    top += 1;
                }
            }
            if (okm != 0ull) {
                asm volatile("" ::: "memory");  // keep the force arithmetic inside this branch
                // lanes outside okm get a zero scale (their own may be inf or NaN: d2 can be 0)
                const float sc = __builtin_amdgcn_inverse_ballot_w64(okm) ? force_scale<MODE>(d2, T::mass(rf), eps2, eps3) : 0.f;
                pt_fma(f, dv, sc);
            }
            if (STATS) {
                n_eval += (unsigned long long)__popcll(mj);
                n_acc += (unsigned long long)__popcll(okm);
                n_visit += 1;
            }
        };
        // ---- cells every lane accepts and that are visited for every lane the wave computes a
        // force for (6 of 7 such cells in 2-D at N = 5M): only the force, no lane mask.  A lane
        // outside m0 is a static or padding lane and sums values nobody reads.  Unrolled by 2 only:
        // the compiler's own choice (8) costs 16 registers, i.e. two of the eight waves per SIMD. ----
#pragma unroll 2
        for (int j = 0; j < nfull; ++j) {
            const uint4 r = lds_read_whole(&batch[j][0]);
            const float4 rf = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            const Pt<DIM> dv = pt_sub(T::point(rf), xi);
            const float d2 = pt_norm2(dv);
            pt_fma(f, dv, force_scale<MODE>(d2, T::mass(rf), eps2, eps3));
            if (STATS) {
                n_eval += (unsigned long long)__popcll(m0);
                n_acc += (unsigned long long)__popcll(m0);
                n_visit += 1;
            }
        }
        // ---- cells every lane accepts: only the force, for the lanes the cell is visited for ----
        for (int j = nfull; j < nsure; ++j) {
            const uint4 r = lds_read_whole(&batch[j][0]);
            const uint2 q2 = *(const uint2 *)&batch[j][1];
            const float4 rf = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            const uint64_t mj = lanes_of(make_uint4(q2.x, q2.y, 0u, 0u));
            const Pt<DIM> dv = pt_sub(T::point(rf), xi);
            const float d2 = pt_norm2(dv);
            const float sc = __builtin_amdgcn_inverse_ballot_w64(mj) ? force_scale<MODE>(d2, T::mass(rf), eps2, eps3) : 0.f;
            pt_fma(f, dv, sc);
            if (STATS) {
                n_eval += (unsigned long long)__popcll(mj);
                n_acc += (unsigned long long)__popcll(mj);
                n_visit += 1;
            }
        }
        for (int j = nsure; j < nfar; ++j)
            visit_cell(j, std::true_type{});
        // ---- cells whose acceptance is settled: counted outside 2 r_i + th of the centre of mass ----
        for (int j = nfar; j < ndrop; ++j) {
            const uint4 r = lds_read_whole(&batch[j][0]);
            const uint2 q2 = *(const uint2 *)&batch[j][1];
            const float4 rf = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            const uint64_t mj = lanes_of(make_uint4(q2.x, q2.y, 0u, 0u));
            const Pt<DIM> dv = pt_sub(T::point(rf), xi);
            const float d2 = pt_norm2(dv);
            const float g2 = d2 - thn2;
            uint64_t farm = __builtin_amdgcn_ballot_w64(g2 >= 0.f);
            if ((__builtin_amdgcn_ballot_w64(__builtin_fabsf(g2) <= thn2_band) & mj) != 0ull) {
                const double *nc = a.node_com + T::COM * (size_t)node_of(j);
                farm = __builtin_amdgcn_ballot_w64(!(a.radius[body] * 2 + a.collision_th > dist64(nc)));
                if (STATS)
                    n_recheck += 1;
            }
            const uint64_t okm = mj & farm;
            if (okm != 0ull) {
                asm volatile("" ::: "memory");
                const float sc = __builtin_amdgcn_inverse_ballot_w64(okm) ? force_scale<MODE>(d2, T::mass(rf), eps2, eps3) : 0.f;
                pt_fma(f, dv, sc);
            }
            if (STATS) {
                n_eval += (unsigned long long)__popcll(mj);
                n_acc += (unsigned long long)__popcll(okm);
                n_visit += 1;
            }
        }
        for (int j = ndrop; j < ncell; ++j)
            visit_cell(j, std::false_type{});
        // ---- bodies: skipped when r_i + r_j + th > dist (barnesHut.cpp:59,125-129) ----
        for (int j = ncell; j < cstaged; ++j) {
            const uint4 r = lds_read_whole(&batch[j][0]), q = lds_read_whole(&batch[j][1]);
            const float4 rf = make_float4(__uint_as_float(r.x), __uint_as_float(r.y), __uint_as_float(r.z), __uint_as_float(r.w));
            const float bw = T::staged_width(rf, q);  // -(r_j + th)
            const uint64_t mj = lanes_of(q);
            const Pt<DIM> dv = pt_sub(T::point(rf), xi);
            const float d2 = pt_norm2(dv);
            const float th = ri - bw;
            const float t2 = th * th;
            const float g = d2 - t2;
            uint64_t passm = __builtin_amdgcn_ballot_w64(g >= 0.f);
            const uint64_t ambm = __builtin_amdgcn_ballot_w64(__builtin_fabsf(g) <= t2 * TR_BAND) & mj;
            if (ambm != 0ull) {
                const double *nc = a.node_com + T::COM * (size_t)node_of(j);
                const double dist = dist64(nc);
                passm = __builtin_amdgcn_ballot_w64(!(a.radius[body] + (-nc[T::COM_W]) + a.collision_th > dist));
                if (STATS)
                    n_recheck += 1;
            }
            uint64_t okm = mj & passm;
            if (self_test)  // the body itself has d2 == 0 (the reference skips it by pointer)
                okm &= __builtin_amdgcn_ballot_w64(d2 > 0.f);
            if (okm != 0ull) {
                asm volatile("" ::: "memory");
                const float sc = __builtin_amdgcn_inverse_ballot_w64(okm) ? force_scale<MODE>(d2, T::mass(rf), eps2, eps3) : 0.f;
                pt_fma(f, dv, sc);
            }
            if (STATS) {
                n_eval += (unsigned long long)__popcll(mj);
                n_acc += (unsigned long long)__popcll(okm);
                n_visit += 1;
            }
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            sum[d] += (double)f[d];
    }

    if (STATS && lane == 0) {
        atomicAdd(&a.counters[0], n_eval);
        atomicAdd(&a.counters[1], n_acc);
        atomicAdd(&a.counters[4], n_recheck);
        atomicAdd(&a.counters[5], n_visit);
        atomicAdd(&a.counters[6], n_batch);
        atomicMax(&a.counters[7], (unsigned long long)max_top);
        for (int k = 0; k < 8; ++k)
            atomicAdd(&a.counters[8 + k], n_kind[k]);
    }
    if (a.wave_cost && lane == 0)
        a.wave_cost[wave] = (uint32_t)(wall_clock64() - t_begin);
    if (!valid)
        return;
    // Where the body's state goes: its own slot, or -- in a step that leaves the bodies in this
    // step's sorted order -- slot k of the other set of arrays (T::carry then moves what the step
    // does not compute).
    const uint32_t dst = T::dest(a, body, (uint32_t)k);
    double acc[DIM];
    if (dynamic) {
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            acc[d] = sum[d] * a.ia.acc_factor;
        T::store_acc(a, dst, acc);
    }
    // a static body keeps the acceleration it was given (calculateAcceleration, barnesHut.cpp:96-106)
    // and every body is advanced with its own velocity (barnesHut.cpp:108-123)
    const bool advance = a.ia.do_integrate && (dynamic || !a.ia.static_fixed);
    if (advance) {
        if (!dynamic)
            T::load_acc(a, body, acc);
        T::integrate(a, body, dst, acc);
    }
    T::carry(a, body, dst, dynamic, advance);
}

}  // namespace nb
