// nb_direct_impl.h -- the all-pairs force kernel with fused integration for gfx950, one source for
// two and three dimensions (internal; included by nb_direct.hip and nb_direct3.hip, which hold the
// traits of their dimension and the host entry points).
//
// Replaces DirectSumTiledKernel (src/gpu/DS/directSum_kernel.cu:61-106) with the semantics of the
// CPU DirectSum::update (src/cpu/directSum.cpp:24-79): accelerations from the OLD positions of
// every body, then v += a dt, x += v dt.  Everything a step reads (posm_cur, pos_cur) is left
// untouched by it; new positions go to posm_next / pos_next.  The reference is two-dimensional
// (src/cpu/vector.h holds x and y); three dimensions are its force law with dz added to the distance.
//
// Data layout: one float4 record per body in internal units (nb_common.h) -- 2-D {x, y, m,
// radius + collision_th}; 3-D {x, y, z, m} and the cutoff radius in an array of its own -- read
// coalesced (16 B per lane) and staged through LDS in tiles of T::TILE records; every lane then
// reads the same record per iteration (an LDS broadcast).  Each thread owns TPT targets, so one
// LDS read feeds TPT interactions.
//
// The reference's near-field cutoff (pairs closer than r_i + r_j + collision_th are skipped,
// directSum.cpp:69-72) costs a third of the pair arithmetic, yet only concerns bodies within
// ~1 % of the domain of each other.  Bodies are held in the order of a Hilbert curve, so a wave's targets and a
// run of T::SUB consecutive sources occupy small boxes; when the two boxes are farther apart than
// the largest possible cutoff, the run is processed by a loop without the test.  Runs whose bodies
// all have one mass (the planets of every reference scenario) also skip the per-pair multiply by
// the mass.  Runs that may hold a pair at the cutoff take the full loop, in which a float
// comparison too close to call is repeated in FP64.
//
// Rounding: FP32 partial sums are folded into FP64 accumulators every T::FLUSH pairs, so a
// dominant term (a sun) cannot swallow the low bits of the thousand terms added after it.
//
// Per interaction, far run: DIM sub, DIM fma, 1 rsq, 2-3 mul, DIM fma; near run: + add, fma, cmp,
// select and the tie test.  Bound: FP32 vector ALU (no MFMA: no dense contraction here).
//
// DirectTraits<DIM> gives: TPT (targets per thread, even), TILE, SUB, FLUSH, NEAR_UNROLL, SPLIT
// (sources per unit a slice of the source range is counted in), W_IN_RECORD (the record carries the
// cutoff radius), the type Lo of a record's low parts, and the accessors coords / mass / cutoff /
// lo_parts / with_coords / make_record / make_lo / pad_record / lo_of.
#pragma once

#include "nb_common.h"

namespace nb {

template <int DIM>
struct DirectTraits;

constexpr int DS_BLOCK = 256;
// A float cutoff comparison closer than this fraction is repeated in FP64: the float distance
// carries the coordinates' rounding (2^-24 of the domain each), a few 1e-6 of the cutoff.
constexpr float DS_BAND = 4.0e-5f;
// safety factor and absolute margin (internal units: coordinates lie in [-8, 8), their float
// rounding is up to 4.8e-7) on the box gap that lets a run skip the cutoff test
#ifndef NB_DS_GAP_SLACK
#define NB_DS_GAP_SLACK 1.0005f
#endif
constexpr float DS_GAP_SLACK = NB_DS_GAP_SLACK;
constexpr float DS_GAP_ABS = 4.0e-6f;
#ifndef NB_DS_MAXSPLIT
#define NB_DS_MAXSPLIT 64
#endif
constexpr int DS_MAX_SPLIT = NB_DS_MAXSPLIT;  // most slices the source range is cut into

// Bounding box of a run of T::SUB sources, its largest cutoff radius and its common mass.
struct RunBox {
    float lim[6];  // lim[2 d], lim[2 d + 1]: least and largest coordinate d (d < DIM; the rest unused)
    float wmax;    // max of radius + collision_th
    float mu;      // the mass every body of the run has, or NaN when they differ
};

template <int DIM>
struct DirectArgs {
    typedef typename DirectTraits<DIM>::Lo Lo;
    const float4 *posm_cur;
    float4 *posm_next;
    const Lo *posl_cur;  // low parts of the coordinates of posm_cur
    Lo *posl_next;
    const float *wcut;  // radius + collision_th per body where the record does not carry it
    const float *rad;   // radius per body
    const uint8_t *dyn;
    const double *pos_cur;  // DIM doubles per body: the FP64 positions the step starts from (never written by it)
    double *pos_next;       // FP64 positions after the step
    const double *radius;
    const RunBox *meta;
    RunBox *meta_out;  // where the step leaves the boxes of the records it writes (null: nowhere)
    double collision_th;
    double *vel;
    double *acc;
    double *part;  // jsplit x nt x DIM partial sums
    int n;
    int t0;
    int nt;
    int units_per_split;  // units of T::SPLIT sources per blockIdx.y
    int jsplit;
    ForceConsts fc;
    IntegrateArgs ia;
};

// DIM doubles of body i: 2-D state is held as double2 (one 16-byte access), 3-D as packed triples.
template <int DIM>
__device__ __forceinline__ void load_vec(const double *p, size_t i, double (&v)[DIM])
{
    if (DIM == 2) {
        const double2 t = ((const double2 *)p)[i];
        v[0] = t.x;
        v[1] = t.y;
    } else {
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            v[d] = p[DIM * i + d];
    }
}

template <int DIM>
__device__ __forceinline__ void store_vec(double *p, size_t i, const double (&v)[DIM])
{
    if (DIM == 2) {
        ((double2 *)p)[i] = make_double2(v[0], v[1]);
    } else {
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            p[DIM * i + d] = v[d];
    }
}

// Extrema of a set of records, gathered lane by lane, reduced over a wave and merged across waves.
template <int DIM>
struct BoxAcc {
    float lo[DIM], hi[DIM], wmax, m0, m1;

    __device__ __forceinline__ void init()
    {
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            lo[d] = __builtin_inff();
            hi[d] = -__builtin_inff();
        }
        wmax = 0.f;
        m0 = __builtin_inff();
        m1 = -__builtin_inff();
    }
    __device__ __forceinline__ void add(const float (&c)[DIM], float w, float m)
    {
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            lo[d] = fminf(lo[d], c[d]);
            hi[d] = fmaxf(hi[d], c[d]);
        }
        wmax = fmaxf(wmax, w);
        m0 = fminf(m0, m);
        m1 = fmaxf(m1, m);
    }
    __device__ __forceinline__ void wave_reduce()
    {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
                hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
            }
            wmax = fmaxf(wmax, __shfl_xor(wmax, o));
            m0 = fminf(m0, __shfl_xor(m0, o));
            m1 = fmaxf(m1, __shfl_xor(m1, o));
        }
    }
    __device__ __forceinline__ bool any() const { return lo[0] <= hi[0]; }
    __device__ __forceinline__ RunBox box() const
    {
        RunBox b;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            b.lim[2 * d] = d < DIM ? lo[d < DIM ? d : 0] : 0.f;
            b.lim[2 * d + 1] = d < DIM ? hi[d < DIM ? d : 0] : 0.f;
        }
        b.wmax = wmax;
        b.mu = (m0 == m1) ? m0 : __builtin_nanf("");
        return b;
    }
};

// One wave per run: extrema of the coordinates, max cutoff radius, common mass.
template <int DIM>
__global__ __launch_bounds__(256) void k_run_boxes(const float4 *posm, const float *wcut, int n, RunBox *meta)
{
    typedef DirectTraits<DIM> T;
    const int lane = threadIdx.x & 63;
    const int sub = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int base = sub * T::SUB;
    if (base >= n)
        return;
    BoxAcc<DIM> acc;
    acc.init();
#pragma unroll
    for (int q = 0; q < T::SUB / 64; ++q) {
        const int j = base + lane + 64 * q;
        if (j < n) {
            const float4 r = posm[j];
            float c[DIM];
            T::coords(r, c);
            acc.add(c, T::cutoff(r, wcut, j), T::mass(r));
        }
    }
    acc.wave_reduce();
    if (lane == 0)
        meta[sub] = acc.box();
}

// v += a dt_kick; x += v dt for one body, and its refreshed float record.
template <int DIM>
__device__ __forceinline__ float4 integrate_one(const DirectArgs<DIM> &a, int i, const double (&s)[DIM], float4 rec)
{
    typedef DirectTraits<DIM> T;
    double acc[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d)
        acc[d] = 0.0;
    if (a.dyn[i]) {
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            acc[d] = s[d] * a.ia.acc_factor;
    }
    store_vec<DIM>(a.acc, i, acc);
    if (a.ia.do_integrate) {
        double v[DIM], p[DIM];
        load_vec<DIM>(a.vel, i, v);
        load_vec<DIM>(a.pos_cur, i, p);
        float q[DIM], l[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            v[d] += acc[d] * a.ia.dt_kick;
            p[d] += v[d] * a.ia.dt;
            const double s64 = (p[d] - a.ia.origin[d]) * a.ia.inv_len_unit;
            q[d] = (float)s64;
            l[d] = (float)(s64 - (double)q[d]);
        }
        store_vec<DIM>(a.vel, i, v);
        store_vec<DIM>(a.pos_next, i, p);
        rec = T::with_coords(rec, q);
        a.posm_next[i] = rec;
        a.posl_next[i] = T::make_lo(l);
    }
    return rec;
}

// The boxes of the runs whose new records a 256-thread block has just written, one record per thread
// (thread t: body first + t, the block's first body a multiple of T::SUB; w: the record's cutoff
// radius): what k_run_boxes would compute from those records at the start of the next step.  All
// threads of the block call.
template <int DIM>
__device__ __forceinline__ void block_boxes_store(RunBox *meta_out, int first, float4 rec, float w, bool valid)
{
    typedef DirectTraits<DIM> T;
    static_assert(T::SUB % 64 == 0 && 256 % T::SUB == 0, "whole waves per run, whole runs per block");
    constexpr int WPR = T::SUB / 64;  // waves per run
    __shared__ BoxAcc<DIM> part[4];
    BoxAcc<DIM> acc;
    acc.init();
    if (valid) {
        float c[DIM];
        T::coords(rec, c);
        acc.add(c, w, T::mass(rec));
    }
    acc.wave_reduce();
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
        part[wv] = acc;
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && wv % WPR == 0) {
#pragma unroll
        for (int k = 1; k < WPR; ++k) {
            const BoxAcc<DIM> o = part[wv + k];
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                acc.lo[d] = fminf(acc.lo[d], o.lo[d]);
                acc.hi[d] = fmaxf(acc.hi[d], o.hi[d]);
            }
            acc.wmax = fmaxf(acc.wmax, o.wmax);
            acc.m0 = fminf(acc.m0, o.m0);
            acc.m1 = fmaxf(acc.m1, o.m1);
        }
        if (acc.any())  // the run holds a body
            meta_out[(first + 64 * wv) / T::SUB] = acc.box();
    }
}

// 1 / (r^2 + eps^2)^(3/2), or the reference's exact denominator r^3 + eps^3 (ForceConsts::exact_denom)
__device__ __forceinline__ float pair_scale(bool exact, float r2, float eps3)
{
    if (exact)
        return __builtin_amdgcn_rcpf(r2 * __builtin_sqrtf(r2) + eps3);
    const float rinv = __builtin_amdgcn_rsqf(r2);
    return rinv * (rinv * rinv);
}

// Adds the pairs of one target with one run that the main loop left out because their float
// cutoff comparison was too close to call: those are decided in FP64 on the FP64 positions, as
// the reference does (Vector::getDistance, vector.cpp:31-35; isCollide, directSum.cpp:69-72).
// Rare (a few pairs per million), so it runs as a separate loop after the hot one.
// (PRECISE as in run_pairs: the same distance must select the pairs here that the hot loop left out.)
template <int DIM, bool EXACT, bool PRECISE>
__device__ __forceinline__ void redo_ambiguous(const DirectArgs<DIM> &a, const float4 *rec, const float *wc,
                                               const typename DirectTraits<DIM>::Lo *lo, int jglobal, int jend,
                                               const float (&ci)[DIM], const float (&cl)[DIM], float ri, int ti,
                                               double (&out)[DIM])
{
    typedef DirectTraits<DIM> T;
    const float eps2 = a.fc.eps2, eps3 = a.fc.eps3;
    double pi[DIM], extra[DIM];
    load_vec<DIM>(a.pos_cur, ti, pi);
#pragma unroll
    for (int d = 0; d < DIM; ++d)
        extra[d] = 0.0;
    const double rdi = a.radius[ti];
#pragma unroll 1
    for (int jj = 0; jj < T::SUB; ++jj) {
        const float4 b = rec[jj];
        float bc[DIM], dc[DIM];
        T::coords(b, bc);
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            dc[d] = bc[d] - ci[d];
        if (PRECISE) {
            float l[DIM];
            T::lo_parts(lo[jj], l);
#pragma unroll
            for (int d = 0; d < DIM; ++d)
                dc[d] += l[d] - cl[d];
        }
        float r2 = eps2;
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            r2 = __builtin_fmaf(dc[d], dc[d], r2);
        const float th = ri + T::cutoff(b, wc, jj);
        const float th2 = __builtin_fmaf(th, th, eps2);
        const int j = jglobal + jj;
        if (__builtin_fabsf(r2 - th2) <= th2 * DS_BAND && j < jend && j != ti) {
            double pj[DIM];
            load_vec<DIM>(a.pos_cur, j, pj);
            double d2 = 0.0;
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                const double e = pj[d] - pi[d];
                d2 += e * e;
            }
            const double dist = sqrt(d2);
            if (!(rdi + a.radius[j] + a.collision_th >= dist)) {
                float s;
                if (EXACT)
                    s = T::mass(b) * pair_scale(true, r2, eps3);
                else {
                    const float rinv = __builtin_amdgcn_rsqf(r2);
                    s = (T::mass(b) * rinv) * (rinv * rinv);
                }
#pragma unroll
                for (int d = 0; d < DIM; ++d)
                    extra[d] += (double)(dc[d] * s);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < DIM; ++d)
        out[d] += extra[d];
}

typedef float v2f __attribute__((ext_vector_type(2)));

// A staged record read whole from LDS.  A loop that needs three of its four words gets a ds_read_b96 from the
// compiler, which the LDS serves in eight cycles per wave where a ds_read_b128 takes four; every wave of a block reads
// the same record at a time, and with four SIMDs issuing 26 cycles of arithmetic per record the eight-cycle read made
// the LDS the busier unit in three dimensions.  (volatile: the read may not be narrowed; the pointer is named as one
// into LDS again, which the cast would otherwise forget.)
static __device__ __forceinline__ float4 lds_record_whole(const float4 *p)  // p: a record in LDS, nowhere else
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const f32x4 v = *(const volatile __attribute__((address_space(3))) f32x4 *)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// All pairs of this thread's TPT targets with one run of T::SUB staged records, on packed FP32
// (v_pk_add/mul/fma_f32: two lanes of arithmetic per issue slot): targets are taken two at a time,
// so every operation but the reciprocal square root and the comparisons serves two pairs.
// CUTOFF: apply the skip test; UNIFORM: every record has mass mu.
// A run that may hold pairs near the cutoff (CUTOFF) takes its coordinates as float pairs hi + lo
// (lo, cl: what rounding the FP64 coordinate to the float record left over), so that the
// difference of two close bodies is exact to float precision of the DIFFERENCE, not of the
// coordinates: the pairs that weigh most are the closest ones.
template <int DIM, int TPT, bool EXACT, bool CUTOFF, bool UNIFORM, bool PRECISE = false>
__device__ __forceinline__ void run_pairs(const DirectArgs<DIM> &a, const float4 *rec, const float *wc,
                                          const typename DirectTraits<DIM>::Lo *lo, int jglobal, int jend, float mu,
                                          const float (&ci)[TPT][DIM], const float (&cl)[TPT][DIM], const float (&ri)[TPT],
                                          const int (&ti)[TPT], double (&acc)[TPT][DIM])
{
    typedef DirectTraits<DIM> T;
    typedef typename T::Lo Lo;
    static_assert(TPT % 2 == 0, "the pair loop takes targets two at a time");
    constexpr int P = TPT / 2;
    constexpr int UNROLL = CUTOFF ? T::NEAR_UNROLL : 16;  // records per unrolled body of the pair loop
    const float eps3 = a.fc.eps3;
    const v2f eps2 = {a.fc.eps2, a.fc.eps2};
    v2f c2[P][DIM], cl2[P][DIM], ri2[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            c2[p][d] = v2f{ci[2 * p][d], ci[2 * p + 1][d]};
            cl2[p][d] = v2f{cl[2 * p][d], cl[2 * p + 1][d]};
        }
        ri2[p] = v2f{ri[2 * p], ri[2 * p + 1]};
    }
    bool any_amb[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k)
        any_amb[k] = false;
#pragma unroll 1
    for (int c = 0; c < T::SUB; c += T::FLUSH) {
        v2f f[P][DIM];
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int d = 0; d < DIM; ++d)
                f[p][d] = v2f{0.f, 0.f};
        // the chunk's low parts: one coalesced load (lane L holds record c + L), handed out lane by lane
        static_assert(T::FLUSH <= 64, "one lane per record of a chunk");
        float lchunk[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            lchunk[d] = 0.f;
        if (CUTOFF && PRECISE) {
            const Lo lv = lo[c + ((int)threadIdx.x & 63 & (T::FLUSH - 1))];
            T::lo_parts(lv, lchunk);
        }
#pragma unroll UNROLL
        for (int jj = 0; jj < T::FLUSH; ++jj) {
            // two words (2-D, uniform mass) are a ds_read_b64; three would be a ds_read_b96: read all four then
            constexpr bool WHOLE = !(DIM == 2 && UNIFORM && !CUTOFF);
            const float4 b = WHOLE ? lds_record_whole(rec + c + jj) : rec[c + jj];
            float bc[DIM];
            T::coords(b, bc);
            const float m = T::mass(b);
            const v2f bm = {m, m};
            const float wj = CUTOFF ? T::cutoff(b, wc, c + jj) : 0.f;
            float l[DIM];
#pragma unroll
            for (int d = 0; d < DIM; ++d)
                l[d] = (CUTOFF && PRECISE) ? __shfl(lchunk[d], jj) : 0.f;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                v2f dc[DIM];
#pragma unroll
                for (int d = 0; d < DIM; ++d) {
                    dc[d] = v2f{bc[d], bc[d]} - c2[p][d];
                    if (CUTOFF && PRECISE)
                        dc[d] += v2f{l[d], l[d]} - cl2[p][d];
                }
                v2f r2 = eps2;
#pragma unroll
                for (int d = 0; d < DIM; ++d)
                    r2 = __builtin_elementwise_fma(dc[d], dc[d], r2);
                v2f s;
                if (EXACT) {
                    s = v2f{pair_scale(true, r2.x, eps3), pair_scale(true, r2.y, eps3)};
                    if (!UNIFORM)
                        s *= bm;
                } else {
                    const v2f rinv = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                    s = UNIFORM ? rinv * (rinv * rinv) : (bm * rinv) * (rinv * rinv);
                }
                if (CUTOFF) {
                    const v2f th = ri2[p] + v2f{wj, wj};
                    const v2f th2 = __builtin_elementwise_fma(th, th, eps2);  // d^2 > th^2 <=> d^2+eps^2 > th^2+eps^2
                    // DirectSum::isCollide skips r_i + r_j + th >= dist (directSum.cpp:69-72).  A
                    // comparison within the band is left out here and settled by redo_ambiguous.
                    const v2f q = r2 - th2, band = th2 * v2f{DS_BAND, DS_BAND};
                    any_amb[2 * p] = any_amb[2 * p] || __builtin_fabsf(q.x) <= band.x;
                    any_amb[2 * p + 1] = any_amb[2 * p + 1] || __builtin_fabsf(q.y) <= band.y;
                    s.x = q.x > band.x ? s.x : 0.f;
                    s.y = q.y > band.y ? s.y : 0.f;
                }
#pragma unroll
                for (int d = 0; d < DIM; ++d)
                    f[p][d] = __builtin_elementwise_fma(dc[d], s, f[p][d]);
            }
        }
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int d = 0; d < DIM; ++d) {
                acc[2 * p][d] += (double)(UNIFORM ? f[p][d].x * mu : f[p][d].x);
                acc[2 * p + 1][d] += (double)(UNIFORM ? f[p][d].y * mu : f[p][d].y);
            }
    }
    if (CUTOFF) {
#pragma unroll
        for (int k = 0; k < TPT; ++k)
            if (any_amb[k])
                redo_ambiguous<DIM, EXACT, PRECISE>(a, rec, wc, lo, jglobal, jend, ci[k], cl[k], ri[k], ti[k], acc[k]);
    }
}

__device__ __forceinline__ float uniform_f(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// Copies one tile of T::TILE records (a multiple of the tile size is always readable: the record
// arrays are padded with records that cannot interact) into LDS, and their cutoff radii where the
// records do not carry them.  global -> LDS without passing through registers: each wave
// instruction moves 64 x 16 B (or 64 x 4 B) to a wave-uniform LDS base + lane * 16 (or * 4).
template <int DIM>
__device__ __forceinline__ void stage_tile(const float4 *src, const float *wsrc, float4 *dst, float *wdst, int tid)
{
    typedef DirectTraits<DIM> T;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#pragma unroll
    for (int l = 0; l < T::TILE / DS_BLOCK; ++l) {
        const int first = (l * (DS_BLOCK / 64) + wave) * 64;
        __builtin_amdgcn_global_load_lds((const void *)(src + first + lane), (__attribute__((address_space(3))) void *)(dst + first),
                                         16, 0, 0);
        if (!T::W_IN_RECORD)
            __builtin_amdgcn_global_load_lds((const void *)(wsrc + first + lane),
                                             (__attribute__((address_space(3))) void *)(wdst + first), 4, 0, 0);
    }
}

#ifdef NB_DS_WAVES
#define NB_DS_OCC __attribute__((amdgpu_waves_per_eu(NB_DS_WAVES, 8)))
#else
#define NB_DS_OCC
#endif
template <int DIM, int TPT, bool EXACT, bool PRECISE>
__global__ __launch_bounds__(DS_BLOCK) NB_DS_OCC void k_direct(DirectArgs<DIM> a)
{
    typedef DirectTraits<DIM> T;
    __shared__ __attribute__((aligned(16))) float4 tile[2][T::TILE];
    __shared__ __attribute__((aligned(16))) float wtile[2][T::W_IN_RECORD ? 4 : T::TILE];
    const int tid = threadIdx.x;
    // a wave's targets are consecutive bodies: lane l takes tbase + 64 TPT wave + l, + 64, ...
    const int wbase = (tid >> 6) * (64 * TPT) + (tid & 63);
    const int tbase = a.t0 + blockIdx.x * (DS_BLOCK * TPT);
    const int tend = a.t0 + a.nt;

    const int jbeg = blockIdx.y * a.units_per_split * T::SPLIT;
    int jend = jbeg + a.units_per_split * T::SPLIT;
    if (jend > a.n)
        jend = a.n;
    const int ntile = (jend - jbeg + T::TILE - 1) / T::TILE;
    stage_tile<DIM>(a.posm_cur + jbeg, a.wcut + jbeg, tile[0], wtile[0], tid);

    float ci[TPT][DIM], cl[TPT][DIM], ri[TPT];
    int ti[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const int i = tbase + wbase + k * 64;
        // threads past the end work on a copy of the last target and discard the result
        ti[k] = i < tend ? i : tend - 1;
        T::coords(a.posm_cur[ti[k]], ci[k]);
        T::lo_parts(a.posl_cur[ti[k]], cl[k]);
        ri[k] = a.rad[ti[k]];
    }
    double acc[TPT][DIM];
#pragma unroll
    for (int k = 0; k < TPT; ++k)
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            acc[k][d] = 0.0;

    // Bounding box of the WAVE's targets (64 TPT consecutive bodies, a compact patch) and their
    // largest radius: every wave decides by itself which runs are far.  (Per block, four times
    // the bodies, the boxes are twice as wide and nearly twice the runs count as near: 9.1 %
    // against 4.9 % of all block-run pairs for the spiral at N = 65 536.)
    float lo[DIM], hi[DIM], rimax = ri[0];
#pragma unroll
    for (int d = 0; d < DIM; ++d)
        lo[d] = hi[d] = ci[0][d];
#pragma unroll
    for (int k = 1; k < TPT; ++k) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            lo[d] = fminf(lo[d], ci[k][d]);
            hi[d] = fmaxf(hi[d], ci[k][d]);
        }
        rimax = fmaxf(rimax, ri[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int d = 0; d < DIM; ++d) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
        }
        rimax = fmaxf(rimax, __shfl_xor(rimax, o));
    }
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        lo[d] = uniform_f(lo[d]);
        hi[d] = uniform_f(hi[d]);
    }
    rimax = uniform_f(rimax);
    __syncthreads();  // completes the staging of tile 0

    for (int t = 0; t < ntile; ++t) {
        if (t + 1 < ntile)
            stage_tile<DIM>(a.posm_cur + jbeg + (t + 1) * T::TILE, a.wcut + jbeg + (t + 1) * T::TILE, tile[(t + 1) & 1],
                            wtile[(t + 1) & 1], tid);
        const float4 *cur = tile[t & 1];
        const float *wcur = wtile[t & 1];
        const int jt = jbeg + t * T::TILE;
#pragma unroll 1
        for (int sub = 0; sub < T::TILE / T::SUB; ++sub) {
            const int j0 = jt + sub * T::SUB;
            if (j0 >= jend)
                break;
            const RunBox m = a.meta[j0 / T::SUB];
            // The boxes are those of the float records; the FP64 positions the reference decides on
            // differ from them by up to 2^-21 units per coordinate, so the gap needs an absolute
            // margin beside the relative one (with a tiny cutoff the relative one alone is smaller
            // than that rounding).
            float g[DIM];
#pragma unroll
            for (int d = 0; d < DIM; ++d)
                g[d] = fmaxf(fmaxf(m.lim[2 * d] - hi[d], lo[d] - m.lim[2 * d + 1]), 0.f);
            float g2 = g[DIM - 1] * g[DIM - 1];
#pragma unroll
            for (int d = DIM - 2; d >= 0; --d)
                g2 = __builtin_fmaf(g[d], g[d], g2);
            const float lim = (rimax + m.wmax) * DS_GAP_SLACK + DS_GAP_ABS;
            // far: no pair of the two boxes can be inside the cutoff (a NaN anywhere means near)
            const bool far = g2 > lim * lim;
            const bool uni = m.mu == m.mu;
            const float4 *rec = cur + sub * T::SUB;
            const float *wc = wcur + (T::W_IN_RECORD ? 0 : sub * T::SUB);
            if (far) {
                if (uni)
                    run_pairs<DIM, TPT, EXACT, false, true>(a, rec, wc, nullptr, j0, jend, m.mu, ci, cl, ri, ti, acc);
                else
                    run_pairs<DIM, TPT, EXACT, false, false>(a, rec, wc, nullptr, j0, jend, 0.f, ci, cl, ri, ti, acc);
            } else {
                // the low parts come straight from memory: every lane reads the same address
                run_pairs<DIM, TPT, EXACT, true, false, PRECISE>(a, rec, wc, a.posl_cur + j0, j0, jend, 0.f, ci, cl, ri, ti, acc);
            }
        }
        // the next tile has landed (the barrier waits for this wave's outstanding copies), and
        // every wave is done reading the current one
        __syncthreads();
    }

#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const int i = tbase + wbase + k * 64;
        if (i >= tend)
            continue;
        if (a.jsplit == 1)
            integrate_one<DIM>(a, i, acc[k], a.posm_cur[i]);
        else
            store_vec<DIM>(a.part, (size_t)blockIdx.y * a.nt + (i - a.t0), acc[k]);
    }
}

// Sums the jsplit partial accelerations of each target in split order, then integrates.
template <int DIM>
__global__ __launch_bounds__(256) void k_direct_finish(DirectArgs<DIM> a)
{
    typedef DirectTraits<DIM> T;
    const int first = a.t0 + blockIdx.x * 256;
    const int i = first + (int)threadIdx.x;
    const bool valid = i < a.t0 + a.nt;
    float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
        double s[DIM];
#pragma unroll
        for (int d = 0; d < DIM; ++d)
            s[d] = 0.0;
        // eight partial sums in flight at a time (a wave of this kernel is one of four on its CU: one
        // load per lane and round trip took 20 us at N = 65 536), added in slice order as before
        constexpr int AHEAD = 8;
        for (int q = 0; q < a.jsplit; q += AHEAD) {
            double p[AHEAD][DIM];
#pragma unroll
            for (int u = 0; u < AHEAD; ++u) {
                const int qq = q + u < a.jsplit ? q + u : a.jsplit - 1;
                load_vec<DIM>(a.part, (size_t)qq * a.nt + (i - a.t0), p[u]);
            }
#pragma unroll
            for (int u = 0; u < AHEAD; ++u)
                if (q + u < a.jsplit) {
#pragma unroll
                    for (int d = 0; d < DIM; ++d)
                        s[d] += p[u][d];
                }
        }
        rec = integrate_one<DIM>(a, i, s, a.posm_cur[i]);
    }
    if (a.meta_out)  // (uniform: every thread of the block gets there)
        block_boxes_store<DIM>(a.meta_out, first, rec, valid ? T::cutoff(rec, a.wcut, i) : 0.f, valid);
}

// Records, low parts and (rad != null) the per-body constants of bodies [i0, i1) from the FP64 state.
template <int DIM>
__global__ __launch_bounds__(256) void k_pack_records(const double *pos, const double *mass, const double *radius, int i0, int i1,
                                                      double ox, double oy, double oz, double inv_len, double inv_mass, double cth,
                                                      float4 *posm, typename DirectTraits<DIM>::Lo *posl, float *wcut, float *rad)
{
    typedef DirectTraits<DIM> T;
    const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i1)
        return;
    const double o[3] = {ox, oy, oz};
    double p[DIM];
    load_vec<DIM>(pos, i, p);
    float h[DIM], l[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) {
        const double q = (p[d] - o[d]) * inv_len;
        h[d] = (float)q;
        l[d] = (float)(q - (double)h[d]);
    }
    const float w = (T::W_IN_RECORD || rad) ? (float)((radius[i] + cth) * inv_len) : 0.f;
    posm[i] = T::make_record(h, (float)(mass[i] * inv_mass), w);
    posl[i] = T::make_lo(l);
    if (rad) {
        rad[i] = (float)(radius[i] * inv_len);
        if (!T::W_IN_RECORD)
            wcut[i] = w;
    }
}

// Records [n, cap) of a record array: zero mass, far away, infinite cutoff -- they cannot interact
// in either pair loop, so whole tiles can be staged without a bounds test.
template <int DIM>
__global__ __launch_bounds__(256) void k_pad_records(float4 *posm, typename DirectTraits<DIM>::Lo *posl, float *wcut, int n, int cap)
{
    typedef DirectTraits<DIM> T;
    const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap) {
        const float zero[DIM] = {};
        posm[i] = T::pad_record();
        posl[i] = T::make_lo(zero);  // (finite: the near loop adds it to a difference it then multiplies by zero)
        if (!T::W_IN_RECORD)
            wcut[i] = __builtin_inff();
    }
}

// ---- host side shared by the two dimensions ----

template <int DIM>
size_t ds_record_capacity(int n)
{
    typedef DirectTraits<DIM> T;
    return ((size_t)n + T::TILE - 1) / T::TILE * T::TILE + T::TILE;
}

template <int DIM>
size_t ds_meta_bytes(int n)
{
    return sizeof(RunBox) * (size_t)((n + DirectTraits<DIM>::SUB - 1) / DirectTraits<DIM>::SUB);
}

// The kernel is bound by vector-ALU issue, so the time of a launch is set by the busiest CU.
// Splitting the source range across blockIdx.y makes blocks many and short: ~65k blocks keep the
// 256 CUs within ~1 % of each other (measured: 1 split 4.4, 16 splits 5.0 T interactions/s at
// N = 1M), and give small N enough waves to fill the chip at all.
template <int DIM>
int ds_choose_jsplit(int n, int nt)
{
    typedef DirectTraits<DIM> T;
    if (nt <= 0)
        return 1;  // a rank without targets launches nothing
    const int tb = (nt + DS_BLOCK * T::TPT - 1) / (DS_BLOCK * T::TPT);
    const int nunits = (n + T::SPLIT - 1) / T::SPLIT;  // a slice is a whole number of units
    int want = (65536 + tb - 1) / tb;
    if (want > DS_MAX_SPLIT)
        want = DS_MAX_SPLIT;
    // partial sums: DIM doubles per target and slice; stay below 1 GiB
    while (want > 1 && (size_t)want * (size_t)nt * sizeof(double) * DIM > ((size_t)1 << 30))
        --want;
    if (want > nunits)
        want = nunits;
    return want < 1 ? 1 : want;
}

// Launches one step for targets [a.t0, a.t0 + a.nt): a.units_per_split, a.jsplit, a.meta and
// a.meta_out are filled in here.  meta / meta_next: ds_meta_bytes(n) each, the boxes of the source
// runs of posm_cur / posm_next; *meta_valid: on entry whether meta holds the boxes of posm_cur
// (computed first otherwise), on return whether the step left in meta_next the boxes of posm_next (a
// step that advances all bodies does) or, for a force evaluation, whether meta still holds those of
// posm_cur.
template <int DIM>
int ds_launch(hipStream_t s, DirectArgs<DIM> a, int jsplit, void *meta, void *meta_next, bool *meta_valid, const char *who)
{
    typedef DirectTraits<DIM> T;
    if (a.nt <= 0)
        return NB_OK;
    if (a.t0 < 0 || a.t0 + a.nt > a.n || jsplit < 1)
        return fail(NB_ERR_ARG, std::string(who) + ": bad target range");
    const int nunits = (a.n + T::SPLIT - 1) / T::SPLIT;
    a.units_per_split = (nunits + jsplit - 1) / jsplit;
    a.jsplit = (nunits + a.units_per_split - 1) / a.units_per_split;  // no empty slices
    if (a.jsplit > 1 && !a.part)
        return fail(NB_ERR_ARG, std::string(who) + ": split sources need a partial buffer");
    // The boxes of the source runs: computed here unless the step before left them (a step that
    // advances every body writes, with the new records, the boxes they will have as sources).
    const int nsub = (a.n + T::SUB - 1) / T::SUB;
    a.meta = (const RunBox *)meta;
    if (!*meta_valid)
        hipLaunchKernelGGL(k_run_boxes<DIM>, dim3((nsub + 3) / 4), dim3(256), 0, s, a.posm_cur, a.wcut, a.n, (RunBox *)meta);
    // (the finishing kernel does that; a step without one -- a few hundred bodies -- computes them as before)
#ifdef NB_DS_NO_FUSED_META  // (tuning aid: the boxes always come from k_run_boxes)
    const bool all_advance = false;
#else
    const bool all_advance = a.ia.do_integrate && a.t0 == 0 && a.nt == a.n && a.jsplit > 1 && meta_next;
#endif
    // (into the other buffer: blocks of this launch still read the present boxes)
    a.meta_out = all_advance ? (RunBox *)meta_next : nullptr;
    // records unchanged (a force evaluation): the boxes stay; only some records rewritten here (a
    // partitioned run: the rest arrives from other ranks): computed again next time
    *meta_valid = all_advance || !a.ia.do_integrate;
    const dim3 grid((a.nt + DS_BLOCK * T::TPT - 1) / (DS_BLOCK * T::TPT), a.jsplit);
    if (a.fc.exact_denom && a.fc.precise_near)
        hipLaunchKernelGGL((k_direct<DIM, T::TPT, true, true>), grid, dim3(DS_BLOCK), 0, s, a);
    else if (a.fc.exact_denom)
        hipLaunchKernelGGL((k_direct<DIM, T::TPT, true, false>), grid, dim3(DS_BLOCK), 0, s, a);
    else if (a.fc.precise_near)
        hipLaunchKernelGGL((k_direct<DIM, T::TPT, false, true>), grid, dim3(DS_BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL((k_direct<DIM, T::TPT, false, false>), grid, dim3(DS_BLOCK), 0, s, a);
    NB_HIP(hipGetLastError());
    if (a.jsplit > 1) {
        hipLaunchKernelGGL(k_direct_finish<DIM>, dim3((a.nt + 255) / 256), dim3(256), 0, s, a);
        NB_HIP(hipGetLastError());
    }
    return NB_OK;
}

// Records of bodies [i0, i1) into posm (and their low parts behind the records); constants: also
// the per-body radii (rad, and wcut where the record does not carry the cutoff radius).
template <int DIM>
int ds_pack(hipStream_t s, const double *pos, const double *mass, const double *radius, int n, int i0, int i1, const Units &units,
            double collision_th, float4 *posm, float *wcut, float *rad)
{
    typedef DirectTraits<DIM> T;
    if (i1 <= i0)
        return NB_OK;
    hipLaunchKernelGGL(k_pack_records<DIM>, dim3((i1 - i0 + 255) / 256), dim3(256), 0, s, pos, mass, radius, i0, i1,
                       units.origin[0], units.origin[1], DIM > 2 ? units.origin[2] : 0.0, 1.0 / units.len_unit,
                       1.0 / units.mass_unit, collision_th, posm, T::lo_of(posm, ds_record_capacity<DIM>(n)), wcut, rad);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

template <int DIM>
int ds_pad(hipStream_t s, float4 *posm, float *wcut, int n)
{
    const int cap = (int)ds_record_capacity<DIM>(n);
    hipLaunchKernelGGL(k_pad_records<DIM>, dim3((cap - n + 255) / 256), dim3(256), 0, s, posm,
                       DirectTraits<DIM>::lo_of(posm, (size_t)cap), wcut, n, cap);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

}  // namespace nb
