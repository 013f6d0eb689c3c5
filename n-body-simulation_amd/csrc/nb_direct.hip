// nb_direct.hip -- all-pairs force kernel with fused integration for gfx950.
//
// Replaces DirectSumTiledKernel (src/gpu/DS/directSum_kernel.cu:61-106) with the semantics
// of the CPU DirectSum::update (src/cpu/directSum.cpp:24-79): accelerations from the OLD
// positions of every body, then v += a dt, x += v dt.  Everything a step reads (posm_cur,
// pos_cur) is left untouched by it; new positions go to posm_next / pos_next.
//
// Data layout: one float4 record {x, y, m, radius + collision_th} per body, in internal
// units (nb_common.h), read coalesced (16 B per lane) and staged through LDS in tiles of
// DS_TILE records; every lane then reads the same record per iteration (an LDS broadcast).
// Each thread owns TPT targets, so one LDS read feeds TPT interactions.
//
// The reference's near-field cutoff (pairs closer than r_i + r_j + collision_th are skipped,
// directSum.cpp:69-72) costs a third of the pair arithmetic, yet only concerns bodies within
// ~1 % of the domain of each other.  Bodies are held in Morton order, so a wave's targets
// and a run of DS_SUB consecutive sources occupy small boxes; when the two boxes are farther
// apart than the largest possible cutoff, the run is processed by a loop without the test.
// Runs whose bodies all have one mass (the planets of every reference scenario) also skip
// the per-pair multiply by the mass.  Runs that may hold a pair at the cutoff take the full
// loop, in which a float comparison too close to call is repeated in FP64.
//
// Rounding: FP32 partial sums are folded into FP64 accumulators every DS_FLUSH pairs, so a
// dominant term (a sun) cannot swallow the low bits of the thousand terms added after it.
//
// Per interaction (2-D), far run: 2 sub, 2 fma, 1 rsq, 2-3 mul, 2 fma; near run: + add, fma,
// cmp, select and the tie test.  Bound: FP32 vector ALU (no MFMA: no dense contraction here).
#include "nb_common.h"

namespace nb {

#ifndef NB_DS_TPT
#define NB_DS_TPT 2  // targets per thread
#endif
#ifndef NB_DS_UNROLL
#define NB_DS_UNROLL 16  // records per unrolled body of the pair loop
#endif
#ifndef NB_DS_DMA
#define NB_DS_DMA 1  // 1: stage tiles with global_load_lds (no staging registers), 0: through registers
#endif

#ifndef NB_DS_PACKED
#define NB_DS_PACKED 1  // 1: cutoff-free runs on packed FP32 arithmetic
#endif

constexpr int DS_BLOCK = 256;
#ifndef NB_DS_TILE
#define NB_DS_TILE 512
#endif
constexpr int DS_TILE = NB_DS_TILE;  // records per LDS tile (8 KiB), two buffers
// (256 -> 128: the boxes are tighter and fewer pairs of a wave and a run count as near, 4.71 -> 4.84
// T interactions/s at N = 65 536, 5.62 -> 5.74 at 262 144, 5.89 -> 5.95 at 1 M; 64: 4.87 / 5.64 /
// 5.82, the per-run scalar work begins to show; 32: 4.64 / 4.98 / 4.98)
#ifndef NB_DS_SUB
#define NB_DS_SUB 128
#endif
constexpr int DS_SUB = NB_DS_SUB;  // records per run with its own bounding box
#ifndef NB_DS_FLUSH
#define NB_DS_FLUSH 32
#endif
constexpr int DS_FLUSH = NB_DS_FLUSH;   // pairs between FP32 -> FP64 folds
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

// Bounding box of a run of DS_SUB sources, its largest cutoff radius and its common mass.
struct SubMeta {
    float x0, x1, y0, y1;
    float wmax;  // max of radius + collision_th
    float mu;    // the mass every body of the run has, or NaN when they differ
    float pad0, pad1;
};

struct DirectLaunch {
    const float4 *posm_cur;
    float4 *posm_next;
    const float2 *posl_cur;  // low parts of the coordinates of posm_cur
    float2 *posl_next;
    const float *rad;
    const uint8_t *dyn;
    const double2 *pos_cur;  // FP64 positions the step starts from (never written by the step)
    double2 *pos_next;       // FP64 positions after the step
    const double *radius;
    const SubMeta *meta;
    SubMeta *meta_out;       // where the step leaves the boxes of the records it writes (null: nowhere)
    double collision_th;
    double2 *vel;
    double2 *acc;
    double2 *part;
    int n;
    int t0;
    int nt;
    int subs_per_split;  // runs of DS_SUB sources per blockIdx.y
    int jsplit;
    ForceConsts fc;
    IntegrateArgs ia;
};

// One wave per run: min/max of x and y, max cutoff radius, common mass.
__global__ __launch_bounds__(256) void k_sub_meta(const float4 *posm, int n, SubMeta *meta)
{
    const int lane = threadIdx.x & 63;
    const int sub = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int base = sub * DS_SUB;
    if (base >= n)
        return;
    float x0 = __builtin_inff(), x1 = -__builtin_inff(), y0 = __builtin_inff(), y1 = -__builtin_inff();
    float wmax = 0.f, m0 = __builtin_inff(), m1 = -__builtin_inff();
#pragma unroll
    for (int q = 0; q < DS_SUB / 64; ++q) {
        const int j = base + lane + 64 * q;
        if (j < n) {
            const float4 r = posm[j];
            x0 = fminf(x0, r.x);
            x1 = fmaxf(x1, r.x);
            y0 = fminf(y0, r.y);
            y1 = fmaxf(y1, r.y);
            wmax = fmaxf(wmax, r.w);
            m0 = fminf(m0, r.z);
            m1 = fmaxf(m1, r.z);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o));
        x1 = fmaxf(x1, __shfl_xor(x1, o));
        y0 = fminf(y0, __shfl_xor(y0, o));
        y1 = fmaxf(y1, __shfl_xor(y1, o));
        wmax = fmaxf(wmax, __shfl_xor(wmax, o));
        m0 = fminf(m0, __shfl_xor(m0, o));
        m1 = fmaxf(m1, __shfl_xor(m1, o));
    }
    if (lane == 0) {
        SubMeta m;
        m.x0 = x0;
        m.x1 = x1;
        m.y0 = y0;
        m.y1 = y1;
        m.wmax = wmax;
        m.mu = (m0 == m1) ? m0 : __builtin_nanf("");
        m.pad0 = m.pad1 = 0.f;
        meta[sub] = m;
    }
}

// v += a dt; x += v dt for one body, and its refreshed float record.
__device__ __forceinline__ float4 integrate_one(const DirectLaunch &a, int i, double sx, double sy, float4 rec)
{
    double ax = 0.0, ay = 0.0;
    if (a.dyn[i]) {
        ax = sx * a.ia.acc_factor;
        ay = sy * a.ia.acc_factor;
    }
    a.acc[i] = make_double2(ax, ay);
    if (a.ia.do_integrate) {
        double2 v = a.vel[i];
        double2 p = a.pos_cur[i];
        v.x += ax * a.ia.dt_kick;
        v.y += ay * a.ia.dt_kick;
        p.x += v.x * a.ia.dt;
        p.y += v.y * a.ia.dt;
        a.vel[i] = v;
        a.pos_next[i] = p;
        const double qx = (p.x - a.ia.origin[0]) * a.ia.inv_len_unit, qy = (p.y - a.ia.origin[1]) * a.ia.inv_len_unit;
        rec.x = (float)qx;
        rec.y = (float)qy;
        a.posm_next[i] = rec;
        a.posl_next[i] = make_float2((float)(qx - (double)rec.x), (float)(qy - (double)rec.y));
    }
    return rec;
}

// The boxes of the runs whose new records a 256-thread block has just written, one record per thread
// (thread t: body first + t, the block's first body a multiple of DS_SUB): what k_sub_meta would
// compute from those records at the start of the next step.  All threads of the block call.
__device__ __forceinline__ void block_meta_store(SubMeta *meta_out, int first, float4 rec, bool valid)
{
    static_assert(DS_SUB % 64 == 0 && 256 % DS_SUB == 0, "whole waves per run, whole runs per block");
    constexpr int WPR = DS_SUB / 64;  // waves per run
    __shared__ float part[4][7];
    float v[7] = {__builtin_inff(), -__builtin_inff(), __builtin_inff(), -__builtin_inff(), 0.f, __builtin_inff(), -__builtin_inff()};
    if (valid) {
        v[0] = v[1] = rec.x;
        v[2] = v[3] = rec.y;
        v[4] = rec.w;
        v[5] = v[6] = rec.z;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        v[0] = fminf(v[0], __shfl_xor(v[0], o));
        v[1] = fmaxf(v[1], __shfl_xor(v[1], o));
        v[2] = fminf(v[2], __shfl_xor(v[2], o));
        v[3] = fmaxf(v[3], __shfl_xor(v[3], o));
        v[4] = fmaxf(v[4], __shfl_xor(v[4], o));
        v[5] = fminf(v[5], __shfl_xor(v[5], o));
        v[6] = fmaxf(v[6], __shfl_xor(v[6], o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 7; ++q)
            part[w][q] = v[q];
    __syncthreads();
    if ((threadIdx.x & 63) == 0 && w % WPR == 0) {
#pragma unroll
        for (int k = 1; k < WPR; ++k) {
            v[0] = fminf(v[0], part[w + k][0]);
            v[1] = fmaxf(v[1], part[w + k][1]);
            v[2] = fminf(v[2], part[w + k][2]);
            v[3] = fmaxf(v[3], part[w + k][3]);
            v[4] = fmaxf(v[4], part[w + k][4]);
            v[5] = fminf(v[5], part[w + k][5]);
            v[6] = fmaxf(v[6], part[w + k][6]);
        }
        if (v[0] <= v[1]) {  // the run holds a body
            SubMeta m;
            m.x0 = v[0];
            m.x1 = v[1];
            m.y0 = v[2];
            m.y1 = v[3];
            m.wmax = v[4];
            m.mu = (v[5] == v[6]) ? v[5] : __builtin_nanf("");
            m.pad0 = m.pad1 = 0.f;
            meta_out[(first + 64 * w) / DS_SUB] = m;
        }
    }
}

// Adds the pairs of one target with one run that the main loop left out because their float
// cutoff comparison was too close to call: those are decided in FP64 on the FP64 positions, as
// the reference does (Vector::getDistance, vector.cpp:31-35; isCollide, directSum.cpp:69-72).
// Rare (a few pairs per million), so it runs as a separate loop after the hot one.
// (PRECISE as in run_pairs: the same distance must select the pairs here that the hot loop left out.)
template <bool EXACT, bool PRECISE>
__device__ __forceinline__ double2 redo_ambiguous(const DirectLaunch &a, const float4 *rec, const float2 *lo, int jglobal,
                                                  int jend, float xi, float yi, float xl, float yl, float ri, int ti)
{
    double ox = 0.0, oy = 0.0;
    const float eps2 = a.fc.eps2, eps3 = a.fc.eps3;
    const double2 pi = a.pos_cur[ti];
    const double rdi = a.radius[ti];
#pragma unroll 1
    for (int jj = 0; jj < DS_SUB; ++jj) {
        const float4 b = rec[jj];
        float dx = b.x - xi, dy = b.y - yi;
        if (PRECISE) {
            const float2 l = lo[jj];
            dx += l.x - xl;
            dy += l.y - yl;
        }
        float r2 = __builtin_fmaf(dx, dx, eps2);
        r2 = __builtin_fmaf(dy, dy, r2);
        const float th = ri + b.w;
        const float th2 = __builtin_fmaf(th, th, eps2);
        const int j = jglobal + jj;
        if (__builtin_fabsf(r2 - th2) <= th2 * DS_BAND && j < jend && j != ti) {
            const double2 pj = a.pos_cur[j];
            const double ex = pj.x - pi.x, ey = pj.y - pi.y;
            const double dist = sqrt(ex * ex + ey * ey);
            if (!(rdi + a.radius[j] + a.collision_th >= dist)) {
                float s;
                if (EXACT)
                    s = b.z * __builtin_amdgcn_rcpf(r2 * __builtin_sqrtf(r2) + eps3);
                else {
                    const float rinv = __builtin_amdgcn_rsqf(r2);
                    s = (b.z * rinv) * (rinv * rinv);
                }
                ox += (double)(dx * s);
                oy += (double)(dy * s);
            }
        }
    }
    return make_double2(ox, oy);
}

// All pairs of this thread's TPT targets with one run of DS_SUB staged records.
// CUTOFF: apply the skip test; UNIFORM: every record has mass mu.
// A run that may hold pairs near the cutoff (CUTOFF) takes its coordinates as float pairs hi + lo
// (lo, xl, yl: what rounding the FP64 coordinate to the float record left over), so that the
// difference of two close bodies is exact to float precision of the DIFFERENCE, not of the
// coordinates: the pairs that weigh most are the closest ones.
template <int TPT, bool EXACT, bool CUTOFF, bool UNIFORM, bool PRECISE = false>
__device__ __forceinline__ void run_pairs(const DirectLaunch &a, const float4 *rec, const float2 *lo, int jglobal, int jend,
                                          float mu, const float (&xi)[TPT], const float (&yi)[TPT], const float (&xl)[TPT],
                                          const float (&yl)[TPT], const float (&ri)[TPT], const int (&ti)[TPT],
                                          double (&sx)[TPT], double (&sy)[TPT])
{
    const float eps2 = a.fc.eps2, eps3 = a.fc.eps3;
    bool any_amb[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k)
        any_amb[k] = false;
#pragma unroll 1
    for (int c = 0; c < DS_SUB; c += DS_FLUSH) {
        float fx[TPT], fy[TPT];
#pragma unroll
        for (int k = 0; k < TPT; ++k)
            fx[k] = fy[k] = 0.f;
        // the chunk's low parts: one coalesced load (lane L holds record c + L), handed out lane by lane
        static_assert(DS_FLUSH <= 64, "one lane per record of a chunk");
        float2 lchunk = make_float2(0.f, 0.f);
        if (CUTOFF && PRECISE)
            lchunk = lo[c + ((int)threadIdx.x & 63 & (DS_FLUSH - 1))];
#pragma unroll NB_DS_UNROLL
        for (int jj = 0; jj < DS_FLUSH; ++jj) {
            const float4 b = rec[c + jj];
            float2 l = make_float2(0.f, 0.f);
            if (CUTOFF && PRECISE)
                l = make_float2(__shfl(lchunk.x, jj), __shfl(lchunk.y, jj));
#pragma unroll
            for (int k = 0; k < TPT; ++k) {
                float dx = b.x - xi[k];
                float dy = b.y - yi[k];
                if (CUTOFF && PRECISE) {
                    dx += l.x - xl[k];
                    dy += l.y - yl[k];
                }
                float r2 = __builtin_fmaf(dx, dx, eps2);
                r2 = __builtin_fmaf(dy, dy, r2);
                float s;
                if (EXACT) {
                    const float r3 = r2 * __builtin_sqrtf(r2);
                    s = __builtin_amdgcn_rcpf(r3 + eps3);
                    if (!UNIFORM)
                        s *= b.z;
                } else {
                    const float rinv = __builtin_amdgcn_rsqf(r2);
                    s = UNIFORM ? rinv * (rinv * rinv) : (b.z * rinv) * (rinv * rinv);
                }
                if (CUTOFF) {
                    const float th = ri[k] + b.w;
                    const float th2 = __builtin_fmaf(th, th, eps2);  // d^2 > th^2 <=> d^2+eps^2 > th^2+eps^2
                    // DirectSum::isCollide skips r_i + r_j + th >= dist (directSum.cpp:69-72).  A
                    // comparison within the band is left out here and settled by redo_ambiguous.
                    const float q = r2 - th2, band = th2 * DS_BAND;
                    any_amb[k] = any_amb[k] || __builtin_fabsf(q) <= band;
                    s = q > band ? s : 0.f;
                }
                fx[k] = __builtin_fmaf(dx, s, fx[k]);
                fy[k] = __builtin_fmaf(dy, s, fy[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            sx[k] += (double)(UNIFORM ? fx[k] * mu : fx[k]);
            sy[k] += (double)(UNIFORM ? fy[k] * mu : fy[k]);
        }
    }
    if (CUTOFF) {
#pragma unroll
        for (int k = 0; k < TPT; ++k)
            if (any_amb[k]) {
                const double2 extra = redo_ambiguous<EXACT, PRECISE>(a, rec, lo, jglobal, jend, xi[k], yi[k], xl[k], yl[k], ri[k], ti[k]);
                sx[k] += extra.x;
                sy[k] += extra.y;
            }
    }
}

typedef float v2f __attribute__((ext_vector_type(2)));

// The cutoff-free loop on packed FP32 (v_pk_add/mul/fma_f32: two lanes of arithmetic per issue
// slot): targets are taken two at a time, so every operation but the reciprocal square root
// serves two pairs.  Same operations and rounding per pair as run_pairs<.., false, ..>.
template <int TPT, bool UNIFORM>
__device__ __forceinline__ void run_pairs_far(const DirectLaunch &a, const float4 *rec, float mu,
                                              const float (&xi)[TPT], const float (&yi)[TPT], double (&sx)[TPT],
                                              double (&sy)[TPT])
{
    static_assert(TPT % 2 == 0, "packed loop takes targets in pairs");
    constexpr int P = TPT / 2;
    const v2f eps2 = {a.fc.eps2, a.fc.eps2};
    v2f x2[P], y2[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        x2[p] = v2f{xi[2 * p], xi[2 * p + 1]};
        y2[p] = v2f{yi[2 * p], yi[2 * p + 1]};
    }
#pragma unroll 1
    for (int c = 0; c < DS_SUB; c += DS_FLUSH) {
        v2f fx[P], fy[P];
#pragma unroll
        for (int p = 0; p < P; ++p)
            fx[p] = fy[p] = v2f{0.f, 0.f};
#pragma unroll NB_DS_UNROLL
        for (int jj = 0; jj < DS_FLUSH; ++jj) {
            const float4 b = rec[c + jj];
            const v2f bx = {b.x, b.x}, by = {b.y, b.y}, bm = {b.z, b.z};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const v2f dx = bx - x2[p];
                const v2f dy = by - y2[p];
                v2f r2 = __builtin_elementwise_fma(dx, dx, eps2);
                r2 = __builtin_elementwise_fma(dy, dy, r2);
                const v2f rinv = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                const v2f s = UNIFORM ? rinv * (rinv * rinv) : (bm * rinv) * (rinv * rinv);
                fx[p] = __builtin_elementwise_fma(dx, s, fx[p]);
                fy[p] = __builtin_elementwise_fma(dy, s, fy[p]);
            }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            sx[2 * p] += (double)(UNIFORM ? fx[p].x * mu : fx[p].x);
            sx[2 * p + 1] += (double)(UNIFORM ? fx[p].y * mu : fx[p].y);
            sy[2 * p] += (double)(UNIFORM ? fy[p].x * mu : fy[p].x);
            sy[2 * p + 1] += (double)(UNIFORM ? fy[p].y * mu : fy[p].y);
        }
    }
}

__device__ __forceinline__ float uniform_f(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// Copies one tile of DS_TILE records (a multiple of the tile size is always readable: the
// record arrays are padded with records that cannot interact) into LDS.
__device__ __forceinline__ void stage_tile(const float4 *src, float4 *dst, int tid)
{
#if NB_DS_DMA
    // global -> LDS without passing through registers: each wave instruction moves 64 x 16 B
    // to a wave-uniform LDS base + lane * 16
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#pragma unroll
    for (int l = 0; l < DS_TILE / DS_BLOCK; ++l) {
        const int first = (l * (DS_BLOCK / 64) + wave) * 64;
        __builtin_amdgcn_global_load_lds((const void *)(src + first + lane),
                                         (__attribute__((address_space(3))) void *)(dst + first), 16, 0, 0);
    }
#else
    float4 stage[DS_TILE / DS_BLOCK];
#pragma unroll
    for (int l = 0; l < DS_TILE / DS_BLOCK; ++l)
        stage[l] = src[tid + l * DS_BLOCK];
#pragma unroll
    for (int l = 0; l < DS_TILE / DS_BLOCK; ++l)
        dst[tid + l * DS_BLOCK] = stage[l];
#endif
}

#ifdef NB_DS_WAVES
#define NB_DS_OCC __attribute__((amdgpu_waves_per_eu(NB_DS_WAVES, 8)))
#else
#define NB_DS_OCC
#endif
template <int TPT, bool EXACT, bool PRECISE>
__global__ __launch_bounds__(DS_BLOCK) NB_DS_OCC void k_direct(DirectLaunch a)
{
    __shared__ __attribute__((aligned(16))) float4 tile[2][DS_TILE];
    const int tid = threadIdx.x;
    // a wave's targets are consecutive bodies: lane l takes tbase + 64 TPT wave + l, + 64, ...
    const int wbase = (tid >> 6) * (64 * TPT) + (tid & 63);
    const int tbase = a.t0 + blockIdx.x * (DS_BLOCK * TPT);
    const int tend = a.t0 + a.nt;

    const int jbeg = blockIdx.y * a.subs_per_split * DS_SUB;
    int jend = jbeg + a.subs_per_split * DS_SUB;
    if (jend > a.n)
        jend = a.n;
    const int ntile = (jend - jbeg + DS_TILE - 1) / DS_TILE;
    stage_tile(a.posm_cur + jbeg, tile[0], tid);

    float xi[TPT], yi[TPT], xl[TPT], yl[TPT], ri[TPT];
    int ti[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const int i = tbase + wbase + k * 64;
        // threads past the end work on a copy of the last target and discard the result
        ti[k] = i < tend ? i : tend - 1;
        const float4 self = a.posm_cur[ti[k]];
        const float2 selfl = a.posl_cur[ti[k]];
        xi[k] = self.x;
        yi[k] = self.y;
        xl[k] = selfl.x;
        yl[k] = selfl.y;
        ri[k] = a.rad[ti[k]];
    }
    double sx[TPT], sy[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k)
        sx[k] = sy[k] = 0.0;

    // Bounding box of the WAVE's targets (64 TPT consecutive bodies, a compact patch) and their
    // largest radius: every wave decides by itself which runs are far.  (Per block, four times
    // the bodies, the boxes are twice as wide and nearly twice the runs count as near: 9.1 %
    // against 4.9 % of all block-run pairs for the spiral at N = 65 536.)
    float bx0 = xi[0], bx1 = xi[0], by0 = yi[0], by1 = yi[0], rimax = ri[0];
#pragma unroll
    for (int k = 1; k < TPT; ++k) {
        bx0 = fminf(bx0, xi[k]);
        bx1 = fmaxf(bx1, xi[k]);
        by0 = fminf(by0, yi[k]);
        by1 = fmaxf(by1, yi[k]);
        rimax = fmaxf(rimax, ri[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        bx0 = fminf(bx0, __shfl_xor(bx0, o));
        bx1 = fmaxf(bx1, __shfl_xor(bx1, o));
        by0 = fminf(by0, __shfl_xor(by0, o));
        by1 = fmaxf(by1, __shfl_xor(by1, o));
        rimax = fmaxf(rimax, __shfl_xor(rimax, o));
    }
    bx0 = uniform_f(bx0);
    bx1 = uniform_f(bx1);
    by0 = uniform_f(by0);
    by1 = uniform_f(by1);
    rimax = uniform_f(rimax);
    __syncthreads();  // completes the staging of tile 0

    for (int t = 0; t < ntile; ++t) {
        if (t + 1 < ntile)
            stage_tile(a.posm_cur + jbeg + (t + 1) * DS_TILE, tile[(t + 1) & 1], tid);
        const float4 *cur = tile[t & 1];
        const int jt = jbeg + t * DS_TILE;
#pragma unroll 1
        for (int sub = 0; sub < DS_TILE / DS_SUB; ++sub) {
            const int j0 = jt + sub * DS_SUB;
            if (j0 >= jend)
                break;
            const SubMeta m = a.meta[j0 / DS_SUB];
            const float gx = fmaxf(fmaxf(m.x0 - bx1, bx0 - m.x1), 0.f);
            const float gy = fmaxf(fmaxf(m.y0 - by1, by0 - m.y1), 0.f);
            // The boxes are those of the float records; the FP64 positions the reference decides on
            // differ from them by up to 2^-21 units per coordinate, so the gap needs an absolute
            // margin beside the relative one (with a tiny cutoff the relative one alone is smaller
            // than that rounding).
            const float lim = (rimax + m.wmax) * DS_GAP_SLACK + DS_GAP_ABS;
            // far: no pair of the two boxes can be inside the cutoff (a NaN anywhere means near)
            const bool far = __builtin_fmaf(gx, gx, gy * gy) > lim * lim;
            const bool uni = m.mu == m.mu;
            const float4 *rec = cur + sub * DS_SUB;
            if (far) {
                if (NB_DS_PACKED && !EXACT && TPT % 2 == 0) {
                    if (uni)
                        run_pairs_far<TPT, true>(a, rec, m.mu, xi, yi, sx, sy);
                    else
                        run_pairs_far<TPT, false>(a, rec, 0.f, xi, yi, sx, sy);
                } else if (uni)
                    run_pairs<TPT, EXACT, false, true>(a, rec, nullptr, j0, jend, m.mu, xi, yi, xl, yl, ri, ti, sx, sy);
                else
                    run_pairs<TPT, EXACT, false, false>(a, rec, nullptr, j0, jend, 0.f, xi, yi, xl, yl, ri, ti, sx, sy);
            } else {
                // the low parts come straight from memory: every lane reads the same address
                run_pairs<TPT, EXACT, true, false, PRECISE>(a, rec, a.posl_cur + j0, j0, jend, 0.f, xi, yi, xl, yl, ri, ti, sx, sy);
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
            integrate_one(a, i, sx[k], sy[k], a.posm_cur[i]);
        else
            a.part[(size_t)blockIdx.y * a.nt + (i - a.t0)] = make_double2(sx[k], sy[k]);
    }
}

// Sums the jsplit partial accelerations of each target in split order, then integrates.
__global__ __launch_bounds__(256) void k_direct_finish(DirectLaunch a)
{
    const int first = a.t0 + blockIdx.x * 256;
    const int i = first + (int)threadIdx.x;
    const bool valid = i < a.t0 + a.nt;
    float4 rec = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) {
        double sx = 0.0, sy = 0.0;
        for (int s = 0; s < a.jsplit; ++s) {
            const double2 p = a.part[(size_t)s * a.nt + (i - a.t0)];
            sx += p.x;
            sy += p.y;
        }
        rec = integrate_one(a, i, sx, sy, a.posm_cur[i]);
    }
    if (a.meta_out)  // (uniform: every thread of the block gets there)
        block_meta_store(a.meta_out, first, rec, valid);
}

__global__ __launch_bounds__(256) void k_pack_posm(const double2 *pos, const double *mass, const double *radius, int i0,
                                                   int i1, double ox, double oy, double inv_len, double inv_mass,
                                                   double cth, float4 *posm, float2 *posl, float *rad)
{
    const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i1)
        return;
    const double2 p = pos[i];
    const double qx = (p.x - ox) * inv_len, qy = (p.y - oy) * inv_len;
    const float hx = (float)qx, hy = (float)qy;
    posm[i] = make_float4(hx, hy, (float)(mass[i] * inv_mass), (float)((radius[i] + cth) * inv_len));
    posl[i] = make_float2((float)(qx - (double)hx), (float)(qy - (double)hy));
    rad[i] = (float)(radius[i] * inv_len);
}

// Records [n, cap) of a record array: zero mass, far away, infinite cutoff -- they cannot
// interact in either pair loop, so whole tiles can be staged without a bounds test.
__global__ __launch_bounds__(256) void k_pad_posm(float4 *posm, int n, int cap)
{
    const int i = n + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cap)
        posm[i] = make_float4(1.0e15f, 1.0e15f, 0.f, __builtin_inff());
}

size_t direct_record_capacity(int n) { return ((size_t)n + DS_TILE - 1) / DS_TILE * DS_TILE + DS_TILE; }
size_t direct_record_floats4(int n) { return direct_record_capacity(n) + direct_record_capacity(n) / 2; }

static float2 *direct_lo(const float4 *posm, int n) { return (float2 *)(posm + direct_record_capacity(n)); }

int pad_posm(hipStream_t s, float4 *posm, int n)
{
    const int cap = (int)direct_record_capacity(n);
    hipLaunchKernelGGL(k_pad_posm, dim3((cap - n + 255) / 256), dim3(256), 0, s, posm, n, cap);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

int pack_posm(hipStream_t s, const double2 *pos, const double *mass, const double *radius, int n, int i0, int i1,
              const Units &units, double collision_th, float4 *posm, float *rad)
{
    if (i1 <= i0)
        return NB_OK;
    const int blocks = (i1 - i0 + 255) / 256;
    hipLaunchKernelGGL(k_pack_posm, dim3(blocks), dim3(256), 0, s, pos, mass, radius, i0, i1, units.origin[0],
                       units.origin[1], 1.0 / units.len_unit, 1.0 / units.mass_unit, collision_th, posm, direct_lo(posm, n), rad);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

constexpr int DS_TPT = NB_DS_TPT;
#ifndef NB_DS_MAXSPLIT
#define NB_DS_MAXSPLIT 64
#endif
constexpr int DS_MAX_SPLIT = NB_DS_MAXSPLIT;  // most slices the source range is cut into

size_t direct_meta_bytes(int n) { return sizeof(SubMeta) * (size_t)((n + DS_SUB - 1) / DS_SUB); }

int direct_choose_jsplit(int n, int nt)
{
    // The kernel is bound by vector-ALU issue, so the time of a launch is set by the busiest CU.
    // Splitting the source range across blockIdx.y makes blocks many and short: ~65k blocks
    // keep the 256 CUs within ~1 % of each other (measured: 1 split 4.4, 16 splits 5.0
    // T interactions/s at N = 1M), and give small N enough waves to fill the chip at all.
    if (nt <= 0)
        return 1;  // a rank without targets launches nothing
    const int tb = (nt + DS_BLOCK * DS_TPT - 1) / (DS_BLOCK * DS_TPT);
    const int ntiles = (n + DS_SUB - 1) / DS_SUB;  // a split is a whole number of runs
    int want = (65536 + tb - 1) / tb;
    if (want > DS_MAX_SPLIT)
        want = DS_MAX_SPLIT;
    // partial sums: 16 B per target and split; stay below 1 GiB
    while (want > 1 && (size_t)want * (size_t)nt * sizeof(double2) > ((size_t)1 << 30))
        --want;
    if (want > ntiles)
        want = ntiles;
    if (want < 1)
        want = 1;
    return want;
}

int direct_step(hipStream_t s, const float4 *posm_cur, float4 *posm_next, const float *rad, int n, int t0, int nt,
                const ForceConsts &fc, const IntegrateArgs &ia, const uint8_t *dyn, const double2 *pos_cur,
                double2 *pos_next, const double *radius, double collision_th, double2 *vel, double2 *acc,
                double2 *part, int jsplit, void *meta, void *meta_next, bool *meta_valid)
{
    if (nt <= 0)
        return NB_OK;
    if (t0 < 0 || t0 + nt > n || jsplit < 1)
        return fail(NB_ERR_ARG, "direct_step: bad target range");
    DirectLaunch a;
    a.posm_cur = posm_cur;
    a.posm_next = posm_next;
    a.posl_cur = direct_lo(posm_cur, n);
    a.posl_next = direct_lo(posm_next, n);
    a.rad = rad;
    a.dyn = dyn;
    a.pos_cur = pos_cur;
    a.pos_next = pos_next;
    a.radius = radius;
    a.meta = (const SubMeta *)meta;
    a.collision_th = collision_th;
    a.vel = vel;
    a.acc = acc;
    a.part = part;
    a.n = n;
    a.t0 = t0;
    a.nt = nt;
    const int nsubs = (n + DS_SUB - 1) / DS_SUB;
    a.subs_per_split = (nsubs + jsplit - 1) / jsplit;
    a.jsplit = (nsubs + a.subs_per_split - 1) / a.subs_per_split;  // no empty splits
    a.fc = fc;
    a.ia = ia;
    if (a.jsplit > 1 && !part)
        return fail(NB_ERR_ARG, "direct_step: split sources need a partial buffer");
    // The boxes of the source runs: computed here unless the step before left them (a step that
    // advances every body writes, with the new records, the boxes they will have as sources).
    const int nsub = (n + DS_SUB - 1) / DS_SUB;
    if (!*meta_valid)
        hipLaunchKernelGGL(k_sub_meta, dim3((nsub + 3) / 4), dim3(256), 0, s, posm_cur, n, (SubMeta *)meta);
    // (the finishing kernel does that; a step without one -- a few hundred bodies -- computes them as before)
#ifdef NB_DS_NO_FUSED_META  // (tuning aid: the boxes always come from k_sub_meta)
    const bool all_advance = false;
#else
    const bool all_advance = ia.do_integrate && t0 == 0 && nt == n && a.jsplit > 1;
#endif
    // (into the other buffer: blocks of this launch still read the present boxes)
    a.meta_out = all_advance ? (SubMeta *)meta_next : nullptr;
    // records unchanged (a force evaluation): the boxes stay; only some records rewritten here (a
    // partitioned run: the rest arrives from other ranks): computed again next time
    *meta_valid = all_advance || !ia.do_integrate;
    dim3 grid((nt + DS_BLOCK * DS_TPT - 1) / (DS_BLOCK * DS_TPT), a.jsplit);
    if (fc.exact_denom && fc.precise_near)
        hipLaunchKernelGGL((k_direct<DS_TPT, true, true>), grid, dim3(DS_BLOCK), 0, s, a);
    else if (fc.exact_denom)
        hipLaunchKernelGGL((k_direct<DS_TPT, true, false>), grid, dim3(DS_BLOCK), 0, s, a);
    else if (fc.precise_near)
        hipLaunchKernelGGL((k_direct<DS_TPT, false, true>), grid, dim3(DS_BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL((k_direct<DS_TPT, false, false>), grid, dim3(DS_BLOCK), 0, s, a);
    NB_HIP(hipGetLastError());
    if (a.jsplit > 1) {
        hipLaunchKernelGGL(k_direct_finish, dim3((nt + 255) / 256), dim3(256), 0, s, a);
        NB_HIP(hipGetLastError());
    }
    return NB_OK;
}

}  // namespace nb
