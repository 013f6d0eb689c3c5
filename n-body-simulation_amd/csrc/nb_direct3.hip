// nb_direct3.hip -- the all-pairs force kernel of nb_direct.hip for three dimensions.
//
// The reference is two-dimensional (src/cpu/vector.h holds x and y); this is its force law with
// dz added to the distance: DirectSum::calculateAcceleration / isCollide
// (src/cpu/directSum.cpp:24-72) and the update order of DirectSum::update (:74-79).
//
// Layout: per body a float4 record {x, y, z, m} and a float {radius + collision_th}, internal
// units (nb_common.h), staged through LDS in tiles of DS3_TILE bodies (two buffers, filled by
// global_load_lds without passing through registers); every lane reads the same record per
// iteration (an LDS broadcast) and owns TPT targets.  As in the 2-D kernel, bodies are held in
// Morton order, a run of DS3_SUB consecutive sources has a small bounding box, and runs farther
// from the block's targets than the largest cutoff are processed by a loop without the cutoff
// test (and without the multiply by the mass when the run's bodies share one mass).  A float
// cutoff comparison too close to call is repeated in FP64.  FP32 partial sums are folded into
// FP64 accumulators every DS3_FLUSH pairs.
//
// Per interaction, far run: 3 sub, 3 fma, 1 rsq, 2-3 mul, 3 fma.  Bound: FP32 vector ALU.
#include "nb_common.h"
#include "nb_dim3.h"

namespace nb {

constexpr int DS3_BLOCK = 256;
constexpr int DS3_TPT = 2;
constexpr int DS3_TILE = 1024;
#ifndef NB_DS3_SUB
#define NB_DS3_SUB 256
#endif
constexpr int DS3_SUB = NB_DS3_SUB;
constexpr int DS3_FLUSH = 32;
constexpr float DS3_BAND = 4.0e-5f;      // see DS_BAND in nb_direct.hip
constexpr float DS3_GAP_SLACK = 1.0005f;  // relative margin on the box gap of a far run ...
constexpr float DS3_GAP_ABS = 4.0e-6f;     // ... and the absolute one (float rounding of the coordinates, internal units)

struct SubMeta3 {
    float x0, x1, y0, y1, z0, z1;
    float wmax;  // max of radius + collision_th
    float mu;    // the mass every body of the run has, or NaN when they differ
};

struct Direct3Launch {
    const float4 *posm_cur;
    float4 *posm_next;
    const float4 *posl_cur;  // low parts of the coordinates of posm_cur
    float4 *posl_next;
    const float *wcut;  // radius + collision_th per body
    const float *rad;   // radius per body
    const uint8_t *dyn;
    const double *pos_cur;  // 3 doubles per body; never written by the step
    double *pos_next;
    const double *radius;
    const SubMeta3 *meta;
    double collision_th;
    double *vel;
    double *acc;
    double *part;  // jsplit x nt x 3
    int n, t0, nt;
    int tiles_per_split, jsplit;
    ForceConsts fc;
    IntegrateArgs ia;
};

__global__ __launch_bounds__(256) void k_sub_meta3(const float4 *posm, const float *wcut, int n, SubMeta3 *meta)
{
    const int lane = threadIdx.x & 63;
    const int sub = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int base = sub * DS3_SUB;
    if (base >= n)
        return;
    const float inf = __builtin_inff();
    float x0 = inf, x1 = -inf, y0 = inf, y1 = -inf, z0 = inf, z1 = -inf, wmax = 0.f, m0 = inf, m1 = -inf;
#pragma unroll
    for (int q = 0; q < DS3_SUB / 64; ++q) {
        const int j = base + lane + 64 * q;
        if (j < n) {
            const float4 r = posm[j];
            x0 = fminf(x0, r.x);
            x1 = fmaxf(x1, r.x);
            y0 = fminf(y0, r.y);
            y1 = fmaxf(y1, r.y);
            z0 = fminf(z0, r.z);
            z1 = fmaxf(z1, r.z);
            wmax = fmaxf(wmax, wcut[j]);
            m0 = fminf(m0, r.w);
            m1 = fmaxf(m1, r.w);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        x0 = fminf(x0, __shfl_xor(x0, o));
        x1 = fmaxf(x1, __shfl_xor(x1, o));
        y0 = fminf(y0, __shfl_xor(y0, o));
        y1 = fmaxf(y1, __shfl_xor(y1, o));
        z0 = fminf(z0, __shfl_xor(z0, o));
        z1 = fmaxf(z1, __shfl_xor(z1, o));
        wmax = fmaxf(wmax, __shfl_xor(wmax, o));
        m0 = fminf(m0, __shfl_xor(m0, o));
        m1 = fmaxf(m1, __shfl_xor(m1, o));
    }
    if (lane == 0)
        meta[sub] = SubMeta3{x0, x1, y0, y1, z0, z1, wmax, (m0 == m1) ? m0 : __builtin_nanf("")};
}

// v += a dt_kick; x += v dt for one body, and its refreshed float record.
__device__ __forceinline__ void integrate_one3(const Direct3Launch &a, int i, const double (&s)[3], float4 rec)
{
    double acc[3] = {0.0, 0.0, 0.0};
    if (a.dyn[i])
        for (int k = 0; k < 3; ++k)
            acc[k] = s[k] * a.ia.acc_factor;
    for (int k = 0; k < 3; ++k)
        a.acc[3 * (size_t)i + k] = acc[k];
    if (a.ia.do_integrate) {
        float q[3], l[3];
        for (int k = 0; k < 3; ++k) {
            const double v = a.vel[3 * (size_t)i + k] + acc[k] * a.ia.dt_kick;
            const double p = a.pos_cur[3 * (size_t)i + k] + v * a.ia.dt;
            a.vel[3 * (size_t)i + k] = v;
            a.pos_next[3 * (size_t)i + k] = p;
            const double s64 = (p - a.ia.origin[k]) * a.ia.inv_len_unit;
            q[k] = (float)s64;
            l[k] = (float)(s64 - (double)q[k]);
        }
        a.posm_next[i] = make_float4(q[0], q[1], q[2], rec.w);
        a.posl_next[i] = make_float4(l[0], l[1], l[2], 0.f);
    }
}

__device__ __forceinline__ float pair_scale(bool exact, float r2, float eps3)
{
    if (exact)
        return __builtin_amdgcn_rcpf(r2 * __builtin_sqrtf(r2) + eps3);
    const float rinv = __builtin_amdgcn_rsqf(r2);
    return rinv * (rinv * rinv);
}

// The pairs of one target with one run whose float cutoff comparison was too close to call,
// decided in FP64 on the FP64 positions (isCollide, directSum.cpp:69-72: >=).
// (PRECISE as in run_pairs3: the same distance must select the pairs here that the hot loop left out.)
template <bool EXACT, bool PRECISE>
__device__ __forceinline__ void redo_ambiguous3(const Direct3Launch &a, const float4 *rec, const float *wc, const float4 *lo,
                                                int jglobal, int jend, float xi, float yi, float zi, const float (&tl)[3],
                                                float ri, int ti, double (&s)[3])
{
    const float eps2 = a.fc.eps2, eps3 = a.fc.eps3;
    const double *pi = a.pos_cur + 3 * (size_t)ti;
    const double rdi = a.radius[ti];
#pragma unroll 1
    for (int jj = 0; jj < DS3_SUB; ++jj) {
        const float4 b = rec[jj];
        float dx = b.x - xi, dy = b.y - yi, dz = b.z - zi;
        if (PRECISE) {
            const float4 l = lo[jj];
            dx += l.x - tl[0];
            dy += l.y - tl[1];
            dz += l.z - tl[2];
        }
        float r2 = __builtin_fmaf(dx, dx, eps2);
        r2 = __builtin_fmaf(dy, dy, r2);
        r2 = __builtin_fmaf(dz, dz, r2);
        const float th = ri + wc[jj];
        const float th2 = __builtin_fmaf(th, th, eps2);
        const int j = jglobal + jj;
        if (__builtin_fabsf(r2 - th2) <= th2 * DS3_BAND && j < jend && j != ti) {
            const double *pj = a.pos_cur + 3 * (size_t)j;
            const double ex = pj[0] - pi[0], ey = pj[1] - pi[1], ez = pj[2] - pi[2];
            const double dist = sqrt(ex * ex + ey * ey + ez * ez);
            if (!(rdi + a.radius[j] + a.collision_th >= dist)) {
                const float sc = b.w * pair_scale(EXACT, r2, eps3);
                s[0] += (double)(dx * sc);
                s[1] += (double)(dy * sc);
                s[2] += (double)(dz * sc);
            }
        }
    }
}

// All pairs of this thread's TPT targets with one run of DS3_SUB staged records.  A run that may
// hold pairs near the cutoff (CUTOFF) takes its coordinates as float pairs hi + lo (lo, tl: what
// rounding the FP64 coordinate to the float record left over), so that the difference of two close
// bodies is exact to float precision of the difference, not of the coordinates.
template <int TPT, bool EXACT, bool CUTOFF, bool UNIFORM, bool PRECISE = false>
__device__ __forceinline__ void run_pairs3(const Direct3Launch &a, const float4 *rec, const float *wc, const float4 *lo,
                                           int jglobal, int jend, float mu, const float (&xi)[TPT], const float (&yi)[TPT],
                                           const float (&zi)[TPT], const float (&tl)[TPT][3], const float (&ri)[TPT],
                                           const int (&ti)[TPT], double (&acc)[TPT][3])
{
    const float eps2 = a.fc.eps2, eps3 = a.fc.eps3;
    bool any_amb[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k)
        any_amb[k] = false;
#pragma unroll 1
    for (int c = 0; c < DS3_SUB; c += DS3_FLUSH) {
        float fx[TPT], fy[TPT], fz[TPT];
#pragma unroll
        for (int k = 0; k < TPT; ++k)
            fx[k] = fy[k] = fz[k] = 0.f;
        // the chunk's low parts: one coalesced load (lane L holds record c + L), handed out lane by lane
        static_assert(DS3_FLUSH <= 64, "one lane per record of a chunk");
        float4 lchunk = make_float4(0.f, 0.f, 0.f, 0.f);
        if (CUTOFF && PRECISE)
            lchunk = lo[c + ((int)threadIdx.x & 63 & (DS3_FLUSH - 1))];
#pragma unroll 8
        for (int jj = 0; jj < DS3_FLUSH; ++jj) {
            const float4 b = rec[c + jj];
            const float wj = CUTOFF ? wc[c + jj] : 0.f;
            float4 l = make_float4(0.f, 0.f, 0.f, 0.f);
            if (CUTOFF && PRECISE)
                l = make_float4(__shfl(lchunk.x, jj), __shfl(lchunk.y, jj), __shfl(lchunk.z, jj), 0.f);
#pragma unroll
            for (int k = 0; k < TPT; ++k) {
                float dx = b.x - xi[k], dy = b.y - yi[k], dz = b.z - zi[k];
                if (CUTOFF && PRECISE) {
                    dx += l.x - tl[k][0];
                    dy += l.y - tl[k][1];
                    dz += l.z - tl[k][2];
                }
                float r2 = __builtin_fmaf(dx, dx, eps2);
                r2 = __builtin_fmaf(dy, dy, r2);
                r2 = __builtin_fmaf(dz, dz, r2);
                float s = pair_scale(EXACT, r2, eps3);
                if (!UNIFORM)
                    s *= b.w;
                if (CUTOFF) {
                    const float th = ri[k] + wj;
                    const float th2 = __builtin_fmaf(th, th, eps2);
                    const float q = r2 - th2, band = th2 * DS3_BAND;
                    any_amb[k] = any_amb[k] || __builtin_fabsf(q) <= band;
                    s = q > band ? s : 0.f;
                }
                fx[k] = __builtin_fmaf(dx, s, fx[k]);
                fy[k] = __builtin_fmaf(dy, s, fy[k]);
                fz[k] = __builtin_fmaf(dz, s, fz[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            acc[k][0] += (double)(UNIFORM ? fx[k] * mu : fx[k]);
            acc[k][1] += (double)(UNIFORM ? fy[k] * mu : fy[k]);
            acc[k][2] += (double)(UNIFORM ? fz[k] * mu : fz[k]);
        }
    }
    if (CUTOFF) {
#pragma unroll
        for (int k = 0; k < TPT; ++k)
            if (any_amb[k])
                redo_ambiguous3<EXACT, PRECISE>(a, rec, wc, lo, jglobal, jend, xi[k], yi[k], zi[k], tl[k], ri[k], ti[k], acc[k]);
    }
}

typedef float v2f3 __attribute__((ext_vector_type(2)));

// The cutoff-free loop on packed FP32 (v_pk_add/mul/fma_f32): targets are taken two at a time, so
// every operation but the reciprocal square root serves two pairs.  Same operations and rounding
// per pair as run_pairs3<.., false, ..> without the second softening term.
template <int TPT, bool UNIFORM>
__device__ __forceinline__ void run_pairs3_far(const Direct3Launch &a, const float4 *rec, float mu, const float (&xi)[TPT],
                                               const float (&yi)[TPT], const float (&zi)[TPT], double (&acc)[TPT][3])
{
    static_assert(TPT % 2 == 0, "packed loop takes targets in pairs");
    constexpr int P = TPT / 2;
    const v2f3 eps2 = {a.fc.eps2, a.fc.eps2};
    v2f3 x2[P], y2[P], z2[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        x2[p] = v2f3{xi[2 * p], xi[2 * p + 1]};
        y2[p] = v2f3{yi[2 * p], yi[2 * p + 1]};
        z2[p] = v2f3{zi[2 * p], zi[2 * p + 1]};
    }
#pragma unroll 1
    for (int c = 0; c < DS3_SUB; c += DS3_FLUSH) {
        v2f3 fx[P], fy[P], fz[P];
#pragma unroll
        for (int p = 0; p < P; ++p)
            fx[p] = fy[p] = fz[p] = v2f3{0.f, 0.f};
#pragma unroll 16
        for (int jj = 0; jj < DS3_FLUSH; ++jj) {
            const float4 b = rec[c + jj];
            const v2f3 bx = {b.x, b.x}, by = {b.y, b.y}, bz = {b.z, b.z}, bm = {b.w, b.w};
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const v2f3 dx = bx - x2[p], dy = by - y2[p], dz = bz - z2[p];
                v2f3 r2 = __builtin_elementwise_fma(dx, dx, eps2);
                r2 = __builtin_elementwise_fma(dy, dy, r2);
                r2 = __builtin_elementwise_fma(dz, dz, r2);
                const v2f3 rinv = {__builtin_amdgcn_rsqf(r2.x), __builtin_amdgcn_rsqf(r2.y)};
                const v2f3 s = UNIFORM ? rinv * (rinv * rinv) : (bm * rinv) * (rinv * rinv);
                fx[p] = __builtin_elementwise_fma(dx, s, fx[p]);
                fy[p] = __builtin_elementwise_fma(dy, s, fy[p]);
                fz[p] = __builtin_elementwise_fma(dz, s, fz[p]);
            }
        }
#pragma unroll
        for (int p = 0; p < P; ++p) {
            acc[2 * p][0] += (double)(UNIFORM ? fx[p].x * mu : fx[p].x);
            acc[2 * p + 1][0] += (double)(UNIFORM ? fx[p].y * mu : fx[p].y);
            acc[2 * p][1] += (double)(UNIFORM ? fy[p].x * mu : fy[p].x);
            acc[2 * p + 1][1] += (double)(UNIFORM ? fy[p].y * mu : fy[p].y);
            acc[2 * p][2] += (double)(UNIFORM ? fz[p].x * mu : fz[p].x);
            acc[2 * p + 1][2] += (double)(UNIFORM ? fz[p].y * mu : fz[p].y);
        }
    }
}

__device__ __forceinline__ float uniform_f3(float v)
{
    return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v)));
}

// One tile of records and cutoff radii into LDS (the arrays are padded to whole tiles).
__device__ __forceinline__ void stage_tile3(const float4 *src, const float *wsrc, float4 *dst, float *wdst, int tid)
{
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
#pragma unroll
    for (int l = 0; l < DS3_TILE / DS3_BLOCK; ++l) {
        const int first = (l * (DS3_BLOCK / 64) + wave) * 64;
        __builtin_amdgcn_global_load_lds((const void *)(src + first + lane),
                                         (__attribute__((address_space(3))) void *)(dst + first), 16, 0, 0);
        __builtin_amdgcn_global_load_lds((const void *)(wsrc + first + lane),
                                         (__attribute__((address_space(3))) void *)(wdst + first), 4, 0, 0);
    }
}

template <int TPT, bool EXACT, bool PRECISE>
__global__ __launch_bounds__(DS3_BLOCK) void k_direct3(Direct3Launch a)
{
    __shared__ __attribute__((aligned(16))) float4 tile[2][DS3_TILE];
    __shared__ __attribute__((aligned(16))) float wtile[2][DS3_TILE];
    const int tid = threadIdx.x;
    // a wave's targets are consecutive bodies: lane l takes tbase + 64 TPT wave + l, + 64, ...
    const int wbase = (tid >> 6) * (64 * TPT) + (tid & 63);
    const int tbase = a.t0 + blockIdx.x * (DS3_BLOCK * TPT);
    const int tend = a.t0 + a.nt;

    const int jbeg = blockIdx.y * a.tiles_per_split * DS3_TILE;
    int jend = jbeg + a.tiles_per_split * DS3_TILE;
    if (jend > a.n)
        jend = a.n;
    const int ntile = (jend - jbeg + DS3_TILE - 1) / DS3_TILE;
    stage_tile3(a.posm_cur + jbeg, a.wcut + jbeg, tile[0], wtile[0], tid);

    float xi[TPT], yi[TPT], zi[TPT], tl[TPT][3], ri[TPT];
    int ti[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const int i = tbase + wbase + k * 64;
        ti[k] = i < tend ? i : tend - 1;  // threads past the end repeat the last target and discard it
        const float4 self = a.posm_cur[ti[k]];
        xi[k] = self.x;
        yi[k] = self.y;
        zi[k] = self.z;
        const float4 selfl = a.posl_cur[ti[k]];
        tl[k][0] = selfl.x;
        tl[k][1] = selfl.y;
        tl[k][2] = selfl.z;
        ri[k] = a.rad[ti[k]];
    }
    double acc[TPT][3];
#pragma unroll
    for (int k = 0; k < TPT; ++k)
        acc[k][0] = acc[k][1] = acc[k][2] = 0.0;

    // bounding box of the WAVE's targets (64 TPT consecutive bodies) and their largest radius:
    // every wave decides by itself which runs are far (tighter boxes, fewer near runs)
    float lo[3] = {xi[0], yi[0], zi[0]}, hi[3] = {xi[0], yi[0], zi[0]}, rimax = ri[0];
#pragma unroll
    for (int k = 1; k < TPT; ++k) {
        lo[0] = fminf(lo[0], xi[k]);
        hi[0] = fmaxf(hi[0], xi[k]);
        lo[1] = fminf(lo[1], yi[k]);
        hi[1] = fmaxf(hi[1], yi[k]);
        lo[2] = fminf(lo[2], zi[k]);
        hi[2] = fmaxf(hi[2], zi[k]);
        rimax = fmaxf(rimax, ri[k]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            lo[d] = fminf(lo[d], __shfl_xor(lo[d], o));
            hi[d] = fmaxf(hi[d], __shfl_xor(hi[d], o));
        }
        rimax = fmaxf(rimax, __shfl_xor(rimax, o));
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        lo[d] = uniform_f3(lo[d]);
        hi[d] = uniform_f3(hi[d]);
    }
    rimax = uniform_f3(rimax);
    __syncthreads();  // completes the staging of tile 0

    for (int t = 0; t < ntile; ++t) {
        if (t + 1 < ntile)
            stage_tile3(a.posm_cur + jbeg + (t + 1) * DS3_TILE, a.wcut + jbeg + (t + 1) * DS3_TILE, tile[(t + 1) & 1],
                        wtile[(t + 1) & 1], tid);
        const float4 *cur = tile[t & 1];
        const float *wcur = wtile[t & 1];
        const int jt = jbeg + t * DS3_TILE;
#pragma unroll 1
        for (int sub = 0; sub < DS3_TILE / DS3_SUB; ++sub) {
            const int j0 = jt + sub * DS3_SUB;
            if (j0 >= jend)
                break;
            const SubMeta3 m = a.meta[j0 / DS3_SUB];
            const float gx = fmaxf(fmaxf(m.x0 - hi[0], lo[0] - m.x1), 0.f);
            const float gy = fmaxf(fmaxf(m.y0 - hi[1], lo[1] - m.y1), 0.f);
            const float gz = fmaxf(fmaxf(m.z0 - hi[2], lo[2] - m.z1), 0.f);
            const float lim = (rimax + m.wmax) * DS3_GAP_SLACK + DS3_GAP_ABS;
            // far: no pair of the two boxes can be inside the cutoff (a NaN anywhere means near)
            const bool far = __builtin_fmaf(gx, gx, __builtin_fmaf(gy, gy, gz * gz)) > lim * lim;
            const bool uni = m.mu == m.mu;
            const float4 *rec = cur + sub * DS3_SUB;
            const float *wc = wcur + sub * DS3_SUB;
            if (far && !EXACT && TPT % 2 == 0) {
                if (uni)
                    run_pairs3_far<TPT, true>(a, rec, m.mu, xi, yi, zi, acc);
                else
                    run_pairs3_far<TPT, false>(a, rec, 0.f, xi, yi, zi, acc);
            } else if (far && uni)
                run_pairs3<TPT, EXACT, false, true>(a, rec, wc, nullptr, j0, jend, m.mu, xi, yi, zi, tl, ri, ti, acc);
            else if (far)
                run_pairs3<TPT, EXACT, false, false>(a, rec, wc, nullptr, j0, jend, 0.f, xi, yi, zi, tl, ri, ti, acc);
            else  // the low parts come straight from memory: every lane reads the same address
                run_pairs3<TPT, EXACT, true, false, PRECISE>(a, rec, wc, a.posl_cur + j0, j0, jend, 0.f, xi, yi, zi, tl, ri, ti,
                                                             acc);
        }
        __syncthreads();  // the next tile has landed and every wave is done with the current one
    }

#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const int i = tbase + wbase + k * 64;
        if (i >= tend)
            continue;
        if (a.jsplit == 1) {
            integrate_one3(a, i, acc[k], a.posm_cur[i]);
        } else {
            double *o = a.part + 3 * ((size_t)blockIdx.y * a.nt + (i - a.t0));
            o[0] = acc[k][0];
            o[1] = acc[k][1];
            o[2] = acc[k][2];
        }
    }
}

// Sums the jsplit partial accelerations of each target in split order, then integrates.
__global__ __launch_bounds__(256) void k_direct3_finish(Direct3Launch a)
{
    const int i = a.t0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.t0 + a.nt)
        return;
    double s[3] = {0.0, 0.0, 0.0};
    for (int q = 0; q < a.jsplit; ++q) {
        const double *p = a.part + 3 * ((size_t)q * a.nt + (i - a.t0));
        s[0] += p[0];
        s[1] += p[1];
        s[2] += p[2];
    }
    integrate_one3(a, i, s, a.posm_cur[i]);
}

// Records of bodies [i0, i1) from the FP64 state; records [n, cap) cannot interact (zero mass,
// far away, infinite cutoff), so whole tiles can be staged without a bounds test.
__global__ __launch_bounds__(256) void k_pack_posm3(const double *pos, const double *mass, const double *radius, int i0, int i1,
                                                    int n, int cap, double ox, double oy, double oz, double inv_len,
                                                    double inv_mass, double cth, float4 *posm, float4 *posl, float *wcut,
                                                    float *rad)
{
    const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < i1 && i < n) {
        const double q[3] = {(pos[3 * (size_t)i] - ox) * inv_len, (pos[3 * (size_t)i + 1] - oy) * inv_len,
                             (pos[3 * (size_t)i + 2] - oz) * inv_len};
        const float h[3] = {(float)q[0], (float)q[1], (float)q[2]};
        posm[i] = make_float4(h[0], h[1], h[2], (float)(mass[i] * inv_mass));
        posl[i] = make_float4((float)(q[0] - (double)h[0]), (float)(q[1] - (double)h[1]), (float)(q[2] - (double)h[2]), 0.f);
        wcut[i] = (float)((radius[i] + cth) * inv_len);
        rad[i] = (float)(radius[i] * inv_len);
    } else if (i >= n && i < cap) {
        posm[i] = make_float4(1.0e15f, 1.0e15f, 1.0e15f, 0.f);
        posl[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        wcut[i] = __builtin_inff();
    }
}

size_t direct3_record_capacity(int n) { return ((size_t)n + DS3_TILE - 1) / DS3_TILE * DS3_TILE + DS3_TILE; }
size_t direct3_meta_bytes(int n) { return sizeof(SubMeta3) * (size_t)((n + DS3_SUB - 1) / DS3_SUB); }

int pack_posm3(hipStream_t s, const double *pos, const double *mass, const double *radius, int n, const Units &units,
               double collision_th, float4 *posm, float *wcut, float *rad)
{
    const int cap = (int)direct3_record_capacity(n);
    hipLaunchKernelGGL(k_pack_posm3, dim3((cap + 255) / 256), dim3(256), 0, s, pos, mass, radius, 0, cap, n, cap,
                       units.origin[0], units.origin[1], units.origin[2], 1.0 / units.len_unit, 1.0 / units.mass_unit,
                       collision_th, posm, posm + cap, wcut, rad);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

int direct3_choose_jsplit(int n, int nt)
{
    // many short blocks keep the 256 CUs level (see direct_choose_jsplit)
    if (nt <= 0)
        return 1;  // a rank without targets launches nothing
    const int tb = (nt + DS3_BLOCK * DS3_TPT - 1) / (DS3_BLOCK * DS3_TPT);
    const int ntiles = (n + DS3_TILE - 1) / DS3_TILE;
    int want = (65536 + tb - 1) / tb;
    if (want > 64)
        want = 64;
    while (want > 1 && (size_t)want * (size_t)nt * 24 > ((size_t)1 << 30))
        --want;
    if (want > ntiles)
        want = ntiles;
    return want < 1 ? 1 : want;
}

__global__ __launch_bounds__(256) void k_pack_posm3_range(const double *pos, const double *mass, int i0, int i1, double ox,
                                                          double oy, double oz, double inv_len, double inv_mass, float4 *posm,
                                                          float4 *posl)
{
    const int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i < i1) {
        const double q[3] = {(pos[3 * (size_t)i] - ox) * inv_len, (pos[3 * (size_t)i + 1] - oy) * inv_len,
                             (pos[3 * (size_t)i + 2] - oz) * inv_len};
        const float h[3] = {(float)q[0], (float)q[1], (float)q[2]};
        posm[i] = make_float4(h[0], h[1], h[2], (float)(mass[i] * inv_mass));
        posl[i] = make_float4((float)(q[0] - (double)h[0]), (float)(q[1] - (double)h[1]), (float)(q[2] - (double)h[2]), 0.f);
    }
}

int pack_posm3_range(hipStream_t s, const double *pos, const double *mass, int n, int i0, int i1, const Units &units,
                     float4 *posm)
{
    if (i1 <= i0)
        return NB_OK;
    hipLaunchKernelGGL(k_pack_posm3_range, dim3((i1 - i0 + 255) / 256), dim3(256), 0, s, pos, mass, i0, i1, units.origin[0],
                       units.origin[1], units.origin[2], 1.0 / units.len_unit, 1.0 / units.mass_unit, posm,
                       posm + direct3_record_capacity(n));
    NB_HIP(hipGetLastError());
    return NB_OK;
}

int direct3_step(hipStream_t s, const Direct3State &st, int n, int t0, int nt, const ForceConsts &fc, const IntegrateArgs &ia,
                 double collision_th)
{
    if (nt <= 0)
        return NB_OK;
    if (t0 < 0 || t0 + nt > n)
        return fail(NB_ERR_ARG, "direct3_step: bad target range");
    Direct3Launch a;
    a.posm_cur = st.posm[st.cur];
    a.posm_next = st.posm[st.cur ^ 1];
    a.posl_cur = a.posm_cur + direct3_record_capacity(n);
    a.posl_next = a.posm_next + direct3_record_capacity(n);
    a.wcut = st.wcut;
    a.rad = st.rad;
    a.dyn = st.dyn;
    a.pos_cur = st.pos_cur;
    a.pos_next = st.pos_next;
    a.radius = st.radius;
    a.meta = (const SubMeta3 *)st.meta;
    a.collision_th = collision_th;
    a.vel = st.vel;
    a.acc = st.acc;
    a.part = st.part;
    a.n = n;
    a.t0 = t0;
    a.nt = nt;
    const int ntiles = (n + DS3_TILE - 1) / DS3_TILE;
    a.tiles_per_split = (ntiles + st.jsplit - 1) / st.jsplit;
    a.jsplit = (ntiles + a.tiles_per_split - 1) / a.tiles_per_split;  // no empty splits
    a.fc = fc;
    a.ia = ia;
    if (a.jsplit > 1 && !st.part)
        return fail(NB_ERR_ARG, "direct3_step: split sources need a partial buffer");
    const int nsub = (n + DS3_SUB - 1) / DS3_SUB;
    hipLaunchKernelGGL(k_sub_meta3, dim3((nsub + 3) / 4), dim3(256), 0, s, a.posm_cur, a.wcut, n, (SubMeta3 *)st.meta);
    dim3 grid((nt + DS3_BLOCK * DS3_TPT - 1) / (DS3_BLOCK * DS3_TPT), a.jsplit);
    if (fc.exact_denom && fc.precise_near)
        hipLaunchKernelGGL((k_direct3<DS3_TPT, true, true>), grid, dim3(DS3_BLOCK), 0, s, a);
    else if (fc.exact_denom)
        hipLaunchKernelGGL((k_direct3<DS3_TPT, true, false>), grid, dim3(DS3_BLOCK), 0, s, a);
    else if (fc.precise_near)
        hipLaunchKernelGGL((k_direct3<DS3_TPT, false, true>), grid, dim3(DS3_BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL((k_direct3<DS3_TPT, false, false>), grid, dim3(DS3_BLOCK), 0, s, a);
    NB_HIP(hipGetLastError());
    if (a.jsplit > 1) {
        hipLaunchKernelGGL(k_direct3_finish, dim3((nt + 255) / 256), dim3(256), 0, s, a);
        NB_HIP(hipGetLastError());
    }
    return NB_OK;
}

}  // namespace nb
