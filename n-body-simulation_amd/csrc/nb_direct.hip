// nb_direct.hip -- all-pairs force kernel with fused integration for gfx950.
//
// Replaces DirectSumTiledKernel (src/gpu/DS/directSum_kernel.cu:61-106) with the semantics
// of the CPU DirectSum::update (src/cpu/directSum.cpp:24-79): accelerations from the OLD
// positions of every body, then v += a dt, x += v dt.  Positions are double-buffered
// (records are read from posm_cur, new ones go to posm_next), so no block ever reads a
// position another block has already advanced.
//
// Data layout: one float4 record {x, y, m, radius + collision_th} per body, in internal
// units (nb_common.h), read coalesced (16 B per lane) and staged through LDS in tiles of
// TILE records; every lane then reads the same record per iteration (an LDS broadcast).
// Each thread owns TPT targets, so one LDS read feeds TPT interactions.  The FP32 partial
// sums of one tile are added to FP64 accumulators, which keeps the rounding of a
// million-term sum at the level of a 1024-term one.
//
// Per interaction (2-D): 2 sub, 3 fma, 1 add, 3 mul, 1 rsq, 1 cmp, 1 select, 2 fma.
// Bound: FP32 vector ALU (no MFMA: there is no dense contraction here).
#include "nb_common.h"

namespace nb {

constexpr int DS_BLOCK = 256;
constexpr int DS_TILE = 1024;  // records per LDS tile (16 KiB), two buffers
// A float cutoff comparison closer than this fraction is repeated in FP64: the float distance
// carries the coordinates' rounding (2^-24 of the domain each), a few 1e-6 of the cutoff.
constexpr float DS_BAND = 4.0e-5f;

struct DirectLaunch {
    const float4 *posm_cur;
    float4 *posm_next;
    const float *rad;
    const uint8_t *dyn;
    const double2 *pos_cur;  // FP64 positions the step starts from (never written by the step)
    double2 *pos_next;       // FP64 positions after the step
    const double *radius;
    double collision_th;
    double2 *vel;
    double2 *acc;
    double2 *part;
    int n;
    int t0;
    int nt;
    int tiles_per_split;
    int jsplit;
    ForceConsts fc;
    IntegrateArgs ia;
};

// v += a dt; x += v dt for one body, and its refreshed float record.
__device__ __forceinline__ void integrate_one(const DirectLaunch &a, int i, double sx, double sy, float4 rec)
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
        v.x += ax * a.ia.dt;
        v.y += ay * a.ia.dt;
        p.x += v.x * a.ia.dt;
        p.y += v.y * a.ia.dt;
        a.vel[i] = v;
        a.pos_next[i] = p;
        rec.x = (float)(p.x * a.ia.inv_len_unit);
        rec.y = (float)(p.y * a.ia.inv_len_unit);
        a.posm_next[i] = rec;
    }
}

template <int TPT, bool EXACT>
__global__ __launch_bounds__(DS_BLOCK) void k_direct(DirectLaunch a)
{
    __shared__ float4 tile[2][DS_TILE];
    const int tid = threadIdx.x;
    const int tbase = a.t0 + blockIdx.x * (DS_BLOCK * TPT);
    const int tend = a.t0 + a.nt;

    float xi[TPT], yi[TPT], ri[TPT];
    float4 self[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        int i = tbase + tid + k * DS_BLOCK;
        // threads past the end work on a copy of the last target and discard the result
        int ic = i < tend ? i : tend - 1;
        self[k] = a.posm_cur[ic];
        xi[k] = self[k].x;
        yi[k] = self[k].y;
        ri[k] = a.rad[ic];
    }
    double sx[TPT], sy[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k)
        sx[k] = sy[k] = 0.0;
    int ti[TPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        int i = tbase + tid + k * DS_BLOCK;
        ti[k] = i < tend ? i : tend - 1;
    }

    const int jbeg = blockIdx.y * a.tiles_per_split * DS_TILE;
    int jend = jbeg + a.tiles_per_split * DS_TILE;
    if (jend > a.n)
        jend = a.n;
    const int ntile = (jend - jbeg + DS_TILE - 1) / DS_TILE;
    // a record that can never interact: zero mass, infinite cutoff
    const float4 pad = make_float4(0.f, 0.f, 0.f, __builtin_inff());
    constexpr int LPT = DS_TILE / DS_BLOCK;  // records each thread stages per tile
    float4 stage[LPT];

#pragma unroll
    for (int l = 0; l < LPT; ++l) {
        int j = jbeg + tid + l * DS_BLOCK;
        stage[l] = j < jend ? a.posm_cur[j] : pad;
    }
#pragma unroll
    for (int l = 0; l < LPT; ++l)
        tile[0][tid + l * DS_BLOCK] = stage[l];
    __syncthreads();

    const float eps2 = a.fc.eps2, eps3 = a.fc.eps3;
    for (int t = 0; t < ntile; ++t) {
        const bool more = t + 1 < ntile;
        if (more) {
#pragma unroll
            for (int l = 0; l < LPT; ++l) {
                int j = jbeg + (t + 1) * DS_TILE + tid + l * DS_BLOCK;
                stage[l] = j < jend ? a.posm_cur[j] : pad;
            }
        }
        const float4 *cur = tile[t & 1];
        float fx[TPT], fy[TPT];
#pragma unroll
        for (int k = 0; k < TPT; ++k)
            fx[k] = fy[k] = 0.f;
#pragma unroll 8
        for (int jj = 0; jj < DS_TILE; ++jj) {
            const float4 b = cur[jj];
#pragma unroll
            for (int k = 0; k < TPT; ++k) {
                float dx = b.x - xi[k];
                float dy = b.y - yi[k];
                float r2 = __builtin_fmaf(dx, dx, eps2);
                r2 = __builtin_fmaf(dy, dy, r2);
                float th = ri[k] + b.w;
                float th2 = __builtin_fmaf(th, th, eps2);  // d^2 > th^2  <=>  d^2+eps^2 > th^2+eps^2
                float s;
                if (EXACT) {
                    float r3 = r2 * __builtin_sqrtf(r2);
                    s = b.z * __builtin_amdgcn_rcpf(r3 + eps3);
                } else {
                    float rinv = __builtin_amdgcn_rsqf(r2);
                    s = (b.z * rinv) * (rinv * rinv);
                }
                // DirectSum::isCollide skips r_i + r_j + th >= dist (directSum.cpp:69-72)
                bool pass = r2 > th2;
                const bool amb = __builtin_fabsf(r2 - th2) <= th2 * DS_BAND;
                if (__ballot(amb) != 0ull) {
                    // Rare: the float distance is within its own rounding of the cutoff; decide
                    // in FP64 on the FP64 positions as the reference does (vector.cpp:31-35).
                    const int j = jbeg + t * DS_TILE + jj;
                    if (amb && j < jend) {
                        const double2 pj = a.pos_cur[j];
                        const double2 pi = a.pos_cur[ti[k]];
                        const double ex = pj.x - pi.x, ey = pj.y - pi.y;
                        const double dist = sqrt(ex * ex + ey * ey);
                        pass = !(a.radius[ti[k]] + a.radius[j] + a.collision_th >= dist) && j != ti[k];
                    }
                }
                s = pass ? s : 0.f;
                fx[k] = __builtin_fmaf(dx, s, fx[k]);
                fy[k] = __builtin_fmaf(dy, s, fy[k]);
            }
        }
#pragma unroll
        for (int k = 0; k < TPT; ++k) {
            sx[k] += (double)fx[k];
            sy[k] += (double)fy[k];
        }
        if (more) {
#pragma unroll
            for (int l = 0; l < LPT; ++l)
                tile[(t + 1) & 1][tid + l * DS_BLOCK] = stage[l];
        }
        __syncthreads();
    }

#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        int i = tbase + tid + k * DS_BLOCK;
        if (i >= tend)
            continue;
        if (a.jsplit == 1)
            integrate_one(a, i, sx[k], sy[k], self[k]);
        else
            a.part[(size_t)blockIdx.y * a.nt + (i - a.t0)] = make_double2(sx[k], sy[k]);
    }
}

// Sums the jsplit partial accelerations of each target in split order, then integrates.
__global__ __launch_bounds__(256) void k_direct_finish(DirectLaunch a)
{
    int i = a.t0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= a.t0 + a.nt)
        return;
    double sx = 0.0, sy = 0.0;
    for (int s = 0; s < a.jsplit; ++s) {
        double2 p = a.part[(size_t)s * a.nt + (i - a.t0)];
        sx += p.x;
        sy += p.y;
    }
    integrate_one(a, i, sx, sy, a.posm_cur[i]);
}

__global__ __launch_bounds__(256) void k_pack_posm(const double2 *pos, const double *mass, const double *radius, int i0,
                                                   int i1, double inv_len, double inv_mass, double cth, float4 *posm,
                                                   float *rad)
{
    int i = i0 + blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= i1)
        return;
    double2 p = pos[i];
    posm[i] = make_float4((float)(p.x * inv_len), (float)(p.y * inv_len), (float)(mass[i] * inv_mass),
                          (float)((radius[i] + cth) * inv_len));
    rad[i] = (float)(radius[i] * inv_len);
}

int pack_posm(hipStream_t s, const double2 *pos, const double *mass, const double *radius, int i0, int i1,
              double inv_len_unit, double inv_mass_unit, double collision_th, float4 *posm, float *rad)
{
    if (i1 <= i0)
        return NB_OK;
    int blocks = (i1 - i0 + 255) / 256;
    hipLaunchKernelGGL(k_pack_posm, dim3(blocks), dim3(256), 0, s, pos, mass, radius, i0, i1, inv_len_unit,
                       inv_mass_unit, collision_th, posm, rad);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

constexpr int DS_TPT = 2;

int direct_choose_jsplit(int n, int nt)
{
    // Enough blocks to give every SIMD of the 256 CUs several waves: when the targets
    // alone do not, the source range is split across blockIdx.y.
    int tb = (nt + DS_BLOCK * DS_TPT - 1) / (DS_BLOCK * DS_TPT);
    int ntiles = (n + DS_TILE - 1) / DS_TILE;
    int want = (2048 + tb - 1) / tb;
    if (want > ntiles)
        want = ntiles;
    if (want < 1)
        want = 1;
    if (want > 64)
        want = 64;
    return want;
}

int direct_step(hipStream_t s, const float4 *posm_cur, float4 *posm_next, const float *rad, int n, int t0, int nt,
                const ForceConsts &fc, const IntegrateArgs &ia, const uint8_t *dyn, const double2 *pos_cur,
                double2 *pos_next, const double *radius, double collision_th, double2 *vel, double2 *acc,
                double2 *part, int jsplit)
{
    if (nt <= 0)
        return NB_OK;
    if (t0 < 0 || t0 + nt > n || jsplit < 1)
        return fail(NB_ERR_ARG, "direct_step: bad target range");
    DirectLaunch a;
    a.posm_cur = posm_cur;
    a.posm_next = posm_next;
    a.rad = rad;
    a.dyn = dyn;
    a.pos_cur = pos_cur;
    a.pos_next = pos_next;
    a.radius = radius;
    a.collision_th = collision_th;
    a.vel = vel;
    a.acc = acc;
    a.part = part;
    a.n = n;
    a.t0 = t0;
    a.nt = nt;
    int ntiles = (n + DS_TILE - 1) / DS_TILE;
    a.tiles_per_split = (ntiles + jsplit - 1) / jsplit;
    a.jsplit = (ntiles + a.tiles_per_split - 1) / a.tiles_per_split;  // no empty splits
    a.fc = fc;
    a.ia = ia;
    if (a.jsplit > 1 && !part)
        return fail(NB_ERR_ARG, "direct_step: split sources need a partial buffer");
    dim3 grid((nt + DS_BLOCK * DS_TPT - 1) / (DS_BLOCK * DS_TPT), a.jsplit);
    if (fc.exact_denom)
        hipLaunchKernelGGL((k_direct<DS_TPT, true>), grid, dim3(DS_BLOCK), 0, s, a);
    else
        hipLaunchKernelGGL((k_direct<DS_TPT, false>), grid, dim3(DS_BLOCK), 0, s, a);
    NB_HIP(hipGetLastError());
    if (a.jsplit > 1) {
        hipLaunchKernelGGL(k_direct_finish, dim3((nt + 255) / 256), dim3(256), 0, s, a);
        NB_HIP(hipGetLastError());
    }
    return NB_OK;
}

}  // namespace nb
