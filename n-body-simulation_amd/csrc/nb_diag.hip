// nb_diag.hip -- diagnostics that are not on the hot path: the potential energy of the current
// state as an FP64 all-pairs sum (the reference has no diagnostics at all; this exists to check
// the integrators).  Pairs the force law skips (closer than r_i + r_j + collision_th,
// directSum.cpp:69-72) are skipped here too.
#include "nb_common.h"

namespace nb {

__global__ __launch_bounds__(256) void k_potential(const double2 *pos, const double *mass, const double *radius, int n,
                                                   double gravity, double eps2, double cth, double *out)
{
    __shared__ double2 sp[256];
    __shared__ double sm[256], sr[256];
    __shared__ double wsum[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    const double2 pi = live ? pos[i] : make_double2(0.0, 0.0);
    const double mi = live ? mass[i] : 0.0, ri = live ? radius[i] : 0.0;
    double e = 0.0;
    for (int base = 0; base < n; base += 256) {
        const int j = base + threadIdx.x;
        sp[threadIdx.x] = j < n ? pos[j] : make_double2(0.0, 0.0);
        sm[threadIdx.x] = j < n ? mass[j] : 0.0;
        sr[threadIdx.x] = j < n ? radius[j] : 0.0;
        __syncthreads();
        const int cnt = n - base < 256 ? n - base : 256;
        for (int k = 0; k < cnt; ++k) {
            const int jj = base + k;
            const double dx = sp[k].x - pi.x, dy = sp[k].y - pi.y;
            const double d2 = dx * dx + dy * dy;
            if (live && jj > i && !(ri + sr[k] + cth >= sqrt(d2)))
                e -= gravity * mi * sm[k] / sqrt(d2 + eps2);
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        e += __shfl_xor(e, o);
    if ((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(out, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

// The same sum over packed triples (the three-dimensional state).
__global__ __launch_bounds__(256) void k_potential3(const double *pos, const double *mass, const double *radius, int n,
                                                    double gravity, double eps2, double cth, double *out)
{
    __shared__ double sp[256][3];
    __shared__ double sm[256], sr[256];
    __shared__ double wsum[4];
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool live = i < n;
    double pi[3] = {0.0, 0.0, 0.0};
    if (live)
        for (int d = 0; d < 3; ++d)
            pi[d] = pos[3 * (size_t)i + d];
    const double mi = live ? mass[i] : 0.0, ri = live ? radius[i] : 0.0;
    double e = 0.0;
    for (int base = 0; base < n; base += 256) {
        const int j = base + threadIdx.x;
        for (int d = 0; d < 3; ++d)
            sp[threadIdx.x][d] = j < n ? pos[3 * (size_t)j + d] : 0.0;
        sm[threadIdx.x] = j < n ? mass[j] : 0.0;
        sr[threadIdx.x] = j < n ? radius[j] : 0.0;
        __syncthreads();
        const int cnt = n - base < 256 ? n - base : 256;
        for (int k = 0; k < cnt; ++k) {
            const int jj = base + k;
            const double dx = sp[k][0] - pi[0], dy = sp[k][1] - pi[1], dz = sp[k][2] - pi[2];
            const double d2 = dx * dx + dy * dy + dz * dz;
            if (live && jj > i && !(ri + sr[k] + cth >= sqrt(d2)))
                e -= gravity * mi * sm[k] / sqrt(d2 + eps2);
        }
        __syncthreads();
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
        e += __shfl_xor(e, o);
    if ((threadIdx.x & 63) == 0)
        wsum[threadIdx.x >> 6] = e;
    __syncthreads();
    if (threadIdx.x == 0)
        atomicAdd(out, wsum[0] + wsum[1] + wsum[2] + wsum[3]);
}

int potential_energy3(hipStream_t s, const double *pos, const double *mass, const double *radius, int n, double gravity,
                      double eps, double collision_th, double *out)
{
    NB_HIP(hipMemsetAsync(out, 0, sizeof(double), s));
    hipLaunchKernelGGL(k_potential3, dim3((n + 255) / 256), dim3(256), 0, s, pos, mass, radius, n, gravity, eps * eps,
                       collision_th, out);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

int potential_energy(hipStream_t s, const double2 *pos, const double *mass, const double *radius, int n, double gravity,
                     double eps, double collision_th, double *out)
{
    NB_HIP(hipMemsetAsync(out, 0, sizeof(double), s));
    hipLaunchKernelGGL(k_potential, dim3((n + 255) / 256), dim3(256), 0, s, pos, mass, radius, n, gravity, eps * eps,
                       collision_th, out);
    NB_HIP(hipGetLastError());
    return NB_OK;
}

}  // namespace nb
