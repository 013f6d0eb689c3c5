// nb_refresh.hip -- see nb_refresh.h.
#include "nb_refresh.h"

#include "nb_bh.h"
#include "nb_dev.h"

namespace nb {

namespace {

constexpr int RF_BLOCKS = 1024;

template <int DIMS>
__global__ __launch_bounds__(256) void k_rf_minmax(const double *pos, int n, double *part)
{
    double lo[DIMS], hi[DIMS];
#pragma unroll
    for (int d = 0; d < DIMS; ++d) {
        lo[d] = __builtin_inf();
        hi[d] = -__builtin_inf();
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256)
#pragma unroll
        for (int d = 0; d < DIMS; ++d) {
            const double v = pos[(size_t)DIMS * i + d];
            lo[d] = fmin(lo[d], v);
            hi[d] = fmax(hi[d], v);
        }
    __shared__ double sh[4][2 * DIMS];
#pragma unroll
    for (int d = 0; d < DIMS; ++d) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            lo[d] = fmin(lo[d], __shfl_xor(lo[d], o));
            hi[d] = fmax(hi[d], __shfl_xor(hi[d], o));
        }
        if (lane_id() == 0) {
            sh[threadIdx.x >> 6][d] = lo[d];
            sh[threadIdx.x >> 6][DIMS + d] = hi[d];
        }
    }
    __syncthreads();
    if (threadIdx.x < 2 * DIMS) {
        const int d = threadIdx.x;
        double v = sh[0][d];
        for (int w = 1; w < 4; ++w)
            v = d < DIMS ? fmin(v, sh[w][d]) : fmax(v, sh[w][d]);
        part[(size_t)blockIdx.x * 2 * DIMS + d] = v;
    }
}

// One block: the box of all bodies, the sort's first pass, and the middle of the box for the host.
template <int DIMS>
__global__ __launch_bounds__(256) void k_rf_box(const double *part, int nblk, double *box, uint32_t *cut, double *origin_host, int n)
{
    __shared__ double sh[256][2 * DIMS];
    double v[2 * DIMS];
#pragma unroll
    for (int d = 0; d < 2 * DIMS; ++d)
        v[d] = d < DIMS ? __builtin_inf() : -__builtin_inf();
    for (int b = threadIdx.x; b < nblk; b += 256)
#pragma unroll
        for (int d = 0; d < 2 * DIMS; ++d) {
            const double p = part[(size_t)b * 2 * DIMS + d];
            v[d] = d < DIMS ? fmin(v[d], p) : fmax(v[d], p);
        }
#pragma unroll
    for (int d = 0; d < 2 * DIMS; ++d)
        sh[threadIdx.x][d] = v[d];
    __syncthreads();
    if (threadIdx.x < 2 * DIMS) {
        const int d = threadIdx.x;
        double r = sh[0][d];
        for (int t = 1; t < 256; ++t)
            r = d < DIMS ? fmin(r, sh[t][d]) : fmax(r, sh[t][d]);
        box[d] = r;
        sh[0][d] = r;
    }
    __syncthreads();
    if (threadIdx.x < DIMS)
        origin_host[threadIdx.x] = 0.5 * (sh[0][threadIdx.x] + sh[0][DIMS + threadIdx.x]);
    if (threadIdx.x == 0) {
        cut[0] = 32u;
        cut[1] = 16u;
        cut[2] = 4u;  // the codes fill the top 32 bits of the sorted word: passes 4..7 have work
        cut[3] = 0u;
        cut[5] = n <= (1 << 30) ? 1u : 0u;  // the index rides in the 30 bits below the code (k_sort_scatter in nb_bh.hip)
    }
}

// Position of every body along a Hilbert curve through a grid of 2 x 16 bits (2-D; every axis scaled by its
// own extent) or 3 x 10 bits (3-D; one scale for all axes, so that a thin disc is not cut into slabs), in the
// top 32 bits of the word the radix sort orders.
template <int DIMS>
__global__ __launch_bounds__(256) void k_rf_keys(const double *pos, int n, const double *box, uint64_t *keys, uint32_t *idx)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n)
        return;
    uint64_t code;
    if (DIMS == 2) {
        const double ex = box[2] - box[0], ey = box[3] - box[1];
        const double sx = ex > 0 ? 65535.0 / ex : 0.0, sy = ey > 0 ? 65535.0 / ey : 0.0;
        const uint32_t qx = (uint32_t)((pos[2 * (size_t)i] - box[0]) * sx), qy = (uint32_t)((pos[2 * (size_t)i + 1] - box[1]) * sy);
        code = hilbert2_index(qx, qy, 16);
    } else {
        double ext = 0.0;
        for (int d = 0; d < 3; ++d)
            ext = fmax(ext, box[3 + d] - box[d]);
        const double sc = ext > 0 ? 1023.0 / ext : 0.0;
        uint32_t q[3];
        for (int d = 0; d < 3; ++d)
            q[d] = (uint32_t)((pos[3 * (size_t)i + d] - box[d]) * sc);
        code = hilbert3_index(q[0], q[1], q[2], 10) << 2;  // 30 bits, at the top of the 32
    }
    if (n <= (1 << 30)) {
        keys[i] = (code << 30) | (uint64_t)(uint32_t)i;  // sort_pairs orders key << 2; packed (k_rf_box's cut[5])
        return;
    }
    keys[i] = code << 30;
    idx[i] = (uint32_t)i;
}

template <typename T>
__global__ __launch_bounds__(256) void k_rf_gather(const T *src, const uint32_t *idx, int n, T *dst)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n)
        dst[k] = src[idx[k]];
}

struct Triple {
    double x, y, z;
};

}  // namespace

int refresh_alloc(RefreshBuffers &r, int n)
{
    r.n = n;
    for (int k = 0; k < 2; ++k) {
        NB_TRY(dev_alloc(r.keys[k], (size_t)n));
        NB_TRY(dev_alloc(r.idx[k], (size_t)n));
    }
    NB_TRY(dev_alloc(r.hist, sort_hist_words(n)));
    NB_TRY(dev_alloc(r.totals, sort_totals_words()));
    NB_TRY(dev_alloc(r.cut, 8));
    NB_TRY(dev_alloc(r.part, (size_t)RF_BLOCKS * 6));
    NB_TRY(dev_alloc(r.box, 6));
    char *t = nullptr;
    NB_TRY(dev_alloc(t, (size_t)n * 24));
    r.tmp = t;
    void *h = nullptr;
    if (hipHostMalloc(&h, 64, hipHostMallocDefault) != hipSuccess)
        return fail(NB_ERR_NOMEM, "hipHostMalloc failed");
    r.origin_host = (double *)h;
    for (int d = 0; d < 8; ++d)
        r.origin_host[d] = 0.0;
    if (hipEventCreateWithFlags(&r.box_done, hipEventDisableTiming) != hipSuccess)
        return fail(NB_ERR_HIP, "hipEventCreate failed");
    r.origin_set = false;
    return NB_OK;
}

void refresh_free(RefreshBuffers &r)
{
    for (int k = 0; k < 2; ++k) {
        (void)hipFree(r.keys[k]);
        (void)hipFree(r.idx[k]);
    }
    (void)hipFree(r.hist);
    (void)hipFree(r.totals);
    (void)hipFree(r.cut);
    (void)hipFree(r.part);
    (void)hipFree(r.box);
    (void)hipFree(r.tmp);
    if (r.origin_host)
        (void)hipHostFree(r.origin_host);
    if (r.box_done)
        (void)hipEventDestroy(r.box_done);
    r = RefreshBuffers();
}

int refresh_sort(hipStream_t s, RefreshBuffers &r, const double *pos, int dims, int n)
{
    if (n != r.n || (dims != 2 && dims != 3))
        return fail(NB_ERR_ARG, "refresh_sort: bad arguments");
    int blocks = (n + 255) / 256;
    if (blocks > RF_BLOCKS)
        blocks = RF_BLOCKS;
    const dim3 grid((n + 255) / 256);
    if (dims == 2) {
        hipLaunchKernelGGL(k_rf_minmax<2>, dim3(blocks), dim3(256), 0, s, pos, n, r.part);
        hipLaunchKernelGGL(k_rf_box<2>, dim3(1), dim3(256), 0, s, r.part, blocks, r.box, r.cut, r.origin_host, n);
        hipLaunchKernelGGL(k_rf_keys<2>, grid, dim3(256), 0, s, pos, n, r.box, r.keys[0], r.idx[0]);
    } else {
        hipLaunchKernelGGL(k_rf_minmax<3>, dim3(blocks), dim3(256), 0, s, pos, n, r.part);
        hipLaunchKernelGGL(k_rf_box<3>, dim3(1), dim3(256), 0, s, r.part, blocks, r.box, r.cut, r.origin_host, n);
        hipLaunchKernelGGL(k_rf_keys<3>, grid, dim3(256), 0, s, pos, n, r.box, r.keys[0], r.idx[0]);
    }
    NB_HIP(hipGetLastError());
    NB_HIP(hipEventRecord(r.box_done, s));  // the box kernel is the second of the three: origin_host is written by then
    r.origin_set = true;
    return sort_pairs(s, r.keys, r.idx, n, r.cut, r.hist, r.totals, 4);  // passes 4..7 (k_rf_box's cut[2]); result in buffer 0
}

int refresh_gather(hipStream_t s, RefreshBuffers &r, void *array, size_t elem_bytes, int n)
{
    if (!array)
        return NB_OK;
    const dim3 grid((n + 255) / 256);
    const uint32_t *idx = r.idx[0];
    switch (elem_bytes) {
    case 1:
        hipLaunchKernelGGL(k_rf_gather<uint8_t>, grid, dim3(256), 0, s, (const uint8_t *)array, idx, n, (uint8_t *)r.tmp);
        break;
    case 4:
        hipLaunchKernelGGL(k_rf_gather<uint32_t>, grid, dim3(256), 0, s, (const uint32_t *)array, idx, n, (uint32_t *)r.tmp);
        break;
    case 8:
        hipLaunchKernelGGL(k_rf_gather<double>, grid, dim3(256), 0, s, (const double *)array, idx, n, (double *)r.tmp);
        break;
    case 16:
        hipLaunchKernelGGL(k_rf_gather<double2>, grid, dim3(256), 0, s, (const double2 *)array, idx, n, (double2 *)r.tmp);
        break;
    case 24:
        hipLaunchKernelGGL(k_rf_gather<Triple>, grid, dim3(256), 0, s, (const Triple *)array, idx, n, (Triple *)r.tmp);
        break;
    default:
        return fail(NB_ERR_ARG, "refresh_gather: unsupported element size");
    }
    NB_HIP(hipGetLastError());
    NB_HIP(hipMemcpyAsync(array, r.tmp, elem_bytes * (size_t)n, hipMemcpyDeviceToDevice, s));
    return NB_OK;
}

}  // namespace nb
