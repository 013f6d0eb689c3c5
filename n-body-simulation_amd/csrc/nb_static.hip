// nb_static.hip -- nb_compute_accel_static: a force evaluation that allocates nothing.
//
// The reference has no counterpart.  Every other entry point works on buffers that nb_create
// obtained from hipMalloc, and takes its input from / returns its output to host memory.  Code
// that runs under HIP graph capture, a tight serving loop or a harness that forbids device
// allocation can do neither; this entry point takes positions and returns accelerations as device
// arrays and keeps everything in between -- the permuted positions, keys, sort buffers, tree,
// centres of mass, traversal records: what BhBuffers holds for a handle -- in one static device
// array.  The kernels are the handle's own (nb_bh.hip); they get their buffers as arguments either
// way.
#include <mutex>

#include "nb_sim.h"

using namespace nb;

// Device memory of the static path, held by every GPU the library is loaded on until the process
// ends.  In host code this name is only a handle for hipGetSymbolAddress: the array's device
// address differs per GPU and is looked up once per device (g_ptr); host code passes that address,
// never the name, to kernels.
constexpr size_t STATIC_SCRATCH_BYTES = (size_t)40 << 20;
static __device__ __align__(256) unsigned char scratch[STATIC_SCRATCH_BYTES];

namespace {

constexpr int STATIC_MAX_DEVICES = 64;
void *g_ptr[STATIC_MAX_DEVICES];  // device address of `scratch` on each device (null: not looked up yet)
std::mutex g_mu;

// positions caller order -> exchange order; the handle's stored accelerations as the starting
// value of the output (a static body keeps the acceleration it was given)
__global__ __launch_bounds__(256) void k_static_gather(const double2 *pos_in, const uint32_t *order, const double2 *acc_sim,
                                                       int n, double2 *pos_x, double2 *acc_x)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n) {
        pos_x[k] = pos_in[order[k]];
        acc_x[k] = acc_sim[k];
    }
}

__global__ __launch_bounds__(256) void k_static_scatter(const double2 *acc_x, const uint32_t *order, int n, double2 *acc_out)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k < n)
        acc_out[order[k]] = acc_x[k];
}

}  // namespace

int nb::static_scratch_init(int device)
{
    if (device < 0 || device >= STATIC_MAX_DEVICES)
        return fail(NB_ERR_ARG, "static scratch: device index beyond " + std::to_string(STATIC_MAX_DEVICES - 1));
    std::lock_guard<std::mutex> lock(g_mu);
    if (g_ptr[device])
        return NB_OK;
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !p)
        return fail(NB_ERR_HIP, std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e));
    g_ptr[device] = p;
    return NB_OK;
}

extern "C" int nb_compute_accel_static(nb_sim *s, const double *pos_xy, double *acc_xy, void *hip_stream)
{
    if (!s || !pos_xy || !acc_xy)
        return fail(NB_ERR_ARG, "nb_compute_accel_static: NULL argument");
    if (s->alg != NB_ALG_BARNES_HUT)
        return fail(NB_ERR_ARG, "nb_compute_accel_static: the handle must be a Barnes-Hut simulation");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb_compute_accel_static: no bodies were set");
    if (s->n > NB_STATIC_MAX_BODIES)
        return fail(NB_ERR_OVERFLOW, "nb_compute_accel_static: " + std::to_string(s->n) + " bodies exceed the static "
                                     "scratch array's limit of " + std::to_string(NB_STATIC_MAX_BODIES));
    int device = -1;
    NB_HIP(hipGetDevice(&device));
    if (device != s->device)
        return fail(NB_ERR_STATE, "nb_compute_accel_static: the handle's device is not the current one");
    NB_TRY(static_scratch_init(device));  // a lookup; nb_create did the work
    char *const ptr = (char *)g_ptr[device];
    hipStream_t stream = hip_stream ? (hipStream_t)hip_stream : s->stream;

    // layout of the scratch array: positions and accelerations in exchange order, then the pipeline's buffers
    const int n = s->n;
    const size_t rec_bytes = (sizeof(double2) * (size_t)n + 255) & ~(size_t)255;
    double2 *pos_x = (double2 *)ptr;
    double2 *acc_x = (double2 *)(ptr + rec_bytes);
    BhBuffers bh;
    NB_TRY(bh_carve(bh, n, ptr + 2 * rec_bytes, STATIC_SCRATCH_BYTES - 2 * rec_bytes, nullptr));

    const dim3 grid((n + 255) / 256);
    hipLaunchKernelGGL(k_static_gather, grid, dim3(256), 0, stream, (const double2 *)pos_xy, s->order_d, s->acc2d(), n, pos_x, acc_x);
    NB_HIP(hipGetLastError());
    BhStepArgs a = bh_args(s, 0);
    a.pos = pos_x;
    a.pos_out = pos_x;
    a.acc = acc_x;
    a.vel = nullptr;  // never touched without integration
    a.t0 = 0;
    a.t1 = n;
    a.collect_stats = 0;
    NB_TRY(bh_bbox(stream, bh, a));
    NB_TRY(bh_keys(stream, bh, a));
    NB_TRY(bh_sort(stream, bh, a));
    NB_TRY(bh_build(stream, bh, a));
    NB_TRY(bh_com(stream, bh, a));
    NB_TRY(bh_force(stream, bh, a));
    hipLaunchKernelGGL(k_static_scatter, grid, dim3(256), 0, stream, acc_x, s->order_d, n, (double2 *)acc_xy);
    NB_HIP(hipGetLastError());
    return NB_OK;
}
