// nb_handle.h -- what the two-dimensional and the three-dimensional simulation handle share
// (internal): the per-body state on the device, and the entry points of include/nbody.h written
// once for both.  nb_sim (nb_sim.h) and nb_sim3 (nb_api3.hip) derive from SimCore<2> / SimCore<3>
// and add what differs: the body type, the force pipelines' buffers and launches, and in 2-D the
// cell exchange and the reference GPU program's tree.
//
// A handle type S offers to the functions below:
//   S::PFX                       "nb_" or "nb3_": the prefix of the entry points' names in messages
//   S::Body                      nb_body or nb_body3
//   S::coord / S::velocity / S::acceleration (const Body &, d), S::store(Body &, p, v, a)
//   s->static_fixed()            static bodies keep a = 0 and their place (2-D bucket_depth mode)
//   s->direct_step(do_integrate), s->bh_step(do_integrate)   one force evaluation or step
//   s->direct_pack_all()         float records of all bodies from the FP64 state, both buffers
//   s->direct_pack_others()      ... of the bodies other ranks advanced, into the next buffer
//   s->direct_flip()             the next buffer becomes the current one
//   s->direct_set_split(nt)      the source split for nt targets per rank
//   s->upload_extra(mass, radius), s->gather_extra(), s->after_load()   per-dimension arrays
//   s->count_step()              a step was finished
//   s->keeps_order_fresh()       the steps leave the bodies in the order of their own sort: no renewal
#pragma once

#include <algorithm>
#include <cmath>
#include <cstring>
#include <string>
#include <vector>

#include "nb_comm.h"
#include "nb_common.h"
#include "nb_refresh.h"
#include "nb_timing.h"

namespace nb {

template <int DIMS>
struct SimCore {
    static constexpr int D = DIMS;
    nb_params p;
    int n = 0;
    int alg = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool have_bodies = false;
    bool first_kick = true;   // leapfrog: the next step is the first since the bodies were set
    bool stepped = false;     // a step was taken since the bodies were set
    long steps_in_order = 0;  // steps since the exchange order was last derived from the positions

    Units units{};
    ForceConsts fc{};

    // FP64 state in exchange order, D doubles per body and array; room for world * chunk records
    size_t cap = 0;
    double *pos = nullptr, *vel = nullptr, *acc = nullptr;
    double *pos2 = nullptr;  // direct sum: the other position buffer (pos is the current one)
    double *mass = nullptr, *radius = nullptr;
    uint8_t *dyn = nullptr;
    std::vector<uint32_t> order;  // caller index of the body in each exchange-order slot
    uint32_t *order_d = nullptr;  // the same on the device
    bool order_stale = false;     // the device renewed the order; `order` is fetched when next needed
    RefreshBuffers rb;            // buffers of the renewal (made at creation for the direct sum, else on first use)

    // partition: this process integrates slots [t0, t1)
    int rank = 0, world = 1;
    size_t chunk = 0;
    int t0 = 0, t1 = 0;
    bool mid_step = false;
    bool own_loop = false;  // the step in flight is one of nb_step / nb_step_distributed (which may move bodies between slots)
    Comm *comm = nullptr;  // RCCL communicator of the comm_init entry point (null: none)
    size_t part_cap = 0;   // direct sum: partial sums allocated

    int collect_stats = 0;
    StageTimer timer;

    // Barnes-Hut: the host sizes a step's launches by what the device reported about the steps
    // before it; it may enqueue at most PACE_AHEAD steps beyond the last finished one, so that what
    // it reads is never older than that (enough to keep the GPU fed: a step's launches take the
    // host a small fraction of the step's time on the device).
    static constexpr int PACE_AHEAD = 3, PACE_RING = 4;
    hipEvent_t pace[PACE_RING] = {};
    long paced = 0;  // steps whose end was marked
};

// Before the launches of a Barnes-Hut step: waits until at most PACE_AHEAD - 1 marked steps are unfinished.
template <class S>
int pace_wait(S *s)
{
    if (s->alg == NB_ALG_BARNES_HUT && s->paced >= S::PACE_AHEAD)
        NB_HIP(hipEventSynchronize(s->pace[(s->paced - S::PACE_AHEAD) % S::PACE_RING]));
    return NB_OK;
}

// After the launches of a Barnes-Hut step.
template <class S>
int pace_mark(S *s)
{
    if (s->alg == NB_ALG_BARNES_HUT) {
        NB_HIP(hipEventRecord(s->pace[s->paced % S::PACE_RING], s->stream));
        s->paced += 1;
    }
    return NB_OK;
}

template <class S>
std::string fn(const char *name)
{
    return std::string(S::PFX) + name;
}

// The origin of the internal coordinates (the middle of the bodies' bounding box) and the powers
// of two that bring the largest coordinate from there to [4, 8) and the largest mass to about 2^20.
template <class S>
void choose_units(S *s, const typename S::Body *b, int n)
{
    constexpr int D = S::D;
    double lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
    for (int d = 0; d < D; ++d)
        lo[d] = hi[d] = S::coord(b[0], d);
    for (int i = 1; i < n; ++i)
        for (int d = 0; d < D; ++d) {
            lo[d] = std::min(lo[d], S::coord(b[i], d));
            hi[d] = std::max(hi[d], S::coord(b[i], d));
        }
    for (int d = 0; d < 3; ++d)
        s->units.origin[d] = d < D ? 0.5 * (lo[d] + hi[d]) : 0.0;
    double maxc = 1.0, maxm = 1.0;
    for (int i = 0; i < n; ++i) {
        for (int d = 0; d < D; ++d)
            maxc = std::max(maxc, std::fabs(S::coord(b[i], d) - s->units.origin[d]));
        maxm = std::max(maxm, std::fabs(b[i].mass));
    }
    s->units.len_unit = std::ldexp(1.0, (int)std::floor(std::log2(maxc)) - 2);
    s->units.mass_unit = std::ldexp(1.0, (int)std::floor(std::log2(maxm)) - 20);
    s->units.acc_factor = s->p.gravity * s->units.mass_unit / (s->units.len_unit * s->units.len_unit);
    const double lu = s->units.len_unit;
    s->fc.eps2 = (float)(s->p.eps * s->p.eps / (lu * lu));
    s->fc.eps3 = (float)(s->p.eps * s->p.eps / (lu * lu * lu));
    s->fc.cth = (float)(s->p.collision_th / lu);
    // Every evaluated pair is farther apart than collision_th, so r^3 >= cth^3: eps3 below
    // 2^-26 of that cannot change a float sum r^3 + eps3 (and eps2 below 2^-26 of cth^2 none r^2 + eps2).
    const double c3 = (double)s->fc.cth * s->fc.cth * s->fc.cth;
    s->fc.exact_denom = !((double)s->fc.eps3 <= std::ldexp(c3, -26));
    s->fc.eps2_negligible = (double)s->fc.eps2 <= std::ldexp((double)s->fc.cth * s->fc.cth, -26);
    // coordinates lie in [-8, 8): their float rounding is up to 2.4e-7 units, i.e. a relative 5e-6 of a
    // distance of 0.05 units (the reference's cutoff in its own scenarios is 0.145 units)
    s->fc.precise_near = s->fc.cth < 0.05f;
}

// Exchange order = bodies sorted by their position along a Hilbert curve (nb_common.h) through a grid over
// their initial positions (2 x 16 bits, every axis scaled to its own range; 3 x 10 bits with one scale for
// all axes, so that a thin disc is not cut into slabs).  Index ranges of it are compact patches of space, which keeps the 64 bodies a
// wave walks the tree for close together when the range is cut across GPUs, and gives the
// direct-sum runs small bounding boxes.  Only locality matters here, not the reference's cells.
template <class S>
void spatial_order(const typename S::Body *b, int n, std::vector<uint32_t> &order)
{
    constexpr int D = S::D;
    order.resize(n);
    double lo[3], hi[3];
    for (int d = 0; d < D; ++d)
        lo[d] = hi[d] = S::coord(b[0], d);
    for (int i = 1; i < n; ++i)
        for (int d = 0; d < D; ++d) {
            lo[d] = std::min(lo[d], S::coord(b[i], d));
            hi[d] = std::max(hi[d], S::coord(b[i], d));
        }
    double sc[3];
    if (D == 2) {
        for (int d = 0; d < D; ++d)
            sc[d] = hi[d] > lo[d] ? 65535.0 / (hi[d] - lo[d]) : 0.0;
    } else {
        double ext = 0.0;
        for (int d = 0; d < D; ++d)
            ext = std::max(ext, hi[d] - lo[d]);
        for (int d = 0; d < D; ++d)
            sc[d] = ext > 0 ? 1023.0 / ext : 0.0;
    }
    std::vector<uint64_t> kv(n);
    for (int i = 0; i < n; ++i) {
        uint32_t q[3] = {0u, 0u, 0u};
        for (int d = 0; d < D; ++d)
            q[d] = (uint32_t)((S::coord(b[i], d) - lo[d]) * sc[d]);
        const uint64_t code = D == 2 ? hilbert2_index(q[0], q[1], 16) : hilbert3_index(q[0], q[1], q[2], 10);
        kv[i] = (code << 32) | (uint32_t)i;
    }
    std::sort(kv.begin(), kv.end());
    for (int i = 0; i < n; ++i)
        order[i] = (uint32_t)(kv[i] & 0xffffffffu);
}

// Arguments of the fused integration of a force evaluation (do_integrate = 0) or a step (1).
template <class S>
IntegrateArgs make_integrate_args(const S *s, int do_integrate)
{
    IntegrateArgs ia;
    ia.acc_factor = s->units.acc_factor;
    ia.dt = s->p.dt;
    ia.dt_kick = (s->p.integrator == NB_INT_LEAPFROG && s->first_kick) ? 0.5 * s->p.dt : s->p.dt;
    ia.inv_len_unit = 1.0 / s->units.len_unit;
    for (int d = 0; d < 3; ++d)
        ia.origin[d] = s->units.origin[d];
    ia.do_integrate = do_integrate;
    ia.static_fixed = s->static_fixed() ? 1 : 0;
    return ia;
}

// Steps between two renewals of the exchange order (device_refresh).  Direct sum: its run-against-wave
// box tests rely on slot ranges being compact patches of space, and bodies wander: at N = 65 536 the
// spiral loses 18 % of its speed within 760 steps (4.72 -> 3.89 T interactions/s) and holds 4.7 when
// the order is renewed every 50.  A renewal costs a few passes over the per-body arrays and some
// twenty launches: every 32 steps where a step is long against that, less often for small systems,
// whose steps are as short as the renewal (about 0.2 % of the time either way).
// Barnes-Hut sorts every step anyway, but its leaf pass and the traversal's epilogue gather and
// scatter the FP64 state through that sort's permutation, which drifts away from the identity as
// the bodies mix (collision discs N = 5 M: 1.74 -> 1.65 ms/step at step 300, 2.21 -> 1.95 at step
// 620 right after a renewal, which takes 0.9 ms): every 256 steps.
inline long refresh_interval(int n, int alg)
{
    if (alg != NB_ALG_DIRECT)
        return 256;
    const double interval = 1.4e11 / ((double)n * (double)n);
    return interval < 32.0 ? 32 : (interval > 4096.0 ? 4096 : (long)interval);
}

template <class S>
int step_local(S *s, int do_integrate)
{
    NB_TRY(pace_wait(s));
    return s->alg == NB_ALG_DIRECT ? s->direct_step(do_integrate) : s->bh_step(do_integrate);
}

// After the positions other ranks advanced have been gathered into the new position buffer.
template <class S>
int step_finish(S *s)
{
    s->first_kick = false;
    s->stepped = true;
    s->count_step();
    if (s->alg == NB_ALG_DIRECT) {
        if (s->world > 1) {  // records of the bodies other ranks advanced come from the gathered positions
            s->timer.mark(s->stream, 3);  // (timed as the direct sum's "build" stage)
            NB_TRY(s->direct_pack_others());
            s->timer.mark(s->stream, StageTimer::STAGE_END);
        }
        s->direct_flip();
        std::swap(s->pos, s->pos2);
    }
    return pace_mark(s);
}

template <class S>
int collect_timing(S *s)
{
    return s->timer.step_done(s->stream);
}

template <class S>
void release(S *s)
{
    (void)hipFree(s->pos);
    (void)hipFree(s->pos2);
    (void)hipFree(s->vel);
    (void)hipFree(s->acc);
    (void)hipFree(s->mass);
    (void)hipFree(s->radius);
    (void)hipFree(s->dyn);
    (void)hipFree(s->order_d);
    s->free_pipelines();
    refresh_free(s->rb);
    comm_destroy(s->comm);
    s->comm = nullptr;
    s->timer.destroy();
    for (auto &e : s->pace)
        if (e) {
            (void)hipEventDestroy(e);
            e = nullptr;
        }
    if (s->own_stream && s->stream)
        (void)hipStreamDestroy(s->stream);
}

// Units, exchange order and device arrays from bodies given in the caller's order.
template <class S>
int load_bodies(S *s, const typename S::Body *b, int n)
{
    constexpr int D = S::D;
    choose_units(s, b, n);
    spatial_order<S>(b, n, s->order);
    std::vector<double> pos(D * (size_t)n), vel(D * (size_t)n), acc(D * (size_t)n), mass(n), radius(n);
    std::vector<uint8_t> dyn(n);
    for (int k = 0; k < n; ++k) {
        const typename S::Body &x = b[s->order[k]];
        for (int d = 0; d < D; ++d) {
            pos[D * (size_t)k + d] = S::coord(x, d);
            vel[D * (size_t)k + d] = S::velocity(x, d);
            acc[D * (size_t)k + d] = S::acceleration(x, d);
        }
        mass[k] = x.mass;
        radius[k] = x.radius;
        dyn[k] = x.is_dynamic ? 1 : 0;
    }
    const size_t vec = sizeof(double) * D * (size_t)n;
    NB_HIP(hipStreamSynchronize(s->stream));
    NB_HIP(hipMemcpy(s->pos, pos.data(), vec, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->vel, vel.data(), vec, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->acc, acc.data(), vec, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->mass, mass.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->radius, radius.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->dyn, dyn.data(), n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->order_d, s->order.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    s->order_stale = false;
    NB_TRY(s->upload_extra(mass.data(), radius.data()));
    if (s->alg == NB_ALG_DIRECT)
        NB_TRY(s->direct_pack_all());
    NB_HIP(hipStreamSynchronize(s->stream));
    NB_TRY(s->after_load());
    s->steps_in_order = 0;
    s->rb.origin_set = false;  // the box of a renewal of the previous bodies says nothing about these
    return NB_OK;
}

// The slot -> caller index map on the host (fetched again after the device renewed it).
template <class S>
int host_order(S *s)
{
    if (!s->order_stale)
        return NB_OK;
    NB_HIP(hipStreamSynchronize(s->stream));
    NB_HIP(hipMemcpy(s->order.data(), s->order_d, sizeof(uint32_t) * s->n, hipMemcpyDeviceToHost));
    s->order_stale = false;
    return NB_OK;
}

// Renews the exchange order on the device, everything enqueued on the stream (nb_refresh.h): new
// order from the current positions, every per-body array gathered into it, the direct sum's
// records packed again.  The float records' origin moves to the middle of the box the PREVIOUS
// renewal saw, so a drifting system keeps its precision.  The device leaves that box in
// host-visible memory; the host reads it only behind the event recorded after the kernel that wrote
// it (a renewal is at least 32 steps old by then, so the wait ends at once unless the host has run
// a whole interval ahead of the device).  Renewal k therefore always takes the box of renewal
// k - 1, however far ahead the host runs: the same values on every run and on every rank.
template <class S>
int device_refresh(S *s)
{
    constexpr int D = S::D;
    if (s->rb.n != s->n)
        NB_TRY(refresh_alloc(s->rb, s->n));
    if (s->rb.origin_set && s->alg == NB_ALG_DIRECT) {
        NB_HIP(hipEventSynchronize(s->rb.box_done));
        for (int d = 0; d < D; ++d)
            s->units.origin[d] = ((volatile double *)s->rb.origin_host)[d];
    }
    NB_TRY(refresh_sort(s->stream, s->rb, s->pos, D, s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->pos, D * sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->vel, D * sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->acc, D * sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->mass, sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->radius, sizeof(double), s->n));
    NB_TRY(s->gather_extra());
    NB_TRY(refresh_gather(s->stream, s->rb, s->dyn, 1, s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->order_d, sizeof(uint32_t), s->n));
    s->order_stale = true;
    if (s->alg == NB_ALG_DIRECT)
        NB_TRY(s->direct_pack_all());
    s->steps_in_order = 0;
    return NB_OK;
}

// The device state in the caller's order; half_kick > 0 adds a(x) * half_kick to the velocities
// of the slots this rank integrates (other ranks' velocities are not held here).
template <class S>
int read_state(S *s, typename S::Body *out, double half_kick)
{
    constexpr int D = S::D;
    const int n = s->n;
    std::vector<double> pos(D * (size_t)n), vel(D * (size_t)n), acc(D * (size_t)n), mass(n), radius(n);
    std::vector<uint8_t> dyn(n);
    const size_t vec = sizeof(double) * D * (size_t)n;
    NB_HIP(hipStreamSynchronize(s->stream));
    NB_TRY(host_order(s));
    NB_HIP(hipMemcpy(pos.data(), s->pos, vec, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(vel.data(), s->vel, vec, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(acc.data(), s->acc, vec, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(mass.data(), s->mass, sizeof(double) * n, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(radius.data(), s->radius, sizeof(double) * n, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(dyn.data(), s->dyn, n, hipMemcpyDeviceToHost));
    for (int k = 0; k < n; ++k) {
        typename S::Body &x = out[s->order[k]];
        std::memset(&x, 0, sizeof(x));
        x.is_dynamic = dyn[k];
        x.mass = mass[k];
        x.radius = radius[k];
        const double *p = &pos[D * (size_t)k], *a = &acc[D * (size_t)k];
        double v[3] = {0, 0, 0};
        const double hk = (k >= s->t0 && k < s->t1) ? half_kick : 0.0;
        for (int d = 0; d < D; ++d)
            v[d] = hk != 0.0 ? vel[D * (size_t)k + d] + hk * a[d] : vel[D * (size_t)k + d];
        S::store(x, p, v, a);
    }
    return NB_OK;
}

// ---- entry points ----

// Checks shared by the create entry points; *out is a new handle with the common arrays allocated
// and S's own (alloc_pipelines) too.
template <class S>
int api_create(const nb_params *params, int n, int alg, S **out)
{
    if (!out)
        return fail(NB_ERR_ARG, fn<S>("create: out is NULL"));
    *out = nullptr;
    if (n < 1)
        return fail(NB_ERR_ARG, "ERROR: need to have at least 1 body");
    if (alg != NB_ALG_DIRECT && alg != NB_ALG_BARNES_HUT)
        return fail(NB_ERR_ARG, fn<S>("create: alg must be 0 (direct sum) or 1 (Barnes-Hut)"));
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(NB_ERR_HIP, fn<S>("create: no HIP device is visible; this library has no CPU path"));
    nb_params p;
    if (params)
        p = *params;
    else
        nb_default_params(&p);
    if (!(p.theta >= 0) || !(p.collision_th >= 0) || !(p.eps >= 0))
        return fail(NB_ERR_ARG, fn<S>("create: theta, collision_th and eps must be >= 0"));
    if (p.integrator != NB_INT_EULER && p.integrator != NB_INT_LEAPFROG)
        return fail(NB_ERR_ARG, fn<S>("create: integrator must be NB_INT_EULER or NB_INT_LEAPFROG"));
    NB_TRY(S::check_params(p));
    S *s = new S();
    s->p = p;
    s->n = n;
    s->alg = alg;
    s->t1 = n;
    s->chunk = (size_t)n;
    s->cap = (size_t)n + 64;  // room for world * chunk records (world <= 64)
    int rc = NB_OK;
    auto chk = [&](int r) {
        if (rc == NB_OK)
            rc = r;
    };
    if (hipGetDevice(&s->device) != hipSuccess)
        chk(fail(NB_ERR_HIP, "hipGetDevice failed"));
    if (rc == NB_OK && hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess)
        chk(fail(NB_ERR_HIP, "hipStreamCreate failed"));
    s->own_stream = rc == NB_OK;
    for (auto &e : s->pace)
        if (rc == NB_OK && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
            chk(fail(NB_ERR_HIP, "hipEventCreate failed"));
    constexpr int D = S::D;
    chk(dev_alloc(s->pos, D * s->cap));
    chk(dev_alloc(s->vel, D * s->cap));
    chk(dev_alloc(s->acc, D * s->cap));
    chk(dev_alloc(s->mass, s->cap));
    chk(dev_alloc(s->radius, s->cap));
    chk(dev_alloc(s->dyn, s->cap));
    chk(dev_alloc(s->order_d, s->cap));
    if (alg == NB_ALG_DIRECT) {
        chk(dev_alloc(s->pos2, D * s->cap));
        chk(refresh_alloc(s->rb, n));  // the step entry points renew the order by themselves for the direct sum
    }
    if (rc == NB_OK)
        chk(s->alloc_pipelines());
    if (rc == NB_OK)
        chk(s->timer.create());
    if (rc != NB_OK) {
        release(s);
        delete s;
        return rc;
    }
    *out = s;
    return NB_OK;
}

template <class S>
int api_destroy(S *s)
{
    if (!s)
        return NB_OK;
    (void)hipStreamSynchronize(s->stream);
    release(s);
    delete s;
    return NB_OK;
}

template <class S>
int api_set_stream(S *s, void *hip_stream)
{
    if (!s)
        return fail(NB_ERR_ARG, fn<S>("set_stream: sim is NULL"));
    NB_HIP(hipStreamSynchronize(s->stream));
    if (s->own_stream && s->stream)
        (void)hipStreamDestroy(s->stream);
    s->stream = (hipStream_t)hip_stream;
    s->own_stream = false;
    return NB_OK;
}

template <class S>
int api_set_bodies(S *s, const typename S::Body *b, int n)
{
    if (!s || !b)
        return fail(NB_ERR_ARG, fn<S>("set_bodies: NULL argument"));
    if (n != s->n)
        return fail(NB_ERR_ARG, fn<S>("set_bodies: n differs from the n given to ") + S::PFX + "create");
    for (int i = 0; i < n; ++i) {
        bool finite = std::isfinite(b[i].mass);
        for (int d = 0; d < S::D; ++d)
            finite = finite && std::isfinite(S::coord(b[i], d));
        if (!finite)
            return fail(NB_ERR_ARG, fn<S>("set_bodies: body ") + std::to_string(i) + " is not finite");
    }
    NB_TRY(load_bodies(s, b, n));
    s->have_bodies = true;
    s->mid_step = false;
    s->first_kick = true;
    s->stepped = false;
    return NB_OK;
}

template <class S>
int api_refresh_order(S *s)
{
    if (!s)
        return fail(NB_ERR_ARG, fn<S>("refresh_order: sim is NULL"));
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, fn<S>("refresh_order: no bodies were set"));
    if (s->mid_step)
        return fail(NB_ERR_STATE, fn<S>("refresh_order: a split step is in flight"));
    // (a partitioned simulation: the caller has all-gathered the velocity buffer, see nbody.h)
    return device_refresh(s);
}

template <class S>
int api_buffer(S *s, const char *what, double *ptr, void **device_ptr, size_t *chunk_records, size_t *record_bytes)
{
    if (!s || !device_ptr || !chunk_records || !record_bytes)
        return fail(NB_ERR_ARG, fn<S>(what) + ": NULL argument");
    *device_ptr = ptr;
    *chunk_records = s->chunk;
    *record_bytes = S::D * sizeof(double);
    return NB_OK;
}

template <class S>
int api_step_local(S *s)
{
    if (!s)
        return fail(NB_ERR_ARG, fn<S>("step_local: sim is NULL"));
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, fn<S>("step_local: no bodies were set"));
    if (s->mid_step)
        return fail(NB_ERR_STATE, fn<S>("step_local: the previous step was not finished"));
    NB_TRY(step_local(s, 1));
    s->mid_step = true;
    return NB_OK;
}

template <class S>
int api_step_finish(S *s)
{
    if (!s)
        return fail(NB_ERR_ARG, fn<S>("step_finish: sim is NULL"));
    if (!s->mid_step)
        return fail(NB_ERR_STATE, fn<S>("step_finish: ") + S::PFX + "step_local was not called");
    NB_TRY(step_finish(s));
    s->mid_step = false;
    return collect_timing(s);
}

template <class S>
int api_step(S *s, int nsteps)
{
    if (!s)
        return fail(NB_ERR_ARG, fn<S>("step: sim is NULL"));
    if (nsteps < 0)
        return fail(NB_ERR_ARG, fn<S>("step: nsteps < 0"));
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, fn<S>("step: no bodies were set"));
    if (s->world != 1)
        return fail(NB_ERR_STATE, fn<S>("step: a partitioned simulation steps with ") + S::PFX + "step_local / " + S::PFX +
                                      "step_finish");
    if (s->mid_step)
        return fail(NB_ERR_STATE, fn<S>("step: a split step is in flight"));
    const long interval = refresh_interval(s->n, s->alg);
    for (int k = 0; k < nsteps; ++k) {
        s->own_loop = true;
        const int rc = step_local(s, 1);
        s->own_loop = false;
        NB_TRY(rc);
        NB_TRY(step_finish(s));
        NB_TRY(collect_timing(s));
        if (!s->keeps_order_fresh() && ++s->steps_in_order >= interval)
            NB_TRY(device_refresh(s));
    }
    return NB_OK;
}

template <class S>
int api_set_partition(S *s, int rank, int world)
{
    if (!s)
        return fail(NB_ERR_ARG, fn<S>("set_partition: sim is NULL"));
    if (world < 1 || world > 64 || rank < 0 || rank >= world)
        return fail(NB_ERR_ARG, fn<S>("set_partition: need 0 <= rank < world <= 64"));
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, fn<S>("set_partition: set the bodies first"));
    if (s->mid_step)
        return fail(NB_ERR_STATE, fn<S>("set_partition: a split step is in flight"));
    if (s->stepped)
        return fail(NB_ERR_STATE, fn<S>("set_partition: the simulation was stepped already; set the bodies again first"));
    s->rank = rank;
    s->world = world;
    s->chunk = ((size_t)s->n + world - 1) / world;
    s->t0 = (int)std::min((size_t)s->n, rank * s->chunk);
    s->t1 = (int)std::min((size_t)s->n, (rank + 1) * s->chunk);
    if (s->alg == NB_ALG_DIRECT)  // fewer targets per rank: split the sources further so the chip stays full
        NB_TRY(s->direct_set_split(s->t1 - s->t0));
    return NB_OK;
}

template <class S>
int api_comm_init(S *s, int rank, int world, const void *id)
{
    if (!s || !id)
        return fail(NB_ERR_ARG, fn<S>("comm_init: NULL argument"));
    if (s->comm)
        return fail(NB_ERR_STATE, fn<S>("comm_init: the simulation already has a communicator"));
    int device = -1;
    NB_HIP(hipGetDevice(&device));
    if (device != s->device)
        return fail(NB_ERR_STATE, fn<S>("comm_init: the handle's device is not the current one"));
    NB_TRY(api_set_partition(s, rank, world));
    return comm_create(&s->comm, rank, world, id);
}

template <class S>
int check_distributed(S *s, int nsteps)
{
    if (!s)
        return fail(NB_ERR_ARG, fn<S>("step_distributed: sim is NULL"));
    if (nsteps < 0)
        return fail(NB_ERR_ARG, fn<S>("step_distributed: nsteps < 0"));
    if (!s->comm)
        return fail(NB_ERR_STATE, fn<S>("step_distributed: call ") + S::PFX + "comm_init first");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, fn<S>("step_distributed: no bodies were set"));
    if (s->mid_step)
        return fail(NB_ERR_STATE, fn<S>("step_distributed: a split step is in flight"));
    return NB_OK;
}

// One step of every rank with the position exchange done by the library's communicator.
template <class S>
int distributed_position_step(S *s)
{
    s->own_loop = true;
    const int rc = step_local(s, 1);
    s->own_loop = false;
    NB_TRY(rc);
    // the new positions of every rank's slice, in place, ordered on the stream with the kernels
    // before and after: the host does not wait
    double *buf = s->alg == NB_ALG_DIRECT ? s->pos2 : s->pos;
    NB_TRY(comm_all_gather_f64(s->comm, buf, S::D * s->chunk, s->stream));
    s->timer.mark(s->stream, StageTimer::STAGE_END);
    NB_TRY(step_finish(s));
    NB_TRY(collect_timing(s));
    if (!s->keeps_order_fresh() && ++s->steps_in_order >= refresh_interval(s->n, s->alg)) {
        // every rank renews the order from the same positions; the new slices hold bodies other
        // ranks advanced so far, whose velocities come with one more all-gather
        NB_TRY(comm_all_gather_f64(s->comm, s->vel, S::D * s->chunk, s->stream));
        NB_TRY(device_refresh(s));
    }
    return NB_OK;
}

template <class S>
int api_compute_accel(S *s)
{
    if (!s)
        return fail(NB_ERR_ARG, fn<S>("compute_accel: sim is NULL"));
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, fn<S>("compute_accel: no bodies were set"));
    if (s->mid_step)
        return fail(NB_ERR_STATE, fn<S>("compute_accel: a split step is in flight"));
    NB_TRY(step_local(s, 0));
    return collect_timing(s);
}

template <class S>
int api_sync(S *s)
{
    if (!s)
        return fail(NB_ERR_ARG, fn<S>("sync: sim is NULL"));
    NB_HIP(hipStreamSynchronize(s->stream));
    return NB_OK;
}

template <class S>
int api_read_bodies(S *s, typename S::Body *out)
{
    if (!s || !out)
        return fail(NB_ERR_ARG, fn<S>("read_bodies: NULL argument"));
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, fn<S>("read_bodies: no bodies were set"));
    // leapfrog keeps velocities half a step behind the positions; bring them level with the
    // force at the current positions
    const bool half_step = s->p.integrator == NB_INT_LEAPFROG && !s->first_kick && !s->mid_step;
    if (half_step) {
        // this force evaluation is not a step: keep it out of the per-stage timing
        const int timing = s->timer.on;
        s->timer.on = 0;
        const int rc = step_local(s, 0);
        s->timer.on = timing;
        NB_TRY(rc);
    }
    return read_state(s, out, half_step ? 0.5 * s->p.dt : 0.0);
}

template <class S>
int api_set_stage_timing(S *s, int on)
{
    if (!s)
        return fail(NB_ERR_ARG, fn<S>("set_stage_timing: sim is NULL"));
    s->collect_stats = (on & 2) ? 1 : 0;
    return s->timer.set(s->stream, on & 1);
}

template <class S>
int api_stage_ms(S *s, float *out, int count)
{
    if (!s || !out)
        return fail(NB_ERR_ARG, fn<S>("stage_ms: NULL argument"));
    return s->timer.read(s->stream, out, count);
}

}  // namespace nb
