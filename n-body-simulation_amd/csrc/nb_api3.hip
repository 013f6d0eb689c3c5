// nb_api3.hip -- the nb3_ entry points of include/nbody.h: the C ABI of nb_api.hip for
// three-dimensional bodies.  nb_sim3 owns the device state (FP64 positions, velocities and
// accelerations as packed triples, in an order sorted once by a Morton code of the initial
// positions) and drives nb_direct3.hip or nb_bh3.hip; bodies are copied back only on request.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "nb_bh.h"
#include "nb_comm.h"
#include "nb_common.h"
#include "nb_refresh.h"
#include "nb_timing.h"
#include "nb_dim3.h"

using namespace nb;

struct nb_sim3 {
    nb_params p;
    int n = 0;
    int alg = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool have_bodies = false;
    bool first_kick = true;   // leapfrog: the next step is the first since the bodies were set
    bool stepped = false;     // a step was taken since the bodies were set
    long steps_in_order = 0;  // direct sum: steps since the order was last derived from the positions

    Units units{};
    ForceConsts fc{};

    double *pos = nullptr, *pos2 = nullptr, *vel = nullptr, *acc = nullptr;  // packed triples
    double *mass = nullptr, *radius = nullptr;
    uint8_t *dyn = nullptr;
    std::vector<uint32_t> order;  // caller index of the body in each slot
    uint32_t *order_d = nullptr;  // the same on the device
    bool order_stale = false;     // the device renewed the order; `order` is fetched when next needed
    RefreshBuffers rb;            // buffers of the renewal (made at nb3_create for the direct sum, else on first use)

    // partition: this process integrates slots [t0, t1)
    int rank = 0, world = 1;
    size_t chunk = 0;
    int t0 = 0, t1 = 0;
    bool mid_step = false;
    Comm *comm = nullptr;  // RCCL communicator of nb3_comm_init (null: none)
    int device = 0;
    size_t part_cap = 0;  // doubles in ds.part

    Direct3State ds;
    Bh3Buffers bh;
    int collect_stats = 0;

    StageTimer timer;
};

namespace {

// Powers of two that bring the largest coordinate to [4, 8) and the largest mass to about 2^20
// (same rule as the 2-D path).
void choose_units(nb_sim3 *s, const nb_body3 *b, int n)
{
    double lo[3] = {b[0].position.x, b[0].position.y, b[0].position.z}, hi[3] = {lo[0], lo[1], lo[2]};
    for (int i = 1; i < n; ++i) {
        const double p[3] = {b[i].position.x, b[i].position.y, b[i].position.z};
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], p[d]);
            hi[d] = std::max(hi[d], p[d]);
        }
    }
    for (int d = 0; d < 3; ++d)
        s->units.origin[d] = 0.5 * (lo[d] + hi[d]);
    const double *o = s->units.origin;
    double maxc = 1.0, maxm = 1.0;
    for (int i = 0; i < n; ++i) {
        maxc = std::max({maxc, std::fabs(b[i].position.x - o[0]), std::fabs(b[i].position.y - o[1]),
                         std::fabs(b[i].position.z - o[2])});
        maxm = std::max(maxm, std::fabs(b[i].mass));
    }
    s->units.len_unit = std::ldexp(1.0, (int)std::floor(std::log2(maxc)) - 2);
    s->units.mass_unit = std::ldexp(1.0, (int)std::floor(std::log2(maxm)) - 20);
    s->units.acc_factor = s->p.gravity * s->units.mass_unit / (s->units.len_unit * s->units.len_unit);
    const double lu = s->units.len_unit;
    s->fc.eps2 = (float)(s->p.eps * s->p.eps / (lu * lu));
    s->fc.eps3 = (float)(s->p.eps * s->p.eps / (lu * lu * lu));
    s->fc.cth = (float)(s->p.collision_th / lu);
    // every evaluated pair is farther apart than collision_th: a term below 2^-26 of cth^3
    // (cth^2) cannot change the float sum it is added to
    const double c3 = (double)s->fc.cth * s->fc.cth * s->fc.cth;
    s->fc.exact_denom = !((double)s->fc.eps3 <= std::ldexp(c3, -26));
    s->fc.eps2_negligible = (double)s->fc.eps2 <= std::ldexp((double)s->fc.cth * s->fc.cth, -26);
    // coordinates lie in [-8, 8): their float rounding is up to 2.4e-7 units, i.e. a relative 5e-6 of a
    // distance of 0.05 units (the reference's cutoff in its own scenarios is 0.145 units)
    s->fc.precise_near = s->fc.cth < 0.05f;
}

// Bodies sorted by a 3 x 10-bit Morton code of their positions: index ranges are compact patches
// of space (small boxes for the direct-sum runs, close targets for a traversal wave).
void spatial_order(const nb_body3 *b, int n, std::vector<uint32_t> &order)
{
    order.resize(n);
    double lo[3], hi[3];
    auto coord = [&](int i, int d) { return d == 0 ? b[i].position.x : (d == 1 ? b[i].position.y : b[i].position.z); };
    for (int d = 0; d < 3; ++d)
        lo[d] = hi[d] = coord(0, d);
    for (int i = 1; i < n; ++i)
        for (int d = 0; d < 3; ++d) {
            lo[d] = std::min(lo[d], coord(i, d));
            hi[d] = std::max(hi[d], coord(i, d));
        }
    // one scale for all axes, so that a thin disc is not cut into slabs
    double ext = 0.0;
    for (int d = 0; d < 3; ++d)
        ext = std::max(ext, hi[d] - lo[d]);
    const double sc = ext > 0 ? 1023.0 / ext : 0.0;
    auto spread = [](uint64_t v) {
        v = (v | (v << 16)) & 0x030000FFull;
        v = (v | (v << 8)) & 0x0300F00Full;
        v = (v | (v << 4)) & 0x030C30C3ull;
        v = (v | (v << 2)) & 0x09249249ull;
        return v;
    };
    std::vector<uint64_t> kv(n);
    for (int i = 0; i < n; ++i) {
        uint64_t code = 0;
        for (int d = 0; d < 3; ++d)
            code |= spread((uint64_t)((coord(i, d) - lo[d]) * sc)) << d;
        kv[i] = (code << 32) | (uint32_t)i;
    }
    std::sort(kv.begin(), kv.end());
    for (int i = 0; i < n; ++i)
        order[i] = (uint32_t)(kv[i] & 0xffffffffu);
}

IntegrateArgs integrate_args(const nb_sim3 *s, int do_integrate)
{
    IntegrateArgs ia;
    ia.acc_factor = s->units.acc_factor;
    ia.dt = s->p.dt;
    ia.dt_kick = (s->p.integrator == NB_INT_LEAPFROG && s->first_kick) ? 0.5 * s->p.dt : s->p.dt;
    ia.inv_len_unit = 1.0 / s->units.len_unit;
    for (int d = 0; d < 3; ++d)
        ia.origin[d] = s->units.origin[d];
    ia.do_integrate = do_integrate;
    ia.static_fixed = 0;
    return ia;
}

void mark(nb_sim3 *s, int k) { s->timer.mark(s->stream, k); }

int step_local(nb_sim3 *s, int do_integrate)
{
    if (s->alg == NB_ALG_DIRECT) {
        s->ds.pos_cur = s->pos;
        s->ds.pos_next = s->pos2;
        mark(s, 5);
        NB_TRY(direct3_step(s->stream, s->ds, s->n, s->t0, s->t1 - s->t0, s->fc, integrate_args(s, do_integrate),
                            s->p.collision_th));
        mark(s, 6);
        return NB_OK;
    }
    Bh3StepArgs a;
    a.pos = s->pos;
    a.mass = s->mass;
    a.radius = s->radius;
    a.dyn = s->dyn;
    a.vel = s->vel;
    a.acc = s->acc;
    a.n = s->n;
    a.t0 = s->t0;
    a.t1 = s->t1;
    a.units = s->units;
    a.fc = s->fc;
    a.ia = integrate_args(s, do_integrate);
    a.theta = s->p.theta;
    a.bbox_margin = s->p.bbox_margin;
    a.collision_th = s->p.collision_th;
    a.collect_stats = s->collect_stats;
    mark(s, 0);
    NB_TRY(bh3_bbox(s->stream, s->bh, a));
    mark(s, 1);
    NB_TRY(bh3_keys(s->stream, s->bh, a));
    mark(s, 2);
    NB_TRY(bh3_sort(s->stream, s->bh, a));
    mark(s, 3);
    NB_TRY(bh3_build(s->stream, s->bh, a));
    mark(s, 4);
    NB_TRY(bh3_com(s->stream, s->bh, a));
    mark(s, 5);
    NB_TRY(bh3_force(s->stream, s->bh, a));
    mark(s, 6);
    return NB_OK;
}

// After the positions other ranks advanced have been gathered into the new position buffer.
int step_finish(nb_sim3 *s)
{
    s->first_kick = false;
    s->stepped = true;
    if (s->alg == NB_ALG_DIRECT) {
        if (s->world > 1) {
            NB_TRY(pack_posm3_range(s->stream, s->pos2, s->mass, s->n, 0, s->t0, s->units, s->ds.posm[s->ds.cur ^ 1]));
            NB_TRY(pack_posm3_range(s->stream, s->pos2, s->mass, s->n, s->t1, s->n, s->units, s->ds.posm[s->ds.cur ^ 1]));
        }
        s->ds.cur ^= 1;
        std::swap(s->pos, s->pos2);
    }
    return NB_OK;
}

int collect_timing(nb_sim3 *s) { return s->timer.step_done(s->stream); }

void release(nb_sim3 *s)
{
    (void)hipFree(s->pos);
    (void)hipFree(s->pos2);
    (void)hipFree(s->vel);
    (void)hipFree(s->acc);
    (void)hipFree(s->mass);
    (void)hipFree(s->radius);
    (void)hipFree(s->dyn);
    (void)hipFree(s->ds.posm[0]);
    (void)hipFree(s->ds.posm[1]);
    (void)hipFree(s->ds.wcut);
    (void)hipFree(s->ds.rad);
    (void)hipFree(s->ds.part);
    (void)hipFree(s->ds.meta);
    bh3_free(s->bh);
    (void)hipFree(s->order_d);
    refresh_free(s->rb);
    comm_destroy(s->comm);
    s->comm = nullptr;
    s->timer.destroy();
    if (s->own_stream && s->stream)
        (void)hipStreamDestroy(s->stream);
}

// Units, order and device arrays from bodies given in the caller's order.
int load_bodies(nb_sim3 *s, const nb_body3 *b, int n)
{
    choose_units(s, b, n);
    spatial_order(b, n, s->order);
    std::vector<double> pos(3 * (size_t)n), vel(3 * (size_t)n), acc(3 * (size_t)n), mass(n), radius(n);
    std::vector<uint8_t> dyn(n);
    for (int k = 0; k < n; ++k) {
        const nb_body3 &x = b[s->order[k]];
        const double p[3] = {x.position.x, x.position.y, x.position.z}, v[3] = {x.velocity.x, x.velocity.y, x.velocity.z},
                     a[3] = {x.acceleration.x, x.acceleration.y, x.acceleration.z};
        for (int d = 0; d < 3; ++d) {
            pos[3 * (size_t)k + d] = p[d];
            vel[3 * (size_t)k + d] = v[d];
            acc[3 * (size_t)k + d] = a[d];
        }
        mass[k] = x.mass;
        radius[k] = x.radius;
        dyn[k] = x.is_dynamic ? 1 : 0;
    }
    const size_t tri = sizeof(double) * 3 * (size_t)n;
    NB_HIP(hipStreamSynchronize(s->stream));
    NB_HIP(hipMemcpy(s->pos, pos.data(), tri, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->vel, vel.data(), tri, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->acc, acc.data(), tri, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->mass, mass.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->radius, radius.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->dyn, dyn.data(), n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->order_d, s->order.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    s->order_stale = false;
    if (s->alg == NB_ALG_DIRECT) {
        s->ds.cur = 0;
        NB_TRY(pack_posm3(s->stream, s->pos, s->mass, s->radius, n, s->units, s->p.collision_th, s->ds.posm[0], s->ds.wcut,
                          s->ds.rad));
        // masses and padding never change: the other buffer starts as a copy
        NB_HIP(hipMemcpyAsync(s->ds.posm[1], s->ds.posm[0], sizeof(float4) * 2 * direct3_record_capacity(n),
                              hipMemcpyDeviceToDevice, s->stream));
    }
    NB_HIP(hipStreamSynchronize(s->stream));
    s->steps_in_order = 0;
    return NB_OK;
}

// The slot -> caller index map on the host (fetched again after the device renewed it).
int host_order(nb_sim3 *s)
{
    if (!s->order_stale)
        return NB_OK;
    NB_HIP(hipStreamSynchronize(s->stream));
    NB_HIP(hipMemcpy(s->order.data(), s->order_d, sizeof(uint32_t) * s->n, hipMemcpyDeviceToHost));
    s->order_stale = false;
    return NB_OK;
}

long refresh_interval(int n)
{
    const long interval = 50000000L / (n > 0 ? n : 1);
    return interval < 64 ? 64 : (interval > 4096 ? 4096 : interval);
}

// Renews the order on the device, asynchronously on the stream (nb_refresh.h; the 2-D handle does
// the same): new order from the current positions, every per-body array gathered into it, the
// direct sum's records packed again relative to the middle of the box the previous renewal saw.
int device_refresh(nb_sim3 *s)
{
    if (s->rb.n != s->n)
        NB_TRY(refresh_alloc(s->rb, s->n));
    if (s->rb.origin_set && s->alg == NB_ALG_DIRECT)
        for (int d = 0; d < 3; ++d)
            s->units.origin[d] = ((volatile double *)s->rb.origin_host)[d];
    NB_TRY(refresh_sort(s->stream, s->rb, s->pos, 3, s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->pos, 3 * sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->vel, 3 * sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->acc, 3 * sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->mass, sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->radius, sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->dyn, 1, s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->order_d, sizeof(uint32_t), s->n));
    s->order_stale = true;
    if (s->alg == NB_ALG_DIRECT) {
        s->ds.cur = 0;
        NB_TRY(pack_posm3(s->stream, s->pos, s->mass, s->radius, s->n, s->units, s->p.collision_th, s->ds.posm[0], s->ds.wcut,
                          s->ds.rad));
        NB_HIP(hipMemcpyAsync(s->ds.posm[1], s->ds.posm[0], sizeof(float4) * 2 * direct3_record_capacity(s->n),
                              hipMemcpyDeviceToDevice, s->stream));
    }
    s->steps_in_order = 0;
    return NB_OK;
}

// The device state in the caller's order; half_kick > 0 adds a(x) * half_kick to the velocities.
int read_state(nb_sim3 *s, nb_body3 *out, double half_kick)
{
    const int n = s->n;
    std::vector<double> pos(3 * (size_t)n), vel(3 * (size_t)n), acc(3 * (size_t)n), mass(n), radius(n);
    std::vector<uint8_t> dyn(n);
    const size_t tri = sizeof(double) * 3 * (size_t)n;
    NB_HIP(hipStreamSynchronize(s->stream));
    NB_TRY(host_order(s));
    NB_HIP(hipMemcpy(pos.data(), s->pos, tri, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(vel.data(), s->vel, tri, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(acc.data(), s->acc, tri, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(mass.data(), s->mass, sizeof(double) * n, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(radius.data(), s->radius, sizeof(double) * n, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(dyn.data(), s->dyn, n, hipMemcpyDeviceToHost));
    for (int k = 0; k < n; ++k) {
        nb_body3 &x = out[s->order[k]];
        std::memset(&x, 0, sizeof(x));
        x.is_dynamic = dyn[k];
        x.mass = mass[k];
        x.radius = radius[k];
        const double *p = &pos[3 * (size_t)k], *v = &vel[3 * (size_t)k], *a = &acc[3 * (size_t)k];
        x.position = {p[0], p[1], p[2]};
        const double hk = (k >= s->t0 && k < s->t1) ? half_kick : 0.0;  // other ranks' velocities are not held here
        x.velocity = {v[0] + hk * a[0], v[1] + hk * a[1], v[2] + hk * a[2]};
        x.acceleration = {a[0], a[1], a[2]};
    }
    return NB_OK;
}

}  // namespace

extern "C" {

int nb3_create(const nb_params *params, int n, int alg, nb_sim3 **out)
{
    if (!out)
        return fail(NB_ERR_ARG, "nb3_create: out is NULL");
    *out = nullptr;
    if (n < 1)
        return fail(NB_ERR_ARG, "ERROR: need to have at least 1 body");
    if (alg != NB_ALG_DIRECT && alg != NB_ALG_BARNES_HUT)
        return fail(NB_ERR_ARG, "nb3_create: alg must be 0 (direct sum) or 1 (Barnes-Hut)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(NB_ERR_HIP, "nb3_create: no HIP device is visible; this library has no CPU path");
    nb_params p;
    if (params)
        p = *params;
    else
        nb_default_params(&p);
    if (!(p.theta >= 0) || !(p.collision_th >= 0) || !(p.eps >= 0))
        return fail(NB_ERR_ARG, "nb3_create: theta, collision_th and eps must be >= 0");
    if (p.bucket_depth != 0)
        return fail(NB_ERR_ARG, "nb3_create: bucket_depth (the reference GPU program's 2-D tree) must be 0 in 3-D");
    if (p.integrator != NB_INT_EULER && p.integrator != NB_INT_LEAPFROG)
        return fail(NB_ERR_ARG, "nb3_create: integrator must be NB_INT_EULER or NB_INT_LEAPFROG");
    nb_sim3 *s = new nb_sim3();
    s->p = p;
    s->n = n;
    s->alg = alg;
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
    s->t1 = n;
    s->chunk = (size_t)n;
    const size_t cap = (size_t)n + 64;  // room for world * chunk records (world <= 64)
    chk(dev_alloc(s->pos, 3 * cap));
    chk(dev_alloc(s->vel, 3 * cap));
    chk(dev_alloc(s->acc, 3 * cap));
    chk(dev_alloc(s->mass, cap));
    chk(dev_alloc(s->radius, cap));
    chk(dev_alloc(s->dyn, cap));
    chk(dev_alloc(s->order_d, cap));
    if (alg == NB_ALG_DIRECT) {
        chk(refresh_alloc(s->rb, n));  // nb3_step renews the order by itself for the direct sum
        chk(dev_alloc(s->pos2, 3 * cap));
        chk(dev_alloc(s->ds.posm[0], 2 * direct3_record_capacity(n)));
        chk(dev_alloc(s->ds.posm[1], 2 * direct3_record_capacity(n)));
        chk(dev_alloc(s->ds.wcut, direct3_record_capacity(n)));
        chk(dev_alloc(s->ds.rad, cap));
        s->ds.jsplit = direct3_choose_jsplit(n, n);
        s->part_cap = 3 * (size_t)s->ds.jsplit * n;
        chk(dev_alloc(s->ds.part, s->part_cap));
        uint8_t *meta = nullptr;
        chk(dev_alloc(meta, direct3_meta_bytes(n)));
        s->ds.meta = meta;
        s->ds.dyn = s->dyn;
        s->ds.radius = s->radius;
        s->ds.vel = s->vel;
        s->ds.acc = s->acc;
    } else {
        chk(bh3_alloc(s->bh, n));
    }
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

int nb3_destroy(nb_sim3 *s)
{
    if (!s)
        return NB_OK;
    (void)hipStreamSynchronize(s->stream);
    release(s);
    delete s;
    return NB_OK;
}

int nb3_set_bodies(nb_sim3 *s, const nb_body3 *b, int n)
{
    if (!s || !b)
        return fail(NB_ERR_ARG, "nb3_set_bodies: NULL argument");
    if (n != s->n)
        return fail(NB_ERR_ARG, "nb3_set_bodies: n differs from the n given to nb3_create");
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(b[i].position.x) || !std::isfinite(b[i].position.y) || !std::isfinite(b[i].position.z) ||
            !std::isfinite(b[i].mass))
            return fail(NB_ERR_ARG, "nb3_set_bodies: body " + std::to_string(i) + " is not finite");
    NB_TRY(load_bodies(s, b, n));
    s->have_bodies = true;
    s->first_kick = true;
    s->stepped = false;
    s->mid_step = false;
    return NB_OK;
}

int nb3_setup(nb_sim3 *s, int scenario, unsigned seed, double thickness)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_setup: sim is NULL");
    if (scenario == NB_SIM_SOLAR && s->n != 5)
        return fail(NB_ERR_ARG, "nb3_setup: the solar system has 5 bodies (create the simulation with n = 5)");
    std::vector<nb_body3> b(s->n);
    NB_TRY(nb3_init_scenario(scenario, s->n, seed, thickness, b.data()));
    return nb3_set_bodies(s, b.data(), s->n);
}

int nb3_step(nb_sim3 *s, int nsteps)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_step: sim is NULL");
    if (nsteps < 0)
        return fail(NB_ERR_ARG, "nb3_step: nsteps < 0");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb3_step: no bodies were set");
    if (s->world != 1)
        return fail(NB_ERR_STATE, "nb3_step: a partitioned simulation steps with nb3_step_local / nb3_step_finish");
    // direct sum: bodies wander, so the spatial order its run-against-wave box tests rely on is
    // renewed now and then (same rule as nb_step)
    const long interval = refresh_interval(s->n);
    for (int k = 0; k < nsteps; ++k) {
        NB_TRY(step_local(s, 1));
        NB_TRY(step_finish(s));
        NB_TRY(collect_timing(s));
        if (s->alg == NB_ALG_DIRECT && ++s->steps_in_order >= interval)
            NB_TRY(device_refresh(s));
    }
    return NB_OK;
}

int nb3_refresh_order(nb_sim3 *s)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_refresh_order: sim is NULL");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb3_refresh_order: no bodies were set");
    if (s->mid_step)
        return fail(NB_ERR_STATE, "nb3_refresh_order: a split step is in flight");
    return device_refresh(s);
}

int nb3_velocity_buffer(nb_sim3 *s, void **device_ptr, size_t *chunk_records, size_t *record_bytes)
{
    if (!s || !device_ptr || !chunk_records || !record_bytes)
        return fail(NB_ERR_ARG, "nb3_velocity_buffer: NULL argument");
    *device_ptr = s->vel;
    *chunk_records = s->chunk;
    *record_bytes = 3 * sizeof(double);
    return NB_OK;
}

int nb3_comm_init(nb_sim3 *s, int rank, int world, const void *id)
{
    if (!s || !id)
        return fail(NB_ERR_ARG, "nb3_comm_init: NULL argument");
    if (s->comm)
        return fail(NB_ERR_STATE, "nb3_comm_init: the simulation already has a communicator");
    int device = -1;
    NB_HIP(hipGetDevice(&device));
    if (device != s->device)
        return fail(NB_ERR_STATE, "nb3_comm_init: the handle's device is not the current one");
    NB_TRY(nb3_set_partition(s, rank, world));
    return comm_create(&s->comm, rank, world, id);
}

int nb3_step_distributed(nb_sim3 *s, int nsteps)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_step_distributed: sim is NULL");
    if (nsteps < 0)
        return fail(NB_ERR_ARG, "nb3_step_distributed: nsteps < 0");
    if (!s->comm)
        return fail(NB_ERR_STATE, "nb3_step_distributed: call nb3_comm_init first");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb3_step_distributed: no bodies were set");
    if (s->mid_step)
        return fail(NB_ERR_STATE, "nb3_step_distributed: a split step is in flight");
    for (int k = 0; k < nsteps; ++k) {
        NB_TRY(step_local(s, 1));
        double *buf = s->alg == NB_ALG_DIRECT ? s->pos2 : s->pos;
        NB_TRY(comm_all_gather_f64(s->comm, buf, 3 * s->chunk, s->stream));
        mark(s, StageTimer::STAGE_END);
        NB_TRY(step_finish(s));
        NB_TRY(collect_timing(s));
        if (s->alg == NB_ALG_DIRECT && ++s->steps_in_order >= refresh_interval(s->n)) {
            NB_TRY(comm_all_gather_f64(s->comm, s->vel, 3 * s->chunk, s->stream));
            NB_TRY(device_refresh(s));
        }
    }
    return NB_OK;
}

int nb3_compute_accel(nb_sim3 *s)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_compute_accel: sim is NULL");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb3_compute_accel: no bodies were set");
    if (s->mid_step)
        return fail(NB_ERR_STATE, "nb3_compute_accel: a split step is in flight");
    NB_TRY(step_local(s, 0));
    return collect_timing(s);
}

int nb3_sync(nb_sim3 *s)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_sync: sim is NULL");
    NB_HIP(hipStreamSynchronize(s->stream));
    return NB_OK;
}

int nb3_read_bodies(nb_sim3 *s, nb_body3 *out)
{
    if (!s || !out)
        return fail(NB_ERR_ARG, "nb3_read_bodies: NULL argument");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb3_read_bodies: no bodies were set");
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

int nb3_diagnostics(nb_sim3 *s, double *out, int count)
{
    if (!s || !out || count < 1)
        return fail(NB_ERR_ARG, "nb3_diagnostics: bad argument");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb3_diagnostics: no bodies were set");
    std::vector<nb_body3> b(s->n);
    NB_TRY(nb3_read_bodies(s, b.data()));
    double v[12] = {};
    for (const nb_body3 &x : b) {
        const nb_vec3 &p = x.position, &w = x.velocity;
        v[0] += x.mass;
        v[1] += x.mass * p.x;
        v[2] += x.mass * p.y;
        v[3] += x.mass * p.z;
        v[4] += x.mass * w.x;
        v[5] += x.mass * w.y;
        v[6] += x.mass * w.z;
        v[7] += x.mass * (p.y * w.z - p.z * w.y);
        v[8] += x.mass * (p.z * w.x - p.x * w.z);
        v[9] += x.mass * (p.x * w.y - p.y * w.x);
        v[10] += 0.5 * x.mass * (w.x * w.x + w.y * w.y + w.z * w.z);
    }
    if (v[0] != 0)
        for (int k = 1; k <= 3; ++k)
            v[k] /= v[0];
    v[11] = std::nan("");
    if (count > 11 && s->n <= 131072) {
        double *d = nullptr;
        NB_TRY(dev_alloc(d, 1));
        int rc = potential_energy3(s->stream, s->pos, s->mass, s->radius, s->n, s->p.gravity, s->p.eps, s->p.collision_th, d);
        if (rc == NB_OK && hipStreamSynchronize(s->stream) == hipSuccess)
            (void)hipMemcpy(&v[11], d, sizeof(double), hipMemcpyDeviceToHost);
        (void)hipFree(d);
        NB_TRY(rc);
    }
    for (int k = 0; k < count && k < 12; ++k)
        out[k] = v[k];
    return NB_OK;
}

int nb3_set_stream(nb_sim3 *s, void *hip_stream)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_set_stream: sim is NULL");
    NB_HIP(hipStreamSynchronize(s->stream));
    if (s->own_stream && s->stream)
        (void)hipStreamDestroy(s->stream);
    s->stream = (hipStream_t)hip_stream;
    s->own_stream = false;
    return NB_OK;
}

int nb3_set_partition(nb_sim3 *s, int rank, int world)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_set_partition: sim is NULL");
    if (world < 1 || world > 64 || rank < 0 || rank >= world)
        return fail(NB_ERR_ARG, "nb3_set_partition: need 0 <= rank < world <= 64");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb3_set_partition: set the bodies first");
    if (s->mid_step)
        return fail(NB_ERR_STATE, "nb3_set_partition: a split step is in flight");
    if (s->stepped)
        return fail(NB_ERR_STATE, "nb3_set_partition: the simulation was stepped already; set the bodies again first");
    s->rank = rank;
    s->world = world;
    s->chunk = ((size_t)s->n + world - 1) / world;
    s->t0 = (int)std::min((size_t)s->n, rank * s->chunk);
    s->t1 = (int)std::min((size_t)s->n, (rank + 1) * s->chunk);
    if (s->alg == NB_ALG_DIRECT) {
        // fewer targets per rank: split the sources further so the chip stays full
        const int js = direct3_choose_jsplit(s->n, s->t1 - s->t0);
        const size_t need = 3 * (size_t)js * (size_t)(s->t1 - s->t0);
        if (need > s->part_cap) {
            (void)hipFree(s->ds.part);
            s->ds.part = nullptr;
            NB_TRY(dev_alloc(s->ds.part, need));
            s->part_cap = need;
        }
        s->ds.jsplit = js;
    }
    return NB_OK;
}

int nb3_exchange_buffer(nb_sim3 *s, void **device_ptr, size_t *chunk_records, size_t *record_bytes)
{
    if (!s || !device_ptr || !chunk_records || !record_bytes)
        return fail(NB_ERR_ARG, "nb3_exchange_buffer: NULL argument");
    // direct sum writes the new positions to the other buffer; mid-step that is the one to gather
    *device_ptr = (s->alg == NB_ALG_DIRECT && s->mid_step) ? s->pos2 : s->pos;
    *chunk_records = s->chunk;
    *record_bytes = 3 * sizeof(double);
    return NB_OK;
}

int nb3_step_local(nb_sim3 *s)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_step_local: sim is NULL");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb3_step_local: no bodies were set");
    if (s->mid_step)
        return fail(NB_ERR_STATE, "nb3_step_local: the previous step was not finished");
    NB_TRY(step_local(s, 1));
    s->mid_step = true;
    return NB_OK;
}

int nb3_step_finish(nb_sim3 *s)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_step_finish: sim is NULL");
    if (!s->mid_step)
        return fail(NB_ERR_STATE, "nb3_step_finish: nb3_step_local was not called");
    NB_TRY(step_finish(s));
    s->mid_step = false;
    return collect_timing(s);
}

int nb3_set_stage_timing(nb_sim3 *s, int on)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb3_set_stage_timing: sim is NULL");
    s->collect_stats = (on & 2) ? 1 : 0;
    return s->timer.set(s->stream, on & 1);
}

int nb3_stage_ms(nb_sim3 *s, float *out, int count)
{
    if (!s || !out)
        return fail(NB_ERR_ARG, "nb3_stage_ms: NULL argument");
    return s->timer.read(s->stream, out, count);
}

int nb3_debug_read(nb_sim3 *s, int which, void *out, size_t bytes, size_t *written)
{
    if (!s || !out)
        return fail(NB_ERR_ARG, "nb3_debug_read: NULL argument");
    NB_HIP(hipStreamSynchronize(s->stream));
    auto give = [&](const void *src, size_t need, bool device) -> int {
        if (bytes < need)
            return fail(NB_ERR_ARG, "nb3_debug_read: buffer too small");
        if (need && device)
            NB_HIP(hipMemcpy(out, src, need, hipMemcpyDeviceToHost));
        else if (need)
            std::memcpy(out, src, need);
        if (written)
            *written = need;
        return NB_OK;
    };
    if (which == NB_DBG_ORDER) {
        NB_TRY(host_order(s));
        return give(s->order.data(), sizeof(uint32_t) * s->n, false);
    }
    if (s->alg != NB_ALG_BARNES_HUT)
        return fail(NB_ERR_STATE, "nb3_debug_read: not a Barnes-Hut simulation");
    constexpr int LS = BH_MAX_DEPTH + 2;
    uint32_t ls[LS] = {};
    NB_HIP(hipMemcpy(ls, s->bh.level_start, sizeof(ls), hipMemcpyDeviceToHost));
    size_t nnodes = 0;
    for (int d = 0; d + 1 < LS; ++d)
        nnodes = std::max(nnodes, (size_t)ls[d + 1]);
    if (nnodes > s->bh.node_cap)
        nnodes = 0;  // no tree was built yet
    switch (which) {
    case NB_DBG_BBOX:
        return give(s->bh.box, 6 * sizeof(double), true);
    case NB_DBG_SORTED_KEYS:
        return give(s->bh.keys[0], sizeof(uint64_t) * s->n, true);
    case NB_DBG_SORTED_IDX:
        return give(s->bh.idx[0], sizeof(uint32_t) * s->n, true);
    case NB_DBG_NODE_RANGE:
    case NB_DBG_NODE_CHILD: {
        std::vector<uint4> meta(nnodes);
        if (nnodes)
            NB_HIP(hipMemcpy(meta.data(), s->bh.node_meta, sizeof(uint4) * nnodes, hipMemcpyDeviceToHost));
        std::vector<uint32_t> o(2 * nnodes);
        for (size_t u = 0; u < nnodes; ++u) {
            o[2 * u] = which == NB_DBG_NODE_RANGE ? meta[u].x : meta[u].z;
            o[2 * u + 1] = which == NB_DBG_NODE_RANGE ? meta[u].y : (meta[u].w & 0xffu);
        }
        return give(o.data(), sizeof(uint32_t) * 2 * nnodes, false);
    }
    case NB_DBG_NODE_COM:
        return give(s->bh.node_com, sizeof(double) * 6 * nnodes, true);
    case NB_DBG_LEVEL_START:
        return give(ls, sizeof(ls), false);
    case NB_DBG_KEY_LEVELS:
        return give(s->bh.cut, sizeof(uint32_t) * 2, true);
    case NB_DBG_COUNTERS:
        return give(s->bh.counters, sizeof(unsigned long long) * 8, true);
    default:
        return fail(NB_ERR_ARG, "nb3_debug_read: array not available in 3-D");
    }
}

}  // extern "C"
