// nb_api.hip -- the C ABI of libnbody_hip.so (declared in include/nbody.h).
//
// nb_sim plays the role of the reference's BarnesHutCuda object
// (src/gpu/BH/barnesHutCuda.cuh:49-80): it owns the device arrays, takes the initial
// state once (setup) and advances it with update()-like steps, copying bodies back only
// when asked.  State on the device is SoA: positions and velocities in FP64 (the
// reference's precision, so integration adds no rounding of its own), and per step a
// float4 record per body for the FP32 force kernels.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <numeric>
#include <vector>

#include "nb_sim.h"

namespace nb {

static thread_local std::string g_error;

void set_error(const std::string &msg) { g_error = msg; }

int fail(int status, const std::string &msg)
{
    g_error = msg;
    return status;
}

}  // namespace nb

using namespace nb;

namespace nb {

IntegrateArgs integrate_args(const nb_sim *s, int do_integrate)
{
    IntegrateArgs ia;
    ia.acc_factor = s->units.acc_factor;
    ia.dt = s->p.dt;
    ia.dt_kick = (s->p.integrator == NB_INT_LEAPFROG && s->first_kick) ? 0.5 * s->p.dt : s->p.dt;
    ia.inv_len_unit = 1.0 / s->units.len_unit;
    for (int d = 0; d < 3; ++d)
        ia.origin[d] = s->units.origin[d];
    ia.do_integrate = do_integrate;
    ia.static_fixed = (s->alg == NB_ALG_BARNES_HUT && s->p.bucket_depth > 0) ? 1 : 0;
    return ia;
}

BhStepArgs bh_args(nb_sim *s, int do_integrate)
{
    BhStepArgs a;
    a.pos = s->pos;
    a.mass = s->mass;
    a.radius = s->radius;
    a.mass_radius = s->mass_radius;
    a.dyn = s->dyn;
    a.vel = s->vel;
    a.acc = s->acc;
    a.pos_out = s->pos;
    a.n = s->n;
    a.t0 = s->t0;
    a.t1 = s->t1;
    a.units = s->units;
    a.fc = s->fc;
    a.ia = integrate_args(s, do_integrate);
    a.theta = s->p.theta;
    a.bbox_margin = s->p.bbox_margin;
    a.collision_th = s->p.collision_th;
    a.bucket_depth = s->p.bucket_depth;
    a.collect_stats = s->collect_stats;
    return a;
}

}  // namespace nb

namespace {

double pow2_floor_exp(double v) { return v > 0 ? std::floor(std::log2(v)) : 0.0; }

// The origin of the internal coordinates (the middle of the bodies' bounding box) and the powers
// of two that bring the largest coordinate from there to [4, 8) and the largest mass to about 2^20.
void choose_units(nb_sim *s, const nb_body *b, int n)
{
    double lo[2] = {b[0].position.x, b[0].position.y}, hi[2] = {lo[0], lo[1]};
    for (int i = 1; i < n; ++i) {
        lo[0] = std::min(lo[0], b[i].position.x);
        hi[0] = std::max(hi[0], b[i].position.x);
        lo[1] = std::min(lo[1], b[i].position.y);
        hi[1] = std::max(hi[1], b[i].position.y);
    }
    s->units.origin[0] = 0.5 * (lo[0] + hi[0]);
    s->units.origin[1] = 0.5 * (lo[1] + hi[1]);
    s->units.origin[2] = 0.0;
    double maxc = 1.0, maxm = 1.0;
    for (int i = 0; i < n; ++i) {
        maxc = std::max(maxc, std::max(std::fabs(b[i].position.x - s->units.origin[0]),
                                       std::fabs(b[i].position.y - s->units.origin[1])));
        maxm = std::max(maxm, std::fabs(b[i].mass));
    }
    s->units.len_unit = std::ldexp(1.0, (int)pow2_floor_exp(maxc) - 2);
    s->units.mass_unit = std::ldexp(1.0, (int)pow2_floor_exp(maxm) - 20);
    s->units.acc_factor = s->p.gravity * s->units.mass_unit / (s->units.len_unit * s->units.len_unit);
    const double lu = s->units.len_unit;
    s->fc.eps2 = (float)(s->p.eps * s->p.eps / (lu * lu));
    s->fc.eps3 = (float)(s->p.eps * s->p.eps / (lu * lu * lu));
    s->fc.cth = (float)(s->p.collision_th / lu);
    // Every evaluated pair is farther apart than collision_th, so r^3 >= cth^3: eps3 below
    // 2^-26 of that cannot change a float sum r^3 + eps3.
    const double c3 = (double)s->fc.cth * s->fc.cth * s->fc.cth;
    s->fc.exact_denom = !((double)s->fc.eps3 <= std::ldexp(c3, -26));
    s->fc.eps2_negligible = (double)s->fc.eps2 <= std::ldexp((double)s->fc.cth * s->fc.cth, -26);
    // coordinates lie in [-8, 8): their float rounding is up to 2.4e-7 units, i.e. a relative 5e-6 of a
    // distance of 0.05 units (the reference's cutoff in its own scenarios is 0.145 units)
    s->fc.precise_near = s->fc.cth < 0.05f;
}

// Exchange order = bodies sorted by a 2 x 16-bit Morton code of their initial positions.
// Index ranges of it are compact patches of space, which keeps the 64 bodies a wave walks the
// tree for close together when the range is cut across GPUs, and gives the direct-sum tiles
// small bounding boxes.  Only locality matters here, not the reference's cell boundaries.
void exchange_order(const nb_body *b, int n, std::vector<uint32_t> &order)
{
    order.resize(n);
    double mnx = b[0].position.x, mxx = mnx, mny = b[0].position.y, mxy = mny;
    for (int i = 1; i < n; ++i) {
        mnx = std::min(mnx, b[i].position.x);
        mxx = std::max(mxx, b[i].position.x);
        mny = std::min(mny, b[i].position.y);
        mxy = std::max(mxy, b[i].position.y);
    }
    const double sx = mxx > mnx ? 65535.0 / (mxx - mnx) : 0.0, sy = mxy > mny ? 65535.0 / (mxy - mny) : 0.0;
    auto spread = [](uint32_t v) {
        v = (v | (v << 8)) & 0x00FF00FFu;
        v = (v | (v << 4)) & 0x0F0F0F0Fu;
        v = (v | (v << 2)) & 0x33333333u;
        v = (v | (v << 1)) & 0x55555555u;
        return v;
    };
    std::vector<uint64_t> kv(n);
    for (int i = 0; i < n; ++i) {
        const uint32_t qx = (uint32_t)((b[i].position.x - mnx) * sx), qy = (uint32_t)((b[i].position.y - mny) * sy);
        kv[i] = ((uint64_t)(spread(qx) | (spread(qy) << 1)) << 32) | (uint32_t)i;
    }
    std::sort(kv.begin(), kv.end());
    for (int i = 0; i < n; ++i)
        order[i] = (uint32_t)(kv[i] & 0xffffffffu);
}

void mark(nb_sim *s, int k) { s->timer.mark(s->stream, k); }

// Direct sum: its run-against-wave box tests rely on slot ranges being compact patches of space;
// bodies wander, so the order is renewed now and then.  Steps between two renewals: they cost a
// few passes over the per-body arrays, well below 1 % of the steps between at this interval.
long refresh_interval(int n)
{
    const long interval = 50000000L / (n > 0 ? n : 1);
    return interval < 64 ? 64 : (interval > 4096 ? 4096 : interval);
}

int step_local(nb_sim *s, int do_integrate)
{
    if (s->alg == NB_ALG_DIRECT) {
        mark(s, 5);
        NB_TRY(direct_step(s->stream, s->posm[s->cur], s->posm[s->cur ^ 1], s->rad_s, s->n, s->t0, s->t1 - s->t0, s->fc,
                           integrate_args(s, do_integrate), s->dyn, s->pos, s->pos2, s->radius, s->p.collision_th,
                           s->vel, s->acc, s->part, s->jsplit, s->ds_meta));
        mark(s, 6);
        return NB_OK;
    }
    BhStepArgs a = bh_args(s, do_integrate);
    mark(s, 0);
    NB_TRY(bh_bbox(s->stream, s->bh, a));
    mark(s, 1);
    NB_TRY(bh_keys(s->stream, s->bh, a));
    mark(s, 2);
    NB_TRY(bh_sort(s->stream, s->bh, a));
    mark(s, 3);
    NB_TRY(bh_build(s->stream, s->bh, a));
    mark(s, 4);
    NB_TRY(bh_com(s->stream, s->bh, a));
    mark(s, 5);
    NB_TRY(bh_force(s->stream, s->bh, a));
    mark(s, 6);
    return NB_OK;
}

int step_finish(nb_sim *s)
{
    s->first_kick = false;
    s->stepped = true;
    if (s->exchange != NB_EXCHANGE_POSITIONS)
        s->cells_steps += 1;  // (the cell exchange is suspended: its level is still looked at every 16 steps)
    if (s->alg == NB_ALG_DIRECT) {
        // records of the bodies other ranks advanced come from the gathered positions
        if (s->world > 1) {
            NB_TRY(pack_posm(s->stream, s->pos2, s->mass, s->radius, s->n, 0, s->t0, s->units, s->p.collision_th,
                             s->posm[s->cur ^ 1], s->rad_s));
            NB_TRY(pack_posm(s->stream, s->pos2, s->mass, s->radius, s->n, s->t1, s->n, s->units, s->p.collision_th,
                             s->posm[s->cur ^ 1], s->rad_s));
        }
        s->cur ^= 1;
        std::swap(s->pos, s->pos2);
    }
    return NB_OK;
}

int collect_timing(nb_sim *s) { return s->timer.step_done(s->stream); }

void release(nb_sim *s)
{
    (void)hipFree(s->pos);
    (void)hipFree(s->pos2);
    (void)hipFree(s->vel);
    (void)hipFree(s->acc);
    (void)hipFree(s->mass);
    (void)hipFree(s->radius);
    (void)hipFree(s->mass_radius);
    (void)hipFree(s->dyn);
    (void)hipFree(s->order_d);
    (void)hipFree(s->posm[0]);
    (void)hipFree(s->posm[1]);
    (void)hipFree(s->rad_s);
    (void)hipFree(s->part);
    (void)hipFree(s->ds_meta);
    bh_free(s->bh);
    refresh_free(s->rb);
    comm_destroy(s->comm);
    s->comm = nullptr;
    s->timer.destroy();
    if (s->own_stream && s->stream)
        (void)hipStreamDestroy(s->stream);
}

}  // namespace

extern "C" {

const char *nb_last_error(void) { return g_error.c_str(); }

const char *nb_version(void) { return "nbody_hip 0.1 (gfx950)"; }

int nb_create(const nb_params *params, int n, int alg, nb_sim **out)
{
    if (!out)
        return fail(NB_ERR_ARG, "nb_create: out is NULL");
    *out = nullptr;
    if (n < 1)
        return fail(NB_ERR_ARG, "ERROR: need to have at least 1 body");
    if (alg != NB_ALG_DIRECT && alg != NB_ALG_BARNES_HUT)
        return fail(NB_ERR_ARG, "nb_create: alg must be 0 (direct sum) or 1 (Barnes-Hut)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(NB_ERR_HIP, "nb_create: no HIP device is visible; this library has no CPU path");
    nb_sim *s = new nb_sim();
    if (params)
        s->p = *params;
    else
        nb_default_params(&s->p);
    if (!(s->p.theta >= 0) || !(s->p.collision_th >= 0) || !(s->p.eps >= 0)) {
        delete s;
        return fail(NB_ERR_ARG, "nb_create: theta, collision_th and eps must be >= 0");
    }
    if (s->p.bucket_depth < 0 || s->p.bucket_depth > 31) {
        delete s;
        return fail(NB_ERR_ARG, "nb_create: bucket_depth must be 0..31");
    }
    if (s->p.integrator != NB_INT_EULER && s->p.integrator != NB_INT_LEAPFROG) {
        delete s;
        return fail(NB_ERR_ARG, "nb_create: integrator must be NB_INT_EULER or NB_INT_LEAPFROG");
    }
    s->n = n;
    s->alg = alg;
    s->t0 = 0;
    s->t1 = n;
    s->chunk = (size_t)n;
    s->cap = (size_t)n + 64;
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
    chk(dev_alloc(s->pos, s->cap));
    chk(dev_alloc(s->vel, s->cap));
    chk(dev_alloc(s->acc, s->cap));
    chk(dev_alloc(s->mass, s->cap));
    chk(dev_alloc(s->radius, s->cap));
    chk(dev_alloc(s->dyn, s->cap));
    chk(dev_alloc(s->order_d, s->cap));
    if (alg == NB_ALG_DIRECT) {
        chk(dev_alloc(s->pos2, s->cap));
        for (int k = 0; k < 2; ++k) {
            chk(dev_alloc(s->posm[k], direct_record_floats4(n)));
            if (rc == NB_OK && hipMemset(s->posm[k], 0, sizeof(float4) * direct_record_floats4(n)) != hipSuccess)
                chk(fail(NB_ERR_HIP, "hipMemset failed"));
        }
        chk(dev_alloc(s->rad_s, s->cap));
        s->jsplit = direct_choose_jsplit(n, n);
        s->part_cap = (size_t)s->jsplit * n;
        chk(dev_alloc(s->part, s->part_cap));
        chk(dev_alloc(s->ds_meta, direct_meta_bytes(n)));
        chk(refresh_alloc(s->rb, n));  // nb_step renews the order by itself for the direct sum
    } else {
        chk(dev_alloc(s->mass_radius, s->cap));
        chk(bh_alloc(s->bh, n));
    }
    if (rc == NB_OK)
        chk(s->timer.create());
    if (rc == NB_OK)
        chk(static_scratch_init(s->device));
    if (rc != NB_OK) {
        release(s);
        delete s;
        return rc;
    }
    *out = s;
    return NB_OK;
}

int nb_destroy(nb_sim *s)
{
    if (!s)
        return NB_OK;
    (void)hipStreamSynchronize(s->stream);
    release(s);
    delete s;
    return NB_OK;
}

int nb_set_stream(nb_sim *s, void *hip_stream)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_set_stream: sim is NULL");
    NB_HIP(hipStreamSynchronize(s->stream));
    if (s->own_stream && s->stream)
        (void)hipStreamDestroy(s->stream);
    s->stream = (hipStream_t)hip_stream;
    s->own_stream = false;
    return NB_OK;
}

// Units, exchange order and device arrays from bodies given in the caller's order.
static int load_bodies(nb_sim *s, const nb_body *b, int n)
{
    choose_units(s, b, n);
    exchange_order(b, n, s->order);
    std::vector<double2> pos(n), vel(n), acc(n);
    std::vector<double> mass(n), radius(n);
    std::vector<uint8_t> dyn(n);
    for (int k = 0; k < n; ++k) {
        const nb_body &x = b[s->order[k]];
        pos[k] = make_double2(x.position.x, x.position.y);
        vel[k] = make_double2(x.velocity.x, x.velocity.y);
        acc[k] = make_double2(x.acceleration.x, x.acceleration.y);
        mass[k] = x.mass;
        radius[k] = x.radius;
        dyn[k] = x.is_dynamic ? 1 : 0;
    }
    NB_HIP(hipStreamSynchronize(s->stream));
    NB_HIP(hipMemcpy(s->pos, pos.data(), sizeof(double2) * n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->vel, vel.data(), sizeof(double2) * n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->acc, acc.data(), sizeof(double2) * n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->mass, mass.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->radius, radius.data(), sizeof(double) * n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->dyn, dyn.data(), n, hipMemcpyHostToDevice));
    NB_HIP(hipMemcpy(s->order_d, s->order.data(), sizeof(uint32_t) * n, hipMemcpyHostToDevice));
    s->order_stale = false;
    if (s->alg == NB_ALG_BARNES_HUT) {
        std::vector<double2> mr(n);
        for (int k = 0; k < n; ++k)
            mr[k] = make_double2(mass[k], radius[k]);
        NB_HIP(hipMemcpy(s->mass_radius, mr.data(), sizeof(double2) * n, hipMemcpyHostToDevice));
    }
    if (s->alg == NB_ALG_DIRECT) {
        s->cur = 0;
        NB_TRY(pack_posm(s->stream, s->pos, s->mass, s->radius, n, 0, n, s->units, s->p.collision_th, s->posm[0], s->rad_s));
        NB_TRY(pad_posm(s->stream, s->posm[0], n));
        // mass and cutoff radius never change: give the second buffer the same records
        NB_HIP(hipMemcpyAsync(s->posm[1], s->posm[0], sizeof(float4) * direct_record_floats4(n),
                              hipMemcpyDeviceToDevice, s->stream));
    }
    NB_HIP(hipStreamSynchronize(s->stream));
    if (s->alg == NB_ALG_BARNES_HUT)
        NB_TRY(bh_prime(s->stream, s->bh, bh_args(s, 0)));
    s->steps_in_order = 0;
    return NB_OK;
}

int nb_set_bodies(nb_sim *s, const nb_body *b, int n)
{
    if (!s || !b)
        return fail(NB_ERR_ARG, "nb_set_bodies: NULL argument");
    if (n != s->n)
        return fail(NB_ERR_ARG, "nb_set_bodies: n differs from the n given to nb_create");
    for (int i = 0; i < n; ++i)
        if (!std::isfinite(b[i].position.x) || !std::isfinite(b[i].position.y) || !std::isfinite(b[i].mass))
            return fail(NB_ERR_ARG, "nb_set_bodies: body " + std::to_string(i) + " is not finite");
    NB_TRY(load_bodies(s, b, n));
    s->have_bodies = true;
    s->mid_step = false;
    s->first_kick = true;
    s->stepped = false;
    return NB_OK;
}

// The slot -> caller index map on the host (fetched again after the device renewed it).
static int host_order(nb_sim *s)
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
// renewal saw (the device leaves it in host-visible memory; the host never waits for it), so a
// drifting system keeps its precision.
static int device_refresh(nb_sim *s)
{
    if (s->rb.n != s->n)
        NB_TRY(refresh_alloc(s->rb, s->n));
    if (s->rb.origin_set && s->alg == NB_ALG_DIRECT) {
        s->units.origin[0] = ((volatile double *)s->rb.origin_host)[0];
        s->units.origin[1] = ((volatile double *)s->rb.origin_host)[1];
    }
    NB_TRY(refresh_sort(s->stream, s->rb, (const double *)s->pos, 2, s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->pos, sizeof(double2), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->vel, sizeof(double2), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->acc, sizeof(double2), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->mass, sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->radius, sizeof(double), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->mass_radius, sizeof(double2), s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->dyn, 1, s->n));
    NB_TRY(refresh_gather(s->stream, s->rb, s->order_d, sizeof(uint32_t), s->n));
    s->order_stale = true;
    if (s->alg == NB_ALG_DIRECT) {
        s->cur = 0;
        NB_TRY(pack_posm(s->stream, s->pos, s->mass, s->radius, s->n, 0, s->n, s->units, s->p.collision_th, s->posm[0],
                         s->rad_s));
        NB_TRY(pad_posm(s->stream, s->posm[0], s->n));
        NB_HIP(hipMemcpyAsync(s->posm[1], s->posm[0], sizeof(float4) * direct_record_floats4(s->n), hipMemcpyDeviceToDevice,
                              s->stream));
    }
    s->steps_in_order = 0;
    return NB_OK;
}

int nb_refresh_order(nb_sim *s)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_refresh_order: sim is NULL");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb_refresh_order: no bodies were set");
    if (s->mid_step)
        return fail(NB_ERR_STATE, "nb_refresh_order: a split step is in flight");
    // (a partitioned simulation: the caller has all-gathered the velocity buffer, see nbody.h)
    return device_refresh(s);
}

int nb_velocity_buffer(nb_sim *s, void **device_ptr, size_t *chunk_records, size_t *record_bytes)
{
    if (!s || !device_ptr || !chunk_records || !record_bytes)
        return fail(NB_ERR_ARG, "nb_velocity_buffer: NULL argument");
    *device_ptr = s->vel;
    *chunk_records = s->chunk;
    *record_bytes = sizeof(double2);
    return NB_OK;
}

int nb_setup(nb_sim *s, int scenario, unsigned seed)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_setup: sim is NULL");
    if (scenario == NB_SIM_SOLAR && s->n != 5)
        return fail(NB_ERR_ARG, "nb_setup: the solar system has 5 bodies (create the simulation with n = 5)");
    std::vector<nb_body> b(s->n);
    NB_TRY(nb_init_scenario(scenario, s->n, seed, b.data()));
    return nb_set_bodies(s, b.data(), s->n);
}

int nb_step_local(nb_sim *s)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_step_local: sim is NULL");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb_step_local: no bodies were set");
    if (s->mid_step)
        return fail(NB_ERR_STATE, "nb_step_local: the previous step was not finished");
    if (s->cells_active)
        return fail(NB_ERR_STATE, "nb_step_local: with the cell exchange a step goes through nb_step_cells");
    NB_TRY(step_local(s, 1));
    s->mid_step = true;
    return NB_OK;
}

int nb_step_finish(nb_sim *s)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_step_finish: sim is NULL");
    if (!s->mid_step)
        return fail(NB_ERR_STATE, "nb_step_finish: nb_step_local was not called");
    NB_TRY(step_finish(s));
    s->mid_step = false;
    return collect_timing(s);
}

int nb_step(nb_sim *s, int nsteps)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_step: sim is NULL");
    if (nsteps < 0)
        return fail(NB_ERR_ARG, "nb_step: nsteps < 0");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb_step: no bodies were set");
    if (s->world != 1)
        return fail(NB_ERR_STATE, "nb_step: a partitioned simulation steps with nb_step_local / nb_step_finish");
    if (s->mid_step)
        return fail(NB_ERR_STATE, "nb_step: a split step is in flight");
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

// ---- cell exchange ----

// The rank's own bodies as a simulation of their own: arrays offset to slot t0, n = their number.
static BhStepArgs cells_args(nb_sim *s, int do_integrate)
{
    BhStepArgs a = bh_args(s, do_integrate);
    const size_t off = (size_t)s->t0;
    a.pos += off;
    a.mass += off;
    a.radius += off;
    a.mass_radius += off;
    a.dyn += off;
    a.vel += off;
    a.acc += off;
    a.pos_out += off;
    a.n = s->t1 - s->t0;
    a.t0 = 0;
    a.t1 = a.n;
    return a;
}

// Digits for the keys of the coming steps, from the box of the last finished step (the caller has
// waited for the stream, so every rank reads the same box): the level at which a box half as large
// again would close the tree, so that the box may grow for a while.  -1: the positions are the
// better thing to exchange -- the tree closes below the dense pyramid's last level, or
// (NB_EXCHANGE_AUTO) the cell records, which an all-reduce moves about twice, are more data than
// the positions an all-gather moves once.
static int choose_cells_levels(nb_sim *s)
{
    const double w = ((volatile double *)s->bh.hint_box)[0], h = ((volatile double *)s->bh.hint_box)[1];
    const int l = closed_level_for(1.5 * w, 1.5 * h, s->p.theta, s->p.collision_th);
    if (l > cells_max_level())
        return -1;
    if (s->exchange == NB_EXCHANGE_AUTO && 2 * cells_agg_doubles(l) * sizeof(double) > sizeof(double2) * (size_t)s->n)
        return -1;
    return l;
}

int nb_set_exchange(nb_sim *s, int mode)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_set_exchange: sim is NULL");
    if (mode != NB_EXCHANGE_POSITIONS && mode != NB_EXCHANGE_CELLS && mode != NB_EXCHANGE_AUTO)
        return fail(NB_ERR_ARG, "nb_set_exchange: unknown mode");
    if (s->mid_step || s->cells_phase != 0)
        return fail(NB_ERR_STATE, "nb_set_exchange: a split step is in flight");
    if (mode == NB_EXCHANGE_POSITIONS) {
        if (s->cells_active && s->stepped)
            return fail(NB_ERR_STATE, "nb_set_exchange: after steps with the cell exchange this rank no longer holds the "
                                      "other ranks' positions; set the bodies again");
        s->exchange = mode;
        s->cells_active = false;
        return NB_OK;
    }
    if (s->alg != NB_ALG_BARNES_HUT || s->p.bucket_depth != 0 || s->p.integrator != NB_INT_EULER)
        return fail(NB_ERR_ARG, "nb_set_exchange: the cell exchange is for Barnes-Hut with the reference's CPU tree and "
                                "its Euler integration");
    if (!s->have_bodies || s->stepped)
        return fail(NB_ERR_STATE, "nb_set_exchange: switch the cell exchange on after the bodies and the partition were "
                                  "set and before the first step");
    NB_TRY(bh_cells_alloc(s->bh));
    NB_HIP(hipStreamSynchronize(s->stream));
    const int before = s->exchange;
    s->exchange = mode;
    const int levels = choose_cells_levels(s);
    if (levels < 0 && mode == NB_EXCHANGE_CELLS) {
        s->exchange = before;
        return fail(NB_ERR_OVERFLOW, "nb_set_exchange: with these parameters the tree closes below level " +
                                         std::to_string(cells_max_level()) + "; use the position exchange");
    }
    s->cells_levels = levels;
    s->cells_active = levels >= 0;
    s->cells_steps = 0;
    s->cells_checked = -1;
    return NB_OK;
}

// Every 16 steps the host waits for the stream and looks at the box the device left: the level of
// the exchanged cells follows it, and when the records of that level stop being the cheaper thing
// to exchange (or the dense pyramid cannot hold the tree any more) the cell exchange is suspended
// until they are again.  The decision is the same on every rank: the box is the reduced one, or,
// while suspended, the box of all positions, which every rank holds.  *gather: the cell exchange
// was suspended just now, so the ranks must all-gather their positions before the next step.
static int cells_housekeeping(nb_sim *s, bool *gather)
{
    *gather = false;
    if (s->exchange == NB_EXCHANGE_POSITIONS || s->cells_steps == 0 || s->cells_steps % 16 != 0 ||
        s->cells_checked == s->cells_steps)
        return NB_OK;
    s->cells_checked = s->cells_steps;
    NB_HIP(hipStreamSynchronize(s->stream));
    if (s->cells_active && *(volatile uint32_t *)s->bh.err_host != 0u)
        return fail(NB_ERR_OVERFLOW, "cell exchange: the bounding box grew by more than half within 16 steps; the last "
                                     "steps' tree was cut above the level it closes at");
    const int levels = choose_cells_levels(s);
    if (levels < 0) {
        *gather = s->cells_active;
        s->cells_active = false;
    } else {
        s->cells_levels = levels;
        s->cells_active = true;
    }
    return NB_OK;
}

int nb_cells_level(nb_sim *s, int *level, int *gather_positions_first)
{
    if (!s || !level || !gather_positions_first)
        return fail(NB_ERR_ARG, "nb_cells_level: NULL argument");
    if (s->mid_step || s->cells_phase != 0)
        return fail(NB_ERR_STATE, "nb_cells_level: a split step is in flight");
    bool gather = false;
    NB_TRY(cells_housekeeping(s, &gather));
    *level = s->cells_active ? s->cells_levels : -1;
    *gather_positions_first = gather ? 1 : 0;
    return NB_OK;
}

int nb_cells_buffer(nb_sim *s, int which, void **device_ptr, size_t *count)
{
    if (!s || !device_ptr || !count)
        return fail(NB_ERR_ARG, "nb_cells_buffer: NULL argument");
    if (!s->cells_active)
        return fail(NB_ERR_STATE, "nb_cells_buffer: the cell exchange is not switched on");
    if (which == 0) {
        *device_ptr = s->bh.red4;
        *count = 4;
    } else if (which == 1) {
        *device_ptr = s->bh.agg;
        *count = cells_agg_doubles(s->cells_levels);
    } else
        return fail(NB_ERR_ARG, "nb_cells_buffer: which must be 0 or 1");
    return NB_OK;
}

int nb_step_cells(nb_sim *s, int phase)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_step_cells: sim is NULL");
    if (s->exchange == NB_EXCHANGE_POSITIONS)
        return fail(NB_ERR_STATE, "nb_step_cells: the cell exchange is not switched on");
    if (phase != s->cells_phase)
        return fail(NB_ERR_STATE, "nb_step_cells: the parts of a step go 0, 1, 2");
    if (phase == 0) {
        bool gather = false;
        NB_TRY(cells_housekeeping(s, &gather));
        if (!s->cells_active)
            return fail(NB_ERR_STATE, "nb_step_cells: the cell exchange is suspended (nb_cells_level says so before "
                                      "a step): all-gather the positions and step with nb_step_local / nb_step_finish");
    }
    const BhStepArgs a = cells_args(s, 1);
    constexpr int EXCHANGE = 6;  // the caller's collectives run between the parts
    if (phase == 0) {
        mark(s, 0);
        NB_TRY(bh_cells_begin(s->stream, s->bh, a));
        mark(s, EXCHANGE);
    } else if (phase == 1) {
        mark(s, 1);
        NB_TRY(bh_cells_middle(s->stream, s->bh, a, s->cells_levels, 0));
        mark(s, 2);
        NB_TRY(bh_cells_middle(s->stream, s->bh, a, s->cells_levels, 1));
        mark(s, 4);
        NB_TRY(bh_cells_middle(s->stream, s->bh, a, s->cells_levels, 2));
        mark(s, EXCHANGE);
    } else {
        mark(s, 3);
        NB_TRY(bh_cells_end(s->stream, s->bh, a, s->cells_levels, 0));
        mark(s, 5);
        NB_TRY(bh_cells_end(s->stream, s->bh, a, s->cells_levels, 1));
        mark(s, StageTimer::STAGE_END);
        s->first_kick = false;
        s->stepped = true;
        s->cells_steps += 1;
        NB_TRY(collect_timing(s));
    }
    s->cells_phase = (phase + 1) % 3;
    return NB_OK;
}

int nb_comm_unique_id(void *id, size_t bytes)
{
    if (bytes < COMM_ID_BYTES)
        return fail(NB_ERR_ARG, "nb_comm_unique_id: the buffer must hold NB_COMM_ID_BYTES bytes");
    return comm_unique_id(id);
}

int nb_comm_init(nb_sim *s, int rank, int world, const void *id)
{
    if (!s || !id)
        return fail(NB_ERR_ARG, "nb_comm_init: NULL argument");
    if (s->comm)
        return fail(NB_ERR_STATE, "nb_comm_init: the simulation already has a communicator");
    int device = -1;
    NB_HIP(hipGetDevice(&device));
    if (device != s->device)
        return fail(NB_ERR_STATE, "nb_comm_init: the handle's device is not the current one");
    NB_TRY(nb_set_partition(s, rank, world));
    return comm_create(&s->comm, rank, world, id);
}

int nb_step_distributed(nb_sim *s, int nsteps)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_step_distributed: sim is NULL");
    if (nsteps < 0)
        return fail(NB_ERR_ARG, "nb_step_distributed: nsteps < 0");
    if (!s->comm)
        return fail(NB_ERR_STATE, "nb_step_distributed: call nb_comm_init first");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb_step_distributed: no bodies were set");
    if (s->mid_step || s->cells_phase != 0)
        return fail(NB_ERR_STATE, "nb_step_distributed: a split step is in flight");
    for (int k = 0; k < nsteps; ++k) {
        bool gather = false;
        NB_TRY(cells_housekeeping(s, &gather));
        if (gather)  // the cell exchange was suspended: from here on every rank holds all positions again
            NB_TRY(comm_all_gather_f64(s->comm, (double *)s->pos, 2 * s->chunk, s->stream));
        if (s->cells_active) {
            // two reductions per step instead of the gather of all positions: the ranks' extents (4
            // doubles, maximum) and the records of the cells of the keys' last level (sum)
            NB_TRY(nb_step_cells(s, 0));
            NB_TRY(comm_all_reduce_max_f64(s->comm, s->bh.red4, 4, s->stream));
            NB_TRY(nb_step_cells(s, 1));
            NB_TRY(comm_all_reduce_sum_f64(s->comm, s->bh.agg, cells_agg_doubles(s->cells_levels), s->stream));
            NB_TRY(nb_step_cells(s, 2));
            continue;
        }
        NB_TRY(step_local(s, 1));
        // the new positions of every rank's slice, in place, ordered on the stream with the kernels
        // before and after: the host does not wait
        double2 *buf = s->alg == NB_ALG_DIRECT ? s->pos2 : s->pos;
        NB_TRY(comm_all_gather_f64(s->comm, (double *)buf, 2 * s->chunk, s->stream));
        mark(s, StageTimer::STAGE_END);
        NB_TRY(step_finish(s));
        NB_TRY(collect_timing(s));
        if (s->alg == NB_ALG_DIRECT && ++s->steps_in_order >= refresh_interval(s->n)) {
            // every rank renews the order from the same positions; the new slices hold bodies other
            // ranks advanced so far, whose velocities come with one more all-gather
            NB_TRY(comm_all_gather_f64(s->comm, (double *)s->vel, 2 * s->chunk, s->stream));
            NB_TRY(device_refresh(s));
        }
    }
    return NB_OK;
}

int nb_compute_accel(nb_sim *s)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_compute_accel: sim is NULL");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb_compute_accel: no bodies were set");
    if (s->mid_step)
        return fail(NB_ERR_STATE, "nb_compute_accel: a split step is in flight");
    if (s->cells_active)
        return fail(NB_ERR_STATE, "nb_compute_accel: with the cell exchange this rank does not hold the other ranks' "
                                  "positions");
    NB_TRY(step_local(s, 0));
    return collect_timing(s);
}

int nb_sync(nb_sim *s)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_sync: sim is NULL");
    NB_HIP(hipStreamSynchronize(s->stream));
    return NB_OK;
}

int nb_read_bodies(nb_sim *s, nb_body *out)
{
    if (!s || !out)
        return fail(NB_ERR_ARG, "nb_read_bodies: NULL argument");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb_read_bodies: no bodies were set");
    const int n = s->n;
    std::vector<double2> pos(n), vel(n), acc(n);
    std::vector<double> mass(n), radius(n);
    std::vector<uint8_t> dyn(n);
    // leapfrog keeps velocities half a step behind the positions; bring them level with the
    // force at the current positions
    if (s->cells_phase != 0)
        return fail(NB_ERR_STATE, "nb_read_bodies: a split step is in flight");
    if (s->cells_active && s->comm)  // the other ranks' positions (a collective call)
        NB_TRY(comm_all_gather_f64(s->comm, (double *)s->pos, 2 * s->chunk, s->stream));
    const bool half_step = s->p.integrator == NB_INT_LEAPFROG && !s->first_kick && !s->mid_step &&
                           s->exchange == NB_EXCHANGE_POSITIONS;
    if (half_step) {
        // this force evaluation is not a step: keep it out of the per-stage timing
        const int timing = s->timer.on;
        s->timer.on = 0;
        const int rc = step_local(s, 0);
        s->timer.on = timing;
        NB_TRY(rc);
    }
    NB_HIP(hipStreamSynchronize(s->stream));
    NB_TRY(host_order(s));
    NB_HIP(hipMemcpy(pos.data(), s->pos, sizeof(double2) * n, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(vel.data(), s->vel, sizeof(double2) * n, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(acc.data(), s->acc, sizeof(double2) * n, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(mass.data(), s->mass, sizeof(double) * n, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(radius.data(), s->radius, sizeof(double) * n, hipMemcpyDeviceToHost));
    NB_HIP(hipMemcpy(dyn.data(), s->dyn, n, hipMemcpyDeviceToHost));
    for (int k = 0; k < n; ++k) {
        nb_body &x = out[s->order[k]];
        std::memset(&x, 0, sizeof(x));
        x.is_dynamic = dyn[k];
        x.mass = mass[k];
        x.radius = radius[k];
        x.position = {pos[k].x, pos[k].y};
        x.velocity = {vel[k].x, vel[k].y};
        x.acceleration = {acc[k].x, acc[k].y};
        if (half_step && k >= s->t0 && k < s->t1) {
            x.velocity.x += 0.5 * s->p.dt * acc[k].x;
            x.velocity.y += 0.5 * s->p.dt * acc[k].y;
        }
    }
    return NB_OK;
}

int nb_diagnostics(nb_sim *s, double *out, int count)
{
    if (!s || !out || count < 1)
        return fail(NB_ERR_ARG, "nb_diagnostics: bad argument");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb_diagnostics: no bodies were set");
    std::vector<nb_body> b(s->n);
    NB_TRY(nb_read_bodies(s, b.data()));
    double v[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (const nb_body &x : b) {
        v[0] += x.mass;
        v[1] += x.mass * x.position.x;
        v[2] += x.mass * x.position.y;
        v[3] += x.mass * x.velocity.x;
        v[4] += x.mass * x.velocity.y;
        v[5] += x.mass * (x.position.x * x.velocity.y - x.position.y * x.velocity.x);
        v[6] += 0.5 * x.mass * (x.velocity.x * x.velocity.x + x.velocity.y * x.velocity.y);
    }
    if (v[0] != 0) {
        v[1] /= v[0];
        v[2] /= v[0];
    }
    v[7] = std::nan("");
    if (count > 7 && s->n <= 131072) {
        double *d = nullptr;
        NB_TRY(dev_alloc(d, 1));
        int rc = potential_energy(s->stream, s->pos, s->mass, s->radius, s->n, s->p.gravity, s->p.eps,
                                  s->p.collision_th, d);
        if (rc == NB_OK && hipStreamSynchronize(s->stream) == hipSuccess)
            (void)hipMemcpy(&v[7], d, sizeof(double), hipMemcpyDeviceToHost);
        (void)hipFree(d);
        NB_TRY(rc);
    }
    for (int k = 0; k < count && k < 8; ++k)
        out[k] = v[k];
    return NB_OK;
}

int nb_set_partition(nb_sim *s, int rank, int world)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_set_partition: sim is NULL");
    if (world < 1 || world > 64 || rank < 0 || rank >= world)
        return fail(NB_ERR_ARG, "nb_set_partition: need 0 <= rank < world <= 64");
    if (!s->have_bodies)
        return fail(NB_ERR_STATE, "nb_set_partition: set the bodies first");
    if (s->mid_step)
        return fail(NB_ERR_STATE, "nb_set_partition: a split step is in flight");
    if (s->stepped)
        return fail(NB_ERR_STATE, "nb_set_partition: the simulation was stepped already; set the bodies again first");
    s->rank = rank;
    s->world = world;
    s->chunk = ((size_t)s->n + world - 1) / world;
    s->t0 = (int)std::min((size_t)s->n, rank * s->chunk);
    s->t1 = (int)std::min((size_t)s->n, (rank + 1) * s->chunk);
    if (s->alg == NB_ALG_DIRECT) {
        // fewer targets per rank: split the sources further so the chip stays full
        const int js = direct_choose_jsplit(s->n, s->t1 - s->t0);
        const size_t need = (size_t)js * (size_t)(s->t1 - s->t0);
        if (need > s->part_cap) {
            (void)hipFree(s->part);
            s->part = nullptr;
            s->part_cap = 0;
            NB_TRY(dev_alloc(s->part, need));
            s->part_cap = need;
        }
        s->jsplit = js;
    }
    return NB_OK;
}

int nb_exchange_buffer(nb_sim *s, void **device_ptr, size_t *chunk_records, size_t *record_bytes)
{
    if (!s || !device_ptr || !chunk_records || !record_bytes)
        return fail(NB_ERR_ARG, "nb_exchange_buffer: NULL argument");
    // direct sum writes the new positions to the other buffer; mid-step that is the one to gather
    *device_ptr = (s->alg == NB_ALG_DIRECT && s->mid_step) ? s->pos2 : s->pos;
    *chunk_records = s->chunk;
    *record_bytes = sizeof(double2);
    return NB_OK;
}

int nb_set_stage_timing(nb_sim *s, int on)
{
    if (!s)
        return fail(NB_ERR_ARG, "nb_set_stage_timing: sim is NULL");
    s->collect_stats = (on & 2) ? 1 : 0;
    return s->timer.set(s->stream, on & 1);
}

int nb_stage_ms(nb_sim *s, float *out, int count)
{
    if (!s || !out)
        return fail(NB_ERR_ARG, "nb_stage_ms: NULL argument");
    return s->timer.read(s->stream, out, count);
}

int nb_debug_read(nb_sim *s, int which, void *out, size_t bytes, size_t *written)
{
    if (!s || !out)
        return fail(NB_ERR_ARG, "nb_debug_read: NULL argument");
    NB_HIP(hipStreamSynchronize(s->stream));
    const void *src = nullptr;
    size_t need = 0;
    std::vector<uint32_t> tmp;
    const bool is_bh = s->alg == NB_ALG_BARNES_HUT;
    uint32_t ls[BH_MAX_DEPTH + 2] = {};
    size_t nnodes = 0;
    if (is_bh) {
        NB_HIP(hipMemcpy(ls, s->bh.level_start, sizeof(ls), hipMemcpyDeviceToHost));
        for (int d = 0; d < BH_MAX_DEPTH + 1; ++d)
            nnodes = std::max(nnodes, (size_t)ls[d + 1]);
        if (nnodes > s->bh.node_cap)
            nnodes = 0;  // no tree was built yet
    }
    switch (which) {
    case NB_DBG_ORDER:
        src = nullptr;
        need = sizeof(uint32_t) * s->n;
        if (bytes < need)
            return fail(NB_ERR_ARG, "nb_debug_read: buffer too small");
        NB_TRY(host_order(s));
        std::memcpy(out, s->order.data(), need);
        if (written)
            *written = need;
        return NB_OK;
    case NB_DBG_BBOX:
        src = s->bh.box;
        need = 4 * sizeof(double);
        break;
    case NB_DBG_KEYS: {
        // the sort ping-pongs over the unsorted keys; rebuild them from the sorted pairs
        if (!is_bh)
            return fail(NB_ERR_STATE, "nb_debug_read: not a Barnes-Hut simulation");
        need = sizeof(uint64_t) * s->n;
        if (bytes < need)
            return fail(NB_ERR_ARG, "nb_debug_read: buffer too small");
        std::vector<uint64_t> sk(s->n);
        std::vector<uint32_t> si(s->n);
        NB_HIP(hipMemcpy(sk.data(), s->bh.keys[s->bh.sorted], need, hipMemcpyDeviceToHost));
        NB_HIP(hipMemcpy(si.data(), s->bh.idx[s->bh.sorted], sizeof(uint32_t) * s->n, hipMemcpyDeviceToHost));
        uint64_t *o = (uint64_t *)out;
        for (int k = 0; k < s->n; ++k)
            o[si[k]] = sk[k];
        if (written)
            *written = need;
        return NB_OK;
    }
    case NB_DBG_SORTED_KEYS:
        src = s->bh.keys[s->bh.sorted];
        need = sizeof(uint64_t) * s->n;
        break;
    case NB_DBG_SORTED_IDX:
        src = s->bh.idx[s->bh.sorted];
        need = sizeof(uint32_t) * s->n;
        break;
    case NB_DBG_NODE_RANGE:
    case NB_DBG_NODE_CHILD: {
        if (!is_bh)
            return fail(NB_ERR_STATE, "nb_debug_read: not a Barnes-Hut simulation");
        need = sizeof(uint32_t) * 2 * nnodes;
        if (bytes < need)
            return fail(NB_ERR_ARG, "nb_debug_read: buffer too small");
        std::vector<uint4> meta(nnodes);
        NB_HIP(hipMemcpy(meta.data(), s->bh.node_meta, sizeof(uint4) * nnodes, hipMemcpyDeviceToHost));
        uint32_t *o = (uint32_t *)out;
        for (size_t u = 0; u < nnodes; ++u) {
            if (which == NB_DBG_NODE_RANGE) {
                o[2 * u] = meta[u].x;
                o[2 * u + 1] = meta[u].y;
            } else {
                o[2 * u] = meta[u].z;
                o[2 * u + 1] = meta[u].w & 0xffu;
            }
        }
        if (written)
            *written = need;
        return NB_OK;
    }
    case NB_DBG_NODE_COM:
        src = s->bh.node_com;
        need = sizeof(double) * 4 * nnodes;
        break;
    case NB_DBG_LEVEL_START:
        need = sizeof(uint32_t) * (BH_MAX_DEPTH + 2);
        if (bytes < need)
            return fail(NB_ERR_ARG, "nb_debug_read: buffer too small");
        std::memcpy(out, ls, need);
        if (written)
            *written = need;
        return NB_OK;
    case NB_DBG_KEY_LEVELS:
        src = s->bh.cut;
        need = sizeof(uint32_t) * 2;
        break;
    case NB_DBG_COUNTERS:
        src = s->bh.counters;
        need = sizeof(unsigned long long) * 8;
        break;
    default:
        return fail(NB_ERR_ARG, "nb_debug_read: unknown array");
    }
    if (!is_bh)
        return fail(NB_ERR_STATE, "nb_debug_read: not a Barnes-Hut simulation");
    if (bytes < need)
        return fail(NB_ERR_ARG, "nb_debug_read: buffer too small");
    if (need)
        NB_HIP(hipMemcpy(out, src, need, hipMemcpyDeviceToHost));
    if (written)
        *written = need;
    return NB_OK;
}

}  // extern "C"
