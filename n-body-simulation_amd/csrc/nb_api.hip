// nb_api.hip -- the C ABI of libnbody_hip.so (declared in include/nbody.h), two-dimensional bodies.
//
// nb_sim plays the role of the reference's BarnesHutCuda object
// (src/gpu/BH/barnesHutCuda.cuh:49-80): it owns the device arrays, takes the initial
// state once (setup) and advances it with update()-like steps, copying bodies back only
// when asked.  State on the device is SoA: positions and velocities in FP64 (the
// reference's precision, so integration adds no rounding of its own), and per step a
// float4 record per body for the FP32 force kernels.  The entry points the 3-D handle has too
// are written once, in nb_handle.h; this file holds the 2-D handle's own parts: the launches of
// nb_direct.hip and nb_bh.hip, the cell exchange between GPUs, diagnostics and introspection.
#include <algorithm>
#include <cmath>
#include <cstdlib>
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

BhStepArgs bh_args(nb_sim *s, int do_integrate)
{
    BhStepArgs a;
    a.pos = s->pos2d();
    a.mass = s->mass;
    a.radius = s->radius;
    a.mass_radius = s->mass_radius;
    a.dyn = s->dyn;
    a.vel = s->vel2d();
    a.acc = s->acc2d();
    a.pos_out = s->pos2d();
    a.n = s->n;
    a.t0 = s->t0;
    a.t1 = s->t1;
    a.units = s->units;
    a.fc = s->fc;
    a.ia = make_integrate_args(s, do_integrate);
    a.theta = s->p.theta;
    a.bbox_margin = s->p.bbox_margin;
    a.collision_th = s->p.collision_th;
    a.bucket_depth = s->p.bucket_depth;
    a.collect_stats = s->collect_stats;
    a.from = s->body_arrays(false);
    // (not in a step the caller drives through nb_step_local: the exchange buffer it was given stays the positions)
    if (do_integrate && s->own_loop && s->keeps_order_fresh() && s->bh_steps % s->resort_every() == 0)
        a.to = s->body_arrays(true);
    return a;
}

}  // namespace nb

using namespace nb;

// ---- what the shared entry points call (nb_handle.h) ----

int nb_sim::check_params(const nb_params &p)
{
    if (p.bucket_depth < 0 || p.bucket_depth > 31)
        return fail(NB_ERR_ARG, "nb_create: bucket_depth must be 0..31");
    return NB_OK;
}

int nb_sim::alloc_pipelines()
{
    if (alg == NB_ALG_DIRECT) {
        for (int k = 0; k < 2; ++k) {
            NB_TRY(dev_alloc(posm[k], direct_record_floats4(n)));
            NB_HIP(hipMemset(posm[k], 0, sizeof(float4) * direct_record_floats4(n)));
        }
        NB_TRY(dev_alloc(rad_s, cap));
        jsplit = direct_choose_jsplit(n, n);
        part_cap = (size_t)jsplit * n;
        NB_TRY(dev_alloc(part, part_cap));
        NB_TRY(dev_alloc(ds_meta, 2 * direct_meta_bytes(n)));  // one set of boxes per record buffer
    } else {
        NB_TRY(dev_alloc(mass_radius, cap));
        NB_TRY(bh_alloc(bh, n));
        // the second set of per-body arrays (85 bytes per body); NB_KEEP_SLOTS in the environment: none,
        // every body keeps its slot and the order is renewed every so many steps (tuning aid)
        if (std::getenv("NB_KEEP_SLOTS"))
            return static_scratch_init(device);
        NB_TRY(dev_alloc(pos2, 2 * cap));
        NB_TRY(dev_alloc(vel_b, 2 * cap));
        NB_TRY(dev_alloc(acc_b, 2 * cap));
        NB_TRY(dev_alloc(mass_b, cap));
        NB_TRY(dev_alloc(radius_b, cap));
        NB_TRY(dev_alloc(mass_radius_b, cap));
        NB_TRY(dev_alloc(dyn_b, cap));
        NB_TRY(dev_alloc(order_b, cap));
    }
    return static_scratch_init(device);
}

void nb_sim::free_pipelines()
{
    (void)hipFree(mass_radius);
    (void)hipFree(vel_b);
    (void)hipFree(acc_b);
    (void)hipFree(mass_b);
    (void)hipFree(radius_b);
    (void)hipFree(mass_radius_b);
    (void)hipFree(dyn_b);
    (void)hipFree(order_b);
    (void)hipFree(posm[0]);
    (void)hipFree(posm[1]);
    (void)hipFree(rad_s);
    (void)hipFree(part);
    (void)hipFree(ds_meta);
    bh_free(bh);
}

int nb_sim::direct_step(int do_integrate)
{
    timer.mark(stream, 5);
    NB_TRY(nb::direct_step(stream, posm[cur], posm[cur ^ 1], rad_s, n, t0, t1 - t0, fc, make_integrate_args(this, do_integrate),
                           dyn, pos2d(), pos2d_next(), radius, p.collision_th, vel2d(), acc2d(), part, jsplit,
                           ds_meta + cur * direct_meta_bytes(n), ds_meta + (cur ^ 1) * direct_meta_bytes(n), &ds_meta_valid));
    timer.mark(stream, 6);
    return NB_OK;
}

int nb_sim::bh_step(int do_integrate)
{
    BhStepArgs a = bh_args(this, do_integrate);
    bh.ring_step = (exchange != NB_EXCHANGE_POSITIONS && do_integrate) ? cells_steps : -1;
    timer.mark(stream, 0);
    NB_TRY(bh_bbox(stream, bh, a));
    timer.mark(stream, 1);
    NB_TRY(bh_keys(stream, bh, a));
    timer.mark(stream, 2);
    NB_TRY(bh_sort(stream, bh, a));
    timer.mark(stream, 3);
    NB_TRY(bh_build(stream, bh, a));
    timer.mark(stream, 4);
    NB_TRY(bh_com(stream, bh, a));
    timer.mark(stream, 5);
    NB_TRY(bh_force(stream, bh, a));
    timer.mark(stream, 6);
    if (do_integrate)
        bh_steps += 1;
    if (a.to.pos) {
        // the step left the bodies in the other set of arrays, in the order of its tree
        std::swap(pos, pos2);
        std::swap(vel, vel_b);
        std::swap(acc, acc_b);
        std::swap(mass, mass_b);
        std::swap(radius, radius_b);
        std::swap(mass_radius, mass_radius_b);
        std::swap(dyn, dyn_b);
        std::swap(order_d, order_b);
        order_stale = true;
    }
    return NB_OK;
}

BodyArrays2 nb_sim::body_arrays(bool other) const
{
    BodyArrays2 r;
    r.pos = (double2 *)(other ? pos2 : pos);
    r.vel = (double2 *)(other ? vel_b : vel);
    r.acc = (double2 *)(other ? acc_b : acc);
    r.mass = other ? mass_b : mass;
    r.radius = other ? radius_b : radius;
    r.mass_radius = other ? mass_radius_b : mass_radius;
    r.dyn = other ? dyn_b : dyn;
    r.order = other ? order_b : order_d;
    return r;
}

int nb_sim::direct_pack_all()
{
    cur = 0;
    ds_meta_valid = false;
    NB_TRY(pack_posm(stream, pos2d(), mass, radius, n, 0, n, units, p.collision_th, posm[0], rad_s));
    NB_TRY(pad_posm(stream, posm[0], n));
    // mass and cutoff radius never change: give the second buffer the same records
    NB_HIP(hipMemcpyAsync(posm[1], posm[0], sizeof(float4) * direct_record_floats4(n), hipMemcpyDeviceToDevice, stream));
    return NB_OK;
}

int nb_sim::direct_pack_others()
{
    ds_meta_valid = false;
    NB_TRY(pack_posm(stream, pos2d_next(), mass, radius, n, 0, t0, units, p.collision_th, posm[cur ^ 1], rad_s));
    return pack_posm(stream, pos2d_next(), mass, radius, n, t1, n, units, p.collision_th, posm[cur ^ 1], rad_s);
}

int nb_sim::direct_set_split(int nt)
{
    const int js = direct_choose_jsplit(n, nt);
    const size_t need = (size_t)js * (size_t)nt;
    if (need > part_cap) {
        (void)hipFree(part);
        part = nullptr;
        part_cap = 0;
        NB_TRY(dev_alloc(part, need));
        part_cap = need;
    }
    jsplit = js;
    return NB_OK;
}

int nb_sim::upload_extra(const double *mass_h, const double *radius_h)
{
    if (alg != NB_ALG_BARNES_HUT)
        return NB_OK;
    std::vector<double2> mr(n);
    for (int k = 0; k < n; ++k)
        mr[k] = make_double2(mass_h[k], radius_h[k]);
    NB_HIP(hipMemcpy(mass_radius, mr.data(), sizeof(double2) * n, hipMemcpyHostToDevice));
    return NB_OK;
}

int nb_sim::gather_extra() { return refresh_gather(stream, rb, mass_radius, sizeof(double2), n); }

// The key depth the first steps will find, so that their launches are sized right from the start.
int nb_sim::after_load()
{
    bh_steps = 0;
    return alg == NB_ALG_BARNES_HUT ? bh_prime(stream, bh, bh_args(this, 0)) : (int)NB_OK;
}

extern "C" {

const char *nb_last_error(void) { return g_error.c_str(); }

const char *nb_version(void) { return "nbody_hip 0.3 (gfx950)"; }

int nb_create(const nb_params *params, int n, int alg, nb_sim **out) { return api_create(params, n, alg, out); }

int nb_destroy(nb_sim *s) { return api_destroy(s); }

int nb_set_stream(nb_sim *s, void *hip_stream) { return api_set_stream(s, hip_stream); }

int nb_set_bodies(nb_sim *s, const nb_body *b, int n)
{
    NB_TRY(api_set_bodies(s, b, n));
    // the exchange mode belongs to a run: the new bodies start with the position exchange
    s->exchange = NB_EXCHANGE_POSITIONS;
    s->cells_active = false;
    s->cells_phase = 0;
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

int nb_refresh_order(nb_sim *s) { return api_refresh_order(s); }

int nb_velocity_buffer(nb_sim *s, void **device_ptr, size_t *chunk_records, size_t *record_bytes)
{
    return api_buffer(s, "velocity_buffer", s ? s->vel : nullptr, device_ptr, chunk_records, record_bytes);
}

int nb_exchange_buffer(nb_sim *s, void **device_ptr, size_t *chunk_records, size_t *record_bytes)
{
    // direct sum writes the new positions to the other buffer; mid-step that is the one to gather
    double *buf = !s ? nullptr : ((s->alg == NB_ALG_DIRECT && s->mid_step) ? s->pos2 : s->pos);
    return api_buffer(s, "exchange_buffer", buf, device_ptr, chunk_records, record_bytes);
}

int nb_step_local(nb_sim *s)
{
    if (s && s->cells_active)
        return fail(NB_ERR_STATE, "nb_step_local: with the cell exchange a step goes through nb_step_cells");
    return api_step_local(s);
}

int nb_step_finish(nb_sim *s) { return api_step_finish(s); }

int nb_step(nb_sim *s, int nsteps) { return api_step(s, nsteps); }

int nb_set_partition(nb_sim *s, int rank, int world) { return api_set_partition(s, rank, world); }

// ---- cell exchange ----

static void mark(nb_sim *s, int k) { s->timer.mark(s->stream, k); }

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
    a.to = BodyArrays2();  // (world > 1 here: every body keeps its slot)
    return a;
}

// The level of the exchanged cells follows the bounding box without the host ever waiting for the
// device and without the ranks talking about it: the device leaves every step's box (the reduced
// one, or while the cell exchange is suspended the box of all positions, which every rank holds --
// the same numbers on every rank) in a pinned ring, and step t uses the box of step t - CELLS_LAG,
// which has finished on every rank that enqueues step t (the host runs at most three steps ahead of
// the device, pace_wait), enlarged by CELLS_MARGIN: the box may grow 2.4 % a step.  The first steps
// use the box the bodies were set with.
constexpr long CELLS_LAG = 4;
constexpr double CELLS_MARGIN = 1.1;
static_assert(CELLS_LAG < nb::BOX_RING && CELLS_LAG > nb_sim::PACE_AHEAD, "the slot read is written and not yet reused");

// Digits for the keys of a step whose box is at most w x h: the level at which that box closes the
// tree.  -1: the positions are the better thing to exchange -- the tree closes below the dense
// pyramid's last level, or (NB_EXCHANGE_AUTO) the cell records, which an all-reduce moves about
// twice, are more data than the positions an all-gather moves once.
static int choose_cells_levels(nb_sim *s, double w, double h)
{
    const int l = closed_level_for(w, h, s->p.theta, s->p.collision_th);
    if (l > cells_max_level())
        return -1;
    if (s->exchange == NB_EXCHANGE_AUTO && 2 * cells_agg_doubles(l) * sizeof(double) > 2 * sizeof(double) * (size_t)s->n)
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
    // (the box nb_set_bodies left: of all bodies, the same on every rank)
    s->cells_box0[0] = ((volatile double *)s->bh.hint_box)[0];
    s->cells_box0[1] = ((volatile double *)s->bh.hint_box)[1];
    const int levels = choose_cells_levels(s, CELLS_MARGIN * s->cells_box0[0], CELLS_MARGIN * s->cells_box0[1]);
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

// Before every step: the level for it (see CELLS_LAG above).  When the records of that level stop
// being the cheaper thing to exchange, or the dense pyramid cannot hold the tree any more, the cell
// exchange is suspended until they are again.  *gather: it was suspended just now, so the ranks
// must all-gather their positions before the step.
static int cells_housekeeping(nb_sim *s, bool *gather)
{
    *gather = false;
    if (s->exchange == NB_EXCHANGE_POSITIONS || s->cells_checked == s->cells_steps)
        return NB_OK;
    NB_TRY(pace_wait(s));  // the step CELLS_LAG back has finished: its box is in the ring
    double w = s->cells_box0[0], h = s->cells_box0[1];
    if (s->cells_steps >= CELLS_LAG) {
        const volatile double *slot = s->bh.box_ring + BOX_SLOT * ((s->cells_steps - CELLS_LAG) % BOX_RING);
        w = slot[0];
        h = slot[1];
        if (slot[2] != 0.0)  // (read from the slot, not from a flag of its own: every rank fails at the same step)
            return fail(NB_ERR_OVERFLOW, "cell exchange: the bounding box grew by more than a tenth within four steps; "
                                         "a tree was cut above the level it closes at");
    }
    // (only now: a slot that reports a tree cut short fails every call for this step, not just the first)
    s->cells_checked = s->cells_steps;
    const int levels = choose_cells_levels(s, CELLS_MARGIN * w, CELLS_MARGIN * h);
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
        NB_TRY(pace_wait(s));
        mark(s, 0);
        NB_TRY(bh_cells_begin(s->stream, s->bh, a));
        mark(s, EXCHANGE);
    } else if (phase == 1) {
        s->bh.ring_step = s->cells_steps;
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
        NB_TRY(pace_mark(s));
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

int nb_comm_init(nb_sim *s, int rank, int world, const void *id) { return api_comm_init(s, rank, world, id); }

int nb_step_distributed(nb_sim *s, int nsteps)
{
    NB_TRY(check_distributed(s, nsteps));
    if (s->cells_phase != 0)
        return fail(NB_ERR_STATE, "nb_step_distributed: a split step is in flight");
    for (int k = 0; k < nsteps; ++k) {
        bool gather = false;
        NB_TRY(cells_housekeeping(s, &gather));
        if (gather)  // the cell exchange was suspended: from here on every rank holds all positions again
            NB_TRY(comm_all_gather_f64(s->comm, s->pos, 2 * s->chunk, s->stream));
        if (!s->cells_active) {
            NB_TRY(distributed_position_step(s));
            continue;
        }
        // two reductions per step instead of the gather of all positions: the ranks' extents (4
        // doubles, maximum) and the records of the cells of the keys' last level (sum)
        NB_TRY(nb_step_cells(s, 0));
        NB_TRY(comm_all_reduce_max_f64(s->comm, s->bh.red4, 4, s->stream));
        NB_TRY(nb_step_cells(s, 1));
        NB_TRY(comm_all_reduce_sum_f64(s->comm, s->bh.agg, cells_agg_doubles(s->cells_levels), s->stream));
        NB_TRY(nb_step_cells(s, 2));
    }
    return NB_OK;
}

int nb_compute_accel(nb_sim *s)
{
    if (s && s->cells_active)
        return fail(NB_ERR_STATE, "nb_compute_accel: with the cell exchange this rank does not hold the other ranks' "
                                  "positions");
    return api_compute_accel(s);
}

int nb_sync(nb_sim *s) { return api_sync(s); }

int nb_read_bodies(nb_sim *s, nb_body *out)
{
    if (s && out && s->have_bodies) {
        if (s->cells_phase != 0)
            return fail(NB_ERR_STATE, "nb_read_bodies: a split step is in flight");
        if (s->cells_active && s->comm)  // the other ranks' positions (a collective call)
            NB_TRY(comm_all_gather_f64(s->comm, s->pos, 2 * s->chunk, s->stream));
    }
    return api_read_bodies(s, out);
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
        int rc = potential_energy(s->stream, s->pos2d(), s->mass, s->radius, s->n, s->p.gravity, s->p.eps,
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

int nb_set_stage_timing(nb_sim *s, int on) { return api_set_stage_timing(s, on); }

int nb_stage_ms(nb_sim *s, float *out, int count) { return api_stage_ms(s, out, count); }

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
    case NB_DBG_WAVE_COST:
        src = s->bh.wave_cost;
        need = sizeof(uint32_t) * (((size_t)(s->t1 - s->t0) + 63) / 64);
        break;
    case NB_DBG_VISIT_KINDS:
        src = s->bh.counters + 8;
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
