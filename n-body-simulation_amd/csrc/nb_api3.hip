// nb_api3.hip -- the nb3_ entry points of include/nbody.h: the C ABI of nb_api.hip for
// three-dimensional bodies.  nb_sim3 owns the device state (FP64 positions, velocities and
// accelerations as packed triples, in an order sorted by their position along a Hilbert curve through the initial
// positions) and drives nb_direct3.hip or nb_bh3.hip; bodies are copied back only on request.
// The entry points themselves are the ones of the 2-D handle (nb_handle.h); this file holds the
// 3-D pipelines' launches, diagnostics and introspection.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "nb_bh.h"
#include "nb_dim3.h"
#include "nb_handle.h"

using namespace nb;

struct nb_sim3 : SimCore<3> {
    static constexpr const char *PFX = "nb3_";
    using Body = nb_body3;
    static double coord(const nb_body3 &b, int d) { return d == 0 ? b.position.x : (d == 1 ? b.position.y : b.position.z); }
    static double velocity(const nb_body3 &b, int d) { return d == 0 ? b.velocity.x : (d == 1 ? b.velocity.y : b.velocity.z); }
    static double acceleration(const nb_body3 &b, int d)
    {
        return d == 0 ? b.acceleration.x : (d == 1 ? b.acceleration.y : b.acceleration.z);
    }
    static void store(nb_body3 &b, const double *p, const double *v, const double *a)
    {
        b.position = {p[0], p[1], p[2]};
        b.velocity = {v[0], v[1], v[2]};
        b.acceleration = {a[0], a[1], a[2]};
    }
    static int check_params(const nb_params &p)
    {
        if (p.bucket_depth != 0)
            return fail(NB_ERR_ARG, "nb3_create: bucket_depth (the reference GPU program's 2-D tree) must be 0 in 3-D");
        return NB_OK;
    }

    Direct3State ds;
    Bh3Buffers bh;

    bool static_fixed() const { return false; }
    bool keeps_order_fresh() const { return false; }

    int alloc_pipelines()
    {
        if (alg == NB_ALG_DIRECT) {
            NB_TRY(dev_alloc(ds.posm[0], 2 * direct3_record_capacity(n)));
            NB_TRY(dev_alloc(ds.posm[1], 2 * direct3_record_capacity(n)));
            NB_TRY(dev_alloc(ds.wcut, direct3_record_capacity(n)));
            NB_TRY(dev_alloc(ds.rad, cap));
            ds.jsplit = direct3_choose_jsplit(n, n);
            part_cap = 3 * (size_t)ds.jsplit * n;
            NB_TRY(dev_alloc(ds.part, part_cap));
            uint8_t *meta = nullptr;
            NB_TRY(dev_alloc(meta, 2 * direct3_meta_bytes(n)));  // one set of boxes per record buffer
            ds.meta = meta;
            return NB_OK;
        }
        return bh3_alloc(bh, n);
    }

    void free_pipelines()
    {
        (void)hipFree(ds.posm[0]);
        (void)hipFree(ds.posm[1]);
        (void)hipFree(ds.wcut);
        (void)hipFree(ds.rad);
        (void)hipFree(ds.part);
        (void)hipFree(ds.meta);
        bh3_free(bh);
    }

    int direct_step(int do_integrate)
    {
        ds.pos_cur = pos;
        ds.pos_next = pos2;
        ds.dyn = dyn;
        ds.radius = radius;
        ds.vel = vel;
        ds.acc = acc;
        timer.mark(stream, 5);
        NB_TRY(direct3_step(stream, ds, n, t0, t1 - t0, fc, make_integrate_args(this, do_integrate), p.collision_th));
        timer.mark(stream, 6);
        return NB_OK;
    }

    int bh_step(int do_integrate)
    {
        Bh3StepArgs a;
        a.pos = pos;
        a.mass = mass;
        a.radius = radius;
        a.dyn = dyn;
        a.vel = vel;
        a.acc = acc;
        a.n = n;
        a.t0 = t0;
        a.t1 = t1;
        a.units = units;
        a.fc = fc;
        a.ia = make_integrate_args(this, do_integrate);
        a.theta = p.theta;
        a.bbox_margin = p.bbox_margin;
        a.collision_th = p.collision_th;
        a.collect_stats = collect_stats;
        timer.mark(stream, 0);
        NB_TRY(bh3_bbox(stream, bh, a));
        timer.mark(stream, 1);
        NB_TRY(bh3_keys(stream, bh, a));
        timer.mark(stream, 2);
        NB_TRY(bh3_sort(stream, bh, a));
        timer.mark(stream, 3);
        NB_TRY(bh3_build(stream, bh, a));
        timer.mark(stream, 4);
        NB_TRY(bh3_com(stream, bh, a));
        timer.mark(stream, 5);
        NB_TRY(bh3_force(stream, bh, a));
        timer.mark(stream, 6);
        return NB_OK;
    }

    int direct_pack_all()
    {
        ds.cur = 0;
        ds.meta_valid = false;
        NB_TRY(pack_posm3(stream, pos, mass, radius, n, units, p.collision_th, ds.posm[0], ds.wcut, ds.rad));
        // masses and padding never change: the other buffer starts as a copy
        NB_HIP(hipMemcpyAsync(ds.posm[1], ds.posm[0], sizeof(float4) * 2 * direct3_record_capacity(n), hipMemcpyDeviceToDevice,
                              stream));
        return NB_OK;
    }

    int direct_pack_others()
    {
        ds.meta_valid = false;
        NB_TRY(pack_posm3_range(stream, pos2, mass, n, 0, t0, units, ds.posm[ds.cur ^ 1]));
        return pack_posm3_range(stream, pos2, mass, n, t1, n, units, ds.posm[ds.cur ^ 1]);
    }

    void direct_flip() { ds.cur ^= 1; }

    int direct_set_split(int nt)
    {
        const int js = direct3_choose_jsplit(n, nt);
        const size_t need = 3 * (size_t)js * (size_t)nt;
        if (need > part_cap) {
            (void)hipFree(ds.part);
            ds.part = nullptr;
            part_cap = 0;
            NB_TRY(dev_alloc(ds.part, need));
            part_cap = need;
        }
        ds.jsplit = js;
        return NB_OK;
    }

    int upload_extra(const double *, const double *) { return NB_OK; }
    int gather_extra() { return NB_OK; }
    int after_load() { return NB_OK; }
    void count_step() {}
};

extern "C" {

int nb3_create(const nb_params *params, int n, int alg, nb_sim3 **out) { return api_create(params, n, alg, out); }

int nb3_destroy(nb_sim3 *s) { return api_destroy(s); }

int nb3_set_bodies(nb_sim3 *s, const nb_body3 *b, int n) { return api_set_bodies(s, b, n); }

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

int nb3_step(nb_sim3 *s, int nsteps) { return api_step(s, nsteps); }

int nb3_refresh_order(nb_sim3 *s) { return api_refresh_order(s); }

int nb3_velocity_buffer(nb_sim3 *s, void **device_ptr, size_t *chunk_records, size_t *record_bytes)
{
    return api_buffer(s, "velocity_buffer", s ? s->vel : nullptr, device_ptr, chunk_records, record_bytes);
}

int nb3_exchange_buffer(nb_sim3 *s, void **device_ptr, size_t *chunk_records, size_t *record_bytes)
{
    // direct sum writes the new positions to the other buffer; mid-step that is the one to gather
    double *buf = !s ? nullptr : ((s->alg == NB_ALG_DIRECT && s->mid_step) ? s->pos2 : s->pos);
    return api_buffer(s, "exchange_buffer", buf, device_ptr, chunk_records, record_bytes);
}

int nb3_comm_init(nb_sim3 *s, int rank, int world, const void *id) { return api_comm_init(s, rank, world, id); }

int nb3_step_distributed(nb_sim3 *s, int nsteps)
{
    NB_TRY(check_distributed(s, nsteps));
    for (int k = 0; k < nsteps; ++k)
        NB_TRY(distributed_position_step(s));
    return NB_OK;
}

int nb3_compute_accel(nb_sim3 *s) { return api_compute_accel(s); }

int nb3_sync(nb_sim3 *s) { return api_sync(s); }

int nb3_read_bodies(nb_sim3 *s, nb_body3 *out) { return api_read_bodies(s, out); }

int nb3_set_stream(nb_sim3 *s, void *hip_stream) { return api_set_stream(s, hip_stream); }

int nb3_set_partition(nb_sim3 *s, int rank, int world) { return api_set_partition(s, rank, world); }

int nb3_step_local(nb_sim3 *s) { return api_step_local(s); }

int nb3_step_finish(nb_sim3 *s) { return api_step_finish(s); }

int nb3_set_stage_timing(nb_sim3 *s, int on) { return api_set_stage_timing(s, on); }

int nb3_stage_ms(nb_sim3 *s, float *out, int count) { return api_stage_ms(s, out, count); }

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
    case NB_DBG_WAVE_COST:
        return give(s->bh.wave_cost, sizeof(uint32_t) * (((size_t)(s->t1 - s->t0) + 63) / 64), true);
    case NB_DBG_VISIT_KINDS:
        return give(s->bh.counters + 8, sizeof(unsigned long long) * 8, true);
    default:
        return fail(NB_ERR_ARG, "nb3_debug_read: array not available in 3-D");
    }
}

}  // extern "C"
