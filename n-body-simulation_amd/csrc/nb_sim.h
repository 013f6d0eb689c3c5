// nb_sim.h -- the 2-D simulation handle behind the nb_ entry points (internal): the state every
// handle has (nb_handle.h) plus the 2-D pipelines' buffers, the cell exchange between GPUs and the
// hooks the shared entry points call (defined in nb_api.hip).
#pragma once

#include "nb_bh.h"
#include "nb_handle.h"

struct nb_sim : nb::SimCore<2> {
    static constexpr const char *PFX = "nb_";
    using Body = nb_body;
    static double coord(const nb_body &b, int d) { return d == 0 ? b.position.x : b.position.y; }
    static double velocity(const nb_body &b, int d) { return d == 0 ? b.velocity.x : b.velocity.y; }
    static double acceleration(const nb_body &b, int d) { return d == 0 ? b.acceleration.x : b.acceleration.y; }
    static void store(nb_body &b, const double *p, const double *v, const double *a)
    {
        b.position = {p[0], p[1]};
        b.velocity = {v[0], v[1]};
        b.acceleration = {a[0], a[1]};
    }
    static int check_params(const nb_params &p);

    // the FP64 arrays as the 2-D kernels take them
    double2 *pos2d() const { return (double2 *)pos; }
    double2 *pos2d_next() const { return (double2 *)pos2; }
    double2 *vel2d() const { return (double2 *)vel; }
    double2 *acc2d() const { return (double2 *)acc; }

    double2 *mass_radius = nullptr;  // Barnes-Hut: {mass, radius} per body (one gather instead of two)
    // Barnes-Hut on one GPU: the second set of per-body arrays (with pos2).  Every step writes the bodies
    // into it in the step's sorted order and the two sets change places (bh_step): the slots follow
    // the bodies, the order never goes stale.
    double *vel_b = nullptr, *acc_b = nullptr, *mass_b = nullptr, *radius_b = nullptr;
    double2 *mass_radius_b = nullptr;
    uint8_t *dyn_b = nullptr;
    uint32_t *order_b = nullptr;
    nb::BodyArrays2 body_arrays(bool other) const;
    long bh_steps = 0;  // Barnes-Hut steps taken since the bodies were set
    // Steps between two that re-order the bodies.  The re-ordering rides in the traversal's shadow while
    // the per-body arrays stay near the last-level cache (N = 5 M: +0.8 % on the traversal, and the
    // order goes stale within five steps once the bodies mix: every step); at N = 50 M its 65 bytes
    // per body show (traversal 4.24 -> 4.54 ms): every eighth step.  By N alone, so that a run is
    // the same whenever it is repeated.
    int resort_every() const { return n > (1 << 24) ? 8 : 1; }
    // the steps themselves keep the slots in the order of the tree: no renewal needed
    bool keeps_order_fresh() const { return alg == NB_ALG_BARNES_HUT && world == 1 && vel_b != nullptr; }

    // direct sum
    float4 *posm[2] = {nullptr, nullptr};
    float *rad_s = nullptr;
    int cur = 0;
    double2 *part = nullptr;  // part_cap records
    int jsplit = 1;
    uint8_t *ds_meta = nullptr;  // per-run bounding boxes (2 x direct_meta_bytes: one set per record buffer)
    bool ds_meta_valid = false;  // the set of the current record buffer is up to date

    // Barnes-Hut
    nb::BhBuffers bh;

    // what the ranks exchange
    int exchange = 0;          // nb_exchange: what the ranks exchange each step
    int cells_levels = -1;     // cell exchange: digits of the keys (the same on every rank)
    bool cells_active = false; // cell exchange: in use for the coming steps (else suspended: positions are exchanged)
    long cells_steps = 0;      // cell exchange: steps since it was switched on
    long cells_checked = -1;   // cell exchange: the step count at which the level was last looked at
    double cells_box0[2] = {0.0, 0.0};  // cell exchange: width and height of the box the bodies were set with
    int cells_phase = 0;       // cell exchange by hand: the next part of the step (0, 1, 2)

    bool static_fixed() const { return alg == NB_ALG_BARNES_HUT && p.bucket_depth > 0; }
    int alloc_pipelines();
    void free_pipelines();
    int direct_step(int do_integrate);
    int bh_step(int do_integrate);
    int direct_pack_all();
    int direct_pack_others();
    void direct_flip() { cur ^= 1; }
    int direct_set_split(int nt);
    int upload_extra(const double *mass_h, const double *radius_h);
    int gather_extra();
    int after_load();
    void count_step()
    {
        if (exchange != NB_EXCHANGE_POSITIONS)
            cells_steps += 1;  // (also while the cell exchange is suspended: its level follows the box all the time)
    }
};

namespace nb {

// Arguments of the Barnes-Hut pipeline for a force evaluation (do_integrate = 0) or a step (1) of
// the handle's current state.
BhStepArgs bh_args(nb_sim *s, int do_integrate);

// Looks up, once per device and with that device current, where the library's static scratch
// array lives on it (nb_static.hip); nb_create calls it so that nb_compute_accel_static never has to.
int static_scratch_init(int device);

}  // namespace nb
