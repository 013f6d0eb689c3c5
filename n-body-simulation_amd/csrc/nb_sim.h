// nb_sim.h -- the 2-D simulation handle behind the nb_ entry points (internal).
#pragma once

#include <vector>

#include "nb_bh.h"
#include "nb_comm.h"
#include "nb_common.h"
#include "nb_refresh.h"
#include "nb_timing.h"

struct nb_sim {
    nb_params p;
    int n = 0;
    int alg = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool have_bodies = false;
    bool first_kick = true;  // leapfrog: the next step is the first since the bodies were set
    bool stepped = false;    // a step was taken since the bodies were set
    long steps_in_order = 0; // steps since the exchange order was last derived from the positions

    nb::Units units{};
    nb::ForceConsts fc{};

    // state in exchange order; pos has room for world*chunk records
    size_t cap = 0;
    double2 *pos = nullptr, *vel = nullptr, *acc = nullptr;
    double2 *pos2 = nullptr;  // direct sum: the other FP64 position buffer (pos is the current one)
    double *mass = nullptr, *radius = nullptr;
    double2 *mass_radius = nullptr;  // Barnes-Hut: {mass, radius} per body (one gather instead of two)
    uint8_t *dyn = nullptr;
    float4 *posm[2] = {nullptr, nullptr};
    float *rad_s = nullptr;
    int cur = 0;
    std::vector<uint32_t> order;  // caller index of the body in each exchange-order slot
    uint32_t *order_d = nullptr;  // the same on the device
    bool order_stale = false;     // the device renewed the order; `order` is fetched when next needed
    nb::RefreshBuffers rb;        // buffers of the renewal (made at nb_create for the direct sum, else on first use)

    // partition
    int rank = 0, world = 1;
    size_t chunk = 0;
    int t0 = 0, t1 = 0;
    bool mid_step = false;
    nb::Comm *comm = nullptr;  // RCCL communicator of nb_comm_init (null: none)
    int exchange = 0;          // nb_exchange: what the ranks exchange each step
    int cells_levels = -1;     // cell exchange: digits of the keys (the same on every rank)
    bool cells_active = false; // cell exchange: in use for the coming steps (else suspended: positions are exchanged)
    long cells_steps = 0;      // cell exchange: steps since it was switched on
    long cells_checked = -1;   // cell exchange: the step count at which the level was last looked at
    int cells_phase = 0;       // cell exchange by hand: the next part of the step (0, 1, 2)

    // direct sum
    double2 *part = nullptr;
    size_t part_cap = 0;  // records allocated in part
    int jsplit = 1;
    uint8_t *ds_meta = nullptr;  // per-run bounding boxes (direct_meta_bytes)

    // Barnes-Hut
    nb::BhBuffers bh;
    int collect_stats = 0;

    nb::StageTimer timer;
};

namespace nb {

// Arguments of a force evaluation (do_integrate = 0) or a step (1) of the handle's current state.
IntegrateArgs integrate_args(const nb_sim *s, int do_integrate);
BhStepArgs bh_args(nb_sim *s, int do_integrate);

// Looks up, once per device and with that device current, where the library's static scratch
// array lives on it (nb_static.hip); nb_create calls it so that nb_compute_accel_static never has to.
int static_scratch_init(int device);

}  // namespace nb
