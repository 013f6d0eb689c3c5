// nbody_sim.hpp -- C++ host object over the C ABI (include/nbody.h), with the surface of the
// reference's simulation classes so that a main() written against them ports line by line:
//
//   BarnesHutCuda(int n); setup(int sim); update(); readDeviceBodies(); getBodies()
//       src/gpu/BH/barnesHutCuda.cuh:49-80
//   NBody(int n, int alg, int sim); update(); bodies
//       src/cpu/nBody.h:25-37
//
// Errors follow the reference's convention for its executables (src/gpu/BH/err.h:21-46):
// print the message and exit(EXIT_FAILURE).
#pragma once

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "nbody.h"

namespace nbody {

inline void check(int rc, const char *what)
{
    if (rc != NB_OK) {
        std::fprintf(stderr, "nbody error at %s: %s\n", what, nb_last_error());
        std::exit(EXIT_FAILURE);
    }
}

class NBodySim {
public:
    NBodySim(int n, int alg, const nb_params *params = nullptr, unsigned seed = 12345)
        : n_(n), seed_(seed), bodies_(n)
    {
        check(nb_create(params, n, alg, &sim_), "nb_create");
    }
    ~NBodySim() { nb_destroy(sim_); }
    NBodySim(const NBodySim &) = delete;
    NBodySim &operator=(const NBodySim &) = delete;

    void setup(int sim) { check(nb_setup(sim_, sim, seed_), "nb_setup"); }
    void setBodies(const nb_body *b) { check(nb_set_bodies(sim_, b, n_), "nb_set_bodies"); }
    void update(int steps = 1)
    {
        if (distributed_)
            check(nb_step_distributed(sim_, steps), "nb_step_distributed");
        else
            check(nb_step(sim_, steps), "nb_step");
    }
    // One process per GPU: joins the RCCL communicator (id from nb_comm_unique_id on rank 0); update()
    // then advances this rank's slice and all-gathers the positions.
    void commInit(int rank, int world, const void *id)
    {
        check(nb_comm_init(sim_, rank, world, id), "nb_comm_init");
        distributed_ = true;
    }
    void readDeviceBodies() { check(nb_read_bodies(sim_, bodies_.data()), "nb_read_bodies"); }
    nb_body *getBodies() { return bodies_.data(); }
    void sync() { check(nb_sync(sim_), "nb_sync"); }
    void diagnostics(double out[8]) { check(nb_diagnostics(sim_, out, 8), "nb_diagnostics"); }
    int size() const { return n_; }
    nb_sim *handle() { return sim_; }

private:
    int n_;
    unsigned seed_;
    nb_sim *sim_ = nullptr;
    bool distributed_ = false;
    std::vector<nb_body> bodies_;
};

// The same object for three-dimensional bodies (the nb3_ entry points).
class NBodySim3 {
public:
    NBodySim3(int n, int alg, const nb_params *params = nullptr, unsigned seed = 12345) : n_(n), seed_(seed), bodies_(n)
    {
        check(nb3_create(params, n, alg, &sim_), "nb3_create");
    }
    ~NBodySim3() { nb3_destroy(sim_); }
    NBodySim3(const NBodySim3 &) = delete;
    NBodySim3 &operator=(const NBodySim3 &) = delete;

    void setup(int sim, double thickness) { check(nb3_setup(sim_, sim, seed_, thickness), "nb3_setup"); }
    void setBodies(const nb_body3 *b) { check(nb3_set_bodies(sim_, b, n_), "nb3_set_bodies"); }
    void update(int steps = 1)
    {
        if (distributed_)
            check(nb3_step_distributed(sim_, steps), "nb3_step_distributed");
        else
            check(nb3_step(sim_, steps), "nb3_step");
    }
    void commInit(int rank, int world, const void *id)
    {
        check(nb3_comm_init(sim_, rank, world, id), "nb3_comm_init");
        distributed_ = true;
    }
    void readDeviceBodies() { check(nb3_read_bodies(sim_, bodies_.data()), "nb3_read_bodies"); }
    nb_body3 *getBodies() { return bodies_.data(); }
    void sync() { check(nb3_sync(sim_), "nb3_sync"); }
    int size() const { return n_; }

private:
    int n_;
    unsigned seed_;
    nb_sim3 *sim_ = nullptr;
    bool distributed_ = false;
    std::vector<nb_body3> bodies_;
};

}  // namespace nbody
