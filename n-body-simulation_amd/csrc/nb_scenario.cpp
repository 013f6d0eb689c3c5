// nb_scenario.cpp -- host-side initial conditions (no GPU).
//
// Restates the four generators of the reference's GPU Barnes-Hut host class
// (src/gpu/BH/barnesHutCuda.cu:73-245; the CPU program's copies are
// src/cpu/nBody.cpp:53-119) so that equal seeds give the same bodies: glibc
// srand(seed)/rand(), the angle drawn before the radius, bodies in index order,
// the same floating-point expressions.  The reference seeds with time(NULL); here
// the seed is an argument.
//
// Built with g++ like the reference's CPU program: the last bit of cos/sin depends on
// whether the compiler pairs them into one sincos call, and the fixtures were made with g++.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "nb_error.h"
#include "nbody.h"

namespace {

// src/gpu/BH/constants.h:23-43
constexpr double kWidth = 10.0e11;
constexpr double kCenterX = 0, kCenterY = 0;
constexpr double kGravity = 6.67E-11;
constexpr double kMinDist = 2.0e10, kMaxDist = 5.0e11;
constexpr double kSunMass = 1.9890e30, kSunDia = 1.3927e6;
constexpr double kEarthMass = 5.974e24, kEarthDia = 12756;

void put(nb_body &b, bool dynamic, double mass, double radius, double px, double py, double vx, double vy)
{
    std::memset(&b, 0, sizeof(b));
    b.is_dynamic = dynamic ? 1 : 0;
    b.mass = mass;
    b.radius = radius;
    b.position = {px, py};
    b.velocity = {vx, vy};
    b.acceleration = {0.0, 0.0};
}

// A planet on a disc round (cx, cy); `orbit` gives it the circular speed round a sun
// at the disc's centre (barnesHutCuda.cu:115-130), otherwise it starts at rest.
void planet(nb_body &b, double cx, double cy, double mind, double maxd, bool orbit)
{
    double angle = 2 * M_PI * (rand() / (double)RAND_MAX);
    double radius = (maxd - mind) * (rand() / (double)RAND_MAX) + mind;
    double x = cx + radius * std::cos(angle);
    double y = cy + radius * std::sin(angle);
    double vx = 0.0, vy = 0.0;
    if (orbit) {
        double distance = std::sqrt(std::pow(x - cx, 2) + std::pow(y - cy, 2));
        double ux = (x - cx) / distance, uy = (y - cy) / distance;
        double esc = std::sqrt((kGravity * kSunMass) / (distance));
        vx = -uy * esc;
        vy = ux * esc;
    }
    put(b, true, kEarthMass, kEarthDia, x, y, vx, vy);
}

}  // namespace

extern "C" void nb_default_params(nb_params *p)
{
    if (!p)
        return;
    p->gravity = 6.67E-11;
    p->eps = 0.5;
    p->dt = 25000.0;
    p->theta = 0.5;
    p->collision_th = 1.0e10;
    p->bbox_margin = 1.0e10;
    p->integrator = NB_INT_EULER;
    p->bucket_depth = 0;
}

extern "C" int nb_init_scenario(int sim, int n, unsigned seed, nb_body *out)
{
    if (!out)
        return nb::fail(NB_ERR_ARG, "nb_init_scenario: out is NULL");
    if (n < 1)
        return nb::fail(NB_ERR_ARG, "ERROR: need to have at least 1 body");
    switch (sim) {
    case NB_SIM_SPIRAL:
    case NB_SIM_RANDOM:
        srand(seed);
        for (int i = 0; i < n - 1; ++i)
            planet(out[i], kCenterX, kCenterY, kMinDist, kMaxDist, sim == NB_SIM_SPIRAL);
        put(out[n - 1], false, kSunMass, kSunDia, kCenterX, kCenterY, 0, 0);
        return NB_OK;
    case NB_SIM_COLLISION: {
        if (n < 2)
            return nb::fail(NB_ERR_ARG, "nb_init_scenario: the collision scenario needs at least 2 bodies");
        srand(seed);
        const double maxd = kMaxDist / 4.0, mind = kMinDist;
        const int galaxy1 = n / 2;
        double cx = -kWidth / 6.0;
        for (int i = 0; i < galaxy1 - 1; ++i)
            planet(out[i], cx, kCenterY, mind, maxd, true);
        put(out[galaxy1 - 1], true, kSunMass, kSunDia, cx, kCenterY, 0, 0);
        cx = kWidth / 6.0;
        for (int i = galaxy1; i < n - 1; ++i)
            planet(out[i], cx, kCenterY, mind, maxd, true);
        put(out[n - 1], true, kSunMass, kSunDia, cx, kCenterY, 0, 0);
        return NB_OK;
    }
    case NB_SIM_SOLAR:
        if (n < 5)
            return nb::fail(NB_ERR_ARG, "nb_init_scenario: the solar system has 5 bodies");
        put(out[0], true, 5.9740e24, 1.3927e6, 1.4960e11, 0, 0, 2.9800e4);
        put(out[1], true, 6.4190e23, 1.3927e6, 2.2790e11, 0, 0, 2.4100e4);
        put(out[2], true, 3.3020e23, 1.3927e6, 5.7900e10, 0, 0, 4.7900e4);
        put(out[3], true, 4.8690e24, 1.3927e6, 1.0820e11, 0, 0, 3.5000e4);
        put(out[4], false, 1.9890e30, 1.3927e6, kCenterX, kCenterY, 0, 0);
        return NB_OK;
    default:
        return nb::fail(NB_ERR_ARG, "ERROR: simulation doesn't exist");
    }
}

// The 2-D scenario for the same seed (identical x, y and velocities), then one more rand() per
// body in index order for its height: z = thickness (2 u - 1); bodies of the suns' mass stay in
// the plane.  The reference's scenarios are flat; this gives its discs a thickness.
extern "C" int nb3_init_scenario(int sim, int n, unsigned seed, double thickness, nb_body3 *out)
{
    if (!out)
        return nb::fail(NB_ERR_ARG, "nb3_init_scenario: out is NULL");
    if (n < 1)
        return nb::fail(NB_ERR_ARG, "ERROR: need to have at least 1 body");
    const int m = sim == NB_SIM_SOLAR ? 5 : n;
    std::vector<nb_body> flat((size_t)std::max(m, n));
    const int rc = nb_init_scenario(sim, n, seed, flat.data());
    if (rc != NB_OK)
        return rc;
    if (sim == NB_SIM_SOLAR)
        srand(seed);  // the solar system draws nothing itself
    for (int i = 0; i < m; ++i) {
        const double u = rand() / (double)RAND_MAX;
        nb_body3 &b = out[i];
        std::memset(&b, 0, sizeof(b));
        b.is_dynamic = flat[i].is_dynamic;
        b.mass = flat[i].mass;
        b.radius = flat[i].radius;
        b.position = {flat[i].position.x, flat[i].position.y, flat[i].mass == kSunMass ? 0.0 : thickness * (2.0 * u - 1.0)};
        b.velocity = {flat[i].velocity.x, flat[i].velocity.y, 0.0};
    }
    return NB_OK;
}
