/*
 * ref_driver.cpp -- TEST INFRASTRUCTURE.  A headless C entry to the REFERENCE's own
 * CPU classes (NBody, DirectSum, BarnesHut of src/cpu), compiled by oracle/Makefile
 * from the sources where they lie in the reference checkout into
 * oracle/_ref/libnbody_ref.so.  Nothing of the reference is copied here: this file
 * only includes its headers and calls its public constructors and update().
 *
 * It exists to (1) produce the golden vectors in tests/golden/, (2) check the plain-C
 * restatement in nbody_oracle.c, and (3) serve as the "reference" CPU baseline that
 * bench.py times beside the GPU path.
 */
#include <ctime>
#include <memory>
#include <vector>

#include "barnesHut.h"
#include "directSum.h"
#include "nBody.h"

#include "nbody_oracle.h"

static unsigned g_seed = 12345;

/* NBody's generators call srand(time(NULL)) (src/cpu/nBody.cpp:55,78).  The library is
 * linked with -Bsymbolic-functions, so their call binds to this definition and the seed
 * becomes a parameter. */
extern "C" time_t time(time_t *t) noexcept
{
    if (t)
        *t = (time_t)g_seed;
    return (time_t)g_seed;
}

static void to_c(const Body &b, orc_body *o)
{
    o->is_dynamic = b.isDynamic ? 1 : 0;
    for (int k = 0; k < 7; ++k)
        o->pad_[k] = 0;
    o->mass = b.mass;
    o->radius = b.radius;
    o->px = b.position.x;
    o->py = b.position.y;
    o->vx = b.velocity.x;
    o->vy = b.velocity.y;
    o->ax = b.acceleration.x;
    o->ay = b.acceleration.y;
}

extern "C" {

/* sim: the CPU program's numbering 0 spiral, 1 random, 2 solar (src/cpu/nBody.cpp:29-41).
 * Returns the number of bodies written (5 for solar). */
int ref_init_scenario(int sim, int n, unsigned seed, orc_body *out)
{
    g_seed = seed;
    NBody nb(n, 0, sim);
    int m = (int)nb.bodies.size();
    for (int i = 0; i < m; ++i)
        to_c(*nb.bodies[i], &out[i]);
    return m;
}

/* alg 0 = DirectSum, 1 = BarnesHut; nsteps of update() on b in place. */
int ref_step(int alg, int n, orc_body *b, int nsteps)
{
    std::vector<std::shared_ptr<Body>> bodies;
    bodies.reserve(n);
    for (int i = 0; i < n; ++i)
        bodies.push_back(std::make_shared<Body>(b[i].mass, b[i].radius, Vector(b[i].px, b[i].py),
                                                Vector(b[i].vx, b[i].vy), Vector(b[i].ax, b[i].ay),
                                                b[i].is_dynamic != 0));
    std::unique_ptr<Algorithm> a;
    if (alg == 0)
        a = std::make_unique<DirectSum>(bodies);
    else
        a = std::make_unique<BarnesHut>(bodies);
    for (int s = 0; s < nsteps; ++s)
        a->update();
    for (int i = 0; i < n; ++i)
        to_c(*bodies[i], &b[i]);
    return 0;
}

} /* extern "C" */
