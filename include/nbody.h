/*
 * nbody.h -- C ABI of the MI355X (gfx950) N-body engine.
 *
 * One shared library, libnbody_hip.so, exports everything declared here.  The
 * signatures use plain pointers and sizes only, so any host language can bind
 * them (ctypes, cgo, JNI, ...).  Every entry point returns 0 on success and a
 * non-zero nb_status otherwise; nb_last_error() returns the message of the
 * last failure on the calling thread.
 *
 * What each entry point replaces in kr1s0404/N-body-simulation (paths under
 * the reference checkout):
 *
 *   nb_body / nb_vec2      struct Body / struct Vector
 *                          src/gpu/BH/barnesHutCuda.cuh:20-35,
 *                          src/gpu/DS/directSum_kernel.cu:33-48,
 *                          class Body src/cpu/body.h:22-34
 *   nb_params              the compile-time constants
 *                          src/gpu/BH/constants.h:23-44, src/cpu/constants.h:23-34,
 *                          src/cpu/directSum.h:27-28, src/cpu/barnesHut.h:28-30
 *   nb_init_scenario       BarnesHutCuda::initSpiralBodies / initRandomBodies /
 *                          initCollideGalaxy / initSolarSystem
 *                          src/gpu/BH/barnesHutCuda.cu:73-245
 *   nb_create/nb_destroy   BarnesHutCuda::BarnesHutCuda(int) / ~BarnesHutCuda
 *                          src/gpu/BH/barnesHutCuda.cu:23-45;
 *                          NBody::NBody(int,int,int) src/cpu/nBody.cpp:26-51
 *   nb_setup               BarnesHutCuda::setup(int) src/gpu/BH/barnesHutCuda.cu:258-279
 *   nb_set_bodies          the cudaMemcpy H2D inside setup(), with caller-made state
 *   nb_step                BarnesHutCuda::update() src/gpu/BH/barnesHutCuda.cu:280-287,
 *                          the DirectSumTiledKernel launch
 *                          src/gpu/DS/directSum_kernel.cu:294-299,
 *                          Algorithm::update() src/cpu/algorithm.h:31
 *   nb_read_bodies         readDeviceBodies() + getBodies()
 *                          src/gpu/BH/barnesHutCuda.cu:247-256
 *   nb_last_error          CHECK_CUDA_ERROR / CHECK_LAST_CUDA_ERROR
 *                          src/gpu/BH/err.h:21-46 (there: print + exit; here: status + text)
 *
 * The reference has no multi-GPU code; nb_set_partition / nb_exchange_* are
 * additions that let one process per GPU own a slice of the target bodies and
 * all-gather positions between steps with whatever collective the host uses.
 */
#ifndef NBODY_H
#define NBODY_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct nb_vec2 {
    double x;
    double y;
} nb_vec2;

/* Same 72-byte layout as the reference's Body (isDynamic@0, mass@8, radius@16,
 * position@24, velocity@40, acceleration@56). */
typedef struct nb_body {
    uint8_t is_dynamic;
    uint8_t pad_[7];
    double mass;
    double radius;
    nb_vec2 position;
    nb_vec2 velocity;
    nb_vec2 acceleration;
} nb_body;

enum nb_alg { NB_ALG_DIRECT = 0, NB_ALG_BARNES_HUT = 1 };

/* Scenario numbering of the GPU Barnes-Hut program (src/gpu/BH/barnesHutCuda.cu:258-275),
 * a superset of the CPU and direct-sum programs' 0/1/2 = spiral/random/solar. */
enum nb_scenario { NB_SIM_SPIRAL = 0, NB_SIM_RANDOM = 1, NB_SIM_COLLISION = 2, NB_SIM_SOLAR = 3 };

enum nb_status {
    NB_OK = 0,
    NB_ERR_ARG = 1,      /* bad argument */
    NB_ERR_HIP = 2,      /* a HIP runtime call failed */
    NB_ERR_STATE = 3,    /* call order (e.g. step before bodies were set) */
    NB_ERR_NOMEM = 4,
    NB_ERR_OVERFLOW = 5  /* an internal capacity was exceeded; state is unchanged */
};

typedef struct nb_params {
    double gravity;       /* GRAVITY      6.67e-11 */
    double eps;           /* E / epsilon  0.5: r = sqrt(dx^2+dy^2+eps^2), f = G m / (r^3 + eps^2) */
    double dt;            /* DT           25000 */
    double theta;         /* THETA        0.5 (Barnes-Hut opening ratio, cell x-width / distance) */
    double collision_th;  /* COLLISION_TH 1.0e10: pairs closer than r_i + r_j + this are skipped */
    double bbox_margin;   /* the 1.0e10 pad added round the bounding box (barnesHut.cpp:28-31) */
    int integrator;       /* nb_integrator; 0 = the reference's scheme */
    int bucket_depth;     /* Barnes-Hut tree: 0 = one body per leaf at any depth, the reference's CPU tree
                           * (src/cpu/quadtree.cpp:73-144).  k = 1..31 = the tree of the reference's GPU program
                           * (src/gpu/BH/barnesHut_kernel.cu:290-423; k = 9 there): cells are not split below
                           * level k, every leaf acts as one point mass at its bodies' centre of mass under the
                           * cell rule 2 r_i + collision_th, the root box is rounded as that program rounds it,
                           * and static bodies are not advanced. */
} nb_params;

/* NB_INT_EULER is the reference's update (v += a dt, then x += v dt: semi-implicit Euler,
 * directSum.cpp:49-67).  NB_INT_LEAPFROG is kick-drift-kick with the same single force
 * evaluation per step: the first step kicks by dt/2, later steps by dt, and nb_read_bodies
 * returns velocities brought to the positions' time (v + a(x) dt/2). */
enum nb_integrator { NB_INT_EULER = 0, NB_INT_LEAPFROG = 1 };

typedef struct nb_sim nb_sim; /* opaque */

/* Fills *p with the reference's constants. */
void nb_default_params(nb_params *p);

/* Host only, no GPU: writes the initial state of scenario `sim` for n bodies
 * (NB_SIM_SOLAR always writes 5 and needs n >= 5).  Uses glibc srand(seed)/rand()
 * in the reference's draw order, so equal seeds give the reference's stream. */
int nb_init_scenario(int sim, int n, unsigned seed, nb_body *out);

/* Creates a simulation of n bodies on the current HIP device.  params may be
 * NULL (defaults).  alg is an nb_alg. */
int nb_create(const nb_params *params, int n, int alg, nb_sim **out);
int nb_destroy(nb_sim *sim);

/* nb_init_scenario + nb_set_bodies. */
int nb_setup(nb_sim *sim, int scenario, unsigned seed);

/* Uploads caller-made state (n must equal the n given to nb_create). */
int nb_set_bodies(nb_sim *sim, const nb_body *bodies, int n);

/* Advances nsteps steps: accelerations from the current positions, then
 * v += a dt, x += v dt.  Asynchronous on the simulation's stream. */
int nb_step(nb_sim *sim, int nsteps);

/* Accelerations of the current state without integrating (also stored so that
 * nb_read_bodies returns them). */
int nb_compute_accel(nb_sim *sim);

/* Copies the current state (acceleration = the one used by the last step) to
 * out[0..n), in the order the bodies were given.  Synchronises the stream. */
int nb_read_bodies(nb_sim *sim, nb_body *out);

/* Blocks until the stream has drained. */
int nb_sync(nb_sim *sim);

/* Use the caller's hipStream_t (passed as void*) for all work of this handle. */
int nb_set_stream(nb_sim *sim, void *hip_stream);

/* ---- multi-GPU: one process per GPU, every process holds all n bodies ---- */

/* Bodies are held in "exchange order": sorted when they are set by their position along a
 * Hilbert curve (NB_DBG_ORDER gives the caller index of each slot), so that an index range is
 * a compact patch of space; the order is renewed as described at nb_refresh_order, and a partitioned
 * simulation changes it only there.  This process integrates only the bodies in slots
 * [rank*chunk, min(n,(rank+1)*chunk)), chunk = ceil(n / world).
 * Must be called after the bodies were set and before the first step. */
int nb_set_partition(nb_sim *sim, int rank, int world);

/* Device pointer to the exchange buffer: world * chunk records of 2 doubles
 * (x, y), record k*chunk+i being body i of rank k's slice.  After
 * nb_step_local() the own slice holds the new positions; the host all-gathers
 * the buffer in place (every rank contributes chunk records at offset
 * rank*chunk) and then calls nb_step_finish().  Ask for the address when it is needed,
 * i.e. between nb_step_local and nb_step_finish: the direct sum alternates between two
 * position buffers, and so does nb_step of an unpartitioned 2-D Barnes-Hut handle (see
 * nb_refresh_order). */
int nb_exchange_buffer(nb_sim *sim, void **device_ptr, size_t *chunk_records, size_t *record_bytes);
int nb_step_local(nb_sim *sim);
int nb_step_finish(nb_sim *sim);

/* The same exchange done by the library itself with RCCL (no counterpart in the reference, which
 * is single-GPU): one in-place ncclAllGather of the position buffer per step, enqueued on the
 * simulation's stream between that step's kernels, so the host never waits inside a step.
 * librccl.so.1 is loaded on first use (by soname: a process that already holds an RCCL, e.g.
 * PyTorch's, shares it); single-GPU users never load it.
 *
 *   rank 0:      nb_comm_unique_id(id, NB_COMM_ID_BYTES); hand the bytes to every rank (any channel)
 *   every rank:  hipSetDevice(its GPU); nb_create; nb_set_bodies (same bodies everywhere);
 *                nb_comm_init(sim, rank, world, id)        -- collective; also sets the partition
 *                nb_step_distributed(sim, nsteps)          -- collective; asynchronous on the stream
 *
 * With world = 1 the steps are those of nb_step. */
#define NB_COMM_ID_BYTES 128
int nb_comm_unique_id(void *id, size_t bytes);
int nb_comm_init(nb_sim *sim, int rank, int world, const void *id);
int nb_step_distributed(nb_sim *sim, int nsteps);

/* What the ranks exchange each step (Barnes-Hut; no counterpart in the reference).
 * NB_EXCHANGE_POSITIONS (default): every rank keeps all positions up to date (one all-gather) and
 * rebuilds the whole tree itself -- sort, build and centres of mass cost every rank the full N.
 * NB_EXCHANGE_CELLS: a rank sorts only the bodies it integrates and describes every cell of the
 * level at which the tree closes (no body ever opens a cell of that level: with the reference's
 * constants level 8-9, a few hundred thousand cells) by one record -- sum m x, sum m y, sum m, count,
 * and, where it holds exactly one body there, that body's position and radius; the records are
 * added up over the ranks (one all-reduce, 64 bytes a cell) and every rank builds the same tree from
 * the totals.  Cells with one body in all keep that body's exact position and radius, so every
 * accept/drop decision is the reference's; centres of mass of shared cells are summed in another
 * order than on one GPU (differences of the order of 1e-16).  Per step: one max-reduction of 4
 * doubles (the bounding box) and the sum-reduction of the records; no rank needs another rank's
 * bodies, so nb_read_bodies all-gathers the positions first when the library holds the communicator
 * (a collective call then) and returns only this rank's velocities up to date.  Needs a tree that
 * closes at level <= 10, Euler integration and bucket_depth 0; switch it on after nb_set_partition /
 * nb_comm_init and before the first step (nb_set_bodies goes back to NB_EXCHANGE_POSITIONS).
 * The level follows the bounding box without the host waiting for the device: step t uses the box
 * of step t - 4 (the same numbers on every rank, left by the device in pinned memory), enlarged by a
 * tenth; a box that grows faster than that makes the next call fail with NB_ERR_OVERFLOW.  A box
 * that has grown so far that the tree closes below level 10 SUSPENDS the cell exchange -- the ranks all-gather their positions once and
 * step by the position exchange -- until the box allows it again.
 * NB_EXCHANGE_AUTO: the same, and the cell exchange is also suspended while its records are more
 * data than the positions (twice 64 bytes x 4^level against 16 bytes x n).
 *
 * With nb_comm_init, nb_step_distributed does all of this.  With another collective, per step:
 *   nb_cells_level(sim, &level, &gather);
 *   if (gather) all-gather nb_exchange_buffer over the ranks   (the cell exchange was suspended just now)
 *   if (level < 0) { nb_step_local; all-gather nb_exchange_buffer; nb_step_finish; }   (suspended)
 *   else {
 *     nb_step_cells(sim, 0); reduce nb_cells_buffer(sim, 0, ..) over the ranks with MAX, in place;
 *     nb_step_cells(sim, 1); reduce nb_cells_buffer(sim, 1, ..) over the ranks with SUM, in place;
 *     nb_step_cells(sim, 2); }
 * (buffers of doubles; ask for buffer 1 after part 1 of each step: its length follows the level).
 * nb_step_cells(sim, 0) fails with NB_ERR_STATE while the cell exchange is suspended. */
enum nb_exchange { NB_EXCHANGE_POSITIONS = 0, NB_EXCHANGE_CELLS = 1, NB_EXCHANGE_AUTO = 2 };
int nb_set_exchange(nb_sim *sim, int mode);
int nb_cells_level(nb_sim *sim, int *level, int *gather_positions_first);
int nb_step_cells(nb_sim *sim, int part);
int nb_cells_buffer(nb_sim *sim, int which, void **device_ptr, size_t *count_doubles);

/* ---- introspection used by the tests and the benchmark ---- */

enum nb_debug_array {
    NB_DBG_BBOX = 0,        /* 4 doubles: topLeft.x, topLeft.y, botRight.x, botRight.y */
    NB_DBG_KEYS = 1,        /* n uint64: key per body (its quadrant path in the step's box, each digit the quadrant's place along a Hilbert curve), unsorted, exchange order */
    NB_DBG_SORTED_KEYS = 2, /* n uint64 */
    NB_DBG_SORTED_IDX = 3,  /* n uint32: exchange-order index of the k-th sorted body */
    NB_DBG_NODE_RANGE = 4,  /* nnodes x 2 uint32: first sorted body, body count */
    NB_DBG_NODE_CHILD = 5,  /* nnodes x 2 uint32: first child node, child count */
    NB_DBG_NODE_COM = 6,    /* nnodes x 4 doubles: com.x, com.y, mass, cell x-width; a leaf holds minus its
                               body's radius there (sign bit set, -0.0 for radius 0) */
    NB_DBG_LEVEL_START = 7, /* 65 uint32: first node of each tree level, then total */
    NB_DBG_ORDER = 8,       /* n uint32: caller index of the body at each exchange-order slot */
    NB_DBG_COUNTERS = 9,    /* 8 uint64: [0] body-node evaluations, [1] accepted interactions,
                               [2] nodes, [3] tree levels, [4] node visits re-decided in FP64,
                               [5] node visits summed over the 64-body groups, [6] node batches fetched,
                               [7] deepest traversal stack,
                               of the last Barnes-Hut step run with stage timing on */
    NB_DBG_VISIT_KINDS = 11, /* 8 uint64: the node visits of NB_DBG_COUNTERS [5] by what the 64-body group made of the
                               node before any lane looked at it: [0] opened by every lane (children pushed, no
                               visit), [1] inside every lane's cutoff distance (dropped), [2] accepted by every
                               lane, visited for all of them (force only), [3] the same under a lane mask,
                               [4] per-lane opening test only (far from every cutoff), [5] all per-lane tests,
                               [6] bodies, [7] per-lane cutoff test only (acceptance settled for the group) */
    NB_DBG_WAVE_COST = 12,  /* ceil(targets / 64) uint32: what the wave of each group of 64 consecutive targets of the last
                             * traversal took, in ticks of 10 ns (a partitioned simulation: of the first
                             * ceil(targets / 64) groups of its list of targets, which pads the groups) */
    NB_DBG_KEY_LEVELS = 10  /* 2 uint32: [0] the first tree level whose cells no body can open (wider cutoff
                               than opening distance; 32 = none), [1] the number of levels the keys hold:
                               bodies that share a cell of that level share a leaf */
};
/* Copies the array into out (bytes = capacity of out); *written gets the size. */
int nb_debug_read(nb_sim *sim, int which, void *out, size_t bytes, size_t *written);

/* on: bit 0 = time every stage with HIP events on the simulation's stream (recorded without
 * waiting: the host only waits when 32 steps are pending or nb_stage_ms is called), bit 1 =
 * collect the NB_DBG_COUNTERS statistics (slower traversal).
 * nb_stage_ms: mean milliseconds per stage over the steps (or nb_compute_accel calls) since the
 * previous nb_stage_ms call or since timing was switched on:
 * [0] bbox [1] keys [2] sort [3] build (direct sum with a partition: re-packing the float records of the
 * bodies other ranks advanced) [4] com [5] force+integrate [6] exchange: the collectives of
 * nb_step_distributed, or with nb_step_cells whatever the stream did between the parts of a step
 * (0 otherwise). */
int nb_set_stage_timing(nb_sim *sim, int on);
int nb_stage_ms(nb_sim *sim, float *out, int count);

/* Re-derives the exchange order (see NB_DBG_ORDER) from the current positions and re-sorts the
 * device state by it, on the device and asynchronously on the simulation's stream (a bounding box,
 * a radix sort of positions along a Hilbert curve, one gather per per-body array).  Nothing the caller sees changes
 * (nb_read_bodies keeps returning bodies in the caller's order; results differ only by the
 * rounding of a different summation order); it restores the speed of long runs: the direct-sum
 * kernel relies on neighbouring slots being neighbours in space, and the Barnes-Hut pipeline gathers
 * the state through its per-step sort, whose permutation drifts away from the identity as the bodies
 * mix.  nb_step and nb_step_distributed do it by themselves, every max(32, min(4096, 1.4e11 / n^2))
 * steps for the direct sum and every 256 steps for Barnes-Hut (not while cells are exchanged: a rank then
 * holds only its own bodies).  A 2-D Barnes-Hut simulation on one GPU needs none of that: each step of
 * its nb_step (every eighth above 2^24 bodies; not a step driven through nb_step_local, whose caller
 * holds the exchange buffer's address) leaves the bodies in the order of the step's own sort, in a
 * second set of per-body arrays that then changes places with the first (85 bytes per body more; made
 * with NB_KEEP_SLOTS set in the environment a handle does without and renews every 256 steps, a
 * switch for comparisons).  The reference keeps bodies in creation order and has nothing of the kind.
 * A partitioned simulation driven through nb_step_local / nb_step_finish may call it between
 * steps on EVERY rank, after all-gathering the velocity buffer the way the position buffer is
 * gathered (nb_velocity_buffer: same layout): every rank then derives the same order, and the
 * slots a rank owns afterwards hold bodies whose velocities other ranks kept so far. */
int nb_refresh_order(nb_sim *sim);
int nb_velocity_buffer(nb_sim *sim, void **device_ptr, size_t *chunk_records, size_t *record_bytes);

/* Conserved quantities of the current state, computed in FP64 (not part of the hot path):
 * out[0] total mass, [1..2] centre of mass, [3..4] momentum, [5] angular momentum about the
 * origin, [6] kinetic energy, [7] potential energy -G sum_{i<j} m_i m_j / sqrt(d^2 + eps^2) over
 * the pairs the force law does not skip (NaN when n > 131072: it is an O(n^2) sum). */
int nb_diagnostics(nb_sim *sim, double *out, int count);

/* ---- a force evaluation that allocates nothing ----
 *
 * No counterpart in the reference.  For callers that cannot allocate device memory or touch host
 * buffers where they call: code under HIP graph capture, tight serving loops, harnesses that forbid
 * hipMalloc.  It is nb_compute_accel with device arrays for input and output: the Barnes-Hut
 * accelerations the bodies of `sim` (their masses, radii, dynamic flags and parameters) have at the
 * positions pos_xy -- n records {x, y} in device memory, in the order the bodies were given to
 * nb_set_bodies -- written to acc_xy (device memory, n records {ax, ay}, same order; a static body
 * gets the acceleration the handle holds for it, as nb_read_bodies would return).  Nothing of the
 * handle is changed, and its positions and velocities are not read; what IS read are the arrays
 * the handle's own steps rewrite on the handle's stream -- the exchange order (see
 * nb_refresh_order: a one-GPU handle's steps change it every time), the stored accelerations, masses,
 * radii and dynamic flags.  This call orders nothing between the two streams (an event wait would not be
 * capturable): when hip_stream is not the handle's stream, call nb_sync after the last nb_step /
 * nb_refresh_order / nb_set_bodies before calling this, and let this call's work finish before
 * the next of those.
 *
 * Everything in between (permuted positions, keys, sort buffers, tree, centres of mass, traversal
 * records) lives in one static device array of 40 MiB that every GPU the library runs on holds
 * until the process ends, so the call makes no hipMalloc/hipFree and only enqueues kernels on
 * hip_stream (a hipStream_t; NULL: the handle's stream).  n above NB_STATIC_MAX_BODIES is refused
 * with NB_ERR_OVERFLOW.  All calls on one GPU share the array: calls that overlap in time on one
 * GPU (two streams, two threads, two handles) race; calls on one stream do not.  `sim` must be a
 * Barnes-Hut handle created on the current device (nb_create looks the array's address up, once
 * per device, so this call does not have to). */
#define NB_STATIC_MAX_BODIES 131072
int nb_compute_accel_static(nb_sim *sim, const double *pos_xy, double *acc_xy, void *hip_stream);

/* ==== three dimensions ====
 *
 * The reference is two-dimensional (struct Vector holds x and y, src/cpu/vector.h); its authors
 * name 3-D with an octree as future work.  The nb3_ entry points run the same rules with a
 * third coordinate: distance sqrt(dx^2+dy^2+dz^2+eps^2), the same cutoff tests, and for
 * Barnes-Hut an octree with one body per leaf whose cells are split at the midpoint of every
 * axis (z treated like y in QuadTree::getQuadrant, src/cpu/quadtree.cpp:41-72), children taken
 * in the order 4 (lower z) + 2 (lower y) + (right x), opening test cell x-width / distance <
 * theta (src/cpu/barnesHut.cpp:70-71).  The root box is padded and rounded on every axis as
 * BarnesHut::computeBoundingBox does (barnesHut.cpp:23-35); its z range is then widened to its
 * x-width when smaller, so that the cells of a thin disc are not slabs.  For bodies that all lie in the plane z = 0 the results
 * are those of the 2-D entry points.  nb_params.bucket_depth must be 0.
 * Each nb3_ function mirrors the nb_ function of the same name (the exchange buffer holds records
 * of 3 doubles). */

typedef struct nb_vec3 {
    double x;
    double y;
    double z;
} nb_vec3;

/* The reference's Body with three-component vectors: 96 bytes (isDynamic@0, mass@8, radius@16,
 * position@24, velocity@48, acceleration@72). */
typedef struct nb_body3 {
    uint8_t is_dynamic;
    uint8_t pad_[7];
    double mass;
    double radius;
    nb_vec3 position;
    nb_vec3 velocity;
    nb_vec3 acceleration;
} nb_body3;

typedef struct nb_sim3 nb_sim3; /* opaque */

/* Host only: nb_init_scenario(sim, n, seed) (identical x, y and velocities), then one more rand()
 * per body in index order: z = thickness (2 u - 1), except that the suns stay at z = 0; vz = 0. */
int nb3_init_scenario(int sim, int n, unsigned seed, double thickness, nb_body3 *out);

int nb3_create(const nb_params *params, int n, int alg, nb_sim3 **out);
int nb3_destroy(nb_sim3 *sim);
int nb3_setup(nb_sim3 *sim, int scenario, unsigned seed, double thickness);
int nb3_set_bodies(nb_sim3 *sim, const nb_body3 *bodies, int n);
int nb3_step(nb_sim3 *sim, int nsteps);
int nb3_compute_accel(nb_sim3 *sim);
int nb3_read_bodies(nb_sim3 *sim, nb_body3 *out);
int nb3_sync(nb_sim3 *sim);
int nb3_set_stream(nb_sim3 *sim, void *hip_stream);
int nb3_set_partition(nb_sim3 *sim, int rank, int world);
int nb3_exchange_buffer(nb_sim3 *sim, void **device_ptr, size_t *chunk_records, size_t *record_bytes);
int nb3_step_local(nb_sim3 *sim);
int nb3_step_finish(nb_sim3 *sim);
int nb3_refresh_order(nb_sim3 *sim);
int nb3_velocity_buffer(nb_sim3 *sim, void **device_ptr, size_t *chunk_records, size_t *record_bytes);
int nb3_comm_init(nb_sim3 *sim, int rank, int world, const void *id); /* id: from nb_comm_unique_id */
int nb3_step_distributed(nb_sim3 *sim, int nsteps);
/* `which` is an nb_debug_array.  Differences from nb_debug_read: NB_DBG_BBOX is 6 doubles
 * (topLeft.x, topLeft.y, botRight.x, botRight.y, upper z, lower z); keys hold 3 bits per level from
 * bit 61 down (at most 20 levels: bodies that share a cell of level 20, a millionth of the box, share a
 * leaf and act as one body at their centre of mass, as bodies sharing all 31 levels do in 2-D); NB_DBG_NODE_COM is 6 doubles per node (com.x, com.y, com.z, mass,
 * cell x-width or minus the body's radius, 0); NB_DBG_KEYS is not available. */
int nb3_debug_read(nb_sim3 *sim, int which, void *out, size_t bytes, size_t *written);
/* out[0] total mass, [1..3] centre of mass, [4..6] momentum, [7..9] angular momentum about the origin,
 * [10] kinetic energy, [11] potential energy (NaN when n > 131072), as nb_diagnostics defines them. */
int nb3_diagnostics(nb_sim3 *sim, double *out, int count);
int nb3_set_stage_timing(nb_sim3 *sim, int on);
int nb3_stage_ms(nb_sim3 *sim, float *out, int count);

const char *nb_last_error(void);
const char *nb_version(void);

#ifdef __cplusplus
}
#endif
#endif /* NBODY_H */
