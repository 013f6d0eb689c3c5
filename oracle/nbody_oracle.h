/*
 * nbody_oracle.h -- TEST INFRASTRUCTURE, not product code.
 *
 * A plain-C, double-precision, single-threaded restatement of the reference's
 * CPU algorithms (kr1s0404/N-body-simulation, src/cpu) and of its scenario
 * generators.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the HIP library never does.
 *
 * Parity pin: every function here is checked by tests/test_oracle_golden.py
 * against outputs of the reference's own CPU classes compiled unmodified
 * (oracle/Makefile -> oracle/_ref/libnbody_ref.so) and stored in tests/golden/.
 */
#ifndef NBODY_ORACLE_H
#define NBODY_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Layout of the reference's Body (src/cpu/body.h:22-34 on x86-64). */
typedef struct orc_body {
    uint8_t is_dynamic;
    uint8_t pad_[7];
    double mass, radius;
    double px, py, vx, vy, ax, ay;
} orc_body;

typedef struct orc_params {
    double gravity, eps, dt, theta, collision_th, bbox_margin;
} orc_params;

void orc_default_params(orc_params *p);

/* Threads for the per-target loops of the *_accel functions (default 1, the reference's own
 * single thread).  Results are independent of it. */
void orc_set_threads(int threads);
int orc_get_threads(void);

/* Scenarios 0 spiral, 1 random, 2 collision, 3 solar (GPU-BH numbering). */
int orc_init_scenario(int sim, int n, unsigned seed, orc_body *out);

/* Direct sum: accelerations of targets [t0,t1) from all n bodies (acc = 2 doubles per target). */
void orc_direct_accel(const orc_params *p, int n, const orc_body *b, int t0, int t1, double *acc);
/* Same for an arbitrary list of targets. */
void orc_direct_accel_list(const orc_params *p, int n, const orc_body *b, int nt, const int *targets, double *acc);
/* nsteps of DirectSum::update on b in place. */
void orc_direct_step(const orc_params *p, int n, orc_body *b, int nsteps);

/* Barnes-Hut: bounding box as the reference rounds it (4 doubles tl.x tl.y br.x br.y). */
void orc_bh_bbox(const orc_params *p, int n, const orc_body *b, double *box);
/* Accelerations of every body (static bodies keep b[i].ax, b[i].ay); returns 0, or -1 when
 * two bodies coincide (the reference recurses without end there). stats may be NULL:
 * [0] node visits, [1] accepted cells + leaf bodies, [2] max depth, [3] tree nodes. */
int orc_bh_accel(const orc_params *p, int n, const orc_body *b, double *acc, double *stats);
/* The same for the bodies targets[0..nt) only (acc = 2 doubles per listed target): the tree of
 * all n bodies is built once and walked for those.  order, when not NULL, is the sequence in which
 * the bodies are inserted (a permutation of 0..n-1; the resulting tree does not depend on it, but
 * inserting in a spatially sorted order is several times faster at millions of bodies). */
int orc_bh_accel_list(const orc_params *p, int n, const orc_body *b, const int *order, int nt, const int *targets,
                      double *acc, double *stats);
/* nsteps of BarnesHut::update on b in place; returns 0 or -1 as above. */
int orc_bh_step(const orc_params *p, int n, orc_body *b, int nsteps);

/* The tree of the reference's GPU program (src/gpu/BH/barnesHut_kernel.cu), restated on the CPU
 * in FP64: root box as ComputeBoundingBoxKernel rounds it (:62-111), cells split by getQuadrant
 * (:118-147) down to level `depth` (9 in the reference: MAX_NODES 349525), a cell with one body or
 * on the last level is a leaf, every node's centre of mass is the mass-weighted mean of its bodies
 * (:196-243, summed here in index order; the reference's order depends on atomics), forces by
 * ComputeForce (:378-423): a leaf acts unless 2 r_i + th > dist, a cell with width/dist < theta
 * likewise, other cells are opened.  No output of that program exists to pin this restatement
 * (it cannot be built here): parity unpinned. */
void orc_bh_gpu_bbox(const orc_params *p, int n, const orc_body *b, double *box);
void orc_bh_gpu_accel(const orc_params *p, int n, const orc_body *b, int depth, double *acc);
/* ... of the bodies targets[0..nt) only (the tree is that of all n bodies); acc: nt records. */
void orc_bh_gpu_accel_list(const orc_params *p, int n, const orc_body *b, int depth, int nt, const int *targets, double *acc);
/* nsteps of the GPU program's update (only dynamic bodies are advanced, :431-443). */
void orc_bh_gpu_step(const orc_params *p, int n, orc_body *b, int depth, int nsteps);

/* 2x31-bit quadrant path of each body by the reference's midpoint descent from box:
 * digit at level l (0 = root split) sits in bits [61-2l, 60-2l] and is
 * 2*(below the y midpoint) + (right of the x midpoint). */
void orc_morton_keys(int n, const orc_body *b, const double *box, uint64_t *keys);

/* ---- three dimensions (oracle/nbody_oracle3.c) ----
 * The reference is two-dimensional throughout; its authors name 3-D and an octree as future
 * work.  These functions extend its rules by the third coordinate and change nothing else:
 * distance sqrt(dx^2+dy^2+dz^2), the same softening and cutoff tests, an octree with one body
 * per leaf whose cells are split at the midpoint of every axis (z like y: mid <= z is the upper
 * half), children visited in the order of 4*(lower z) + 2*(lower y) + (right x) -- for bodies
 * that all have z = 0 that is the reference's TL,TR,BL,BR and the octree's cells are the
 * quadtree's -- and the opening test cell x-width / dist < theta.
 * Parity pin: with every z = 0 these functions return bit for bit what the 2-D functions above
 * return (tests/test_oracle_golden.py), and those are pinned by the compiled reference.  For
 * z != 0 nothing of the reference exists to compare with: parity unpinned. */
typedef struct orc3_body {
    uint8_t is_dynamic;
    uint8_t pad_[7];
    double mass, radius;
    double p[3], v[3], a[3];
} orc3_body;

/* The 2-D scenario `sim` for the same seed, then one more rand() per body in index order:
 * z = thickness * (2 u - 1), except that bodies of the suns' mass stay at z = 0.  vz = 0. */
int orc3_init_scenario(int sim, int n, unsigned seed, double thickness, orc3_body *out);
void orc3_direct_accel(const orc_params *p, int n, const orc3_body *b, int t0, int t1, double *acc);
void orc3_direct_step(const orc_params *p, int n, orc3_body *b, int nsteps);
/* box: tl.x, tl.y, br.x, br.y as in orc_bh_bbox, then the upper and the lower z bound: padded and
 * rounded like y, then widened about its middle to the x-width if it is smaller. */
void orc3_bh_bbox(const orc_params *p, int n, const orc3_body *b, double *box);
/* stats as orc_bh_accel; acc = 3 doubles per body. */
int orc3_bh_accel(const orc_params *p, int n, const orc3_body *b, double *acc, double *stats);
/* ... of the bodies targets[0..nt) only, from the octree of all n bodies (inserted in the sequence
 * order[0..n) when given: same tree, faster build); acc: nt records of 3 doubles. */
int orc3_bh_accel_list(const orc_params *p, int n, const orc3_body *b, const int *order, int nt, const int *targets,
                       double *acc, double *stats);
int orc3_bh_step(const orc_params *p, int n, orc3_body *b, int nsteps);
/* Octant path of each body, 20 levels: the digit of level l sits in bits [61-3l, 59-3l] and is
 * 4*(below the z midpoint) + 2*(below the y midpoint) + (right of the x midpoint). */
void orc3_morton_keys(int n, const orc3_body *b, const double *box, uint64_t *keys);

#ifdef __cplusplus
}
#endif
#endif
