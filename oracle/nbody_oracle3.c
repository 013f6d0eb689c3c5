/*
 * nbody_oracle3.c -- TEST INFRASTRUCTURE, not product code (see nbody_oracle.h).
 *
 * The reference's CPU algorithms with a third coordinate added and nothing else changed
 * (the reference itself is two-dimensional: src/cpu/vector.h has x and y only).  Every
 * function names the 2-D reference lines whose rule it extends.  With all z = 0 the
 * arithmetic is bit for bit that of nbody_oracle.c: the extra terms are added last and
 * are exact zeros, and the octree's children are visited in an order that reduces to
 * TL,TR,BL,BR.
 */
#include "nbody_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define ORC_SUN_MASS 1.9890e30

/* Vector::getDistance (src/cpu/vector.cpp:31-35) with a z term. */
static double dist3d(const double *a, const double *b)
{
    return sqrt(pow(b[0] - a[0], 2) + pow(b[1] - a[1], 2) + pow(b[2] - a[2], 2));
}

int orc3_init_scenario(int sim, int n, unsigned seed, double thickness, orc3_body *out)
{
    int m = sim == 3 ? 5 : n;
    if (n < 1 || !out || (sim == 3 && n < 5))
        return -1;
    orc_body *flat = (orc_body *)malloc(sizeof(orc_body) * (size_t)m);
    if (orc_init_scenario(sim, m, seed, flat) != 0) {
        free(flat);
        return -1;
    }
    if (sim == 3)
        srand(seed); /* the solar system draws nothing itself */
    for (int i = 0; i < m; ++i) {
        double u = rand() / (double)RAND_MAX;
        memset(&out[i], 0, sizeof(out[i]));
        out[i].is_dynamic = flat[i].is_dynamic;
        out[i].mass = flat[i].mass;
        out[i].radius = flat[i].radius;
        out[i].p[0] = flat[i].px;
        out[i].p[1] = flat[i].py;
        out[i].p[2] = flat[i].mass == ORC_SUN_MASS ? 0.0 : thickness * (2.0 * u - 1.0);
        out[i].v[0] = flat[i].vx;
        out[i].v[1] = flat[i].vy;
    }
    free(flat);
    return 0;
}

/* ----------------------------------------------------------------- direct sum */

/* DirectSum::calculateAcceleration for one body (src/cpu/directSum.cpp:27-46), isCollide
 * directSum.cpp:69-72 (>=). */
static void direct3_one(const orc_params *p, int n, const orc3_body *b, int i, double *acc)
{
    const orc3_body *bi = &b[i];
    double fx = 0.0, fy = 0.0, fz = 0.0;
    double eps2 = p->eps * p->eps;
    for (int j = 0; j < n; ++j) {
        const orc3_body *bj = &b[j];
        if (i != j && bi->is_dynamic && !(bi->radius + bj->radius + p->collision_th >= dist3d(bi->p, bj->p))) {
            double rx = bj->p[0] - bi->p[0], ry = bj->p[1] - bi->p[1], rz = bj->p[2] - bi->p[2];
            double r = sqrt((rx * rx) + (ry * ry) + (rz * rz) + eps2);
            double f = (p->gravity * bi->mass * bj->mass) / (r * r * r + eps2);
            fx += rx * f;
            fy += ry * f;
            fz += rz * f;
        }
    }
    acc[0] = 0.0 + fx / bi->mass;
    acc[1] = 0.0 + fy / bi->mass;
    acc[2] = 0.0 + fz / bi->mass;
}

void orc3_direct_accel(const orc_params *p, int n, const orc3_body *b, int t0, int t1, double *acc)
{
    for (int i = t0; i < t1; ++i)
        direct3_one(p, n, b, i, &acc[3 * (i - t0)]);
}

/* directSum.cpp:49-67 / barnesHut.cpp:108-123: every body, velocity first. */
static void integrate3(const orc_params *p, int n, orc3_body *b)
{
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
            b[i].v[k] += b[i].a[k] * p->dt;
    for (int i = 0; i < n; ++i)
        for (int k = 0; k < 3; ++k)
            b[i].p[k] += b[i].v[k] * p->dt;
}

void orc3_direct_step(const orc_params *p, int n, orc3_body *b, int nsteps)
{
    double *acc = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    for (int s = 0; s < nsteps; ++s) {
        orc3_direct_accel(p, n, b, 0, n, acc);
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k)
                b[i].a[k] = acc[3 * i + k];
        integrate3(p, n, b);
    }
    free(acc);
}

/* ---------------------------------------------------------------- Barnes-Hut */

/* BarnesHut::computeBoundingBox (src/cpu/barnesHut.cpp:23-35): bounds through fminf/fmaxf. */
void orc3_bh_bbox(const orc_params *p, int n, const orc3_body *b, double *box)
{
    double tlx = INFINITY, tly = -INFINITY, brx = -INFINITY, bry = INFINITY, zup = -INFINITY, zlo = INFINITY;
    for (int i = 0; i < n; ++i) {
        tlx = fminf((float)tlx, (float)(b[i].p[0] - p->bbox_margin));
        tly = fmaxf((float)tly, (float)(b[i].p[1] + p->bbox_margin));
        brx = fmaxf((float)brx, (float)(b[i].p[0] + p->bbox_margin));
        bry = fminf((float)bry, (float)(b[i].p[1] - p->bbox_margin));
        zup = fmaxf((float)zup, (float)(b[i].p[2] + p->bbox_margin));
        zlo = fminf((float)zlo, (float)(b[i].p[2] - p->bbox_margin));
    }
    box[0] = tlx;
    box[1] = tly;
    box[2] = brx;
    box[3] = bry;
    /* The root is at least as deep in z as it is wide in x (centred on the z range): cells then
     * are no flatter than wide, so a thin disc is not cut into slabs that the opening test, which
     * looks at the x-width only, would all have to accept one by one.  Flat input stays on one
     * side of every z split either way. */
    if (zup - zlo < brx - tlx) {
        double mid = (zup + zlo) / 2.0, half = (brx - tlx) / 2.0;
        zup = mid + half;
        zlo = mid - half;
    }
    box[4] = zup;
    box[5] = zlo;
}

/* QuadTree (src/cpu/quadtree.h:25-54) with eight children. */
typedef struct onode {
    double tlx, tly, brx, bry, zup, zlo;
    double cm[3], mass;
    int is_leaf;
    int body;     /* -1 = none */
    int child[8]; /* 4*(lower z) + 2*(lower y) + (right x); -1 = none */
} onode;

typedef struct otree {
    onode *nd;
    int count, cap;
    int failed;
    int max_depth;
} otree;

static int ot_new(otree *t, double tlx, double tly, double brx, double bry, double zup, double zlo)
{
    if (t->count == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nd = (onode *)realloc(t->nd, sizeof(onode) * (size_t)t->cap);
    }
    onode *q = &t->nd[t->count];
    q->tlx = tlx;
    q->tly = tly;
    q->brx = brx;
    q->bry = bry;
    q->zup = zup;
    q->zlo = zlo;
    q->cm[0] = q->cm[1] = q->cm[2] = -1;
    q->mass = 0.0;
    q->is_leaf = 1;
    q->body = -1;
    for (int c = 0; c < 8; ++c)
        q->child[c] = -1;
    return t->count++;
}

/* QuadTree::getQuadrant (quadtree.cpp:41-72): mid >= x is the left half, mid <= y the top
 * half; z is treated like y. */
static int ot_octant(const onode *q, const double *x)
{
    int right = !((q->tlx + q->brx) / 2.0 >= x[0]);
    int ylow = !((q->tly + q->bry) / 2.0 <= x[1]);
    int zlow = !((q->zup + q->zlo) / 2.0 <= x[2]);
    return 4 * zlow + 2 * ylow + right;
}

/* QuadTree::insert, quadtree.cpp:73-144. */
static void ot_insert(otree *t, int ni, const orc3_body *b, int body, int depth)
{
    if (t->failed)
        return;
    if (depth > 256) {
        t->failed = 1;
        return;
    }
    if (depth > t->max_depth)
        t->max_depth = depth;
    onode *q = &t->nd[ni];
    if (q->body == -1 && q->is_leaf) {
        q->body = body;
        return;
    }
    if (q->body != -1) {
        if (q->body == body) {
            q->body = -1;
        } else {
            ot_insert(t, ni, b, q->body, depth);
            q = &t->nd[ni];
        }
    }
    q->is_leaf = 0;
    int c = ot_octant(q, b[body].p);
    if (q->child[c] == -1) {
        double mx = (q->tlx + q->brx) / 2, my = (q->tly + q->bry) / 2, mz = (q->zup + q->zlo) / 2;
        double tlx = (c & 1) ? mx : q->tlx, brx = (c & 1) ? q->brx : mx;
        double tly = (c & 2) ? my : q->tly, bry = (c & 2) ? q->bry : my;
        double zup = (c & 4) ? mz : q->zup, zlo = (c & 4) ? q->zlo : mz;
        int nn = ot_new(t, tlx, tly, brx, bry, zup, zlo);
        t->nd[ni].child[c] = nn;
    }
    ot_insert(t, t->nd[ni].child[c], b, body, depth + 1);
}

/* updateCenterMass, quadtree.cpp:187-228. */
static void ot_center_mass(otree *t, int ni, const orc3_body *b)
{
    if (ni < 0)
        return;
    onode *q = &t->nd[ni];
    if (q->body != -1) {
        q->mass = b[q->body].mass;
        for (int k = 0; k < 3; ++k)
            q->cm[k] = b[q->body].p[k];
        return;
    }
    for (int c = 0; c < 8; ++c)
        ot_center_mass(t, q->child[c], b);
    /* the reference adds the four child masses, absent ones as 0.0, left to right; absent
     * children are skipped here instead, which is the same sum when four of the eight are absent
     * in the pattern of a flat tree (x + 0.0 == x) */
    double total = 0.0, s[3] = {0.0, 0.0, 0.0};
    int first = 1;
    for (int c = 0; c < 8; ++c) {
        if (q->child[c] < 0)
            continue;
        const onode *ch = &t->nd[q->child[c]];
        total = first ? ch->mass : total + ch->mass;
        first = 0;
        for (int k = 0; k < 3; ++k)
            s[k] += ch->cm[k] * ch->mass;
    }
    q->mass = total;
    for (int k = 0; k < 3; ++k)
        q->cm[k] = s[k] / total;
}

typedef struct trav3 {
    const orc_params *p;
    const orc3_body *b;
    const otree *t;
    double eps2;
    double a[3];
    double visits, accepted;
} trav3;

/* BarnesHut::calculateForceHelper, barnesHut.cpp:50-89; isCollide barnesHut.cpp:125-134. */
static void ot_force(trav3 *tv, int ni, int body)
{
    if (ni < 0)
        return;
    const onode *q = &tv->t->nd[ni];
    const orc3_body *bi = &tv->b[body];
    const orc_params *p = tv->p;
    tv->visits += 1;
    if (q->body != -1) {
        const orc3_body *bj = &tv->b[q->body];
        if ((bi->radius + bj->radius + p->collision_th > dist3d(bi->p, bj->p)) || q->body == body)
            return;
        double rx = bj->p[0] - bi->p[0], ry = bj->p[1] - bi->p[1], rz = bj->p[2] - bi->p[2];
        double r = sqrt((rx * rx) + (ry * ry) + (rz * rz) + tv->eps2);
        double s = (p->gravity * bi->mass * bj->mass) / (r * r * r + tv->eps2);
        tv->a[0] += (rx * s) / bi->mass;
        tv->a[1] += (ry * s) / bi->mass;
        tv->a[2] += (rz * s) / bi->mass;
        tv->accepted += 1;
        return;
    }
    double d = dist3d(bi->p, q->cm);
    double sd = (q->brx - q->tlx) / d;
    if (sd < p->theta) {
        double rx = q->cm[0] - bi->p[0], ry = q->cm[1] - bi->p[1], rz = q->cm[2] - bi->p[2];
        if (!(bi->radius * 2 + p->collision_th > d)) {
            double r = sqrt((rx * rx) + (ry * ry) + (rz * rz) + tv->eps2);
            double s = (p->gravity * bi->mass * q->mass) / (r * r * r + tv->eps2);
            tv->a[0] += (rx * s) / bi->mass;
            tv->a[1] += (ry * s) / bi->mass;
            tv->a[2] += (rz * s) / bi->mass;
        }
        tv->accepted += 1;
        return;
    }
    for (int c = 0; c < 8; ++c)
        ot_force(tv, q->child[c], body);
}

/* The accelerations of the bodies targets[0..nt) (NULL: body k) from the octree of all n bodies,
 * which is built once.  order (may be NULL) gives the sequence in which the bodies are inserted:
 * the one-body-per-leaf tree does not depend on it, only the time the insertions take does. */
int orc3_bh_accel_list(const orc_params *p, int n, const orc3_body *b, const int *order, int nt, const int *targets,
                       double *acc, double *stats)
{
    double box[6];
    orc3_bh_bbox(p, n, b, box);
    otree t;
    memset(&t, 0, sizeof(t));
    int root = ot_new(&t, box[0], box[1], box[2], box[3], box[4], box[5]);
    for (int i = 0; i < n; ++i)
        ot_insert(&t, root, b, order ? order[i] : i, 0);
    if (t.failed) {
        free(t.nd);
        return -1;
    }
    ot_center_mass(&t, root, b);
    double nv = 0.0, na = 0.0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nv, na) num_threads(orc_get_threads()) if (orc_get_threads() > 1)
    for (int j = 0; j < nt; ++j) {
        const int i = targets ? targets[j] : j;
        if (b[i].is_dynamic) {
            trav3 tv;
            tv.p = p;
            tv.b = b;
            tv.t = &t;
            tv.eps2 = p->eps * p->eps;
            tv.visits = tv.accepted = 0;
            tv.a[0] = tv.a[1] = tv.a[2] = 0.0;
            ot_force(&tv, root, i);
            for (int k = 0; k < 3; ++k)
                acc[3 * j + k] = tv.a[k];
            nv += tv.visits;
            na += tv.accepted;
        } else {
            for (int k = 0; k < 3; ++k)
                acc[3 * j + k] = b[i].a[k];
        }
    }
    if (stats) {
        stats[0] = nv;
        stats[1] = na;
        stats[2] = t.max_depth;
        stats[3] = t.count;
    }
    free(t.nd);
    return 0;
}

int orc3_bh_accel(const orc_params *p, int n, const orc3_body *b, double *acc, double *stats)
{
    return orc3_bh_accel_list(p, n, b, NULL, n, NULL, acc, stats);
}

int orc3_bh_step(const orc_params *p, int n, orc3_body *b, int nsteps)
{
    double *acc = (double *)malloc(sizeof(double) * 3 * (size_t)n);
    for (int s = 0; s < nsteps; ++s) {
        if (orc3_bh_accel(p, n, b, acc, NULL) != 0) {
            free(acc);
            return -1;
        }
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < 3; ++k)
                b[i].a[k] = acc[3 * i + k];
        integrate3(p, n, b);
    }
    free(acc);
    return 0;
}

void orc3_morton_keys(int n, const orc3_body *b, const double *box, uint64_t *keys)
{
    for (int i = 0; i < n; ++i) {
        double tlx = box[0], tly = box[1], brx = box[2], bry = box[3], zup = box[4], zlo = box[5];
        uint64_t k = 0;
        for (int l = 0; l < 20; ++l) {
            double mx = (tlx + brx) / 2.0, my = (tly + bry) / 2.0, mz = (zup + zlo) / 2.0;
            uint64_t right = !(mx >= b[i].p[0]);
            uint64_t ylow = !(my <= b[i].p[1]);
            uint64_t zlow = !(mz <= b[i].p[2]);
            if (right)
                tlx = mx;
            else
                brx = mx;
            if (ylow)
                tly = my;
            else
                bry = my;
            if (zlow)
                zup = mz;
            else
                zlo = mz;
            k = (k << 3) | (zlow << 2) | (ylow << 1) | right;
        }
        keys[i] = k << 2;
    }
}
