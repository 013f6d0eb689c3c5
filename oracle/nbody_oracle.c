/*
 * nbody_oracle.c -- TEST INFRASTRUCTURE, not product code (see nbody_oracle.h).
 *
 * Double-precision CPU restatement of the reference's direct-sum and
 * Barnes-Hut steps and of its scenario generators.  Each function names the
 * reference lines it follows (paths under the reference checkout).  Operation
 * order is kept where it decides the last bits (sum order over j, child order
 * TL,TR,BL,BR, force/mass after the product).
 */
#include "nbody_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* src/gpu/BH/constants.h:23-43 */
#define ORC_NBODY_WIDTH 10.0e11
#define ORC_CENTERX 0.0
#define ORC_CENTERY 0.0
#define ORC_MIN_DIST 2.0e10
#define ORC_MAX_DIST 5.0e11
#define ORC_SUN_MASS 1.9890e30
#define ORC_SUN_DIA 1.3927e6
#define ORC_EARTH_MASS 5.974e24
#define ORC_EARTH_DIA 12756

/* Threads the per-target loops use (the reference itself is single-threaded: 1 is its timing;
 * more threads only give the "all host cores" figure of the benchmark's CPU baseline).  The
 * results do not depend on it: every target's sum is formed by one thread in the same order. */
static int g_threads = 1;

void orc_set_threads(int threads) { g_threads = threads < 1 ? 1 : threads; }
int orc_get_threads(void) { return g_threads; }

void orc_default_params(orc_params *p)
{
    p->gravity = 6.67E-11;     /* constants.h:25 */
    p->eps = 0.5;              /* constants.h:26, directSum.h:27, barnesHut.h:28 */
    p->dt = 25000.0;           /* constants.h:27, directSum.h:28, barnesHut.h:29 */
    p->theta = 0.5;            /* constants.h:28, barnesHut.h:30 */
    p->collision_th = 1.0e10;  /* constants.h:37 */
    p->bbox_margin = 1.0e10;   /* literal in barnesHut.cpp:28-31 */
}

/* Vector::getDistance, src/cpu/vector.cpp:31-35 (from a to b: pow(b-a, 2)). */
static double dist2d(double ax, double ay, double bx, double by)
{
    return sqrt(pow(bx - ax, 2) + pow(by - ay, 2));
}

/* ------------------------------------------------------------------ scenarios */

static void set_body(orc_body *b, int dyn, double m, double r, double px, double py, double vx, double vy)
{
    memset(b, 0, sizeof(*b));
    b->is_dynamic = (uint8_t)dyn;
    b->mass = m;
    b->radius = r;
    b->px = px;
    b->py = py;
    b->vx = vx;
    b->vy = vy;
    b->ax = 0.0;
    b->ay = 0.0;
}

/* One disc body: angle is drawn before the radius (barnesHutCuda.cu:115-122). */
static void disc_body(orc_body *b, double cx, double cy, double mind, double maxd, int orbit)
{
    double angle = 2 * M_PI * (rand() / (double)RAND_MAX);
    double radius = (maxd - mind) * (rand() / (double)RAND_MAX) + mind;
    double x = cx + radius * cos(angle);
    double y = cy + radius * sin(angle);
    double vx = 0.0, vy = 0.0;
    if (orbit) {
        /* barnesHutCuda.cu:124-130 */
        double distance = sqrt(pow(x - cx, 2) + pow(y - cy, 2));
        double rx = x - cx, ry = y - cy;
        double axn = rx / distance, ayn = ry / distance;
        double esc = sqrt((6.67E-11 * ORC_SUN_MASS) / (distance));
        vx = -ayn * esc;
        vy = axn * esc;
    }
    set_body(b, 1, ORC_EARTH_MASS, ORC_EARTH_DIA, x, y, vx, vy);
}

int orc_init_scenario(int sim, int n, unsigned seed, orc_body *out)
{
    if (n < 1 || !out)
        return -1;
    if (sim == 0 || sim == 1) {
        /* spiral: barnesHutCuda.cu:105-145 (= cpu/nBody.cpp:76-108);
           random: barnesHutCuda.cu:73-103 (= cpu/nBody.cpp:53-74) */
        srand(seed);
        for (int i = 0; i < n - 1; ++i)
            disc_body(&out[i], ORC_CENTERX, ORC_CENTERY, ORC_MIN_DIST, ORC_MAX_DIST, sim == 0);
        set_body(&out[n - 1], 0, ORC_SUN_MASS, ORC_SUN_DIA, ORC_CENTERX, ORC_CENTERY, 0, 0);
        return 0;
    }
    if (sim == 2) {
        /* collision: barnesHutCuda.cu:147-226; both suns are dynamic */
        if (n < 2)
            return -1;
        srand(seed);
        double maxd = ORC_MAX_DIST / 4.0, mind = ORC_MIN_DIST;
        double cx = -ORC_NBODY_WIDTH / 6.0, cy = ORC_CENTERY;
        int galaxy1 = n / 2;
        for (int i = 0; i < galaxy1 - 1; ++i)
            disc_body(&out[i], cx, cy, mind, maxd, 1);
        set_body(&out[galaxy1 - 1], 1, ORC_SUN_MASS, ORC_SUN_DIA, cx, cy, 0, 0);
        cx = ORC_NBODY_WIDTH / 6.0;
        for (int i = galaxy1; i < n - 1; ++i)
            disc_body(&out[i], cx, cy, mind, maxd, 1);
        set_body(&out[n - 1], 1, ORC_SUN_MASS, ORC_SUN_DIA, cx, cy, 0, 0);
        return 0;
    }
    if (sim == 3) {
        /* solar system: barnesHutCuda.cu:238-245 (= cpu/nBody.cpp:110-119) */
        if (n < 5)
            return -1;
        set_body(&out[0], 1, 5.9740e24, 1.3927e6, 1.4960e11, 0, 0, 2.9800e4);
        set_body(&out[1], 1, 6.4190e23, 1.3927e6, 2.2790e11, 0, 0, 2.4100e4);
        set_body(&out[2], 1, 3.3020e23, 1.3927e6, 5.7900e10, 0, 0, 4.7900e4);
        set_body(&out[3], 1, 4.8690e24, 1.3927e6, 1.0820e11, 0, 0, 3.5000e4);
        set_body(&out[4], 0, 1.9890e30, 1.3927e6, ORC_CENTERX, ORC_CENTERY, 0, 0);
        return 0;
    }
    return -1;
}

/* ----------------------------------------------------------------- direct sum */

/* DirectSum::calculateAcceleration for one body, src/cpu/directSum.cpp:27-46;
 * isCollide is directSum.cpp:69-72 (note >=). */
static void direct_one(const orc_params *p, int n, const orc_body *b, int i, double *ax, double *ay)
{
    const orc_body *bi = &b[i];
    double fx = 0.0, fy = 0.0;
    double eps2 = p->eps * p->eps;
    for (int j = 0; j < n; ++j) {
        const orc_body *bj = &b[j];
        if (i != j && bi->is_dynamic &&
            !(bi->radius + bj->radius + p->collision_th >= dist2d(bi->px, bi->py, bj->px, bj->py))) {
            double rx = bj->px - bi->px, ry = bj->py - bi->py;
            double r = sqrt((rx * rx) + (ry * ry) + eps2);
            double f = (p->gravity * bi->mass * bj->mass) / (r * r * r + eps2);
            fx += rx * f;
            fy += ry * f;
        }
    }
    *ax = 0.0 + fx / bi->mass;
    *ay = 0.0 + fy / bi->mass;
}

void orc_direct_accel(const orc_params *p, int n, const orc_body *b, int t0, int t1, double *acc)
{
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int i = t0; i < t1; ++i)
        direct_one(p, n, b, i, &acc[2 * (i - t0)], &acc[2 * (i - t0) + 1]);
}

void orc_direct_accel_list(const orc_params *p, int n, const orc_body *b, int nt, const int *targets, double *acc)
{
#pragma omp parallel for schedule(static) num_threads(g_threads) if (g_threads > 1)
    for (int k = 0; k < nt; ++k)
        direct_one(p, n, b, targets[k], &acc[2 * k], &acc[2 * k + 1]);
}

/* Velocity then position for every body, directSum.cpp:49-67 / barnesHut.cpp:108-123. */
static void integrate(const orc_params *p, int n, orc_body *b)
{
    for (int i = 0; i < n; ++i) {
        b[i].vx += b[i].ax * p->dt;
        b[i].vy += b[i].ay * p->dt;
    }
    for (int i = 0; i < n; ++i) {
        b[i].px += b[i].vx * p->dt;
        b[i].py += b[i].vy * p->dt;
    }
}

/* DirectSum::update, directSum.cpp:74-79. */
void orc_direct_step(const orc_params *p, int n, orc_body *b, int nsteps)
{
    double *acc = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    for (int s = 0; s < nsteps; ++s) {
        orc_direct_accel(p, n, b, 0, n, acc);
        for (int i = 0; i < n; ++i) {
            b[i].ax = acc[2 * i];
            b[i].ay = acc[2 * i + 1];
        }
        integrate(p, n, b);
    }
    free(acc);
}

/* ---------------------------------------------------------------- Barnes-Hut */

/* BarnesHut::computeBoundingBox, src/cpu/barnesHut.cpp:23-35: the running
 * bounds pass through fminf/fmaxf, i.e. every candidate is rounded to float. */
void orc_bh_bbox(const orc_params *p, int n, const orc_body *b, double *box)
{
    double tlx = INFINITY, tly = -INFINITY, brx = -INFINITY, bry = INFINITY;
    for (int i = 0; i < n; ++i) {
        tlx = fminf((float)tlx, (float)(b[i].px - p->bbox_margin));
        tly = fmaxf((float)tly, (float)(b[i].py + p->bbox_margin));
        brx = fmaxf((float)brx, (float)(b[i].px + p->bbox_margin));
        bry = fminf((float)bry, (float)(b[i].py - p->bbox_margin));
    }
    box[0] = tlx;
    box[1] = tly;
    box[2] = brx;
    box[3] = bry;
}

/* QuadTree, src/cpu/quadtree.h:25-54, as an index-linked arena. */
typedef struct qnode {
    double tlx, tly, brx, bry;
    double cmx, cmy, mass;
    int is_leaf;
    int body;     /* -1 = none */
    int child[4]; /* 0 topLeft, 1 topRight, 2 botLeft, 3 botRight; -1 = none */
} qnode;

typedef struct qtree {
    qnode *nd;
    int count, cap;
    int failed; /* set when the insertion depth runs away (coincident bodies) */
    int max_depth;
} qtree;

static int qt_new(qtree *t, double tlx, double tly, double brx, double bry)
{
    if (t->count == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nd = (qnode *)realloc(t->nd, sizeof(qnode) * (size_t)t->cap);
    }
    qnode *q = &t->nd[t->count];
    q->tlx = tlx;
    q->tly = tly;
    q->brx = brx;
    q->bry = bry;
    q->cmx = -1;
    q->cmy = -1;
    q->mass = 0.0;
    q->is_leaf = 1;
    q->body = -1;
    q->child[0] = q->child[1] = q->child[2] = q->child[3] = -1;
    return t->count++;
}

/* QuadTree::getQuadrant, quadtree.cpp:41-72, mapped to the child slot:
 * 2 (top left) -> 0, 1 (top right) -> 1, 3 (bottom left) -> 2, 4 (bottom right) -> 3. */
static int qt_quadrant(const qnode *q, double x, double y)
{
    if ((q->tlx + q->brx) / 2.0 >= x)
        return ((q->tly + q->bry) / 2.0 <= y) ? 0 : 2;
    return ((q->tly + q->bry) / 2.0 <= y) ? 1 : 3;
}

/* QuadTree::insert, quadtree.cpp:73-144. */
static void qt_insert(qtree *t, int ni, const orc_body *b, int body, int depth)
{
    if (t->failed)
        return;
    if (depth > 256) {
        t->failed = 1;
        return;
    }
    if (depth > t->max_depth)
        t->max_depth = depth;
    qnode *q = &t->nd[ni];
    double x = b[body].px, y = b[body].py;
    if (!(x >= q->tlx && x <= q->brx && y <= q->tly && y >= q->bry)) {
        fprintf(stderr, "ERROR: body out of bound\n");
        return;
    }
    if (q->body == -1 && q->is_leaf) {
        q->body = body;
        return;
    }
    if (q->body != -1) {
        if (q->body == body) {
            q->body = -1;
        } else {
            qt_insert(t, ni, b, q->body, depth);
            q = &t->nd[ni]; /* the arena may have moved */
        }
    }
    q->is_leaf = 0;
    int c = qt_quadrant(q, x, y);
    if (q->child[c] == -1) {
        double mx = (q->tlx + q->brx) / 2, my = (q->tly + q->bry) / 2;
        double tlx = q->tlx, tly = q->tly, brx = q->brx, bry = q->bry;
        int nn;
        if (c == 1)
            nn = qt_new(t, mx, tly, brx, my);
        else if (c == 0)
            nn = qt_new(t, tlx, tly, mx, my);
        else if (c == 2)
            nn = qt_new(t, tlx, my, mx, bry);
        else
            nn = qt_new(t, mx, my, brx, bry);
        t->nd[ni].child[c] = nn;
    }
    qt_insert(t, t->nd[ni].child[c], b, body, depth + 1);
}

/* updateCenterMass, quadtree.cpp:187-228. */
static void qt_center_mass(qtree *t, int ni, const orc_body *b)
{
    if (ni < 0)
        return;
    qnode *q = &t->nd[ni];
    if (q->body != -1) {
        q->mass = b[q->body].mass;
        q->cmx = b[q->body].px;
        q->cmy = b[q->body].py;
        return;
    }
    for (int c = 0; c < 4; ++c)
        qt_center_mass(t, q->child[c], b);
    double m[4];
    for (int c = 0; c < 4; ++c)
        m[c] = q->child[c] >= 0 ? t->nd[q->child[c]].mass : 0.0;
    double total = m[0] + m[1] + m[2] + m[3];
    double sx = 0.0, sy = 0.0;
    for (int c = 0; c < 4; ++c) {
        if (q->child[c] >= 0) {
            const qnode *ch = &t->nd[q->child[c]];
            sx += ch->cmx * ch->mass;
            sy += ch->cmy * ch->mass;
        }
    }
    q->mass = total;
    q->cmx = sx / total;
    q->cmy = sy / total;
}

typedef struct trav {
    const orc_params *p;
    const orc_body *b;
    const qtree *t;
    double eps2;
    double ax, ay;
    double visits, accepted;
} trav;

/* BarnesHut::calculateForceHelper, barnesHut.cpp:50-89; isCollide barnesHut.cpp:125-134. */
static void qt_force(trav *tv, int ni, int body)
{
    if (ni < 0)
        return;
    const qnode *q = &tv->t->nd[ni];
    const orc_body *bi = &tv->b[body];
    const orc_params *p = tv->p;
    tv->visits += 1;
    if (q->body != -1) {
        const orc_body *bj = &tv->b[q->body];
        if ((bi->radius + bj->radius + p->collision_th > dist2d(bi->px, bi->py, bj->px, bj->py)) || q->body == body)
            return;
        double rx = bj->px - bi->px, ry = bj->py - bi->py;
        double r = sqrt((rx * rx) + (ry * ry) + tv->eps2);
        double s = (p->gravity * bi->mass * bj->mass) / (r * r * r + tv->eps2);
        tv->ax += (rx * s) / bi->mass;
        tv->ay += (ry * s) / bi->mass;
        tv->accepted += 1;
        return;
    }
    double d = dist2d(bi->px, bi->py, q->cmx, q->cmy);
    double sd = (q->brx - q->tlx) / d;
    if (sd < p->theta) {
        double rx = q->cmx - bi->px, ry = q->cmy - bi->py;
        if (!(bi->radius * 2 + p->collision_th > d)) {
            double r = sqrt((rx * rx) + (ry * ry) + tv->eps2);
            double s = (p->gravity * bi->mass * q->mass) / (r * r * r + tv->eps2);
            tv->ax += (rx * s) / bi->mass;
            tv->ay += (ry * s) / bi->mass;
        }
        tv->accepted += 1;
        return;
    }
    qt_force(tv, q->child[0], body);
    qt_force(tv, q->child[1], body);
    qt_force(tv, q->child[2], body);
    qt_force(tv, q->child[3], body);
}

/* Bounding box, tree and centres of mass of one step (barnesHut.cpp:136-140).  order (may be
 * NULL) gives the sequence in which the bodies are inserted: the tree of a one-body-per-leaf
 * quadtree does not depend on it, only the time the insertions take does. */
static int qt_build(qtree *t, const orc_params *p, int n, const orc_body *b, const int *order)
{
    double box[4];
    orc_bh_bbox(p, n, b, box);
    memset(t, 0, sizeof(*t));
    int root = qt_new(t, box[0], box[1], box[2], box[3]);
    for (int i = 0; i < n; ++i)
        qt_insert(t, root, b, order ? order[i] : i, 0); /* constructQuadTree, barnesHut.cpp:37-43 */
    if (t->failed) {
        free(t->nd);
        t->nd = NULL;
        return -1;
    }
    qt_center_mass(t, root, b);
    return root;
}

/* calculateAcceleration, barnesHut.cpp:96-106, for the bodies targets[0..nt) (NULL: body k). */
static void qt_accel_targets(const qtree *t, int root, const orc_params *p, const orc_body *b, int nt, const int *targets,
                             double *acc, double *visits, double *accepted)
{
    double nv = 0.0, na = 0.0;
#pragma omp parallel for schedule(dynamic, 256) reduction(+ : nv, na) num_threads(g_threads) if (g_threads > 1)
    for (int k = 0; k < nt; ++k) {
        const int i = targets ? targets[k] : k;
        if (b[i].is_dynamic) {
            trav tv;
            tv.p = p;
            tv.b = b;
            tv.t = t;
            tv.eps2 = p->eps * p->eps;
            tv.visits = tv.accepted = 0;
            tv.ax = 0.0;
            tv.ay = 0.0;
            qt_force(&tv, root, i);
            acc[2 * k] = tv.ax;
            acc[2 * k + 1] = tv.ay;
            nv += tv.visits;
            na += tv.accepted;
        } else {
            acc[2 * k] = b[i].ax;
            acc[2 * k + 1] = b[i].ay;
        }
    }
    *visits = nv;
    *accepted = na;
}

int orc_bh_accel(const orc_params *p, int n, const orc_body *b, double *acc, double *stats)
{
    return orc_bh_accel_list(p, n, b, NULL, n, NULL, acc, stats);
}

int orc_bh_accel_list(const orc_params *p, int n, const orc_body *b, const int *order, int nt, const int *targets,
                      double *acc, double *stats)
{
    qtree t;
    const int root = qt_build(&t, p, n, b, order);
    if (root < 0)
        return -1;
    double visits, accepted;
    qt_accel_targets(&t, root, p, b, nt, targets, acc, &visits, &accepted);
    if (stats) {
        stats[0] = visits;
        stats[1] = accepted;
        stats[2] = t.max_depth;
        stats[3] = t.count;
    }
    free(t.nd);
    return 0;
}

/* BarnesHut::update, barnesHut.cpp:136-144. */
int orc_bh_step(const orc_params *p, int n, orc_body *b, int nsteps)
{
    double *acc = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    for (int s = 0; s < nsteps; ++s) {
        if (orc_bh_accel(p, n, b, acc, NULL) != 0) {
            free(acc);
            return -1;
        }
        for (int i = 0; i < n; ++i) {
            b[i].ax = acc[2 * i];
            b[i].ay = acc[2 * i + 1];
        }
        integrate(p, n, b);
    }
    free(acc);
    return 0;
}

/* The path a body takes through QuadTree::insert's midpoint splits
 * (quadtree.cpp:41-72 and the child bounds of quadtree.cpp:107-143), 31 levels. */
void orc_morton_keys(int n, const orc_body *b, const double *box, uint64_t *keys)
{
    for (int i = 0; i < n; ++i) {
        double tlx = box[0], tly = box[1], brx = box[2], bry = box[3];
        double x = b[i].px, y = b[i].py;
        uint64_t k = 0;
        for (int l = 0; l < 31; ++l) {
            double mx = (tlx + brx) / 2.0, my = (tly + bry) / 2.0;
            uint64_t right = !(mx >= x);
            uint64_t below = !(my <= y);
            if (right)
                tlx = mx;
            else
                brx = mx;
            if (below)
                tly = my;
            else
                bry = my;
            k = (k << 2) | (below << 1) | right;
        }
        keys[i] = k;
    }
}

/* ---------------------------------------------------------------- the GPU program's tree */

/* ComputeBoundingBoxKernel, barnesHut_kernel.cu:62-111: positions are reduced through
 * fminf/fmaxf (rounded to float), then padded, then reduced through fminf/fmaxf again. */
void orc_bh_gpu_bbox(const orc_params *p, int n, const orc_body *b, double *box)
{
    float mnx = INFINITY, mxx = -INFINITY, mny = INFINITY, mxy = -INFINITY;
    for (int i = 0; i < n; ++i) {
        mnx = fminf(mnx, (float)b[i].px);
        mxx = fmaxf(mxx, (float)b[i].px);
        mny = fminf(mny, (float)b[i].py);
        mxy = fmaxf(mxy, (float)b[i].py);
    }
    box[0] = (float)((double)mnx - p->bbox_margin);
    box[1] = (float)((double)mxy + p->bbox_margin);
    box[2] = (float)((double)mxx + p->bbox_margin);
    box[3] = (float)((double)mny - p->bbox_margin);
}

typedef struct gnode {
    double cmx, cmy, mass;
    int leaf;
    int child[4]; /* in the order ComputeForce visits them: 4i+1 TR, 4i+2 TL, 4i+3 BL, 4i+4 BR; -1 = empty */
} gnode;

typedef struct gtree {
    gnode *nd;
    int count, cap;
    int *scratch;
} gtree;

/* ConstructQuadTreeKernel, :290-360, for the bodies ids[0..cnt) (kept in index order). */
static int gt_build(gtree *t, const orc_body *b, int *ids, int cnt, double tlx, double tly, double brx, double bry,
                    int level, int depth)
{
    if (t->count == t->cap) {
        t->cap = t->cap ? t->cap * 2 : 1024;
        t->nd = (gnode *)realloc(t->nd, sizeof(gnode) * (size_t)t->cap);
    }
    const int ni = t->count++;
    double M = 0.0, rx = 0.0, ry = 0.0;
    for (int k = 0; k < cnt; ++k) {
        const orc_body *q = &b[ids[k]];
        M += q->mass;
        rx += q->mass * q->px;
        ry += q->mass * q->py;
    }
    t->nd[ni].mass = M;
    t->nd[ni].cmx = rx / M;
    t->nd[ni].cmy = ry / M;
    t->nd[ni].leaf = 1;
    for (int c = 0; c < 4; ++c)
        t->nd[ni].child[c] = -1;
    if (level >= depth || cnt == 1)
        return ni;
    t->nd[ni].leaf = 0;
    /* getQuadrant :118-147: 1 TR, 2 TL, 3 BL, 4 BR; stable grouping */
    const double mx = (tlx + brx) / 2, my = (tly + bry) / 2;
    int cnt4[4] = {0, 0, 0, 0};
    int *quad = (int *)malloc(sizeof(int) * (size_t)cnt);
    for (int k = 0; k < cnt; ++k) {
        const orc_body *q = &b[ids[k]];
        int c;
        if (mx >= q->px)
            c = (my <= q->py) ? 1 : 2; /* slot of TL, BL in visiting order */
        else
            c = (my <= q->py) ? 0 : 3; /* TR, BR */
        quad[k] = c;
        cnt4[c]++;
    }
    int start[4], fill[4];
    start[0] = 0;
    for (int c = 1; c < 4; ++c)
        start[c] = start[c - 1] + cnt4[c - 1];
    for (int c = 0; c < 4; ++c)
        fill[c] = start[c];
    int *tmp = (int *)malloc(sizeof(int) * (size_t)cnt);
    for (int k = 0; k < cnt; ++k)
        tmp[fill[quad[k]]++] = ids[k];
    memcpy(ids, tmp, sizeof(int) * (size_t)cnt);
    free(tmp);
    free(quad);
    for (int c = 0; c < 4; ++c) {
        if (cnt4[c] == 0)
            continue;
        double ctlx, ctly, cbrx, cbry; /* UpdateChildBound :149-172 */
        if (c == 0) { ctlx = mx; ctly = tly; cbrx = brx; cbry = my; }
        else if (c == 1) { ctlx = tlx; ctly = tly; cbrx = mx; cbry = my; }
        else if (c == 2) { ctlx = tlx; ctly = my; cbrx = mx; cbry = bry; }
        else { ctlx = mx; ctly = my; cbrx = brx; cbry = bry; }
        const int ch = gt_build(t, b, ids + start[c], cnt4[c], ctlx, ctly, cbrx, cbry, level + 1, depth);
        t->nd[ni].child[c] = ch;
    }
    return ni;
}

/* ComputeForce, :378-423 (acceleration = force / m_i). */
static void gt_force(const orc_params *p, const gtree *t, int ni, const orc_body *bi, double width, double *ax, double *ay)
{
    if (ni < 0)
        return;
    const gnode *q = &t->nd[ni];
    const double dist = dist2d(bi->px, bi->py, q->cmx, q->cmy);
    const int collide = bi->radius * 2 + p->collision_th > dist;
    if (!q->leaf && !(width / dist < p->theta)) {
        for (int c = 0; c < 4; ++c)
            gt_force(p, t, q->child[c], bi, width / 2, ax, ay);
        return;
    }
    if (!collide) {
        const double rx = q->cmx - bi->px, ry = q->cmy - bi->py;
        const double r = sqrt(rx * rx + ry * ry + p->eps * p->eps);
        const double f = (p->gravity * bi->mass * q->mass) / (r * r * r + p->eps * p->eps);
        *ax += rx * f / bi->mass;
        *ay += ry * f / bi->mass;
    }
}

/* The accelerations of the bodies targets[0..nt) (NULL: body k) from the tree of all n bodies. */
void orc_bh_gpu_accel_list(const orc_params *p, int n, const orc_body *b, int depth, int nt, const int *targets, double *acc)
{
    double box[4];
    orc_bh_gpu_bbox(p, n, b, box);
    gtree t = {0};
    int *ids = (int *)malloc(sizeof(int) * (size_t)n);
    for (int i = 0; i < n; ++i)
        ids[i] = i;
    gt_build(&t, b, ids, n, box[0], box[1], box[2], box[3], 0, depth);
    const double width = box[2] - box[0];
#pragma omp parallel for schedule(dynamic, 256) num_threads(g_threads) if (g_threads > 1)
    for (int k = 0; k < nt; ++k) {
        const int i = targets ? targets[k] : k;
        double ax = b[i].ax, ay = b[i].ay;
        if (b[i].is_dynamic) {
            ax = ay = 0.0;
            gt_force(p, &t, 0, &b[i], width, &ax, &ay);
        }
        acc[2 * k] = ax;
        acc[2 * k + 1] = ay;
    }
    free(ids);
    free(t.nd);
}

void orc_bh_gpu_accel(const orc_params *p, int n, const orc_body *b, int depth, double *acc)
{
    orc_bh_gpu_accel_list(p, n, b, depth, n, NULL, acc);
}

void orc_bh_gpu_step(const orc_params *p, int n, orc_body *b, int depth, int nsteps)
{
    double *acc = (double *)malloc(sizeof(double) * 2 * (size_t)n);
    for (int s = 0; s < nsteps; ++s) {
        orc_bh_gpu_accel(p, n, b, depth, acc);
        for (int i = 0; i < n; ++i) {
            if (!b[i].is_dynamic)
                continue;
            b[i].ax = acc[2 * i];
            b[i].ay = acc[2 * i + 1];
            b[i].vx += b[i].ax * p->dt;
            b[i].vy += b[i].ay * p->dt;
            b[i].px += b[i].vx * p->dt;
            b[i].py += b[i].vy * p->dt;
        }
    }
    free(acc);
}
