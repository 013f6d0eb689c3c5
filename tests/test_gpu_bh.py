"""Parity of the HIP Barnes-Hut pipeline (through the C ABI) with the reference, stage by
stage and end to end.

Bars: integer/index stages (bounding box, keys, sort order, tree topology) bit-exact against
CPU twins; per-body acceleration within 1e-4 relative of the reference Barnes-Hut at equal
theta (observed ~1e-6: every body accepts exactly the cells the reference accepts, the
remaining difference is FP32 force rounding); trajectories after 100 steps within 1e-5 of the
system's extent.
"""
import numpy as np
import pytest

from conftest import conditioning, sampled_targets, bodies_from_golden, golden, rel_err

pytestmark = pytest.mark.gpu

ACC_TOL = 1e-4
ACC_EXPECTED = 5e-6  # what identical accept/reject decisions leave; asserted too, so a silent
                     # change of tree semantics cannot hide inside the 1e-4 budget
TRAJ_TOL = 1e-5


def run_accel(nb, b, params=None):
    sim = nb.NBodySim(len(b), nb.BARNES_HUT, params)
    sim.set_bodies(b)
    sim.set_stage_timing(True)
    acc = sim.compute_accel()
    return sim, acc


def exchange_bodies(sim, b):
    """The bodies in the order the library holds them (its keys and indices refer to it)."""
    return b[sim.order()]


def tree_arrays(nb, sim):
    n = sim.n
    cap = max(2 * n + 8, (4 ** 11 - 1) // 3)  # the dense pyramid has a node slot per cell of levels 0..10
    ls = sim.debug(nb.DBG_LEVEL_START, np.uint32, 64)
    rng = sim.debug(nb.DBG_NODE_RANGE, np.uint32, 2 * cap).reshape(-1, 2)
    ch = sim.debug(nb.DBG_NODE_CHILD, np.uint32, 2 * cap).reshape(-1, 2)
    com = sim.debug(nb.DBG_NODE_COM, np.float64, 4 * cap).reshape(-1, 4)
    return ls, rng, ch, com


def walk_tree(rng, ch):
    """Node slots reachable from the root (slot 0), their tree depth, and each one's children.
    A node's children are the slots [first child, first child + child slots) that hold bodies: the
    level-by-level tree packs children, the dense pyramid keeps a slot for each quadrant."""
    nodes, depth, kids = [0], [0], {}
    frontier = np.array([0], dtype=np.int64)
    d = 0
    while len(frontier):
        cf = ch[frontier, 0].astype(np.int64)
        nc = ch[frontier, 1].astype(np.int64)
        cand = cf[:, None] + np.arange(4)[None, :]
        ok = (np.arange(4)[None, :] < nc[:, None])
        cand = np.where(ok, cand, 0)
        ok &= rng[cand, 1] > 0
        for u, row, okr in zip(frontier, cand, ok):
            kids[int(u)] = row[okr]
        frontier = cand[ok]
        d += 1
        nodes.extend(int(v) for v in frontier)
        depth.extend([d] * len(frontier))
    return np.array(nodes), np.array(depth), kids


def cut_keys(nb, sim, ref_keys):
    """The reference path of every body, cut at the level the library's keys stop at (cells of
    that level are never opened, so what lies below them is not sorted or built)."""
    closed, levels = (int(v) for v in sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2))
    assert levels == min(closed, 31)
    return ref_keys & ~np.uint64((1 << (62 - 2 * levels)) - 1), levels


@pytest.mark.parametrize("name", ["spiral_n256", "random_n4096", "collision_n4096"])
def test_bbox_keys_sort_are_bit_exact(nb, oracle, name):
    b = bodies_from_golden(golden(name))
    sim, _ = run_accel(nb, b)
    assert np.array_equal(np.sort(sim.order()), np.arange(len(b)))
    b = exchange_bodies(sim, b)
    box = sim.debug(nb.DBG_BBOX, np.float64, 4)
    assert np.array_equal(box, oracle.bh_bbox(b))
    keys = sim.debug(nb.DBG_KEYS, np.uint64, len(b))
    ref_keys, levels = cut_keys(nb, sim, oracle.tree_keys(b, box))
    assert 6 <= levels <= 12  # the reference's constants: cutoff 1e10 in a ~1e12 box, theta 0.5
    assert np.array_equal(keys, ref_keys)
    order = np.argsort(ref_keys, kind="stable")
    assert np.array_equal(sim.debug(nb.DBG_SORTED_IDX, np.uint32, len(b)), order.astype(np.uint32))
    assert np.array_equal(sim.debug(nb.DBG_SORTED_KEYS, np.uint64, len(b)), ref_keys[order])
    sim.close()


@pytest.mark.parametrize("n", [1 << 16, 300001])
def test_radix_sort_matches_cpu_stable_sort_large(nb, oracle, n):
    b = nb.init_scenario(nb.COLLISION, n, 5)
    sim, _ = run_accel(nb, b)
    b = exchange_bodies(sim, b)
    box = sim.debug(nb.DBG_BBOX, np.float64, 4)
    assert np.array_equal(box, oracle.bh_bbox(b))
    ref_keys, _ = cut_keys(nb, sim, oracle.tree_keys(b, box))
    order = np.argsort(ref_keys, kind="stable")
    assert np.array_equal(sim.debug(nb.DBG_SORTED_IDX, np.uint32, n), order.astype(np.uint32))
    assert np.array_equal(sim.debug(nb.DBG_SORTED_KEYS, np.uint64, n), ref_keys[order])
    sim.close()


def test_sort_is_stable_for_equal_keys(nb):
    """Coincident bodies get equal keys; their order must stay the input order."""
    n = 5000
    b = nb.init_scenario(nb.RANDOM, n, 6)
    b["pos"][100:140] = b["pos"][100]
    b["pos"][3000:3003] = b["pos"][7]
    sim, acc = run_accel(nb, b)
    keys = sim.debug(nb.DBG_KEYS, np.uint64, n)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(sim.debug(nb.DBG_SORTED_IDX, np.uint32, n), order.astype(np.uint32))
    assert np.all(np.isfinite(acc))  # the reference recurses without end here; we must not hang
    sim.close()


@pytest.mark.parametrize("name", ["spiral_n256", "collision_n4096"])
def test_tree_invariants_and_centre_of_mass(nb, name):
    b = bodies_from_golden(golden(name))
    n = len(b)
    sim, _ = run_accel(nb, b)
    b = exchange_bodies(sim, b)
    ls, rng, ch, com = tree_arrays(nb, sim)
    keys = sim.debug(nb.DBG_SORTED_KEYS, np.uint64, n)
    idx = sim.debug(nb.DBG_SORTED_IDX, np.uint32, n)
    box = sim.debug(nb.DBG_BBOX, np.float64, 4)
    assert len(rng) == ls.max()
    assert rng[0, 0] == 0 and rng[0, 1] == n
    nodes, depth, kids = walk_tree(rng, ch)
    assert len(set(nodes.tolist())) == len(nodes) <= 2 * n - 1  # a tree: nobody is reached twice
    leaves = np.array([u for u in nodes if len(kids[int(u)]) == 0])
    closed, levels = (int(v) for v in sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2))
    m = b["mass"][idx]
    p = b["pos"][idx]
    width0 = box[2] - box[0]
    # every body is in exactly one leaf; a leaf holds one body, or the bodies of one cell of the
    # level below which no cell is ever opened (then it carries that cell's width and their
    # centre of mass)
    assert rng[leaves, 1].sum() == n
    covered = np.zeros(n, dtype=int)
    for u in leaves:
        covered[rng[u, 0]:rng[u, 0] + rng[u, 1]] += 1
        if rng[u, 1] > 1:
            first, count = rng[u]
            assert closed <= 31 and com[u, 3] == width0 / 2.0 ** closed
            assert len(np.unique(keys[first:first + count])) == 1
            mm = m[first:first + count]
            assert com[u, 0] == pytest.approx((mm * p[first:first + count, 0]).sum() / mm.sum(), rel=1e-9, abs=1e-3)
    assert np.all(covered == 1)
    if name == "collision_n4096":
        assert np.any(rng[leaves, 1] > 1)  # two discs of 2048 bodies: some share a level-9 cell
    for u in nodes:
        first, count = rng[u]
        cr = rng[kids[int(u)]]
        nc = len(cr)
        if nc == 0:
            continue
        assert 2 <= nc <= 4
        assert np.all(kids[int(u)] > u)  # children live behind their parent
        # children tile the parent's range, in order
        assert cr[0, 0] == first and cr[-1, 0] + cr[-1, 1] == first + count
        assert np.all(cr[1:, 0] == cr[:-1, 0] + cr[:-1, 1])
        # the cell is the smallest one holding the range: keys share `lvl` digits, split at digit lvl
        k = keys[first:first + count]
        lvl = int(round(np.log2(width0 / com[u, 3])))
        shift = np.uint64(62 - 2 * lvl)
        assert len(np.unique(k >> shift)) == 1
        digits = (k >> (shift - np.uint64(2))) & np.uint64(3)
        assert len(np.unique(digits)) == nc
        for c in range(nc):
            kk = keys[cr[c, 0]:cr[c, 0] + cr[c, 1]]
            assert len(np.unique((kk >> (shift - np.uint64(2))) & np.uint64(3))) == 1
        # centre of mass and mass of the range
        mm = m[first:first + count]
        assert com[u, 2] == pytest.approx(mm.sum(), rel=1e-12)
        assert com[u, 0] == pytest.approx((mm * p[first:first + count, 0]).sum() / mm.sum(), rel=1e-9, abs=1e-3)
        assert com[u, 1] == pytest.approx((mm * p[first:first + count, 1]).sum() / mm.sum(), rel=1e-9, abs=1e-3)
    sim.close()


@pytest.mark.parametrize("name", ["solar_n5", "spiral_n256", "random_n256", "collision_n256",
                                  "spiral_n4096", "random_n4096", "collision_n4096"])
def test_accel_matches_reference_golden(nb, name):
    g = golden(name)
    b = bodies_from_golden(g)
    sim, acc = run_accel(nb, b)
    dyn = b["is_dynamic"] != 0
    err = rel_err(acc[dyn], g["acc_bh"][dyn])
    assert err.max() <= ACC_TOL, (name, err.max())
    assert err.max() <= ACC_EXPECTED, (name, err.max())
    sim.close()


@pytest.mark.parametrize("name", ["solar_n5", "spiral_n256", "collision_n256", "spiral_n4096", "collision_n4096"])
def test_100_step_trajectory_matches_reference_golden(nb, name):
    g = golden(name)
    b = bodies_from_golden(g)
    with nb.NBodySim(len(b), nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        sim.update(100)
        out = sim.read_bodies()
    extent = np.linalg.norm(g["pos_bh_100"], axis=1).max()
    dpos = np.linalg.norm(out["pos"] - g["pos_bh_100"], axis=1).max() / extent
    assert dpos <= TRAJ_TOL, (name, dpos)


@pytest.mark.parametrize("name", ["random_n256", "random_n4096"])
def test_random_scenario_trajectory(nb, name):
    """See test_gpu_direct.test_random_scenario_trajectory: chaotic near the cutoff sphere."""
    g = golden(name)
    b = bodies_from_golden(g)
    with nb.NBodySim(len(b), nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        sim.update(100)
        out = sim.read_bodies()
    extent = np.linalg.norm(g["pos_bh_100"], axis=1).max()
    dpos = np.linalg.norm(out["pos"] - g["pos_bh_100"], axis=1) / extent
    assert np.median(dpos) <= TRAJ_TOL
    assert np.mean(dpos <= 1e-3) >= 0.95


@pytest.mark.parametrize("n,scenario", [(65536, 0), (200000, 2), (1000000, 0), (2000000, 2)])
def test_accel_matches_oracle_at_scale(nb, oracle, n, scenario):
    """configs: spiral N = 1M theta = 0.5; collision; against the oracle's reference-exact tree."""
    b = nb.init_scenario(scenario, n)
    ref, st = oracle.bh_accel(b, with_stats=True)
    sim, acc = run_accel(nb, b)
    dyn = b["is_dynamic"] != 0
    err = rel_err(acc[dyn], ref[dyn])
    assert err.max() <= ACC_TOL, err.max()
    assert err.max() <= ACC_EXPECTED, err.max()
    cnt = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
    # accepted cells + bodies per target never exceed the reference's (the reference also counts
    # the cells it accepts and then drops inside the cutoff; chains are collapsed here)
    assert cnt[1] <= st[1] + n
    sim.close()


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 64, 65, 100, 257, 1000, 4097])
def test_ragged_sizes(nb, oracle, n):
    b = oracle.scenario(0, n, 31)
    sim, acc = run_accel(nb, b)
    ref = oracle.bh_accel(b)
    dyn = b["is_dynamic"] != 0
    if dyn.any():
        assert rel_err(acc[dyn], ref[dyn]).max() <= ACC_EXPECTED
    sim.update(3)
    out = sim.read_bodies()
    ref3 = oracle.bh_step(b, 3)
    assert np.linalg.norm(out["pos"] - ref3["pos"], axis=1).max() <= 1e-7 * 5e11
    sim.close()


@pytest.mark.parametrize("theta", [0.0, 0.3, 1.0])
def test_theta_is_honoured(nb, oracle, theta):
    n = 2000
    b = oracle.scenario(0, n, 17)
    sim, acc = run_accel(nb, b, nb.default_params(theta=theta))
    ref = oracle.bh_accel(b, oracle.params(theta=theta))
    dyn = b["is_dynamic"] != 0
    assert rel_err(acc[dyn], ref[dyn]).max() <= ACC_EXPECTED
    if theta == 0.0:
        # nothing is ever accepted: Barnes-Hut degenerates to the direct sum (with > in place of >=)
        assert rel_err(acc[dyn], oracle.direct_accel(b)[dyn]).max() <= 1e-5
    sim.close()


def test_decisions_at_the_thresholds_follow_the_reference(nb, oracle):
    """Bodies placed within float rounding of the pair cutoff on both sides."""
    rng = np.random.default_rng(9)
    n = 600
    b = oracle.scenario(1, n, 4)
    th = 2 * 12756 + 1.0e10
    for k in range(0, n - 2, 2):
        ang = rng.uniform(0, 2 * np.pi)
        d = th * (1 + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-12, -6))
        b["pos"][k + 1] = b["pos"][k] + d * np.array([np.cos(ang), np.sin(ang)])
    sim, acc = run_accel(nb, b)
    ref = oracle.bh_accel(b)
    dyn = b["is_dynamic"] != 0
    assert rel_err(acc[dyn], ref[dyn]).max() <= ACC_EXPECTED
    assert sim.debug(nb.DBG_COUNTERS, np.uint64, 8)[4] > 0  # the FP64 re-decision path ran
    sim.close()


def test_softened_force_without_cutoff(nb, oracle):
    n = 900
    b = oracle.scenario(0, n, 11)
    b["radius"] = 0.0
    kw = dict(collision_th=0.0, eps=3.0e9)
    sim, acc = run_accel(nb, b, nb.default_params(**kw))
    ref = oracle.bh_accel(b, oracle.params(**kw))
    dyn = b["is_dynamic"] != 0
    assert rel_err(acc[dyn], ref[dyn]).max() <= ACC_EXPECTED
    sim.close()


def test_bh_close_to_direct_sum(nb, oracle):
    """Sanity of the approximation itself: theta = 0.5 is within 2e-2 of the direct sum."""
    n = 4096
    b = oracle.scenario(0, n, 1)
    sim, acc = run_accel(nb, b)
    dyn = b["is_dynamic"] != 0
    assert np.median(rel_err(acc[dyn], oracle.direct_accel(b)[dyn])) < 2e-3
    sim.close()


def test_two_rank_partition_equals_single_rank(nb, oracle):
    """Each of two handles walks the tree only for its half of the bodies; positions are
    exchanged through the exchange buffers as an all-gather would.  Every body's trajectory
    must match the unpartitioned run to FP32 summation-order noise (the 64-body groups differ)."""
    import ctypes as C

    n = 5003
    b = oracle.scenario(0, n, 8)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    with nb.NBodySim(n, nb.BARNES_HUT) as one, nb.NBodySim(n, nb.BARNES_HUT) as r0, nb.NBodySim(n, nb.BARNES_HUT) as r1:
        for s in (one, r0, r1):
            s.set_bodies(b)
        r0.set_partition(0, 2)
        r1.set_partition(1, 2)
        for _ in range(5):
            one.update()
            for s in (r0, r1):
                s.step_local()
            for s in (r0, r1):
                s.sync()
            p0, chunk, rb = r0.exchange_buffer()
            p1, _, _ = r1.exchange_buffer()
            assert hip.hipMemcpy(p1, p0, chunk * rb, 3) == 0
            assert hip.hipMemcpy(p0 + chunk * rb, p1 + chunk * rb, (n - chunk) * rb, 3) == 0
            assert hip.hipDeviceSynchronize() == 0  # the handles' streams do not wait for the null stream
            for s in (r0, r1):
                s.step_finish()
        ref = one.read_bodies()
        a = r0.read_bodies()
        c = r1.read_bodies()
    assert np.array_equal(a["pos"], c["pos"])
    assert np.linalg.norm(a["pos"] - ref["pos"], axis=1).max() <= 1e-9 * 5e11


def test_tree_deeper_than_the_previous_steps(nb, oracle):
    """The per-level build kernels are launched for as many levels as earlier trees had; a tree
    that suddenly needs more levels (here: bodies on a coarse grid, then a sequence converging on
    a point) must still be built completely and give the reference's accelerations.  With the
    reference's constants the tree stops at level 8-9 and is built as a dense pyramid, which has no
    such launches; a small opening angle (0.05) moves that level beyond what the pyramid covers."""
    n = 65
    theta = 0.05
    wide = oracle.scenario(0, n, 12)
    g = np.arange(n - 1)
    wide["pos"][:-1, 0] = -4.0e11 + (g % 8) * 1.1e11 + 1.0e9 * np.sin(g)       # 8 x 8 grid, jittered
    wide["pos"][:-1, 1] = -4.0e11 + (g // 8) * 1.1e11 + 1.0e9 * np.cos(3 * g)
    with nb.NBodySim(n, nb.BARNES_HUT, nb.default_params(theta=theta)) as sim:
        sim.set_bodies(wide)
        sim.set_stage_timing(True)
        sim.update(3)
        assert int(sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2)[1]) > 10  # the level-by-level build
        shallow = int(sim.debug(nb.DBG_COUNTERS, np.uint64, 8)[3])
        wide = sim.read_bodies()
        tight = wide.copy()
        # a geometric sequence of bodies converging on a point: every cell level splits one body
        # off (a clump would only add a collapsed chain), down to the level below which cells are
        # never opened and the tree is not built
        k = np.arange(26)
        box = sim.debug(nb.DBG_BBOX, np.float64, 4)
        corner = np.array([box[0] + 1.0e10, box[1] - 1.0e10])
        tight["pos"][30:56] = corner + (box[2] - box[0]) * 0.5 ** (k[:, None] + 1) * np.array([1.0, -1.0])
        sim.set_bodies(tight)
        acc = sim.compute_accel()
        deep = int(sim.debug(nb.DBG_COUNTERS, np.uint64, 8)[3])
        assert deep >= shallow + 4
        ref = oracle.bh_accel(tight, oracle.params(theta=theta))
        dyn = tight["is_dynamic"] != 0
        assert rel_err(acc[dyn], ref[dyn]).max() <= ACC_EXPECTED
        # and back to a shallow tree
        sim.set_bodies(wide)
        acc = sim.compute_accel()
        assert rel_err(acc[dyn], oracle.bh_accel(wide, oracle.params(theta=theta))[dyn]).max() <= ACC_EXPECTED


@pytest.mark.parametrize("stretch,level", [(1.6, 10), (0.6, 8), (0.3, 7)])
def test_pyramid_levels_other_than_nine(nb, oracle, stretch, level):
    """The reference's scenarios close the tree at level 9; a wider or narrower system moves that
    level (10 is the deepest the dense pyramid covers: its top kernel then stages one level more
    than it holds).  A clump of bodies inside one closed cell makes a leaf that spans many 64-body
    chunks, whose sums the pyramid kernel adds up wave by wave."""
    n = 6000
    b = oracle.scenario(0, n, 9)
    b["pos"] *= stretch
    rng_ = np.random.default_rng(3)
    b["pos"][100:2100] = b["pos"][100] + rng_.uniform(-1.0e8, 1.0e8, size=(2000, 2))
    sim, acc = run_accel(nb, b)
    closed, levels = (int(v) for v in sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2))
    sim.close()
    assert closed == levels == level
    ref = oracle.bh_accel(b)
    dyn = b["is_dynamic"] != 0
    # (2e-5, not the usual 5e-6: bodies just beyond the cutoff of the 2000-body clump feel it through
    # float coordinates that resolve 1e-5 of that distance in the narrowest of these systems)
    assert rel_err(acc[dyn], ref[dyn]).max() <= 2.0e-5


def test_keys_deeper_than_the_pyramid_fall_back(nb, oracle):
    """The dense-pyramid kernels are launched when the previous steps' keys stopped at a level
    they cover.  One body leaving at 1.6e8 m/s makes the bounding box five times wider within a step,
    the level below which no cell is opened moves from 9 to beyond 10, and the step whose kernels
    were already chosen must still build the complete tree (one block does); later steps use the
    level-by-level build.  Every step's accelerations are the reference's (bar 1e-4: the float
    coordinates now resolve a box five times the system's size)."""
    n = 700
    b = oracle.scenario(0, n, 31)
    b["vel"][5] = (1.6e8, 0.8e8)
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        depths = []
        for step in range(4):
            state = sim.read_bodies()
            acc = sim.compute_accel()  # same kernels, same launch decisions as a step
            depths.append(int(sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2)[1]))
            ref = oracle.bh_accel(state)
            dyn = state["is_dynamic"] != 0
            assert rel_err(acc[dyn], ref[dyn]).max() <= ACC_TOL, step
            sim.update(1)
    assert depths[0] <= 10 < depths[1], depths


@pytest.mark.parametrize("case", ["tiny_system", "wide_and_clump", "theta_small", "theta_large", "no_cutoff", "huge_cutoff"])
def test_closed_level_edge_cases(nb, oracle, case):
    """The level below which the tree is not built depends on box, cutoff and theta; whatever it
    comes out as (none, the root itself, anything between), the accelerations are the full tree's."""
    n = 600
    b = oracle.scenario(0, n, 77)
    kw = {}
    if case == "tiny_system":
        b["pos"] *= 1.0e-3  # everything within a few 1e8 m: all inside each other's cutoff
        kw = dict(bbox_margin=1.0e6)
    elif case == "wide_and_clump":
        b["pos"][:300] = b["pos"][0] + 1.0e-4 * (b["pos"][:300] - b["pos"][0])  # half the bodies in one small cell
    elif case == "theta_small":
        kw = dict(theta=0.05)
    elif case == "theta_large":
        kw = dict(theta=1.5)
    elif case == "no_cutoff":
        kw = dict(collision_th=0.0)
    elif case == "huge_cutoff":
        kw = dict(collision_th=3.0e11)
    with nb.NBodySim(n, nb.BARNES_HUT, nb.default_params(**kw)) as sim:
        sim.set_bodies(b)
        acc = sim.compute_accel()
        closed, levels = (int(v) for v in sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2))
    ref = oracle.bh_accel(b, oracle.params(**kw))
    dyn = b["is_dynamic"] != 0
    if case == "no_cutoff":
        assert closed == 32 and levels == 31
        tol = 1.0e-4  # float coordinates resolve 6e4 m of a 1e12 m box; the cutoff is what keeps pairs apart
    else:
        tol = ACC_EXPECTED
    if case in ("tiny_system", "huge_cutoff"):
        assert closed <= (2 if case == "tiny_system" else 4)
    scale = np.abs(ref[dyn]).max()
    assert np.all(np.abs(acc[dyn] - ref[dyn]) <= tol * np.maximum(np.linalg.norm(ref[dyn], axis=1, keepdims=True), 1e-12 * scale) + 1e-30)


def test_full_size_collision_matches_the_reference_tree(nb, oracle):
    """The headline workload itself (galaxy collision, N = 5,000,000, theta = 0.5): 512 sampled
    bodies against the oracle's reference-exact tree of all 5 M bodies (built once, walked for the
    sampled bodies only; calculateForceHelper, src/cpu/barnesHut.cpp:50-89), at the same bars as
    every smaller case: 1e-4, and the 5e-6 that identical accept/reject decisions leave.  Also: the
    result is finite and reproducible and the tree stops at the level the constants imply."""
    n = 5000000
    b = nb.init_scenario(nb.COLLISION, n, 12345)
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        acc = sim.compute_accel()
        closed, levels = (int(v) for v in sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2))
        again = sim.compute_accel()
        sim.update(2)
        out = sim.read_bodies()
    assert np.all(np.isfinite(acc)) and np.array_equal(acc, again)
    assert 7 <= closed <= 10 and levels == closed
    assert np.all(np.isfinite(out["pos"])) and np.all(np.isfinite(out["vel"]))
    targets = sampled_targets(b, 512, 5)
    ref = oracle.bh_accel_list(b, targets)
    err = rel_err(acc[targets], ref)
    assert err.max() <= ACC_TOL, err.max()
    assert err.max() <= ACC_EXPECTED, err.max()
    # two steps on: the state the library integrated, judged by the oracle's tree of that state
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(out)
        acc2 = sim.compute_accel()
    ref2 = oracle.bh_accel_list(out, targets)
    err2 = rel_err(acc2[targets], ref2)
    assert err2.max() <= ACC_EXPECTED, err2.max()


@pytest.mark.timeout(600)
def test_late_states_of_the_collision_run_match_the_reference_tree(nb, oracle, capsys):
    """The headline workload far into its run, where the benchmark times it again: after 300 and
    after 620 steps of nb_step the discs have collapsed into dense cores (single cells of the
    closed level hold thousands of bodies), ejected bodies have stretched the box so that the
    tree closes two to three levels deeper than at the start, and the device has renewed the
    bodies' slots have followed the tree's order for hundreds of steps.  At both states the accelerations of 256
    sampled bodies (half of them the ones nearest the two suns) are compared with the oracle's
    reference-exact tree of the state the library holds (calculateForceHelper,
    src/cpu/barnesHut.cpp:50-89), bar 1e-4; the dense pyramid must have built both trees."""
    n = 5000000
    b = nb.init_scenario(nb.COLLISION, n, 12345)
    seen = []
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        done = 0
        for upto in (300, 620):
            sim.update(upto - done)
            done = upto
            state = sim.read_bodies()
            acc = sim.compute_accel()
            closed, levels = (int(v) for v in sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2))
            cnt = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
            seen.append((upto, state, acc, closed, levels, int(cnt[2])))
    for upto, state, acc, closed, levels, nodes in seen:
        assert np.all(np.isfinite(acc)) and np.all(np.isfinite(state["pos"]))
        targets = sampled_targets(state, 256, upto)
        ref, stats = oracle.bh_accel_list(state, targets, with_stats=True)
        err = rel_err(acc[targets], ref)
        with capsys.disabled():
            print("\n[collision 5M, step %d: keys of %d levels (closed %d), %d nodes, oracle depth %d, max err %.2e]"
                  % (upto, levels, closed, nodes, stats[2], err.max()), flush=True)
        assert err.max() <= ACC_TOL, (upto, err.max())
        # identical decisions leave float rounding only: ACC_EXPECTED, or for a body of a dense core what
        # the conditioning of its sum allows (measured: err / conditioning = 0.3e-7 .. 4.4e-7, for the
        # bodies with the largest errors as for the median ones)
        cth = nb.default_params().collision_th
        for i in np.flatnonzero(err > ACC_EXPECTED):
            c = conditioning(state, int(targets[i]), cth)
            assert err[i] <= 1.5e-6 * c, (upto, int(targets[i]), err[i], c)
        # every cell of levels 0..levels is a node slot: the dense pyramid built this tree
        assert nodes == (4 ** (levels + 1) - 1) // 3, (upto, levels, nodes)
        if upto == 300:
            assert 9 <= levels <= 11, levels
        else:
            assert closed == 11 and levels == 11, (closed, levels)


@pytest.mark.timeout(600)
def test_spiral_50m_matches_the_reference_tree(nb, oracle, capsys):
    """The largest configuration (spiral galaxy, N = 50,000,000, one GPU's share of the multi-GPU
    case): 256 sampled bodies against the oracle's tree of all 50 M bodies (about 12 GB of host
    memory and a minute to build), same bars.  Properties that need no oracle: finite, the keys
    stop at the closed level, every body belongs to exactly one leaf."""
    n = 50000000
    b = nb.init_scenario(nb.SPIRAL, n, 12345)
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        sim.set_stage_timing(True)
        acc = sim.compute_accel()
        closed, levels = (int(v) for v in sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2))
        cnt = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
        ls, rng_, ch, _ = tree_arrays(nb, sim)
    with capsys.disabled():  # a sign of life: the oracle's tree below takes about a minute
        print("\n[50M: GPU done, closed level %d, %d nodes, %d levels; building the oracle tree]"
              % (closed, cnt[2], cnt[3]), flush=True)
    assert np.all(np.isfinite(acc))
    assert 7 <= closed <= 10 and levels == closed
    nodes, _, kids = walk_tree(rng_, ch)
    leaves = np.array([u for u in nodes if len(kids[int(u)]) == 0])
    assert int(rng_[leaves, 1].astype(np.int64).sum()) == n
    targets = sampled_targets(b, 256, 11)
    ref = oracle.bh_accel_list(b, targets)
    err = rel_err(acc[targets], ref)
    assert err.max() <= ACC_TOL, err.max()
    assert err.max() <= ACC_EXPECTED, err.max()


def test_pyramid_down_to_level_11(nb, oracle):
    """A system large enough that its node slots hold the cells of level 11 anyway (N >= 2.8 M)
    builds the dense pyramid down to that level: one body far outside stretches the box of a
    3 M-body spiral 3.7 times, the tree closes at level 11 instead of 9, and the pyramid kernels
    (with the middle kernel that takes levels 6..3) must give the reference's tree: every body in
    exactly one leaf, sampled bodies at the oracle tree's accelerations (bar 1e-4: the float
    coordinates resolve a box 3.7 times the system's size), also two steps on."""
    n = 3000000
    b = nb.init_scenario(nb.SPIRAL, n, 4242)
    b["pos"][7] = (3.2e12, 1.0e11)
    b["vel"][7] = (0.0, 0.0)
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        acc = sim.compute_accel()
        closed, levels = (int(v) for v in sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2))
        cnt = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
        ls, rng_, ch, _ = tree_arrays(nb, sim)
        sim.update(2)
        mid = sim.read_bodies()
        acc2 = sim.compute_accel()
        levels2 = int(sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2)[1])
        cnt2 = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
    assert closed == 11 and levels == 11 and levels2 == 11, (closed, levels, levels2)
    # every cell of levels 0..11 is a node slot: the dense pyramid built this tree, both times
    assert int(cnt[2]) == (4 ** 12 - 1) // 3 and int(cnt2[2]) == (4 ** 12 - 1) // 3, (cnt[2], cnt2[2])
    assert np.all(np.isfinite(acc)) and np.all(np.isfinite(acc2))
    nodes, _, kids = walk_tree(rng_, ch)
    leaves = np.array([u for u in nodes if len(kids[int(u)]) == 0])
    assert int(rng_[leaves, 1].astype(np.int64).sum()) == n
    targets = sampled_targets(b, 256, 3)
    err = rel_err(acc[targets], oracle.bh_accel_list(b, targets))
    assert err.max() <= ACC_TOL, err.max()
    err2 = rel_err(acc2[targets], oracle.bh_accel_list(mid, targets))
    assert err2.max() <= ACC_TOL, err2.max()


@pytest.mark.parametrize("dims", [2, 3])
def test_deep_clustered_tree_stresses_the_traversal_stack(nb, oracle, dims):
    """No cutoff (so no level is closed), a small opening angle and a clump that shrinks geometrically
    towards the origin (where float coordinates still resolve it): deep levels, many opened cells,
    long stacks.  The stack must stay inside its
    allocation (the kernel's bound is per construction; the counter shows the depth reached) and
    every body must still get the oracle's acceleration."""
    rng = np.random.default_rng(12)
    n = 5000
    pos = rng.uniform(-4.0e11, 4.0e11, size=(n, 3))
    k = np.arange(n // 2)
    scale = 4.0e11 * 2.0 ** (-(k % 14).astype(np.float64))  # the 3-D keys resolve 20 levels: keep bodies apart
    pos[: n // 2] = scale[:, None] * rng.uniform(0.2, 1.0, size=(n // 2, 3))
    theta, eps = 0.1, 1.0e6
    if dims == 2:
        from oracle_lib import make_bodies

        b = make_bodies(n)
        b["pos"] = pos[:, :2]
    else:
        from oracle_lib import BODY3_DTYPE

        b = np.zeros(n, dtype=BODY3_DTYPE)
        b["pos"] = pos
    b["is_dynamic"] = 1
    b["mass"] = rng.uniform(1.0e24, 1.0e25, size=n)
    b["radius"] = 1.0e4
    po = oracle.params(theta=theta, collision_th=0.0, eps=eps)
    pn = nb.default_params(theta=theta, collision_th=0.0, eps=eps)
    ref = oracle.bh_accel(b, po) if dims == 2 else oracle.bh_accel3(b, po)
    Sim = nb.NBodySim if dims == 2 else nb.NBodySim3
    with Sim(n, nb.BARNES_HUT, pn) as s:
        s.set_bodies(b)
        s.set_stage_timing(False, counters=True)
        acc = s.compute_accel()
        c = s.debug(nb.DBG_COUNTERS, np.uint64, 8)
    assert c[3] >= 12  # tree levels
    assert 40 < c[7] < 200  # deepest stack, entries (allocated: 225 in 2-D, 260 in 3-D)
    # without a cutoff a body's nearest neighbours pull in all directions with terms far larger
    # than their sum: float rounding of the terms shows in the sum of a few bodies
    err = rel_err(acc, ref)
    assert np.median(err) <= 5e-6 and np.quantile(err, 0.99) <= 1e-4 and err.max() <= 2e-3, (
        np.quantile(err, [0.5, 0.99, 0.999]), err.max())


def test_leaf_sums_of_runs_that_cross_chunk_boundaries(nb, oracle):
    """Bodies that share a cell nobody opens share a leaf; its mass, centre of mass and largest radius
    are summed by waves of 64 sorted bodies, and a leaf of more than 64 bodies is put together from the
    parts of the chunks it crosses.  Clumps of many sizes (1 .. 5000 bodies, so runs start and end at
    every position of a chunk) against numpy sums, and the accelerations against the oracle's full tree."""
    rng_ = np.random.default_rng(5)
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 191, 193, 640, 1000, 5000, 4097, 3]
    n = sum(sizes) + 1
    b = oracle.scenario(0, n, 21)
    at = 0
    for j, sz in enumerate(sizes):
        # clump j: inside one cell of 1/1024 of the box (the tree closes at level 8 or 9)
        centre = np.array([(-4.0 + 0.55 * j) * 1.0e11 + 3.1e8, (-3.0 + 0.41 * j) * 1.0e11 + 1.7e8])
        b["pos"][at:at + sz] = centre + rng_.uniform(-2.0e8, 2.0e8, size=(sz, 2))
        b["radius"][at:at + sz] = rng_.uniform(1.0e3, 5.0e4, size=sz)
        b["mass"][at:at + sz] *= rng_.uniform(0.5, 2.0, size=sz)
        at += sz
    sim, acc = run_accel(nb, b)
    bx = exchange_bodies(sim, b)
    ls, rng, ch, com = tree_arrays(nb, sim)
    idx = sim.debug(nb.DBG_SORTED_IDX, np.uint32, n)
    closed, levels = (int(v) for v in sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2))
    assert closed <= 10
    m, p = bx["mass"][idx], bx["pos"][idx]
    leaves = np.flatnonzero((ch[:, 1] == 0) & (rng[:, 1] > 1))
    seen = set()
    for u in leaves:
        first, count = (int(v) for v in rng[u])
        mm = m[first:first + count]
        assert com[u, 2] == pytest.approx(mm.sum(), rel=1e-13)
        assert com[u, 0] == pytest.approx((mm * p[first:first + count, 0]).sum() / mm.sum(), rel=1e-12, abs=1e-3)
        assert com[u, 1] == pytest.approx((mm * p[first:first + count, 1]).sum() / mm.sum(), rel=1e-12, abs=1e-3)
        seen.add(count)
    # (a clump that straddles a cell boundary makes two leaves, so only the overall picture is fixed)
    assert max(seen) >= 2000 and sum(c > 64 for c in seen) >= 6 and sum(c <= 64 for c in seen) >= 1
    sim.close()
    ref = oracle.bh_accel(b)
    dyn = b["is_dynamic"] != 0
    assert rel_err(acc[dyn], ref[dyn]).max() <= ACC_EXPECTED


@pytest.mark.parametrize("dims", [2, 3])
def test_long_tree_run_keeps_its_slots_in_the_order_of_the_tree(nb, oracle, dims):
    """The per-step sort's permutation drifts away from the identity as bodies mix, and the leaf pass and
    the traversal gather through it.  2-D: every step of a one-GPU run leaves the bodies in the order of
    its own sort (the traversal writes body k of the sorted order to slot k of a second set of arrays);
    3-D: nb3_step renews the order every 256 steps.  300 steps through nb_step against 300 split steps,
    which leave every body in its slot: the same run up to FP32 summation order, and the slot -> body
    map has changed on the way."""
    n = 3000
    if dims == 2:
        b = oracle.scenario(0, n, 8)
        make = lambda: nb.NBodySim(n, nb.BARNES_HUT)  # noqa: E731
    else:
        b = oracle.scenario3(0, n, 8, thickness=4.0e10)
        make = lambda: nb.NBodySim3(n, nb.BARNES_HUT)  # noqa: E731
    with make() as auto, make() as plain:
        auto.set_bodies(b)
        plain.set_bodies(b)
        first = auto.order().copy()
        auto.update(300)
        for _ in range(300):
            plain.step_local()
            plain.step_finish()
        got, ref = auto.read_bodies(), plain.read_bodies()
        assert not np.array_equal(first, auto.order())
        assert np.array_equal(first, plain.order())
    # (the 64-body groups of the traversal differ between the two: FP32 summation order)
    assert np.linalg.norm(got["pos"] - ref["pos"], axis=1).max() <= 1e-6 * np.abs(ref["pos"]).max()
    assert np.array_equal(got["mass"], ref["mass"]) and np.array_equal(got["radius"], ref["radius"])
    assert np.array_equal(got["is_dynamic"], ref["is_dynamic"])


@pytest.mark.parametrize("speed", [2.5e10, 1.0e12])
def test_sort_is_finished_when_the_keys_outgrow_the_launched_passes(nb, oracle, speed):
    """The host launches only the sort passes the keys of the last step it knows would need with a
    box a tenth larger.  A body leaving at an absurd speed widens the box hundreds (or tens of thousands) of
    times within one step: the keys of the next evaluation need passes that were not launched, and
    the device finishes the order itself.  Keys and sorted order must be the reference's to the bit
    (the accelerations cannot be checked at the usual bar: float coordinates no longer resolve the
    system inside such a box)."""
    n = 20000
    b = nb.init_scenario(nb.SPIRAL, n, 9)
    b["vel"][11] = (speed, -0.4 * speed)
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        sim.compute_accel()
        before = int(sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2)[1])
        sim.update(1)  # launched for the depth above; moves the body away
        for _ in range(2):  # the first evaluation is still launched for the old depth, the second for the new
            sim.compute_accel()
            state = exchange_bodies(sim, sim.read_bodies())
            box = sim.debug(nb.DBG_BBOX, np.float64, 4)
            assert np.array_equal(box, oracle.bh_bbox(state))
            ref_keys, levels = cut_keys(nb, sim, oracle.tree_keys(state, box))
            order = np.argsort(ref_keys, kind="stable")
            assert np.array_equal(sim.debug(nb.DBG_SORTED_KEYS, np.uint64, n), ref_keys[order])
            assert np.array_equal(sim.debug(nb.DBG_SORTED_IDX, np.uint32, n), order.astype(np.uint32))
    assert before <= 10 and levels >= before + 8, (before, levels)


@pytest.mark.parametrize("dims", [2, 3])
def test_every_group_of_targets_reports_what_its_wave_took(nb, dims):
    """NB_DBG_WAVE_COST: one entry per group of 64 consecutive targets of the last traversal, in ticks of 10 ns; a wave
    that walked the tree took at least a microsecond and far less than a second."""
    n = 10000 + 37
    if dims == 2:
        sim = nb.NBodySim(n, nb.BARNES_HUT)
        b = nb.init_scenario(nb.COLLISION, n, 5)
    else:
        sim = nb.NBodySim3(n, nb.BARNES_HUT)
        b = nb.init_scenario3(nb.COLLISION, n, 5, 2.0e10)
    with sim:
        sim.set_bodies(b)
        sim.update(2)
        groups = (n + 63) // 64
        cost = sim.debug(nb.DBG_WAVE_COST, np.uint32, groups)
        assert cost.shape == (groups,)
        assert cost.min() >= 100 and cost.max() < 100_000_000
        # a rank of a partitioned run reports the groups of its own targets
        sim.set_bodies(b)
        sim.set_partition(1, 3)
        sim.compute_accel()
        own = (n + 2) // 3
        cost = sim.debug(nb.DBG_WAVE_COST, np.uint32, (own + 63) // 64)
        assert cost.min() >= 100 and cost.max() < 100_000_000
