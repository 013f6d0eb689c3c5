"""Parity of the three-dimensional HIP paths (the nb3_ entry points, through the C ABI).

The reference is two-dimensional, so the checks are of two kinds:
  * flat input (all z = 0): the 3-D paths must agree with the reference's own golden
    accelerations (tests/golden, made by the compiled reference) to the 2-D bars;
  * z != 0: the 3-D paths against the CPU oracle's 3-D restatement (oracle/nbody_oracle3.c,
    which is pinned to the reference only through the flat case -- parity unpinned beyond it).
Bars as in 2-D: direct sum 1e-5, Barnes-Hut 1e-4 relative per body; FP64 integration.
"""
import numpy as np
import pytest

from conftest import bodies_from_golden, golden, rel_err, sampled_targets
from oracle_lib import lift3

pytestmark = pytest.mark.gpu

DS_TOL = 1e-5
BH_TOL = 1e-4
TRAJ_TOL = 1e-6


def gpu_accel3(nb, b, alg, params=None):
    with nb.NBodySim3(len(b), alg, params) as sim:
        sim.set_bodies(b)
        return sim.compute_accel()


@pytest.mark.parametrize("name", ["solar_n5", "spiral_n256", "random_n256", "collision_n256", "spiral_n4096",
                                  "random_n4096", "collision_n4096"])
def test_flat_input_matches_the_reference_golden(nb, name):
    g = golden(name)
    b3 = lift3(bodies_from_golden(g))
    for alg, key, tol in ((nb.DIRECT, "acc_direct", DS_TOL), (nb.BARNES_HUT, "acc_bh", BH_TOL)):
        acc = gpu_accel3(nb, b3, alg)
        assert not acc[:, 2].any()
        err = rel_err(acc[:, :2], g[key])
        assert err.max() <= tol, (name, alg, err.max())


@pytest.mark.parametrize("sim,n,thick", [(0, 3000, 5.0e10), (1, 4096, 2.0e11), (2, 5000, 2.0e10), (1, 777, 5.0e11),
                                         (3, 5, 3.0e10)])
def test_direct3_matches_the_oracle(nb, oracle, sim, n, thick):
    b = oracle.scenario3(sim, n, 4242, thickness=thick)
    assert np.array_equal(nb.init_scenario3(sim, n, 4242, thick)["pos"], b["pos"])
    err = rel_err(gpu_accel3(nb, b, nb.DIRECT), oracle.direct_accel3(b))
    assert err.max() <= DS_TOL, err.max()


@pytest.mark.parametrize("sim,n,thick", [(0, 3000, 5.0e10), (1, 4096, 2.0e11), (2, 5000, 2.0e10), (1, 777, 5.0e11),
                                         (3, 5, 3.0e10), (2, 20000, 1.0e11)])
def test_bh3_matches_the_oracle(nb, oracle, sim, n, thick):
    b = oracle.scenario3(sim, n, 99, thickness=thick)
    ref, stats = oracle.bh_accel3(b, with_stats=True)
    with nb.NBodySim3(len(b), nb.BARNES_HUT) as s:
        s.set_bodies(b)
        s.set_stage_timing(False, counters=True)
        acc = s.compute_accel()
        counters = s.debug(nb.DBG_COUNTERS, np.uint64, 8)
    err = rel_err(acc, ref)
    assert err.max() <= BH_TOL, err.max()
    # every body accepts exactly the cells the oracle accepts for it: same number of
    # contributing interactions is not observable (pruned cells count in the oracle), but the
    # traversal must at least have evaluated something for every dynamic body
    assert counters[1] > 0 and counters[2] > 0


@pytest.mark.parametrize("theta", [0.3, 0.8, 1.2])
def test_bh3_other_opening_angles(nb, oracle, theta):
    b = oracle.scenario3(0, 2500, 5, thickness=8.0e10)
    ref = oracle.bh_accel3(b, oracle.params(theta=theta))
    err = rel_err(gpu_accel3(nb, b, nb.BARNES_HUT, nb.default_params(theta=theta)), ref)
    assert err.max() <= BH_TOL, (theta, err.max())


def test_bh3_without_cutoff_builds_the_deep_tree(nb, oracle):
    """collision_th = 0: no closed level, one body per leaf down to wherever the bodies separate."""
    b = oracle.scenario3(1, 3000, 8, thickness=3.0e11)
    po, pn = oracle.params(collision_th=0.0), nb.default_params(collision_th=0.0)
    ref = oracle.bh_accel3(b, po)
    with nb.NBodySim3(len(b), nb.BARNES_HUT, pn) as s:
        s.set_bodies(b)
        acc = s.compute_accel()
        cut = s.debug(nb.DBG_KEY_LEVELS, np.uint32, 2)
        child = s.debug(nb.DBG_NODE_CHILD, np.uint32, 4 * len(b) + 16).reshape(-1, 2)
        rng = s.debug(nb.DBG_NODE_RANGE, np.uint32, 4 * len(b) + 16).reshape(-1, 2)
    assert cut[0] == 32 and cut[1] == 20
    leaves = child[:, 1] == 0
    assert rng[leaves, 1].sum() == len(b) and (rng[leaves, 1] == 1).all()
    assert child[:, 1].max() <= 8 and (child[~leaves, 1] >= 2).all()
    assert rel_err(acc, ref).max() <= BH_TOL


def test_octree_invariants_and_keys(nb, oracle):
    b = oracle.scenario3(2, 6000, 17, thickness=6.0e10)
    with nb.NBodySim3(len(b), nb.BARNES_HUT) as s:
        s.set_bodies(b)
        s.compute_accel()
        box = s.debug(nb.DBG_BBOX, np.float64, 6)
        keys = s.debug(nb.DBG_SORTED_KEYS, np.uint64, len(b))
        idx = s.debug(nb.DBG_SORTED_IDX, np.uint32, len(b))
        order = s.order()
        cut = s.debug(nb.DBG_KEY_LEVELS, np.uint32, 2)
        rng = s.debug(nb.DBG_NODE_RANGE, np.uint32, 4 * len(b) + 16).reshape(-1, 2)
        child = s.debug(nb.DBG_NODE_CHILD, np.uint32, 4 * len(b) + 16).reshape(-1, 2)
        com = s.debug(nb.DBG_NODE_COM, np.float64, 6 * (2 * len(b) + 8)).reshape(-1, 6)
    assert np.array_equal(box, oracle.bh_bbox3(b))
    # keys: the oracle's octant paths, cut to the levels the device kept, sorted stably
    levels = int(cut[1])
    full = oracle.tree_keys3(b, box)[order]
    keep = np.uint64(((1 << (3 * levels)) - 1) << (62 - 3 * levels)) if levels else np.uint64(0)
    expect = full & keep
    perm = np.argsort(expect, kind="stable")
    assert np.array_equal(keys, expect[perm]) and np.array_equal(idx, perm.astype(np.uint32))
    # tree: the root holds everything, children partition their parent in order, masses add up
    n_nodes = len(rng)
    assert rng[0, 0] == 0 and rng[0, 1] == len(b)
    mass = b["mass"][order][idx]
    pos = b["pos"][order][idx]
    for u in range(n_nodes):
        f, c = int(rng[u, 0]), int(rng[u, 1])
        cf, nc = int(child[u, 0]), int(child[u, 1])
        if nc:
            assert rng[cf, 0] == f and rng[cf:cf + nc, 1].sum() == c
            assert (rng[cf + 1:cf + nc, 0] == rng[cf:cf + nc - 1, 0] + rng[cf:cf + nc - 1, 1]).all()
    for u in list(range(0, min(n_nodes, 200))) + list(range(max(0, n_nodes - 200), n_nodes)):
        f, c = int(rng[u, 0]), int(rng[u, 1])
        m = mass[f:f + c].sum()
        assert com[u, 3] == pytest.approx(m, rel=1e-12)
        assert np.allclose(com[u, :3], (pos[f:f + c] * mass[f:f + c, None]).sum(0) / m, rtol=1e-9, atol=1e-3)


@pytest.mark.parametrize("alg", [0, 1])
def test_trajectories_follow_the_oracle(nb, oracle, alg):
    b = oracle.scenario3(0, 1500, 21, thickness=4.0e10)
    steps = 100
    ref = (oracle.direct_step3 if alg == 0 else oracle.bh_step3)(b, steps)
    with nb.NBodySim3(len(b), alg) as s:
        s.set_bodies(b)
        s.update(steps)
        out = s.read_bodies()
    dyn = b["is_dynamic"] != 0
    # a body that passes within the cutoff of another may take the other side of a knife-edge
    # decision after many steps; the bulk must follow closely
    d = np.linalg.norm(out["pos"] - ref["pos"], axis=1) / 1.0e12
    assert np.median(d[dyn]) < TRAJ_TOL and np.quantile(d[dyn], 0.99) < 1e-4
    assert np.array_equal(out["pos"][~dyn], ref["pos"][~dyn])


def test_leapfrog3_against_the_oracle_forces(nb, oracle):
    """Kick-drift-kick built from the oracle's accelerations, 20 steps."""
    b = oracle.scenario3(0, 600, 2, thickness=5.0e10)
    dt = 25000.0
    cur = b.copy()
    acc = oracle.direct_accel3(cur)
    for _ in range(20):
        cur["vel"] += 0.5 * dt * acc
        cur["pos"] += dt * cur["vel"]
        acc = oracle.direct_accel3(cur)
        cur["vel"] += 0.5 * dt * acc
    with nb.NBodySim3(len(b), nb.DIRECT, nb.default_params(integrator=nb.LEAPFROG)) as s:
        s.set_bodies(b)
        s.update(20)
        out = s.read_bodies()
    # FP32 forces (1e-5 relative) over 20 steps move a body near the sun by a few km at most
    assert np.abs(out["pos"] - cur["pos"]).max() / 1.0e12 < 2e-8
    assert np.abs(out["vel"] - cur["vel"]).max() / np.abs(cur["vel"]).max() < 1e-6


def test_diagnostics3_and_leapfrog_energy(nb, oracle):
    """Conserved quantities against numpy sums; kick-drift-kick keeps the energy of a tilted solar
    system far better than the reference's Euler step does."""
    b = oracle.scenario3(3, 5, 1, thickness=0.0)
    # tilt two orbits out of the plane
    b["pos"][1] = [0.0, 0.0, 2.2790e11]
    b["vel"][1] = [0.0, 2.4100e4, 0.0]
    b["vel"][3] = [0.0, 3.5000e4 * np.cos(0.5), 3.5000e4 * np.sin(0.5)]
    m = b["mass"]

    def energy(x):
        k = 0.5 * (m * (x["vel"] ** 2).sum(1)).sum()
        u = 0.0
        for i in range(5):
            for j in range(i + 1, 5):
                d = np.linalg.norm(x["pos"][i] - x["pos"][j])
                if not (x["radius"][i] + x["radius"][j] + 1.0e10 >= d):
                    u -= 6.67e-11 * m[i] * m[j] / np.sqrt(d * d + 0.25)
        return k, u

    drift = {}
    for integ in (nb.EULER, nb.LEAPFROG):
        with nb.NBodySim3(5, nb.DIRECT, nb.default_params(integrator=integ)) as s:
            s.set_bodies(b)
            d0 = s.diagnostics()
            k0, u0 = energy(b)
            assert d0["kinetic"] == pytest.approx(k0, rel=1e-12) and d0["potential"] == pytest.approx(u0, rel=1e-12)
            assert d0["mass"] == pytest.approx(m.sum(), rel=1e-15)
            lz = (m[:, None] * np.cross(b["pos"], b["vel"])).sum(0)
            assert np.allclose(d0["angular_momentum"], lz, rtol=1e-12)
            s.update(1500)
            d1 = s.diagnostics()
            out = s.read_bodies()
            k1, u1 = energy(out)
            assert d1["kinetic"] == pytest.approx(k1, rel=1e-9) and d1["potential"] == pytest.approx(u1, rel=1e-9)
            drift[integ] = abs((d1["kinetic"] + d1["potential"]) - (d0["kinetic"] + d0["potential"])) / abs(k0 + u0)
    assert drift[nb.LEAPFROG] < drift[nb.EULER] / 5 and drift[nb.LEAPFROG] < 1e-3


def test_rotating_the_system_rotates_the_answer(nb, oracle):
    """A disc in the x-y plane and the same disc stood up in the y-z plane."""
    b = oracle.scenario3(0, 4000, 31, thickness=1.0e10)
    r = b.copy()
    for f in ("pos", "vel"):
        r[f][:, 0] = -b[f][:, 2]
        r[f][:, 2] = b[f][:, 0]
    for alg, tol in ((nb.DIRECT, 2 * DS_TOL), (nb.BARNES_HUT, None)):
        a = gpu_accel3(nb, b, alg)
        ar = gpu_accel3(nb, r, alg)
        back = np.stack([ar[:, 2], ar[:, 1], -ar[:, 0]], axis=1)
        if tol is not None:
            assert rel_err(back, a).max() <= tol
        else:
            # the octree's cells differ between the two orientations (the opening test uses the
            # x-width): each must match its own oracle, and the two agree like two theta = 0.5 trees
            assert rel_err(ar, oracle.bh_accel3(r)).max() <= BH_TOL
            assert np.median(rel_err(back, a)) < 2e-2


def test_large_direct3_sampled_and_newton(nb, oracle):
    n = 100000
    b = oracle.scenario3(2, n, 1, thickness=5.0e10)
    acc = gpu_accel3(nb, b, nb.DIRECT)
    rng = np.random.default_rng(0)
    pick = np.sort(rng.choice(n, 40, replace=False))
    for i in pick:
        ref = oracle.direct_accel3(b, t0=int(i), t1=int(i) + 1)[0]
        assert np.linalg.norm(acc[i] - ref) <= DS_TOL * np.linalg.norm(ref)
    # Newton's third law: with every body dynamic the forces cancel
    f = (acc * b["mass"][:, None]).sum(0)
    scale = (np.linalg.norm(acc, axis=1) * b["mass"]).sum()
    assert np.linalg.norm(f) / scale < 1e-6


def test_large_bh3_against_direct3(nb, oracle):
    n = 300000
    b = oracle.scenario3(2, n, 3, thickness=5.0e10)
    a_bh = gpu_accel3(nb, b, nb.BARNES_HUT)
    a_ds = gpu_accel3(nb, b, nb.DIRECT)
    err = rel_err(a_bh, a_ds)
    assert np.median(err) < 2e-2 and np.quantile(err, 0.999) < 0.2
    with nb.NBodySim3(n, nb.BARNES_HUT, nb.default_params(theta=0.2)) as s:
        s.set_bodies(b)
        tight = s.compute_accel()
    assert np.median(rel_err(tight, a_ds)) < np.median(err) / 3


@pytest.mark.timeout(600)
def test_full_size_collision3_matches_the_oracle_octree(nb, oracle, capsys):
    """The 3-D workload the benchmark times (collision discs, N = 5,000,000, half-thickness 2e10 m):
    256 sampled bodies (half of them the ones nearest the two suns) against the oracle's octree of
    all 5 M bodies, built once and walked for the sampled bodies only; bar 1e-4."""
    n = 5000000
    b = nb.init_scenario3(nb.COLLISION, n, 12345, thickness=2.0e10)
    acc = gpu_accel3(nb, b, nb.BARNES_HUT)
    assert np.all(np.isfinite(acc))
    targets = sampled_targets(b, 256, 17)
    ref, stats = oracle.bh_accel_list3(b, targets, with_stats=True)
    err = rel_err(acc[targets], ref)
    with capsys.disabled():
        print("\n[3-D collision 5M: oracle octree of %d nodes, depth %d, %.0f accepted per body, max err %.2e]"
              % (stats[3], stats[2], stats[1] / len(targets), err.max()), flush=True)
    assert err.max() <= BH_TOL, err.max()
    assert err.max() <= 5.0e-6, err.max()  # identical decisions leave float rounding only


@pytest.mark.parametrize("alg", [0, 1])
def test_two_rank_partition_equals_single_rank(nb, oracle, alg):
    """Two handles own half the targets each; positions travel through the exchange buffers as an
    all-gather would move them (records of three doubles)."""
    import ctypes as C

    n = 3001
    b = oracle.scenario3(0, n, 8, thickness=5.0e10)
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    with nb.NBodySim3(n, alg) as one, nb.NBodySim3(n, alg) as r0, nb.NBodySim3(n, alg) as r1:
        for s in (one, r0, r1):
            s.set_bodies(b)
        r0.set_partition(0, 2)
        r1.set_partition(1, 2)
        with pytest.raises(nb.NBodyError, match="partitioned"):
            r0.update()
        for _ in range(5):
            one.update()
            for s in (r0, r1):
                s.step_local()
            for s in (r0, r1):
                s.sync()
            p0, chunk, rb = r0.exchange_buffer()
            p1, _, _ = r1.exchange_buffer()
            assert chunk == (n + 1) // 2 and rb == 24
            assert hip.hipMemcpy(p1, p0, chunk * rb, 3) == 0
            assert hip.hipMemcpy(p0 + chunk * rb, p1 + chunk * rb, (n - chunk) * rb, 3) == 0
            # a device-to-device hipMemcpy may return before the copy is done, and the handles'
            # streams do not wait for the null stream
            assert hip.hipDeviceSynchronize() == 0
            for s in (r0, r1):
                s.step_finish()
        ref = one.read_bodies()
        a = r0.read_bodies()
        c = r1.read_bodies()
        order = r0.order()
    assert np.array_equal(a["pos"], c["pos"])
    # Barnes-Hut: the same tree and the same per-body decisions on every rank, only the 64-body
    # groups differ (FP32 summation order); direct sum likewise
    assert np.linalg.norm(a["pos"] - ref["pos"], axis=1).max() <= 1e-9 * 5e11
    own0, own1 = order[:chunk], order[chunk:]  # each rank keeps velocities of its own slots only
    vmax = np.linalg.norm(ref["vel"], axis=1).max()
    assert np.linalg.norm(a["vel"][own0] - ref["vel"][own0], axis=1).max() <= 1e-6 * vmax
    assert np.linalg.norm(c["vel"][own1] - ref["vel"][own1], axis=1).max() <= 1e-6 * vmax


def test_argument_checks(nb):
    with pytest.raises(nb.NBodyError):
        nb.NBodySim3(0, nb.DIRECT)
    with pytest.raises(nb.NBodyError):
        nb.NBodySim3(10, nb.BARNES_HUT, nb.default_params(bucket_depth=9))
    with nb.NBodySim3(8, nb.DIRECT) as s:
        with pytest.raises(nb.NBodyError):
            s.update()
        with pytest.raises(nb.NBodyError):
            s.set_bodies(np.zeros(7, dtype=nb.BODY3_DTYPE))


@pytest.mark.parametrize("dims", [2, 3])
def test_a_system_far_from_the_coordinate_origin_keeps_its_precision(nb, oracle, dims):
    """Float coordinates are taken relative to the middle of the bodies' bounding box: moving the
    whole system by 60 times its size must not cost accuracy (the oracle works in FP64)."""
    n = 3000
    shift = np.array([3.0e13, -2.0e13, 1.0e13])
    if dims == 2:
        b = oracle.scenario(0, n, 6)
        b["pos"] += shift[:2]
        refs = (oracle.direct_accel(b), oracle.bh_accel(b))
        Sim = nb.NBodySim
    else:
        b = oracle.scenario3(0, n, 6, thickness=5.0e10)
        b["pos"] += shift
        refs = (oracle.direct_accel3(b), oracle.bh_accel3(b))
        Sim = nb.NBodySim3
    for alg, ref, tol in ((nb.DIRECT, refs[0], DS_TOL), (nb.BARNES_HUT, refs[1], BH_TOL)):
        with Sim(n, alg) as s:
            s.set_bodies(b)
            acc = s.compute_accel()
            assert rel_err(acc, ref).max() <= tol, (dims, alg, rel_err(acc, ref).max())
            # and after a few steps (the records of moved bodies are refreshed relative to the same origin)
            s.update(3)
            out = s.read_bodies()
        step = {(2, 0): oracle.direct_step, (2, 1): oracle.bh_step, (3, 0): oracle.direct_step3, (3, 1): oracle.bh_step3}[(dims, alg)]
        want = step(b, 3)
        assert np.abs(out["pos"] - want["pos"]).max() <= 1e-7 * 5e11


@pytest.mark.parametrize("alg", [0, 1])
@pytest.mark.parametrize("n", [1, 2, 3, 65, 129])
def test_tiny_systems(nb, oracle, alg, n):
    rng = np.random.default_rng(n)
    b = np.zeros(n, dtype=nb.BODY3_DTYPE)
    b["is_dynamic"] = 1
    b["mass"] = rng.uniform(1e24, 1e26, n)
    b["radius"] = 1.0e4
    b["pos"] = rng.uniform(-3e11, 3e11, (n, 3))
    b["vel"] = rng.uniform(-1e3, 1e3, (n, 3))
    ref = oracle.direct_accel3(b) if alg == 0 else oracle.bh_accel3(b)
    with nb.NBodySim3(n, alg) as s:
        s.set_bodies(b)
        acc = s.compute_accel()
        s.update(2)
        out = s.read_bodies()
    assert rel_err(acc, ref).max() <= (DS_TOL if alg == 0 else BH_TOL)
    want = (oracle.direct_step3 if alg == 0 else oracle.bh_step3)(b, 2)
    assert np.abs(out["pos"] - want["pos"]).max() <= 1e-9 * 5e11


@pytest.mark.parametrize("alg", [0, 1])
def test_coincident_and_static_bodies(nb, oracle, alg):
    """Two bodies at one point (the reference's tree would recurse without end; they are inside each
    other's cutoff, so they exert nothing on each other) and a static body that must not move."""
    b = oracle.scenario3(1, 400, 13, thickness=1.0e11)
    b["pos"][7] = b["pos"][3]
    b["pos"][11] = b["pos"][3]
    b["is_dynamic"][20] = 0
    b["acc"][20] = [1.0e-9, -2.0e-9, 3.0e-9]  # a static body keeps the acceleration it was given ...
    b["vel"][20] = [5.0, 0.0, -5.0]           # ... and moves with its own velocity
    ref = oracle.direct_accel3(b)             # the pairs inside the cutoff drop out of both sums
    with nb.NBodySim3(len(b), alg) as s:
        s.set_bodies(b)
        acc = s.compute_accel()
        s.update(1)
        out = s.read_bodies()
    dyn = b["is_dynamic"] != 0
    tol = DS_TOL if alg == 0 else 2e-2  # Barnes-Hut against the direct sum: the tree's own error
    assert rel_err(acc[dyn], ref[dyn]).max() <= tol
    if alg == 0:
        assert not out["acc"][20].any()       # DirectSum::calculateAcceleration zeroes static bodies
    else:
        assert np.array_equal(out["acc"][20], b["acc"][20])
        assert np.allclose(out["vel"][20], b["vel"][20] + 25000.0 * b["acc"][20], rtol=1e-15)
    assert np.all(np.isfinite(out["pos"]))


@pytest.mark.parametrize("dims", [2, 3])
def test_a_drifting_system_keeps_its_precision_in_the_tree_path(nb, oracle, dims):
    """Barnes-Hut takes its float coordinates relative to the middle of each step's bounding box: a
    system that moves 20 times its own size away from where it was set loses nothing.  (The direct
    sum keeps the origin of the last nb_set_bodies / nb_refresh_order.)"""
    n = 2500
    bulk = np.array([4.0e6, -3.0e6, 2.0e6])  # m/s: 1e11 m per step of 25 000 s
    if dims == 2:
        b = oracle.scenario(0, n, 16)
        b["vel"] += bulk[:2]
        Sim, accel = nb.NBodySim, oracle.bh_accel
    else:
        b = oracle.scenario3(0, n, 16, thickness=5.0e10)
        b["vel"] += bulk
        Sim, accel = nb.NBodySim3, oracle.bh_accel3
    b["is_dynamic"][-1] = 1  # the sun travels with the rest
    with Sim(n, nb.BARNES_HUT) as s:
        s.set_bodies(b)
        s.update(100)
        out = s.read_bodies()
        acc = s.compute_accel()
    assert np.abs(out["pos"]).max() > 8.0e12
    at = out.copy()
    ref = accel(at)
    assert rel_err(acc, ref).max() <= BH_TOL, rel_err(acc, ref).max()


@pytest.mark.parametrize("alg", ["direct", "bh"])
def test_refresh_order_changes_nothing_the_caller_sees_3d(nb, oracle, alg):
    """nb3_refresh_order re-sorts the device state by the current positions, on the device; the run
    goes on as before up to the rounding of a different summation order."""
    n = 3000
    b0 = oracle.scenario3(0, n, 23, thickness=5.0e10)
    algc = nb.DIRECT if alg == "direct" else nb.BARNES_HUT
    with nb.NBodySim3(n, algc) as a:
        a.set_bodies(b0)
        a.update(40)
        ra = a.read_bodies()
    with nb.NBodySim3(n, algc) as c:
        c.set_bodies(b0)
        c.update(20)
        before = c.order().copy()
        mid = c.read_bodies()
        c.refresh_order()
        after = c.order()
        assert np.array_equal(np.sort(after), np.arange(n)) and not np.array_equal(before, after)
        again = c.read_bodies()
        for f in ("pos", "vel", "acc", "mass", "radius", "is_dynamic"):
            assert np.array_equal(mid[f], again[f])
        c.update(20)
        rc = c.read_bodies()
    scale = np.abs(ra["pos"]).max()
    assert np.abs(ra["pos"] - rc["pos"]).max() / scale < 1e-7
    assert np.abs(ra["vel"] - rc["vel"]).max() / np.abs(ra["vel"]).max() < 1e-5
