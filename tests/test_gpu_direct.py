"""Parity of the HIP direct-sum path (through the C ABI) with the reference.

Bar: per-body acceleration within 1e-5 relative of the reference direct sum
(|a - a_ref| / |a_ref|, FP32 kernel vs FP64 reference), every body, no exclusions: a float
cutoff comparison that is too close to call is re-decided in FP64 by the kernel, so the
set of skipped pairs equals the reference's.  Trajectories after 100 steps: positions
within 1e-6 of the system's extent for the orbiting scenarios (state is integrated in FP64;
only the FP32 force rounding accumulates).
"""
import ctypes as C

import numpy as np
import pytest

from conftest import bodies_from_golden, golden, rel_err

pytestmark = pytest.mark.gpu

ACC_TOL = 1e-5
TRAJ_TOL = 1e-6


def gpu_accel(nb, b, alg, params=None):
    with nb.NBodySim(len(b), alg, params) as sim:
        sim.set_bodies(b)
        return sim.compute_accel()


def test_solar_system_known_answers(nb):
    """The values the reference prints for its 5-body scenario, 1 and 100 steps."""
    with nb.NBody(5, nb.DIRECT, 2) as sim:
        sim.update()
        s1 = sim.read_bodies()
        assert s1["pos"][0, 0] == pytest.approx(149596294976.13806, rel=1e-9)
        assert s1["pos"][0, 1] == 745000000.0
        assert s1["vel"][0, 0] == pytest.approx(-148.20095447794509, rel=1e-6)
        assert s1["acc"][0, 0] == pytest.approx(-0.005928038179117804, rel=1e-6)
        assert s1["acc"][2, 0] == pytest.approx(-0.039573234139035962, rel=1e-6)
        assert np.all(s1["pos"][4] == 0) and np.all(s1["vel"][4] == 0) and np.all(s1["acc"][4] == 0)
        sim.update(99)
        s100 = sim.read_bodies()
    expect = np.array(
        [
            [131256301024.06088, 71458106727.478882],
            [219882076684.75839, 59548709506.070251],
            [-28808843896.008045, 49776650490.252937],
            [74258735237.716614, 78218074445.492767],
        ]
    )
    assert np.max(np.linalg.norm(s100["pos"][:4] - expect, axis=1)) / 2.3e11 < TRAJ_TOL


@pytest.mark.parametrize("name", ["solar_n5", "spiral_n256", "random_n256", "collision_n256",
                                  "spiral_n4096", "random_n4096", "collision_n4096"])
def test_accel_matches_reference_golden(nb, name):
    g = golden(name)
    b = bodies_from_golden(g)
    acc = gpu_accel(nb, b, nb.DIRECT)
    err = rel_err(acc, g["acc_direct"])
    assert err.max() <= ACC_TOL, (name, err.max())
    # static bodies: exactly zero, as DirectSum::calculateAcceleration leaves them
    assert np.all(acc[b["is_dynamic"] == 0] == 0)


@pytest.mark.parametrize("name", ["solar_n5", "spiral_n256", "collision_n256", "spiral_n4096", "collision_n4096"])
def test_100_step_trajectory_matches_reference_golden(nb, name):
    g = golden(name)
    b = bodies_from_golden(g)
    with nb.NBodySim(len(b), nb.DIRECT) as sim:
        sim.set_bodies(b)
        sim.update(100)
        out = sim.read_bodies()
    extent = np.linalg.norm(g["pos_direct_100"], axis=1).max()
    dpos = np.linalg.norm(out["pos"] - g["pos_direct_100"], axis=1).max() / extent
    assert dpos <= TRAJ_TOL * (10 if "collision" in name else 1), (name, dpos)
    vscale = np.linalg.norm(g["vel_direct_100"], axis=1).max()
    dvel = np.linalg.norm(out["vel"] - g["vel_direct_100"], axis=1).max() / vscale
    # tight orbits (63 steps per revolution at the inner edge) turn a position lag into a
    # velocity-direction error: v dx / r, an order above the position error
    assert dvel <= 1e-4, (name, dvel)


@pytest.mark.parametrize("name", ["random_n256", "random_n4096"])
def test_random_scenario_trajectory(nb, name):
    """Bodies start at rest and fall through the sun's cutoff sphere, where the force switches
    off discontinuously: trajectories are chaotic there.  Most bodies must still follow the
    reference closely; the few that crossed a cutoff boundary a step apart may not."""
    g = golden(name)
    b = bodies_from_golden(g)
    with nb.NBodySim(len(b), nb.DIRECT) as sim:
        sim.set_bodies(b)
        sim.update(100)
        out = sim.read_bodies()
    extent = np.linalg.norm(g["pos_direct_100"], axis=1).max()
    dpos = np.linalg.norm(out["pos"] - g["pos_direct_100"], axis=1) / extent
    assert np.median(dpos) <= TRAJ_TOL
    assert np.mean(dpos <= 1e-4) >= 0.97


@pytest.mark.parametrize("n", [1, 2, 3, 5, 63, 64, 65, 255, 256, 257, 511, 513, 1023, 1025, 2049, 3000])
def test_ragged_sizes(nb, oracle, n):
    b = oracle.scenario(1, n, 99)
    if n > 1:
        b["vel"][: n - 1] = np.random.default_rng(n).normal(0, 1e4, (n - 1, 2))
    acc = gpu_accel(nb, b, nb.DIRECT)
    ref = oracle.direct_accel(b)
    assert rel_err(acc, ref).max() <= ACC_TOL
    with nb.NBodySim(n, nb.DIRECT) as sim:
        sim.set_bodies(b)
        sim.update(3)
        out = sim.read_bodies()
    ref3 = oracle.direct_step(b, 3)
    assert np.linalg.norm(out["pos"] - ref3["pos"], axis=1).max() <= 1e-7 * 5e11


def test_large_n_sampled_targets(nb, oracle):
    """config: spiral galaxy N = 65,536, FP32 direct sum; 2048 sampled targets against the oracle."""
    n = 65536
    b = nb.init_scenario(nb.SPIRAL, n)
    acc = gpu_accel(nb, b, nb.DIRECT)
    targets = np.sort(np.random.default_rng(7).choice(n, 2048, replace=False)).astype(np.int32)
    ref = oracle.direct_accel_list(b, targets)
    err = rel_err(acc[targets], ref)
    assert err.max() <= ACC_TOL, err.max()


def test_pairs_at_the_cutoff_follow_the_reference(nb, oracle):
    """Pairs placed within float rounding of the cutoff distance r_i + r_j + 1e10 on both
    sides: the skipped set must equal the FP64 reference's (directSum.cpp:69-72, note >=)."""
    rng = np.random.default_rng(5)
    n = 512
    b = oracle.scenario(1, n, 4)
    th = 2 * 12756 + 1.0e10
    for k in range(0, n - 2, 2):
        ang = rng.uniform(0, 2 * np.pi)
        d = th * (1 + rng.choice([-1, 1]) * 10.0 ** rng.uniform(-12, -6))
        b["pos"][k + 1] = b["pos"][k] + d * np.array([np.cos(ang), np.sin(ang)])
    acc = gpu_accel(nb, b, nb.DIRECT)
    ref = oracle.direct_accel(b)
    assert rel_err(acc, ref).max() <= ACC_TOL


@pytest.mark.parametrize("dims", [2, 3])
def test_pairs_at_a_tiny_cutoff_across_run_boundaries(nb, oracle, dims):
    """A cutoff of 1e-5 of the system's extent (1e7 m), 2000 isolated pairs placed 7e2 .. 7e5 m
    (1e-8 .. 1e-5 internal units: below and above the float rounding of the coordinates) either side
    of it.  A pair lies next to itself in the exchange order, so some pairs straddle the boundary of
    two runs of 256 sources, where the box of the one body's wave and the box of the other's run are
    exactly the pair's distance apart: the shortcut that skips the cutoff test for far runs must
    not take such a run, and the near runs' FP64 re-decision must reproduce the reference's skipped
    set (directSum.cpp:69-72).  A pair on the wrong side changes its bodies' accelerations by orders
    of magnitude (the pair is 100 times closer than any other body)."""
    rng = np.random.default_rng(8)
    n = 4001
    cth = 1.0e7
    if dims == 2:
        b = oracle.scenario(1, n, 6)
    else:
        b = oracle.scenario3(1, n, 6, thickness=2.0e11)
    th = 2 * 12756 + cth
    for k in range(0, n - 2, 2):
        u = rng.normal(size=dims)
        u /= np.linalg.norm(u)
        d = th + rng.choice([-1, 1]) * 10.0 ** rng.uniform(2.85, 5.85)
        b["pos"][k + 1] = b["pos"][k] + d * u
    po = oracle.params(collision_th=cth)
    pn = nb.default_params(collision_th=cth)
    if dims == 2:
        acc = gpu_accel(nb, b, nb.DIRECT, pn)
        ref = oracle.direct_accel(b, po)
    else:
        with nb.NBodySim3(n, nb.DIRECT, pn) as sim:
            sim.set_bodies(b)
            acc = sim.compute_accel()
        ref = oracle.direct_accel3(b, po)
    err = rel_err(acc, ref)
    # the pairs just outside the cutoff dominate their bodies' sums and sit 1e-5 of the extent apart:
    # their float distance carries ~1e-4 relative rounding unless the low parts are used, which they are
    assert err.max() <= ACC_TOL, err.max()


def test_softened_force_without_cutoff(nb, oracle):
    """collision_th = 0 with a large epsilon exercises the exact r^3 + eps^2 denominator."""
    n = 700
    b = oracle.scenario(0, n, 11)
    b["radius"] = 0.0
    kw = dict(collision_th=0.0, eps=3.0e9)
    acc = gpu_accel(nb, b, nb.DIRECT, nb.default_params(**kw))
    ref = oracle.direct_accel(b, oracle.params(**kw))
    assert rel_err(acc, ref).max() <= ACC_TOL


def test_parameters_are_honoured(nb, oracle):
    n = 300
    b = oracle.scenario(0, n, 21)
    kw = dict(gravity=1.0e-10, dt=1000.0, collision_th=3.0e10)
    with nb.NBodySim(n, nb.DIRECT, nb.default_params(**kw)) as sim:
        sim.set_bodies(b)
        sim.update(5)
        out = sim.read_bodies()
    ref = oracle.direct_step(b, 5, oracle.params(**kw))
    assert rel_err(out["acc"], ref["acc"]).max() <= ACC_TOL
    assert np.linalg.norm(out["pos"] - ref["pos"], axis=1).max() <= 1e-9 * 5e11


def test_static_body_with_velocity_moves_unaccelerated(nb, oracle):
    b = oracle.scenario(0, 64, 2)
    b["vel"][63] = (1.0e3, -2.0e3)  # the static sun drifts: calculateVelocity/Position treat all bodies
    with nb.NBodySim(64, nb.DIRECT) as sim:
        sim.set_bodies(b)
        sim.update(4)
        out = sim.read_bodies()
    ref = oracle.direct_step(b, 4)
    assert np.allclose(out["pos"][63], ref["pos"][63], rtol=0, atol=1e-3)
    assert np.all(out["vel"][63] == b["vel"][63])


def _hip():
    return C.CDLL("libamdhip64.so")


@pytest.mark.parametrize("cth", [1.0e10, 1.0e9])
def test_two_rank_partition_equals_single_rank(nb, oracle, cth):
    """Two handles own half the targets each; positions are exchanged through the exchange
    buffers as an all-gather would.  The result must equal the unpartitioned run up to FP32
    summation order (blocks hold different targets, so their near/far run choices differ).
    With the small cutoff the records carry the coordinates' low parts, which a rank must also
    refresh for the bodies the other rank advanced."""
    n = 3001
    b = oracle.scenario(0, n, 8)
    hip = _hip()
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    pr = nb.default_params(collision_th=cth)
    with nb.NBodySim(n, nb.DIRECT, pr) as one, nb.NBodySim(n, nb.DIRECT, pr) as r0, nb.NBodySim(n, nb.DIRECT, pr) as r1:
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
            assert chunk == (n + 1) // 2 and rb == 16
            # all-gather: slice k of every rank <- rank k's slice
            assert hip.hipMemcpy(p1, p0, chunk * rb, 3) == 0
            n1 = n - chunk
            assert hip.hipMemcpy(p0 + chunk * rb, p1 + chunk * rb, n1 * rb, 3) == 0
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
    assert np.linalg.norm(a["pos"] - ref["pos"], axis=1).max() <= 1e-9 * 5e11
    own0, own1 = order[:chunk], order[chunk:]  # each rank keeps velocities of its own slots only
    assert np.allclose(a["vel"][own0], ref["vel"][own0], rtol=1e-6, atol=1e-6)
    assert np.allclose(c["vel"][own1], ref["vel"][own1], rtol=1e-6, atol=1e-6)


def test_call_order_errors(nb):
    with nb.NBodySim(16, nb.DIRECT) as sim:
        with pytest.raises(nb.NBodyError, match="no bodies"):
            sim.update()
        sim.setup(nb.RANDOM, 1)
        with pytest.raises(nb.NBodyError, match="nb_step_local was not called"):
            sim.step_finish()
        with pytest.raises(nb.NBodyError):
            sim.set_bodies(nb.make_bodies(15))
        sim.set_partition(0, 2)
        with pytest.raises(nb.NBodyError, match="partitioned"):
            sim.update()


@pytest.mark.parametrize("alg", ["direct", "bh"])
def test_refresh_order_changes_nothing_the_caller_sees(nb, alg):
    """nb_refresh_order re-sorts the device state by the current positions; the run goes on as
    before up to the rounding of a different summation order."""
    b0 = bodies_from_golden(golden("spiral_n4096"))
    a = nb.NBodySim(len(b0), nb.DIRECT if alg == "direct" else nb.BARNES_HUT)
    a.set_bodies(b0)
    a.update(40)
    ra = a.read_bodies()
    a.close()
    c = nb.NBodySim(len(b0), nb.DIRECT if alg == "direct" else nb.BARNES_HUT)
    c.set_bodies(b0)
    c.update(20)
    before = c.order().copy()
    mid = c.read_bodies()
    c.refresh_order()
    after = c.order()
    assert np.array_equal(np.sort(after), np.arange(len(b0))) and not np.array_equal(before, after)
    again = c.read_bodies()
    for f in ("pos", "vel", "acc", "mass", "radius", "is_dynamic"):
        assert np.array_equal(mid[f], again[f])
    c.update(20)
    rc = c.read_bodies()
    c.close()
    # FP32 partial sums in another order: accelerations differ by ~1e-7 relative, which 20 steps
    # of dt = 25000 s turn into ~1e-8 of the system's extent for the innermost bodies
    scale = np.abs(ra["pos"]).max()
    assert np.abs(ra["pos"] - rc["pos"]).max() / scale < 1e-7
    assert np.abs(ra["vel"] - rc["vel"]).max() / np.abs(ra["vel"]).max() < 1e-5


def test_long_direct_run_refreshes_by_itself(nb, oracle):
    """A large system renews its order every 32 steps; for 64 bodies the interval is 4096, so many
    steps of a small system it is: 4200 steps of 64 bodies stay on the oracle's path."""
    b0 = oracle.scenario(0, 64, 3)
    sim = nb.NBodySim(64, nb.DIRECT)
    sim.set_bodies(b0)
    first = sim.order().copy()
    sim.update(4200)
    got = sim.read_bodies()
    assert not np.array_equal(first, sim.order())  # bodies have orbited: the order was renewed
    sim.close()
    want = oracle.direct_step(b0, 4200)
    assert np.abs(got["pos"] - want["pos"]).max() / np.abs(want["pos"]).max() < 1e-4


def test_new_bodies_forget_the_previous_systems_box(nb, oracle):
    """The float records' origin follows the box of the renewal before the last one.  A handle that
    ran system A past two renewals and is then given a far-away system B must treat B exactly as a
    fresh handle does: the same bits, also when many steps with renewals are enqueued at once (the
    origin a renewal takes does not depend on how far the host runs ahead of the device)."""
    n = 4096
    a0 = oracle.scenario(0, n, 21)
    b0 = oracle.scenario(0, n, 22)
    b0["pos"] += np.array([6.0e13, -4.0e13])  # 60 times the system's size away
    used = nb.NBodySim(n, nb.DIRECT)
    used.set_bodies(a0)
    for _ in range(3):
        used.update(2)
        used.refresh_order()
    fresh = nb.NBodySim(n, nb.DIRECT)
    outs = []
    for sim in (used, fresh):
        sim.set_bodies(b0)
        for _ in range(3):
            sim.update(2)
            sim.refresh_order()
        sim.update(1)
        outs.append(sim.read_bodies())
        sim.close()
    for f in ("pos", "vel", "acc"):
        assert np.array_equal(outs[0][f], outs[1][f]), f
    want = oracle.direct_step(b0, 7)
    ext = np.abs(want["pos"] - want["pos"].mean(0)).max()
    assert np.abs(outs[0]["pos"] - want["pos"]).max() / ext < 1e-6
    # accelerations at the state six steps on keep the bar although the system sits far from the origin
    six = oracle.direct_step(b0, 6)
    assert rel_err(outs[0]["acc"], oracle.direct_accel(six)).max() <= ACC_TOL


def test_full_size_direct_properties(nb, oracle):
    """The N = 1,000,000 direct-sum configuration itself: sampled targets against the oracle, and
    Newton's third law over all bodies (the pair rule is symmetric, so sum m a vanishes up to the
    float rounding of the pair forces)."""
    n = 1000000
    b = nb.init_scenario(nb.SPIRAL, n, 12345)
    acc = gpu_accel(nb, b, nb.DIRECT)
    assert np.all(np.isfinite(acc))
    targets = np.sort(np.random.default_rng(11).choice(n, 64, replace=False)).astype(np.int32)
    ref = oracle.direct_accel_list(b, targets)
    err = rel_err(acc[targets], ref)
    assert err.max() <= ACC_TOL, err.max()
    dyn = b["is_dynamic"] != 0
    f = b["mass"][:, None] * acc
    # the static sun gets no acceleration of its own: leave out the force the others feel from it
    sun = np.flatnonzero(~dyn)
    assert len(sun) == 1
    d = b["pos"][sun[0]] - b["pos"][dyn]
    r = np.sqrt((d ** 2).sum(1) + 0.25)
    from_sun = (6.67e-11 * b["mass"][sun[0]] * b["mass"][dyn] / (r ** 3 + 0.25))[:, None] * d
    net = (f[dyn] - from_sun).sum(0)
    assert np.abs(net).max() / np.abs(f[dyn] - from_sun).sum() < 1.0e-5
