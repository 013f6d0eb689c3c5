"""Pins the plain-C oracle (oracle/nbody_oracle.c) to outputs of the reference's own CPU
classes stored in tests/golden/ (made by tests/golden/make_golden.py), and to the
known-answer values of the 5-body solar system after 1 and 100 steps.

The oracle repeats the reference's operation order, so the expected agreement is
bit-for-bit with the same compiler; the asserted bound (1e-12 relative) leaves room for a
different libm or optimisation level only.
"""
import numpy as np
import pytest

from conftest import bodies_from_golden, golden

TOL = 1e-12


def close(a, b, tol=TOL):
    a = np.asarray(a)
    b = np.asarray(b)
    scale = np.maximum(np.abs(b), np.abs(b).max() * 1e-6 + 1e-300)
    return np.max(np.abs(a - b) / scale) <= tol


def test_solar_known_answers(oracle):
    """Values printed by the compiled reference for scenario 2 (solar), DirectSum."""
    b = oracle.scenario(3, 5)
    s1 = oracle.direct_step(b, 1)
    assert s1["pos"][0, 0] == pytest.approx(149596294976.13806, rel=1e-15)
    assert s1["pos"][0, 1] == 745000000.0
    assert s1["vel"][0, 0] == pytest.approx(-148.20095447794509, rel=1e-14)
    assert s1["vel"][0, 1] == 29800.0
    assert s1["acc"][0, 0] == pytest.approx(-0.005928038179117804, rel=1e-14)
    assert s1["acc"][2, 0] == pytest.approx(-0.039573234139035962, rel=1e-14)
    assert np.all(s1["pos"][4] == 0) and np.all(s1["vel"][4] == 0)
    s100 = oracle.direct_step(b, 100)
    expect = np.array(
        [
            [131256301024.06088, 71458106727.478882],
            [219882076684.75839, 59548709506.070251],
            [-28808843896.008045, 49776650490.252937],
            [74258735237.716614, 78218074445.492767],
        ]
    )
    assert np.max(np.abs(s100["pos"][:4] - expect) / np.abs(expect)) < 1e-13
    assert s100["vel"][0] == pytest.approx([-14238.206232464086, 26213.304128239939], rel=1e-13)
    bh100 = oracle.bh_step(b, 100)
    assert bh100["pos"][1, 0] == pytest.approx(219882076737.13251, rel=1e-13)
    # analytic sanity: G M_sun / r^2 for the Earth
    assert abs(s1["acc"][0, 0]) == pytest.approx(6.67e-11 * 1.989e30 / 1.496e11**2, rel=2e-4)


def test_solar_golden(oracle):
    g = golden("solar_n5")
    b = bodies_from_golden(g)
    assert np.array_equal(oracle.scenario(3, 5)["pos"], g["pos0"])
    for name, step in (("direct", oracle.direct_step), ("bh", oracle.bh_step)):
        s1 = step(b, 1)
        assert close(s1["acc"], g["acc_" + name])
        assert close(s1["pos"], g["pos_%s_1" % name])
        assert close(s1["vel"], g["vel_%s_1" % name])
        s100 = step(b, 100)
        assert close(s100["pos"], g["pos_%s_100" % name])
        assert close(s100["vel"], g["vel_%s_100" % name])


@pytest.mark.parametrize("sim,name", [(0, "spiral"), (1, "random"), (2, "collision")])
@pytest.mark.parametrize("n", [256, 4096])
def test_scenarios_and_steps_golden(oracle, sim, name, n):
    g = golden("%s_n%d" % (name, n))
    b0 = oracle.scenario(sim, n, 12345)
    # the generators reproduce the reference's rand() stream and arithmetic exactly
    assert np.array_equal(b0["pos"], g["pos0"])
    assert np.array_equal(b0["vel"], g["vel0"])
    assert np.array_equal(b0["mass"], g["mass"])
    assert np.array_equal(b0["radius"], g["radius"])
    assert np.array_equal(b0["is_dynamic"], g["is_dynamic"])
    b = bodies_from_golden(g)
    assert close(oracle.direct_accel(b), g["acc_direct"])
    assert close(oracle.bh_accel(b), g["acc_bh"])
    nsteps = 100
    d = oracle.direct_step(b, nsteps)
    assert close(d["pos"], g["pos_direct_100"])
    assert close(d["vel"], g["vel_direct_100"])
    h = oracle.bh_step(b, nsteps)
    assert close(h["pos"], g["pos_bh_100"])
    assert close(h["vel"], g["vel_bh_100"])


def test_bbox_is_float_rounded(oracle):
    b = bodies_from_golden(golden("spiral_n256"))
    box = oracle.bh_bbox(b)
    assert np.array_equal(box, box.astype(np.float32).astype(np.float64))
    assert box[0] == np.float32(b["pos"][:, 0].min() - 1.0e10)
    assert box[1] == np.float32(b["pos"][:, 1].max() + 1.0e10)
    assert box[2] == np.float32(b["pos"][:, 0].max() + 1.0e10)
    assert box[3] == np.float32(b["pos"][:, 1].min() - 1.0e10)


def test_morton_keys_follow_the_midpoint_descent(oracle):
    b = bodies_from_golden(golden("random_n256"))
    box = oracle.bh_bbox(b)
    keys = oracle.morton_keys(b, box)
    # level-0 digit against the root midpoints, ties as QuadTree::getQuadrant has them
    mx = (box[0] + box[2]) / 2.0
    my = (box[1] + box[3]) / 2.0
    right = ~(mx >= b["pos"][:, 0])
    below = ~(my <= b["pos"][:, 1])
    assert np.array_equal((keys >> np.uint64(60)) & np.uint64(3), (below.astype(np.uint64) << np.uint64(1)) | right.astype(np.uint64))
    assert len(np.unique(keys)) == len(keys)
    assert keys.max() < (1 << 62)


def test_coincident_bodies_are_reported(oracle):
    b = oracle.scenario(1, 16)
    b["pos"][3] = b["pos"][7]
    with pytest.raises(ValueError):
        oracle.bh_accel(b)


def test_config0_through_the_reference_cpu_classes(oracle):
    """BASELINE.json configs[0] -- random-body init, N = 4096, direct sum, 100 steps on the reference's
    CPU path, headless -- run through the reference's own classes (NBody(n, 0, 1): DirectSum,
    initRandomBodies; compiled unmodified into oracle/_ref by oracle/Makefile) where that library
    was built: the run reproduces the fixture it left (tests/golden/random_n4096.npz) to the last bit
    (test_scenarios_and_steps_golden pins the plain-C restatement to the same fixture at 1e-12).  (The
    product has no CPU path: its run of the same configuration is
    test_gpu_direct.py::test_random_scenario_trajectory[random_n4096].)"""
    from oracle_lib import Reference

    g = golden("random_n4096")
    if Reference.available():
        ref = Reference()
        b0 = ref.scenario(1, 4096, 12345)
        assert np.array_equal(b0["pos"], g["pos0"]) and np.array_equal(b0["mass"], g["mass"])
        end = ref.step(0, b0, 100)
        assert np.array_equal(end["pos"], g["pos_direct_100"])
        assert np.array_equal(end["vel"], g["vel_direct_100"])
    else:  # (test_scenarios_and_steps_golden[4096-1-random] steps the restatement through the same 100 steps)
        pytest.skip("oracle/_ref/libnbody_ref.so was not built here")
