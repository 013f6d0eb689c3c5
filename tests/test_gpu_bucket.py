"""Barnes-Hut with the tree of the reference's GPU program (nb_params.bucket_depth): cells are not
split below a level (9 in src/gpu/BH/barnesHut_kernel.cu), leaves are point masses.

That program cannot be built here (CUDA, device-side launches, OpenCV) and ships no outputs, so
nothing of the reference pins this mode: parity unpinned.  The checker is the oracle's FP64
restatement of that program's kernels.
"""
import numpy as np
import pytest

from conftest import bodies_from_golden, golden, rel_err, sampled_targets

pytestmark = pytest.mark.gpu

ACC_TOL = 1.0e-4       # relative acceleration error allowed against the FP64 restatement
ACC_EXPECTED = 5.0e-6  # what the float path is expected to deliver


def run(nb, b, depth, steps=0):
    sim = nb.NBodySim(len(b), nb.BARNES_HUT, nb.default_params(bucket_depth=depth))
    sim.set_bodies(b)
    if steps:
        sim.update(steps)
        out = sim.read_bodies()
    else:
        out = sim.compute_accel()
    return sim, out


@pytest.mark.parametrize("depth", [9, 6, 3])
@pytest.mark.parametrize("name", ["spiral_n4096", "random_n4096", "collision_n4096", "spiral_n256", "solar_n5"])
def test_accel_matches_the_gpu_tree_restatement(nb, oracle, name, depth):
    b = bodies_from_golden(golden(name))
    sim, acc = run(nb, b, depth)
    ref = oracle.gpu_accel(b, depth)
    dyn = b["is_dynamic"] != 0
    err = rel_err(acc[dyn], ref[dyn])
    assert err.max() <= ACC_TOL, err.max()
    assert err.max() <= ACC_EXPECTED, err.max()
    box = sim.debug(nb.DBG_BBOX, np.float64, 4)
    assert np.array_equal(box, oracle.gpu_bbox(b))
    sim.close()


def test_buckets_really_hold_several_bodies(nb, oracle):
    """At depth 3 there are at most 64 leaves: the tree must be tiny and the result far from the
    one-body-per-leaf tree's, yet equal to the restatement's."""
    b = bodies_from_golden(golden("spiral_n4096"))
    sim, acc = run(nb, b, 3)
    sim.set_stage_timing(False, counters=True)
    sim.compute_accel()
    nodes = int(sim.debug(nb.DBG_COUNTERS, np.uint64, 8)[2])
    sim.close()
    assert nodes <= 1 + 4 + 16 + 64
    exact = oracle.bh_accel(b)
    dyn = b["is_dynamic"] != 0
    assert rel_err(acc[dyn], exact[dyn]).max() > 1.0e-3


def test_trajectory_and_static_bodies(nb, oracle):
    b = bodies_from_golden(golden("spiral_n256"))
    b["vel"][-1] = (123.0, -45.0)  # the static sun is given a velocity: the GPU program never moves it
    sim, got = run(nb, b, 9, steps=50)
    sim.close()
    want = oracle.gpu_step(b, 50, 9)
    assert np.array_equal(got["pos"][-1], b["pos"][-1])
    scale = np.abs(want["pos"]).max()
    assert np.abs(got["pos"] - want["pos"]).max() / scale < 1.0e-6


@pytest.mark.timeout(600)
def test_full_size_bucket_tree_matches_the_restatement(nb, oracle, capsys):
    """The size the benchmark times this mode at (galaxy collision, N = 5,000,000, depth 9: buckets of
    about 19 bodies on average, thousands in the discs' middles, each acting as one point mass --
    barnesHut_kernel.cu:309,387-399): 256 sampled bodies against the restatement's tree of all
    bodies, at the start and three steps on."""
    n = 5000000
    b = nb.init_scenario(nb.COLLISION, n, 12345)
    sim, acc = run(nb, b, 9)
    sim.update(3)
    later = sim.read_bodies()
    acc3 = sim.compute_accel()
    sim.close()
    for state, got, seed in ((b, acc, 23), (later, acc3, 29)):
        assert np.all(np.isfinite(got))
        targets = sampled_targets(state, 256, seed)
        ref = oracle.gpu_accel_list(state, targets, 9)
        err = rel_err(got[targets], ref)
        with capsys.disabled():
            print("\n[bucket tree 5M: max err %.2e]" % err.max(), flush=True)
        assert err.max() <= ACC_TOL, err.max()


def test_bad_depth_is_rejected(nb):
    with pytest.raises(nb.NBodyError):
        nb.NBodySim(16, nb.BARNES_HUT, nb.default_params(bucket_depth=32))
