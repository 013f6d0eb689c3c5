"""The 3-D extension of the CPU oracle (oracle/nbody_oracle3.c), pinned to the 2-D oracle.

The reference has no third dimension, so nothing of it can check z != 0 results.  What can be
checked: for bodies in the plane z = 0 the 3-D functions must return, bit for bit, what the 2-D
functions return -- and those are pinned by the compiled reference (test_oracle_golden.py).
Beyond that, properties any 3-D gravity sum has (rotating the system rotates the answer).
"""
import numpy as np
import pytest

from oracle_lib import lift3


@pytest.mark.parametrize("sim,n", [(0, 256), (1, 1024), (2, 512), (3, 5)])
def test_flat_input_reproduces_the_2d_oracle_bit_for_bit(oracle, sim, n):
    b2 = oracle.scenario(sim, n, 77)
    b3 = lift3(b2)
    flat = oracle.scenario3(sim, n, 77, thickness=0.0)
    assert np.array_equal(flat["pos"], b3["pos"]) and np.array_equal(flat["vel"], b3["vel"])
    a2 = oracle.direct_accel(b2)
    a3 = oracle.direct_accel3(b3)
    assert np.array_equal(a3[:, :2], a2) and not a3[:, 2].any()
    box2, box3 = oracle.bh_bbox(b2), oracle.bh_bbox3(b3)
    assert np.array_equal(box3[:4], box2)
    # the z range (+-1e10 round z = 0) is widened to the root's x-width
    half = (box3[2] - box3[0]) / 2.0
    assert box3[4] == half and box3[5] == -half
    t2, s2 = oracle.bh_accel(b2, with_stats=True)
    t3, s3 = oracle.bh_accel3(b3, with_stats=True)
    assert np.array_equal(t3[:, :2], t2) and not t3[:, 2].any()
    assert np.array_equal(s2, s3)  # same visits, accepted cells, depth and node count
    for step2, step3 in ((oracle.direct_step, oracle.direct_step3), (oracle.bh_step, oracle.bh_step3)):
        e2, e3 = step2(b2, 5), step3(b3, 5)
        assert np.array_equal(e3["pos"][:, :2], e2["pos"]) and np.array_equal(e3["vel"][:, :2], e2["vel"])
        assert not e3["pos"][:, 2].any()


def test_keys3_hold_the_2d_path_for_flat_input(oracle):
    b2 = oracle.scenario(1, 300, 5)
    b3 = lift3(b2)
    k2 = oracle.morton_keys(b2, oracle.bh_bbox(b2))
    k3 = oracle.morton_keys3(b3, oracle.bh_bbox3(b3))
    assert not (k3 & np.uint64(3)).any()
    for i in range(len(b2)):
        digits3 = [(int(k3[i]) >> (59 - 3 * l)) & 7 for l in range(20)]
        digits2 = [(int(k2[i]) >> (60 - 2 * l)) & 3 for l in range(20)]
        # z = 0 is the upper half of the root box and the lower half of every cell below it
        assert [d & 3 for d in digits3] == digits2
        assert [d >> 2 for d in digits3] == [0] + [1] * 19


def test_scenario3_thickness(oracle):
    b = oracle.scenario3(0, 500, 9, thickness=2.0e10)
    z = b["pos"][:, 2]
    assert z[-1] == 0.0 and np.abs(z[:-1]).max() <= 2.0e10 and np.abs(z[:-1]).max() > 1.0e10
    flat = oracle.scenario(0, 500, 9)
    assert np.array_equal(b["pos"][:, :2], flat["pos"]) and np.array_equal(b["vel"][:, :2], flat["vel"])
    assert not b["vel"][:, 2].any()


def rotate_xz(b):
    """x -> z, z -> -x (a quarter turn about the y axis)."""
    r = b.copy()
    for f in ("pos", "vel", "acc"):
        r[f][:, 0] = -b[f][:, 2]
        r[f][:, 2] = b[f][:, 0]
    return r


def test_direct3_is_rotation_covariant(oracle):
    b = oracle.scenario3(1, 400, 3, thickness=8.0e10)
    a = oracle.direct_accel3(b)
    ar = oracle.direct_accel3(rotate_xz(b))
    assert np.allclose(ar[:, 2], a[:, 0], rtol=1e-12, atol=0) and np.allclose(ar[:, 0], -a[:, 2], rtol=1e-12, atol=1e-30)
    assert np.allclose(ar[:, 1], a[:, 1], rtol=1e-12, atol=0)


def test_bh3_tends_to_direct3_as_theta_shrinks(oracle):
    b = oracle.scenario3(1, 600, 11, thickness=1.0e11)
    dyn = b["is_dynamic"] != 0
    ref = oracle.direct_accel3(b)[dyn]
    errs = []
    for theta in (0.8, 0.4, 0.1):
        a = oracle.bh_accel3(b, oracle.params(theta=theta))[dyn]
        errs.append(np.median(np.linalg.norm(a - ref, axis=1) / np.linalg.norm(ref, axis=1)))
    assert errs[0] > errs[1] > errs[2] and errs[2] < 1e-4
    # theta = 0 opens every cell: what remains differs from the direct sum only by the cells'
    # cutoff rule (> against >=) and the order of the additions
    a0 = oracle.bh_accel3(b, oracle.params(theta=0.0))[dyn]
    assert (np.linalg.norm(a0 - ref, axis=1) / np.linalg.norm(ref, axis=1)).max() < 1e-12


def test_target_lists_equal_the_full_walks(oracle):
    """The list variants the full-size GPU tests use (one tree of all bodies, walked for sampled
    bodies only, any thread count, any insertion order) return the full walks' values bit for bit:
    the octree, and the 2-D tree of the reference's GPU program."""
    b3 = oracle.scenario3(2, 3000, 9, thickness=2.0e10)
    full = oracle.bh_accel3(b3)
    targets = np.sort(np.random.default_rng(1).choice(len(b3), 200, replace=False)).astype(np.int32)
    for threads, sorted_insert in ((1, False), (4, True)):
        oracle.set_threads(threads)
        got = oracle.bh_accel_list3(b3, targets, sorted_insert=sorted_insert)
        assert np.array_equal(got, full[targets])
    b2 = oracle.scenario(2, 3000, 9)
    full2 = oracle.gpu_accel(b2, depth=5)
    got2 = oracle.gpu_accel_list(b2, targets, depth=5)
    oracle.set_threads(1)
    assert np.array_equal(got2, full2[targets])
    assert np.array_equal(oracle.gpu_accel_list(b2, targets, depth=5), full2[targets])
