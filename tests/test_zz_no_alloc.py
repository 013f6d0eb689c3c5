"""nb_compute_accel_static: the force evaluation that allocates nothing (device arrays in and
out, every intermediate buffer in the library's static device array).

Checked against the existing entry point it is a variant of, nb_compute_accel on a handle: the same
kernels on the same bodies in the same order, so the accelerations must be bit-identical; two calls
in a row, the second with a smaller system (the static array keeps the first call's contents).
Runs last: a failure here must not hide the tests after it.
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hip():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemGetInfo.argtypes = [C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    return hip


def free_device_bytes(hip):
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def static_accel(sim, pos, torch):
    """Accelerations at `pos` (numpy n x 2, caller order) through the static entry point."""
    dev = torch.device("cuda", 0)
    pos_d = torch.from_numpy(np.ascontiguousarray(pos, dtype=np.float64)).to(dev)
    acc_d = torch.full_like(pos_d, float("nan"))
    torch.cuda.synchronize()
    stream = torch.cuda.current_stream(dev).cuda_stream
    sim.compute_accel_static(pos_d.data_ptr(), acc_d.data_ptr(), stream)
    torch.cuda.synchronize()
    return acc_d.cpu().numpy()


def test_static_entry_point_matches_the_handle_path_twice_in_a_row(nb):
    import torch

    hip = _hip()
    big = nb.init_scenario(nb.SPIRAL, 100000, 21)
    small = nb.init_scenario(nb.COLLISION, 3001, 22)
    with nb.NBodySim(len(big), nb.BARNES_HUT) as a, nb.NBodySim(len(small), nb.BARNES_HUT) as b:
        a.set_bodies(big)
        b.set_bodies(small)
        ref_big = a.compute_accel()
        ref_small = b.compute_accel()
        # first call: the larger system
        out_big = static_accel(a, big["pos"], torch)
        # second call, smaller input, on top of what the first left in the static array
        out_small = static_accel(b, small["pos"], torch)
        # and the larger one again
        out_big2 = static_accel(a, big["pos"], torch)
        assert np.array_equal(out_big, ref_big)
        assert np.array_equal(out_small, ref_small)
        assert np.array_equal(out_big2, ref_big)
        # the handle's own state is untouched: it steps as a handle that never made the call
        a.update(2)
        with nb.NBodySim(len(big), nb.BARNES_HUT) as c:
            c.set_bodies(big)
            c.update(2)
            assert np.array_equal(a.read_bodies()["pos"], c.read_bodies()["pos"])

        # positions other than the handle's own: the state two steps on, against a handle that was
        # given that state (its exchange order differs, so bodies sharing a leaf are summed in
        # another order: agreement to rounding, not to the bit)
        later = a.read_bodies()
        out_later = static_accel(b, b.read_bodies()["pos"], torch)  # keeps the array busy in between
        assert np.array_equal(out_later, ref_small)
        out = static_accel(a, later["pos"], torch)
        with nb.NBodySim(len(big), nb.BARNES_HUT) as d:
            d.set_bodies(later)
            ref = d.compute_accel()
        dyn = big["is_dynamic"] != 0
        err = np.linalg.norm(out[dyn] - ref[dyn], axis=1) / np.linalg.norm(ref[dyn], axis=1)
        assert err.max() <= 1e-6, err.max()

        # no device memory is taken or returned by the call
        dev = torch.device("cuda", 0)
        pos_d = torch.from_numpy(np.ascontiguousarray(big["pos"])).to(dev)
        acc_d = torch.empty_like(pos_d)
        torch.cuda.synchronize()
        before = free_device_bytes(hip)
        for _ in range(3):
            a.compute_accel_static(pos_d.data_ptr(), acc_d.data_ptr(), torch.cuda.current_stream(dev).cuda_stream)
        torch.cuda.synchronize()
        assert free_device_bytes(hip) == before
        # (the handle has stepped since ref_big was made, and a 2-D Barnes-Hut step leaves the bodies in other
        # slots: bodies sharing a leaf are summed in another order)
        got = acc_d.cpu().numpy()
        assert (np.linalg.norm(got[dyn] - ref_big[dyn], axis=1) / np.linalg.norm(ref_big[dyn], axis=1)).max() <= 1e-6


def test_static_entry_point_rejects_what_it_cannot_do(nb):
    import torch

    dev = torch.device("cuda", 0)
    buf = torch.zeros(2 * (nb.STATIC_MAX_BODIES + 1), dtype=torch.float64, device=dev)
    with nb.NBodySim(nb.STATIC_MAX_BODIES + 1, nb.BARNES_HUT) as sim:
        sim.setup(nb.RANDOM, 1)
        with pytest.raises(nb.NBodyError, match="exceed the static scratch array's limit"):
            sim.compute_accel_static(buf.data_ptr(), buf.data_ptr())
    with nb.NBodySim(64, nb.DIRECT) as sim:
        sim.setup(nb.RANDOM, 1)
        with pytest.raises(nb.NBodyError, match="Barnes-Hut"):
            sim.compute_accel_static(buf.data_ptr(), buf.data_ptr())
    with nb.NBodySim(64, nb.BARNES_HUT) as sim:
        with pytest.raises(nb.NBodyError, match="no bodies"):
            sim.compute_accel_static(buf.data_ptr(), buf.data_ptr())
        sim.setup(nb.RANDOM, 1)
        with pytest.raises(nb.NBodyError, match="NULL"):
            sim.compute_accel_static(0, buf.data_ptr())


def test_static_entry_point_at_its_size_limit(nb):
    import torch

    n = nb.STATIC_MAX_BODIES
    b = nb.init_scenario(nb.COLLISION, n, 5)
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        ref = sim.compute_accel()
        out = static_accel(sim, b["pos"], torch)
    assert np.array_equal(out, ref)
