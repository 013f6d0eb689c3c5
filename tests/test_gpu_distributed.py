"""The multi-GPU stepping layer on one GPU: the exchange buffer wrapped as a torch tensor, the
library's kernels on torch's stream, and an in-place RCCL all-gather (world size 1, backend
nccl).  The multi-rank logic itself is covered on CPU by tests/test_distributed_cpu.py."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.fixture(scope="module")
def nccl_group():
    import torch
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


@pytest.mark.parametrize("alg", [0, 1])
def test_distributed_stepper_world1_matches_plain_update(nb, oracle, nccl_group, alg):
    import torch
    from nbody_amd.distributed import DistributedSim, GpuEngine, device_tensor

    n = 4099
    b = oracle.scenario(0, n, 3)
    dev = torch.device("cuda", 0)
    with nb.NBodySim(n, alg) as plain, nb.NBodySim(n, alg) as part:
        plain.set_bodies(b)
        part.set_bodies(b)
        eng = GpuEngine(part, dev)
        stepper = DistributedSim(eng)
        for _ in range(4):
            plain.update()
            eng.step_local()
            stepper.all_gather_positions()  # in place, every slice already at its offset
            eng.step_finish()
        torch.cuda.synchronize()
        # the exchange tensor aliases the library's FP64 positions
        ptr, chunk, rec = part.exchange_buffer()
        t = device_tensor(ptr, n, dev)
        a = part.read_bodies()
        ref = plain.read_bodies()
        assert np.array_equal(t.cpu().numpy(), a["pos"][part.order()])
    if alg == nb.DIRECT:
        assert np.array_equal(a["pos"], ref["pos"])
        assert np.array_equal(a["vel"], ref["vel"])
    else:
        # nb_step leaves the bodies of a one-GPU 2-D Barnes-Hut run in the order of every step's sort, split
        # steps leave them in their slots: bodies of one leaf are summed, and grouped for the traversal,
        # in another order (FP32 summation noise)
        assert np.abs(a["pos"] - ref["pos"]).max() <= 1e-9 * np.abs(ref["pos"]).max()
        assert np.abs(a["vel"] - ref["vel"]).max() <= 1e-6 * np.abs(ref["vel"]).max()


@pytest.mark.parametrize("alg", [0, 1])
def test_distributed_stepper_world1_matches_plain_update_3d(nb, oracle, nccl_group, alg):
    """The same through the nb3_ entry points: the exchange tensor has three columns."""
    import torch
    from nbody_amd.distributed import DistributedSim, GpuEngine, device_tensor

    n = 4099
    b = oracle.scenario3(0, n, 3, thickness=5.0e10)
    dev = torch.device("cuda", 0)
    with nb.NBodySim3(n, alg) as plain, nb.NBodySim3(n, alg) as part:
        plain.set_bodies(b)
        part.set_bodies(b)
        eng = GpuEngine(part, dev)
        stepper = DistributedSim(eng)
        for _ in range(4):
            plain.update()
            eng.step_local()
            stepper.all_gather_positions()
            eng.step_finish()
        torch.cuda.synchronize()
        ptr, chunk, rec = part.exchange_buffer()
        assert rec == 24
        t = device_tensor(ptr, n, dev, 3)
        a = part.read_bodies()
        ref = plain.read_bodies()
        assert np.array_equal(t.cpu().numpy(), a["pos"][part.order()])
    assert np.array_equal(a["pos"], ref["pos"])
    assert np.array_equal(a["vel"], ref["vel"])


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("alg", [0, 1])
def test_native_rccl_world1_matches_plain_update(nb, oracle, alg, dims):
    """nb_comm_init / nb_step_distributed: the library's own RCCL communicator (a second one beside
    torch's, sharing its librccl), world size 1 -- all a one-GPU box can form.  The steps must be
    those of nb_step to the bit, the host must not be needed between them, and the exchange shows up
    as its own timing stage."""
    n = 4099
    b = oracle.scenario(0, n, 3) if dims == 2 else oracle.scenario3(0, n, 3, thickness=5.0e10)
    Sim = nb.NBodySim if dims == 2 else nb.NBodySim3
    cid = nb.comm_unique_id()
    assert len(cid) == nb.COMM_ID_BYTES
    with Sim(n, alg) as plain, Sim(n, alg) as part:
        plain.set_bodies(b)
        part.set_bodies(b)
        with pytest.raises(nb.NBodyError, match="comm_init first"):
            part.step_distributed(1)
        part.comm_init(0, 1, cid)
        with pytest.raises(nb.NBodyError, match="already has a communicator"):
            part.comm_init(0, 1, cid)
        part.set_stage_timing(True, counters=False)
        plain.update(6)
        part.step_distributed(6)  # six steps enqueued in one call
        ms = part.stage_ms()
        a = part.read_bodies()
        ref = plain.read_bodies()
    assert np.array_equal(a["pos"], ref["pos"])
    assert np.array_equal(a["vel"], ref["vel"])
    assert ms["force"] > 0 and ms["exchange"] >= 0


@pytest.mark.parametrize("dims", [2, 3])
def test_native_rccl_world1_renews_the_order_like_the_plain_run(nb, oracle, dims):
    """nb_step_distributed moves the bodies between slots exactly as nb_step does: 3-D crosses the
    Barnes-Hut order renewal (every 256 steps: one more all-gather, of the velocities, then the renewal on
    the device); a 2-D run on one GPU leaves the bodies in the order of every step's own sort."""
    n = 2000
    b = oracle.scenario(0, n, 6) if dims == 2 else oracle.scenario3(0, n, 6, thickness=5.0e10)
    Sim = nb.NBodySim if dims == 2 else nb.NBodySim3
    with Sim(n, nb.BARNES_HUT) as plain, Sim(n, nb.BARNES_HUT) as part:
        plain.set_bodies(b)
        part.set_bodies(b)
        part.comm_init(0, 1, nb.comm_unique_id())
        first = part.order().copy()
        plain.update(260)
        part.step_distributed(260)
        a, ref = part.read_bodies(), plain.read_bodies()
        assert not np.array_equal(first, part.order())
        assert np.array_equal(part.order(), plain.order())
    assert np.array_equal(a["pos"], ref["pos"])
    assert np.array_equal(a["vel"], ref["vel"])


def test_native_rccl_world1_cell_exchange(nb, oracle):
    """The cell exchange through the library's own communicator (ncclAllReduce with MAX and SUM on the
    simulation's stream), world size 1: the same tree as the plain step (the keys carry one more
    digit, so bodies of one closed cell are summed, and grouped for the traversal, in another order:
    the runs agree to FP32 summation noise)."""
    n = 30011
    b = nb.init_scenario(nb.COLLISION, n, 4)
    with nb.NBodySim(n, nb.BARNES_HUT) as plain, nb.NBodySim(n, nb.BARNES_HUT) as part:
        plain.set_bodies(b)
        part.set_bodies(b)
        part.comm_init(0, 1, nb.comm_unique_id())
        part.set_exchange(nb.EXCHANGE_CELLS)
        part.set_stage_timing(True, counters=False)
        plain.update(20)
        part.step_distributed(20)  # crosses the 16-step look at the box
        ms = part.stage_ms()
        a = part.read_bodies()
        ref = plain.read_bodies()
    assert np.abs(a["pos"] - ref["pos"]).max() <= 1e-7 * np.abs(ref["pos"]).max()
    assert np.abs(a["vel"] - ref["vel"]).max() <= 1e-5 * np.abs(ref["vel"]).max()
    assert ms["force"] > 0


def test_native_rccl_world1_cell_exchange_suspends_and_goes_on(nb):
    """nb_step_distributed over a run in which a body leaving at speed makes the box outgrow the cell
    exchange: the library gathers the positions (ncclAllGather) and goes on with the position
    exchange by itself.  Checked against the plain run; `levels` shows that the switch happened."""
    n, steps = 3001, 200
    b = nb.init_scenario(nb.SPIRAL, n, 21)
    extent = np.abs(b["pos"]).max()
    k = int(np.argmax(np.abs(b["pos"][:, 0])))
    b["vel"][k] = (np.sign(b["pos"][k, 0]) * 0.04 * extent / nb.default_params().dt, 0.0)
    with nb.NBodySim(n, nb.BARNES_HUT) as plain, nb.NBodySim(n, nb.BARNES_HUT) as part:
        plain.set_bodies(b)
        part.set_bodies(b)
        part.comm_init(0, 1, nb.comm_unique_id())
        part.set_exchange(nb.EXCHANGE_CELLS)
        first = part.cells_level()
        plain.update(steps)
        part.step_distributed(steps)
        last = part.cells_level()
        a = part.read_bodies()
        ref = plain.read_bodies()
    assert first[0] >= 0 and last == (-1, False)
    # (FP32 sums in another order over 200 steps: 3e-7 of the extent observed)
    assert np.linalg.norm(a["pos"] - ref["pos"], axis=1).max() <= 3e-6 * np.abs(ref["pos"]).max()
    assert np.linalg.norm(a["vel"] - ref["vel"], axis=1).max() <= 5.0


@pytest.mark.parametrize("alg,dims,world,mode", [(0, 2, 2, ""), (1, 2, 2, ""), (1, 2, 3, ""), (0, 3, 2, ""), (1, 3, 2, ""),
                                                 (1, 2, 2, "cells"), (1, 2, 3, "cells")])
def test_separate_processes_step_one_system_together(nb, oracle, tmp_path, alg, dims, world, mode):
    """`world` processes, each with its own handle on the one GPU, advance one system through the
    stepping protocol (positions gathered over gloo).  Every rank must end with the positions of
    the unpartitioned run (up to FP32 summation order: the 64-body groups and the far/near run
    choices differ) and with the velocities of the slots it integrates."""
    import subprocess
    import sys

    n, steps = 3001, 4
    port = _free_port()
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_rank_worker.py")
    procs = [subprocess.Popen([sys.executable, worker, str(r), str(world), str(port), str(alg), str(dims), str(n),
                               str(steps), str(tmp_path), mode]) for r in range(world)]
    codes = [p.wait(timeout=110) for p in procs]
    assert codes == [0] * world
    b = oracle.scenario(0, n, 5) if dims == 2 else oracle.scenario3(0, n, 5, thickness=6.0e10)
    Sim = nb.NBodySim if dims == 2 else nb.NBodySim3
    with Sim(n, alg) as one:
        one.set_bodies(b)
        one.update(steps)
        ref = one.read_bodies()
    chunk = (n + world - 1) // world
    for r in range(world):
        pos = np.load(tmp_path / ("pos_%d.npy" % r))
        vel = np.load(tmp_path / ("vel_%d.npy" % r))
        order = np.load(tmp_path / ("order_%d.npy" % r))
        own = order[r * chunk:(r + 1) * chunk]
        # (with the cell exchange a rank holds only its own bodies' positions up to date)
        sel = own if mode == "cells" else slice(None)
        assert np.linalg.norm(pos[sel] - ref["pos"][sel], axis=1).max() <= 1e-9 * 5e11
        # orbital speeds are 1e4..1e5 m/s; FP32 summation-order noise adds up to ~1e-3 m/s over the steps
        assert np.linalg.norm(vel[own] - ref["vel"][own], axis=1).max() <= 1e-2
