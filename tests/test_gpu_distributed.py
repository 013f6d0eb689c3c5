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
    assert np.array_equal(a["pos"], ref["pos"])
    assert np.array_equal(a["vel"], ref["vel"])


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
