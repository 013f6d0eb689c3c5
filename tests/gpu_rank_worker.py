"""One rank of a multi-process run on ONE GPU (started by tests/test_gpu_distributed.py, not collected
by pytest): every process holds its own simulation handle on cuda:0, the ranks exchange positions
over gloo.  RCCL wants one GPU per rank, so this is as close as a one-GPU box gets to the
multi-GPU launch: separate processes, the real engine, the real stepping protocol.

    python gpu_rank_worker.py RANK WORLD PORT ALG DIMS N STEPS OUT_DIR [cells]
"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from conftest import load_package  # noqa: E402
from oracle_lib import Oracle  # noqa: E402  (scenario generator of the checker: same bodies in every process)


def main():
    rank, world, port, alg, dims, n, steps = (int(x) for x in sys.argv[1:8])
    out_dir = sys.argv[8]
    cells = len(sys.argv) > 9 and sys.argv[9] == "cells"
    nb = load_package()
    from nbody_amd.distributed import DistributedSim, GpuEngine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = Oracle()
        b = o.scenario(0, n, 5) if dims == 2 else o.scenario3(0, n, 5, thickness=6.0e10)
        dev = torch.device("cuda", 0)
        torch.cuda.set_device(dev)
        Sim = nb.NBodySim if dims == 2 else nb.NBodySim3
        with Sim(n, alg) as sim:
            sim.set_bodies(b)
            eng = GpuEngine(sim, dev)
            stepper = DistributedSim(eng)
            if cells:
                eng.use_cell_exchange()
            stepper.update(steps)
            torch.cuda.synchronize()
            out = sim.read_bodies()
            order = sim.order()
        np.save(os.path.join(out_dir, "pos_%d.npy" % rank), out["pos"])
        np.save(os.path.join(out_dir, "vel_%d.npy" % rank), out["vel"])
        np.save(os.path.join(out_dir, "order_%d.npy" % rank), order)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
