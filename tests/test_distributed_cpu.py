"""Host logic of the multi-GPU path on CPU: two gloo ranks drive DistributedSim with an
engine whose force step is the oracle.  Partitioning, the in-place all-gather of positions
and the step protocol must reproduce the single-process run bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


class OracleEngine:
    """The engine protocol of n-body-simulation_amd/distributed.py on top of the CPU oracle."""

    def __init__(self, oracle, bodies, alg):
        self.o = oracle
        self.b = bodies.copy()
        self.n = len(bodies)
        self.alg = alg
        self.p = oracle.params()
        self.dims = bodies["pos"].shape[1]  # 3: the three-dimensional oracle functions

    def set_partition(self, rank, world):
        from nbody_amd.distributed import slice_bounds

        self.t0, self.t1, self.chunk = slice_bounds(self.n, rank, world)
        self.buf = np.zeros((self.chunk * world, self.dims))
        self.buf[: self.n] = self.b["pos"]

    def exchange_tensor(self):
        return torch.from_numpy(self.buf)

    def step_local(self):
        t0, t1 = self.t0, self.t1
        direct, tree = (self.o.direct_accel, self.o.bh_accel) if self.dims == 2 else (self.o.direct_accel3, self.o.bh_accel3)
        if self.alg == 0:
            acc = direct(self.b, self.p, t0, t1)
        else:
            acc = tree(self.b, self.p)[t0:t1]
        own = self.b[t0:t1]
        own["acc"] = acc
        own["vel"] += acc * self.p.dt
        self.buf[t0:t1] = own["pos"] + own["vel"] * self.p.dt

    def step_finish(self):
        self.b["pos"] = self.buf[: self.n]


def _worker(rank, world, port, alg, n, steps, out_dir, dims=2):
    sys.path.insert(0, HERE)
    from conftest import load_package
    from oracle_lib import Oracle

    load_package()
    from nbody_amd.distributed import DistributedSim

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        o = Oracle()
        b = o.scenario(0, n, 5) if dims == 2 else o.scenario3(0, n, 5, thickness=6.0e10)
        eng = OracleEngine(o, b, alg)
        sim = DistributedSim(eng)
        sim.update(steps)
        np.save(os.path.join(out_dir, "pos_%d.npy" % rank), eng.b["pos"])
        np.save(os.path.join(out_dir, "vel_%d.npy" % rank), eng.b["vel"][eng.t0:eng.t1])
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("alg,n", [(0, 301), (1, 1000)])
def test_two_gloo_ranks_equal_one_process(oracle, nb, tmp_path, alg, n):
    world, steps = 2, 3
    mp.spawn(_worker, args=(world, _free_port(), alg, n, steps, str(tmp_path)), nprocs=world, join=True)
    b = oracle.scenario(0, n, 5)
    ref = oracle.direct_step(b, steps) if alg == 0 else oracle.bh_step(b, steps)
    chunk = (n + 1) // 2
    for r in range(world):
        pos = np.load(tmp_path / ("pos_%d.npy" % r))
        assert np.array_equal(pos, ref["pos"])
        vel = np.load(tmp_path / ("vel_%d.npy" % r))
        assert np.array_equal(vel, ref["vel"][r * chunk:(r + 1) * chunk])


@pytest.mark.parametrize("alg,n", [(0, 257), (1, 600)])
def test_two_gloo_ranks_equal_one_process_3d(oracle, nb, tmp_path, alg, n):
    """The same protocol with records of three doubles (the nb3_ exchange buffer)."""
    world, steps = 2, 3
    mp.spawn(_worker, args=(world, _free_port(), alg, n, steps, str(tmp_path), 3), nprocs=world, join=True)
    b = oracle.scenario3(0, n, 5, thickness=6.0e10)
    ref = oracle.direct_step3(b, steps) if alg == 0 else oracle.bh_step3(b, steps)
    chunk = (n + 1) // 2
    for r in range(world):
        pos = np.load(tmp_path / ("pos_%d.npy" % r))
        assert pos.shape == (n, 3) and np.array_equal(pos, ref["pos"])
        vel = np.load(tmp_path / ("vel_%d.npy" % r))
        assert np.array_equal(vel, ref["vel"][r * chunk:(r + 1) * chunk])


def test_slice_bounds_cover_every_body_once(nb):
    from nbody_amd.distributed import slice_bounds

    for n in (1, 2, 5, 64, 1000, 1000003):
        for world in (1, 2, 3, 4, 8):
            seen = 0
            for r in range(world):
                t0, t1, chunk = slice_bounds(n, r, world)
                assert 0 <= t0 <= t1 <= n and t1 - t0 <= chunk
                assert t0 == min(n, r * chunk)
                seen += t1 - t0
            assert seen == n and chunk * world >= n
