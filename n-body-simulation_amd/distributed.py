"""One process per GPU: every rank holds all bodies, integrates its slice of the exchange
order and all-gathers the new positions each step (RCCL over xGMI through
torch.distributed's "nccl" backend; "gloo" for the host-logic tests on CPU and for ranks that
share one GPU, the gathered slices then pass through host memory).

The reference is single-GPU; this layer has no counterpart there.  The engine it drives
only has to offer

    n, set_partition(rank, world), step_local(), step_finish(),
    exchange_tensor() -> torch tensor [world * chunk, 2 or 3] float64 holding every body's position,
                         this rank's slice up to date after step_local()

which NBodySim and NBodySim3 (below, through their exchange buffers) and the CPU test engine all do.
"""
import numpy as np
import torch
import torch.distributed as dist


def slice_bounds(n, rank, world):
    """Exchange-order slots [t0, t1) of a rank; chunk = ceil(n / world) (nb_set_partition)."""
    chunk = (n + world - 1) // world
    return min(n, rank * chunk), min(n, (rank + 1) * chunk), chunk


class _DevicePointer:
    """Exposes a raw device pointer to torch without copying."""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {
            "shape": (nbytes,),
            "typestr": "|u1",
            "data": (ptr, False),
            "version": 2,
        }


def device_tensor(ptr, records, device, width=2):
    """A float64 [records, width] tensor aliasing `records` position records of width doubles at ptr."""
    t = torch.as_tensor(_DevicePointer(ptr, records * 8 * width), device=device)
    return t.view(torch.float64).view(records, width)


class GpuEngine:
    """NBodySim or NBodySim3 behind the engine protocol; its kernels run on torch's current stream."""

    def __init__(self, sim, device):
        self.sim = sim
        self.n = sim.n
        self.device = device
        sim.set_stream(torch.cuda.current_stream(device).cuda_stream)
        self._cache = {}

    def set_partition(self, rank, world):
        self.sim.set_partition(rank, world)
        self.world = world

    def use_cell_exchange(self, mode=None):
        """Switches the simulation to NB_EXCHANGE_CELLS (or NB_EXCHANGE_AUTO) after set_partition and
        before the first step."""
        from . import EXCHANGE_CELLS

        self.sim.set_exchange(EXCHANGE_CELLS if mode is None else mode)
        self.cells = True

    def step_local(self):
        self.sim.step_local()

    def step_finish(self):
        self.sim.step_finish()

    def exchange_tensor(self):
        ptr, chunk, rec = self.sim.exchange_buffer()
        assert rec in (16, 24)
        t = self._cache.get(ptr)
        if t is None:
            t = device_tensor(ptr, chunk * self.world, self.device, rec // 8)
            self._cache[ptr] = t
        return t


class DistributedSim:
    """Steps an engine across the ranks of a torch.distributed process group."""

    def __init__(self, engine, group=None):
        self.engine = engine
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.t0, self.t1, self.chunk = slice_bounds(engine.n, self.rank, self.world)
        engine.set_partition(self.rank, self.world)

    def all_gather_positions(self):
        buf = self.engine.exchange_tensor()
        mine = buf[self.rank * self.chunk:(self.rank + 1) * self.chunk]
        if dist.get_backend(self.group) == "nccl":
            # in place: every rank's slice already sits at its offset in the output
            dist.all_gather_into_tensor(buf, mine, group=self.group)
        else:
            # gloo gathers host tensors: a device slice is staged through the host (.cpu() waits for
            # the kernels queued on torch's current stream, which is the engine's stream)
            send = mine.cpu() if mine.is_cuda else mine.clone()
            parts = [torch.empty_like(send) for _ in range(self.world)]
            dist.all_gather(parts, send, group=self.group)
            for r, p in enumerate(parts):
                if r != self.rank:
                    buf[r * self.chunk:(r + 1) * self.chunk].copy_(p)

    def all_reduce_cells(self, which):
        """In-place reduction over the ranks of the cell-exchange buffer `which` (0: extent, MAX; 1: cell
        records, SUM) of an engine whose simulation runs the cell exchange."""
        ptr, count = self.engine.sim.cells_buffer(which)
        t = device_tensor(ptr, count, self.engine.device, 1).view(-1)
        op = dist.ReduceOp.MAX if which == 0 else dist.ReduceOp.SUM
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(t, op=op, group=self.group)
        else:
            h = t.cpu()  # waits for the kernels queued on torch's current stream (the engine's)
            dist.all_reduce(h, op=op, group=self.group)
            t.copy_(h)

    def update(self, nsteps=1):
        cells = getattr(self.engine, "cells", False)
        for _ in range(nsteps):
            level = -1
            if cells:
                # the library follows the bounding box and suspends the cell exchange when the
                # positions are the cheaper (or the only possible) thing to exchange
                level, gather = self.engine.sim.cells_level()
                if gather:
                    self.all_gather_positions()
            if level >= 0:
                self.engine.sim.step_cells(0)
                self.all_reduce_cells(0)
                self.engine.sim.step_cells(1)
                self.all_reduce_cells(1)
                self.engine.sim.step_cells(2)
                continue
            self.engine.step_local()
            if self.world > 1:
                self.all_gather_positions()
            self.engine.step_finish()
