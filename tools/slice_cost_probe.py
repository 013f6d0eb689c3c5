#!/usr/bin/env python3
"""How the traversal's time is spread along the sorted order, and how much of a launch is not proportional to
its bodies: the force stage of the galaxy collision (N = 5 M by default) evaluated for one slice of the slots
at a time (nb_set_partition with world = SLICES, rank = slice), against the whole system's.  The slots follow
the Hilbert order of the initial state, so a slice is a contiguous piece of the curve.
Usage: python tools/slice_cost_probe.py [N] [SLICES]"""
import importlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nb = importlib.import_module("n-body-simulation_amd")


def force_ms(bodies, rank, world, reps=20):
    with nb.NBodySim(len(bodies), nb.BARNES_HUT) as sim:
        sim.set_bodies(bodies)
        if world > 1:
            sim.set_partition(rank, world)
        for _ in range(60):
            sim.evaluate()
        sim.sync()
        sim.set_stage_timing(True, counters=False)
        for _ in range(reps):
            sim.evaluate()
        sim.sync()
        return sim.stage_ms()["force"]


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    slices = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    bodies = nb.init_scenario(nb.COLLISION, n, 12345)
    whole = force_ms(bodies, 0, 1)
    print("whole system: force %.4f ms" % whole, flush=True)
    for world in (2, slices):
        parts = [force_ms(bodies, r, world) for r in range(world)]
        print("world %d: slices %s" % (world, " ".join("%.3f" % p for p in parts)))
        print("   sum %.4f ms = %.2f x the whole launch; mean slice %.4f, ideal %.4f" %
              (sum(parts), sum(parts) / whole, sum(parts) / world, whole / world), flush=True)


if __name__ == "__main__":
    main()
