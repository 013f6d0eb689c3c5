#!/usr/bin/env python3
"""One slice of a partitioned Barnes-Hut system (rank R of W, default 13 of 16) evaluated repeatedly: for a kernel
trace of what a rank's force evaluation launches, and the per-wave times of its traversal."""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nb = importlib.import_module("n-body-simulation_amd")
rank = int(sys.argv[1]) if len(sys.argv) > 1 else 13
world = int(sys.argv[2]) if len(sys.argv) > 2 else 16
b = nb.init_scenario(nb.COLLISION, 5_000_000, 12345)
with nb.NBodySim(len(b), nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.set_partition(rank, world)
    for _ in range(40):
        sim.evaluate()
    sim.sync()
    nw = (len(b) // world + 63) // 64
    c = sim.debug(nb.DBG_WAVE_COST, np.uint32, nw).astype(np.float64) * 0.01
    print("rank %d of %d: %d waves: mean %.1f us, median %.1f, p90 %.1f, p99 %.1f, max %.1f" %
          (rank, world, nw, c.mean(), np.median(c), np.percentile(c, 90), np.percentile(c, 99), c.max()))
    print("per 16th of the slice: " + " ".join("%.0f" % v for v in np.add.reduceat(c, np.arange(0, nw, nw // 16))[:16] / (nw // 16)))
    top = np.argsort(-c)[:8]
    print("slowest waves (index: us): " + ", ".join("%d: %.0f" % (i, c[i]) for i in top))
