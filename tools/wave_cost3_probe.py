#!/usr/bin/env python3
"""Per-wave times of the octree traversal (NB_DBG_WAVE_COST of an nb_sim3): spread, the slowest waves, and the
list-scheduling model of tools/wave_cost_probe.py for the dispatch order in use and for heaviest-first.
Usage: python tools/wave_cost3_probe.py [N] [thickness] [steps ...]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nb = importlib.import_module("n-body-simulation_amd")
from wave_cost_probe import logical, makespan  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
thick = float(sys.argv[2]) if len(sys.argv) > 2 else 2.0e10
at = [int(v) for v in sys.argv[3:]] or [10]
sim = nb.NBodySim3(n, nb.BARNES_HUT)
sim.set_bodies(nb.init_scenario3(nb.COLLISION, n, 12345, thick))
done = 0
for step in at:
    sim.update(step - done)
    done = step
    sim.sync()
    nw = (n + 63) // 64
    c = sim.debug(nb.DBG_WAVE_COST, np.uint32, nw).astype(np.float64) * 0.01
    print("step %d: %d waves, mean %.0f us, median %.0f, p90 %.0f, p99 %.0f, p99.9 %.0f, max %.0f; sum / 8192 slots = %.0f us" %
          (step, nw, c.mean(), np.median(c), np.percentile(c, 90), np.percentile(c, 99), np.percentile(c, 99.9), c.max(),
           c.sum() / 8192), flush=True)
    now = [logical(b, nw) for b in range(nw)]
    print("   model: as dispatched now %.0f us; heaviest first %.0f us" % (makespan(c, now), makespan(c, list(np.argsort(-c)))))
    q = np.add.reduceat(c, np.arange(0, nw, max(1, nw // 16))) / c.sum()
    print("   share of the time per sixteenth of the order: " + " ".join("%.3f" % v for v in q[:16]))
    print("   slowest (index: us): " + ", ".join("%d: %.0f" % (i, c[i]) for i in np.argsort(-c)[:8]))
sim.close()
