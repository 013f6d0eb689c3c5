#!/usr/bin/env python3
"""3-D Barnes-Hut step time of the collision discs (N = 5 M, thickness 2e10 m) over a stretch of steps: for A/B
runs of one build under different environment switches (e.g. NB_PLAIN_WAVE_ORDER=1)."""
import importlib
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nb = importlib.import_module("n-body-simulation_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
b = nb.init_scenario3(nb.COLLISION, n, 12345, 2.0e10)
with nb.NBodySim3(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(20)
    sim.sync()
    out = []
    for _ in range(4):
        t0 = time.perf_counter()
        sim.update(20)
        sim.sync()
        out.append((time.perf_counter() - t0) / 20 * 1e3)
    print("NB_PLAIN_WAVE_ORDER=%s: ms/step over four stretches of 20 steps: %s" %
          (os.environ.get("NB_PLAIN_WAVE_ORDER"), " ".join("%.3f" % v for v in out)))
