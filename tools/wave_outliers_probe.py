#!/usr/bin/env python3
"""Which waves of a traversal launch are the slow ones: per group of 64 consecutive sorted bodies, the time its
wave took (NB_DBG_WAVE_COST) beside the diagonal of the group's bounding box.
Usage: python tools/wave_outliers_probe.py [N] [step]"""
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nb = importlib.import_module("n-body-simulation_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
step = int(sys.argv[2]) if len(sys.argv) > 2 else 10
sim = nb.NBodySim(n, nb.BARNES_HUT)
sim.set_bodies(nb.init_scenario(nb.COLLISION, n, 12345))
sim.update(step)
sim.sync()
sim.evaluate()  # the state as it is now, sorted again: positions, order and costs belong together
sim.sync()
nw = (n + 63) // 64
c = sim.debug(nb.DBG_WAVE_COST, np.uint32, nw).astype(np.float64) * 0.01
slot_of_sorted = sim.debug(nb.DBG_SORTED_IDX, np.uint32, n)
caller_of_slot = sim.order()
pos = sim.read_bodies()["pos"][caller_of_slot[slot_of_sorted]]
sim.close()
pad = nw * 64 - n
p = np.concatenate([pos, np.repeat(pos[-1:], pad, axis=0)]).reshape(nw, 64, 2)
diag = np.linalg.norm(p.max(axis=1) - p.min(axis=1), axis=1)
# the largest step between consecutive bodies of the group
jump = np.linalg.norm(np.diff(p, axis=1), axis=2).max(axis=1)
print("waves %d: cost mean %.1f us, p99 %.1f, max %.1f; box diagonal median %.3g m, p99 %.3g, max %.3g" %
      (nw, c.mean(), np.percentile(c, 99), c.max(), np.median(diag), np.percentile(diag, 99), diag.max()))
print("correlation of cost with log diagonal: %.2f" % np.corrcoef(c, np.log(diag + 1.0))[0, 1])
for lo, hi in ((0, 50), (50, 90), (90, 99), (99, 99.9), (99.9, 100)):
    a, b = np.percentile(c, lo), np.percentile(c, hi)
    m = (c >= a) & (c <= b)
    print("cost percentiles %5.1f-%5.1f (%.0f-%.0f us): median diagonal %.3g m, median largest jump %.3g m" %
          (lo, hi, a, b, np.median(diag[m]), np.median(jump[m])))
big = diag > 4 * np.median(diag)
print("waves with a diagonal above 4 x the median: %.2f %%, their mean cost %.1f us (others %.1f)" %
      (100 * big.mean(), c[big].mean(), c[~big].mean()))
top = np.argsort(-c)[:10]
print("slowest: " + "; ".join("%d: %.0f us, diag %.2g, jump %.2g" % (i, c[i], diag[i], jump[i]) for i in top))
