"""How many cells of every level of the step's quadtree hold bodies, at a late state of the collision
workload (tuning aid for the pyramid kernels, run by hand on a GPU box): python tools/occupancy_probe.py [step]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
step = int(sys.argv[1]) if len(sys.argv) > 1 else 620
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(step)
    box = sim.debug(nb.DBG_BBOX, np.float64, 4)
    p = sim.read_bodies()["pos"]
tlx, tly, brx, bry = box
for L in range(4, 12):
    cx = np.minimum(((p[:, 0] - tlx) / (brx - tlx) * (1 << L)).astype(np.int64), (1 << L) - 1)
    cy = np.minimum(((tly - p[:, 1]) / (tly - bry) * (1 << L)).astype(np.int64), (1 << L) - 1)
    cnt = np.bincount(cx * (1 << L) + cy, minlength=1 << (2 * L))
    occ = cnt[cnt > 0]
    print("step %d level %2d: %8d of %8d cells occupied (%.1f %%), largest %d, median %d" % (
        step, L, len(occ), 1 << (2 * L), 100.0 * len(occ) / (1 << (2 * L)), occ.max(), int(np.median(occ))), flush=True)
