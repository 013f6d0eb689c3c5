"""Traversal counters and stage times of the collision workload at a late step (tuning aid)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
late = int(sys.argv[1]) if len(sys.argv) > 1 else 390
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(late)
    sim.sync()
    sim.set_stage_timing(True, counters=False)
    sim.update(8)
    st = sim.stage_ms()
    sim.set_stage_timing(False, counters=True)
    sim.update(1)
    sim.sync()
    c = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
    lv = sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2)
    print("step %d: %s | closed %d keys %d | evals %.4g accepted %.4g nodes %d levels %d visits %.4g batches %.4g stack %d" % (
        late, " ".join("%s=%.3f" % kv for kv in st.items() if kv[1] > 0), lv[0], lv[1], c[0], c[1], c[2], c[3], c[5], c[6], c[7]), flush=True)
