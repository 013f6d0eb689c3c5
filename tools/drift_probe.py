"""Stage times of the collision workload (N = 5M, Barnes-Hut) every 16 steps, and what a renewal of
the exchange order costs and gains (tuning aid, run by hand)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(4)
    step = 4
    while step < 480:
        sim.sync()
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter()
        sim.update(8)
        sim.sync()
        dt = (time.perf_counter() - t) / 8 * 1e3
        st = sim.stage_ms()
        sim.set_stage_timing(False, counters=False)
        lv = sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2)
        print("step %4d  %.3f ms/step  sort=%.3f build=%.3f com=%.3f force=%.3f  closed level %d, key levels %d" % (
            step, dt, st["sort"], st["build"], st["com"], st["force"], lv[0], lv[1]), flush=True)
        sim.update(8)
        step += 16
        if step in (260 + 16 * 6, 260 + 16 * 6 + 16):
            sim.sync()
            t = time.perf_counter()
            sim.refresh_order()
            sim.sync()
            print("  renewal by hand: %.3f ms" % ((time.perf_counter() - t) * 1e3), flush=True)
