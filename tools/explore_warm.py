"""Does the order of the bodies inside the closed leaves matter?  (tuning aid, not a test)
The headline workload after 300 steps, timed before and after nb_refresh_order, which re-sorts the
exchange order -- the order the stable sort keeps inside a leaf -- by the current positions."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(300)
    sim.sync()
    for tag in ("before", "after refresh_order", "again"):
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter()
        sim.update(20)
        sim.sync()
        dt = (time.perf_counter() - t) / 20 * 1e3
        print(tag, "%.3f ms/step" % dt, " ".join("%s=%.3f" % kv for kv in sim.stage_ms().items() if kv[1] > 0), flush=True)
        sim.set_stage_timing(False, counters=False)
        if tag == "before":
            sim.refresh_order()
