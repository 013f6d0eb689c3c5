"""Runs the collision workload (N = 5M, Barnes-Hut) to a late step and times it there (tuning aid,
run by hand; `trace` = only run, for a kernel trace of the last steps)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
late = int(sys.argv[1]) if len(sys.argv) > 1 else 300
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)

    def timed(tag, k=10):
        sim.sync()
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter()
        sim.update(k)
        sim.sync()
        dt = (time.perf_counter() - t) / k * 1e3
        st = sim.stage_ms()
        sim.set_stage_timing(False, counters=False)
        print("%-28s %.3f ms/step  %s" % (tag, dt, " ".join("%s=%.3f" % kv for kv in st.items() if kv[1] > 0)), flush=True)

    sim.update(3)
    timed("steps 3-13")
    for done in range(13, late, 50):  # (in pieces, with a sign of life: under a profiler this takes a while)
        sim.update(min(50, late - done))
        sim.sync()
        print("  at step %d" % min(late, done + 50), flush=True)
    if "renew" in sys.argv:
        sim.refresh_order()
    timed("steps %d-%d" % (late, late + 10))
