"""The same force evaluation (collision N = 5 M, initial state, nothing advanced) over and over for a few
seconds: the per-stage times as the GPU's clocks settle (tuning aid, run by hand)."""
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
    t0 = time.perf_counter()
    for block in range(30):
        sim.set_stage_timing(True, counters=False)
        for _ in range(100):
            sim.evaluate()
        sim.sync()
        st = sim.stage_ms()
        print("t = %5.2f s: total %.3f force %.3f sort %.3f" % (time.perf_counter() - t0, sum(st.values()), st["force"], st["sort"]), flush=True)
