"""Per-step stage times of the first steps after the bodies are set (collision N = 5 M): how long the GPU's
clocks take to come up after the idle upload (tuning aid, run by hand): python tools/clock_ramp_probe.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.set_stage_timing(True, counters=False)
    for k in range(steps):
        sim.update(1)
        st = sim.stage_ms()
        print("step %3d: total %.3f force %.3f sort %.3f" % (k + 1, sum(st.values()), st["force"], st["sort"]), flush=True)
