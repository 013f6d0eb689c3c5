"""Barnes-Hut step time over a long run (tuning aid, not a test): the spiral galaxy at N = 1 M for 4000 steps,
timed in blocks, then once more after nb_refresh_order.  Measured: 0.60 -> 0.80 ms/step as the disc spreads
(deeper tree); re-deriving the exchange order changes nothing, i.e. the order is not what costs time."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package
nb = load_package()
n = 1000000
b = nb.init_scenario(nb.SPIRAL, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(5); sim.sync()
    for block in range(8):
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter(); sim.update(100); sim.sync(); dt = (time.perf_counter() - t) / 100
        st = sim.stage_ms()
        print("steps %5d: %.3f ms/step" % ((block + 1) * 500 - 400, dt * 1e3), " ".join("%s=%.3f" % kv for kv in st.items() if kv[1] > 0), flush=True)
        sim.set_stage_timing(False, counters=False)
        sim.update(400); sim.sync()
    sim.refresh_order()
    sim.set_stage_timing(True, counters=False)
    t = time.perf_counter(); sim.update(100); sim.sync(); dt = (time.perf_counter() - t) / 100
    print("after refresh_order: %.3f ms/step" % (dt * 1e3), " ".join("%s=%.3f" % kv for kv in sim.stage_ms().items() if kv[1] > 0), flush=True)
