"""Step time of the headline Barnes-Hut workload in consecutive chunks of steps after a cold start
(tuning aid, not a test): shows how many steps the GPU takes to reach its steady clocks."""
import sys
import time

sys.path.insert(0, "tests")
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.sync()
    out = []
    for chunk in (3, 5, 5, 5, 5, 5, 5, 5, 5, 10, 10):
        t = time.perf_counter()
        sim.update(chunk)
        sim.sync()
        out.append("%d: %.3f" % (chunk, (time.perf_counter() - t) / chunk * 1e3))
print("BH N=%d ms/step per chunk of steps  " % n + " | ".join(out), flush=True)
