"""Average step time of the collision workload (N = 5M, Barnes-Hut) over 512 steps with the exchange
order renewed by hand every `every` steps, beside the library's own schedule (tuning aid)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
b = nb.init_scenario(nb.COLLISION, n, 12345)
for every in [int(a) for a in sys.argv[1:]] or [0]:
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        sim.update(4)
        sim.sync()
        t = time.perf_counter()
        marks = []
        for done in range(0, 512, every or 512):
            sim.update(every or 512)
            if every:
                sim.refresh_order()
            if (done + (every or 512)) % 128 == 0:
                sim.sync()
                marks.append((time.perf_counter() - t) * 1e3)
        sim.sync()
        total = (time.perf_counter() - t) * 1e3
        parts = [marks[0]] + [b2 - a2 for a2, b2 in zip(marks[:-1], marks[1:])]
        print("renewal every %3d steps: %.3f ms/step over 512 steps; per 128 steps: %s" % (
            every, total / 512, " ".join("%.3f" % (p / 128) for p in parts)), flush=True)
