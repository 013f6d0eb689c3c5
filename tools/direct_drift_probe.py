"""Throughput of the direct sum at N = 65536 along a run, with the library's own order renewal and
with one forced every `every` steps (tuning aid, run by hand)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 65536
every = int(sys.argv[1]) if len(sys.argv) > 1 else 0
b = nb.init_scenario(nb.SPIRAL, n, 12345)
with nb.NBodySim(n, nb.DIRECT) as sim:
    sim.set_bodies(b)
    sim.update(20)
    sim.sync()
    for block in range(24):
        t = time.perf_counter()
        for k in range(0, 100, every or 100):
            sim.update(every or 100)
            if every:
                sim.refresh_order()
        sim.sync()
        dt = (time.perf_counter() - t) / 100
        print("steps %4d-%4d  %.3f ms/step  %.0f G/s" % (20 + 100 * block, 120 + 100 * block, dt * 1e3, n * n / dt / 1e9), flush=True)
