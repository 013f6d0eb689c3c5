"""The direct-sum step at N = 1 M (spiral) step by step over a few seconds: the rate as the GPU's clocks settle
under a sustained FP32 load (tuning aid, run by hand): python tools/direct_sustained_probe.py [n] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
b = nb.init_scenario(nb.SPIRAL, n, 12345)
with nb.NBodySim(n, nb.DIRECT) as sim:
    sim.set_bodies(b)
    t0 = time.perf_counter()
    for k in range(steps):
        t = time.perf_counter()
        sim.update(1)
        sim.sync()
        dt = time.perf_counter() - t
        print("t = %5.2f s step %2d: %.2f ms  %.0f G interactions/s" % (time.perf_counter() - t0, k + 1, dt * 1e3, n * float(n) / dt / 1e9), flush=True)
