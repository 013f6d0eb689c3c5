"""A/B aid for direct-sum builds (run by hand on a GPU box, one process per library build):
    NBODY_HIP_LIB=<variant .so> python tools/direct_ab.py <tag>
steps a few systems with the direct sum (2-D and 3-D, near-field heavy and sparse, ragged sizes),
prints interactions/s and stores sampled states in gpurun_out/dab_<tag>.npz;
    python tools/direct_ab.py diff <tag a> <tag b>
compares two such files (largest relative difference per array, and how many entries are bit-identical)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

if len(sys.argv) > 1 and sys.argv[1] == "diff":
    a = np.load("gpurun_out/dab_%s.npz" % sys.argv[2])
    b = np.load("gpurun_out/dab_%s.npz" % sys.argv[3])
    for k in a.files:
        x, y = a[k], b[k]
        scale = np.maximum(np.abs(x).max(), 1e-300)
        print("%-28s max |diff| / max |value| %.3e   identical %d of %d" % (k, np.abs(x - y).max() / scale, int((x == y).sum()), x.size))
    sys.exit(0)

from conftest import load_package  # noqa: E402

nb = load_package()
tag = sys.argv[1] if len(sys.argv) > 1 else "default"
out = {}


def run2(name, scen, n, steps):
    b = nb.init_scenario(scen, n, 777)
    with nb.NBodySim(n, nb.DIRECT) as sim:
        sim.set_bodies(b)
        out[name + "_acc0"] = sim.compute_accel()[:: max(1, n // 4096)]
        sim.update(2)
        sim.sync()
        t = time.perf_counter()
        sim.update(steps)
        sim.sync()
        dt = (time.perf_counter() - t) / steps
        r = sim.read_bodies()
        out[name + "_pos"] = r["pos"][:: max(1, n // 4096)]
        out[name + "_vel"] = r["vel"][:: max(1, n // 4096)]
        print("%-8s %-26s n=%-8d %.3f ms/step  %.0f G interactions/s" % (tag, name, n, dt * 1e3, n * float(n) / dt / 1e9), flush=True)


def run3(name, n, steps, thickness):
    b = nb.init_scenario3(nb.COLLISION, n, 777, thickness)
    with nb.NBodySim3(n, nb.DIRECT) as sim:
        sim.set_bodies(b)
        out[name + "_acc0"] = sim.compute_accel()[:: max(1, n // 4096)]
        sim.update(2)
        sim.sync()
        t = time.perf_counter()
        sim.update(steps)
        sim.sync()
        dt = (time.perf_counter() - t) / steps
        r = sim.read_bodies()
        out[name + "_pos"] = r["pos"][:: max(1, n // 4096)]
        out[name + "_vel"] = r["vel"][:: max(1, n // 4096)]
        print("%-8s %-26s n=%-8d %.3f ms/step  %.0f G interactions/s" % (tag, name, n, dt * 1e3, n * float(n) / dt / 1e9), flush=True)


run2("spiral_65536", nb.SPIRAL, 65536, 200)
run2("spiral_1m", nb.SPIRAL, 1000000, 3)
run2("collision_100001", nb.COLLISION, 100001, 40)
run2("random_30000", nb.RANDOM, 30000, 100)
run2("spiral_777", nb.SPIRAL, 777, 200)
run3("discs3_100001", 100001, 20, 2.0e10)
run3("discs3_1m", 1000000, 2, 2.0e10)
run3("flat3_50000", 50000, 50, 0.0)
np.savez("gpurun_out/dab_%s.npz" % tag, **out)
