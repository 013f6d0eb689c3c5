"""Times the reference benchmark configurations (N = 65536 direct, 1M and 5M Barnes-Hut, 1M direct, 50M Barnes-Hut) on one GPU; `d3` = the same shapes with three-dimensional bodies (tuning/record aid, not a test)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_package  # noqa: E402

nb = load_package()


def run(alg, scenario, n, steps, warm=2):
    t = time.perf_counter()
    b = nb.init_scenario(scenario, n, 12345)
    tgen = time.perf_counter() - t
    with nb.NBodySim(n, alg) as sim:
        t = time.perf_counter()
        sim.set_bodies(b)
        tset = time.perf_counter() - t
        sim.update(warm)
        sim.sync()
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter()
        sim.update(steps)
        sim.sync()
        dt = (time.perf_counter() - t) / steps
        st = sim.stage_ms()
        t = time.perf_counter()
        out = sim.read_bodies()
        tread = time.perf_counter() - t
    assert np.all(np.isfinite(out["pos"]))
    name = {0: "direct", 1: "BH"}[alg] + " " + {0: "spiral", 1: "random", 2: "collision", 3: "solar"}[scenario]
    extra = "%.0f Ginter/s" % (n * n / dt / 1e9) if alg == 0 else " ".join("%s=%.2f" % kv for kv in st.items() if kv[1] > 0)
    print("%-20s N=%-9d %9.3f ms/step  %s  (gen %.1fs set %.1fs read %.2fs)" % (name, n, dt * 1e3, extra, tgen, tset, tread), flush=True)


def run3(alg, scenario, n, steps, thickness, warm=2):
    """The same configurations for three-dimensional bodies (discs of the given half-thickness)."""
    b = nb.init_scenario3(scenario, n, 12345, thickness)
    with nb.NBodySim3(n, alg) as sim:
        sim.set_bodies(b)
        sim.update(warm)
        sim.sync()
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter()
        sim.update(steps)
        sim.sync()
        dt = (time.perf_counter() - t) / steps
        st = sim.stage_ms()
        if alg == 1:
            sim.set_stage_timing(False, counters=True)
            sim.update(1)
            c = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
            cut = sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2)
        out = sim.read_bodies()
    assert np.all(np.isfinite(out["pos"]))
    name = "3-D " + {0: "direct", 1: "BH"}[alg] + " " + {0: "spiral", 1: "random", 2: "collision"}[scenario]
    if alg == 0:
        extra = "%.0f Ginter/s" % (n * n / dt / 1e9)
    else:
        extra = " ".join("%s=%.2f" % kv for kv in st.items() if kv[1] > 0)
        extra += " | evals %.3g accepted %.3g nodes %d levels %d closed %d stack %d" % (c[0], c[1], c[2], c[3], cut[0], c[7])
    print("%-22s N=%-9d thick %.0e %9.3f ms/step  %s" % (name, n, thickness, dt * 1e3, extra), flush=True)


which = sys.argv[1:] or ["c2", "c3", "c4", "c5"]
if "d3" in which:
    run3(nb.DIRECT, nb.SPIRAL, 65536, 20, 5.0e10)
    run3(nb.BARNES_HUT, nb.SPIRAL, 1000000, 10, 5.0e10)
    run3(nb.BARNES_HUT, nb.COLLISION, 5000000, 10, 2.0e10)
    run3(nb.BARNES_HUT, nb.COLLISION, 5000000, 10, 1.0e11)
    run3(nb.DIRECT, nb.SPIRAL, 1000000, 3, 5.0e10, 1)
if "c2" in which:
    run(nb.DIRECT, nb.SPIRAL, 65536, 20)
if "c3" in which:
    run(nb.BARNES_HUT, nb.SPIRAL, 1000000, 10)
if "c4" in which:
    run(nb.BARNES_HUT, nb.COLLISION, 5000000, 10)
if "c5" in which:
    run(nb.DIRECT, nb.SPIRAL, 1000000, 3, 1)
    run(nb.BARNES_HUT, nb.SPIRAL, 50000000, 3, 1)
