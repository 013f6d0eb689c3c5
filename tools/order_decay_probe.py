"""How the Barnes-Hut step slows down between two renewals of the slot order (tuning aid, run by hand):
the collision workload (N = 5 M) is taken to a renewal, then the step and stage times are printed for
blocks of steps until the next one.  python tools/order_decay_probe.py [first_step] [block]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
first = int(sys.argv[1]) if len(sys.argv) > 1 else 256
block = int(sys.argv[2]) if len(sys.argv) > 2 else 16
span = int(sys.argv[3]) if len(sys.argv) > 3 else 256
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(first)
    sim.sync()
    done = 0
    while done < span:
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter()
        sim.update(block)
        sim.sync()
        dt = (time.perf_counter() - t) / block * 1e3
        done += block
        st = sim.stage_ms()
        print("steps %4d..%4d: %.3f ms/step  %s" % (first + done - block, first + done, dt,
              " ".join("%s=%.3f" % kv for kv in st.items() if kv[1] > 0)), flush=True)
