"""The 3-D collision discs (N = 5 M) along their run: stage times every 60 steps, and right after a renewal of
the order asked for by hand (tuning aid, run by hand on a GPU box): python tools/bh3_late_probe.py [steps]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
total = int(sys.argv[1]) if len(sys.argv) > 1 else 240
b = nb.init_scenario3(nb.COLLISION, n, 12345, 2.0e10)
with nb.NBodySim3(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)

    def timed(label):
        sim.set_stage_timing(True, counters=False)
        sim.update(5)
        sim.sync()
        st = sim.stage_ms()
        sim.set_stage_timing(False, counters=False)
        print("%-28s %s total %.3f" % (label, " ".join("%s=%.3f" % kv for kv in st.items() if kv[1] > 0), sum(st.values())), flush=True)

    done = 0
    sim.update(3)
    done = 3
    timed("steps 3-8")
    done += 5
    while done < total:
        sim.update(55)
        done += 55
        timed("steps %d-%d" % (done, done + 5))
        done += 5
    sim.refresh_order()
    timed("after a renewal by hand")
    sim.update(10)
    timed("15 steps later")
