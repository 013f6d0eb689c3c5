"""The collision workload (N = 5M, Barnes-Hut) over 2400 steps in blocks of 200: step time, stage
times, the level the keys stop at and whether the state stays finite (tuning aid, run by hand).
The box keeps growing with the ejected bodies: the tree closes at level 8 at the start, 11 from
step ~400, deeper later -- every hand-over between the builds lies on this path."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = 5000000
total = int(sys.argv[1]) if len(sys.argv) > 1 else 2400
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    done = 0
    while done < total:
        sim.set_stage_timing(False, counters=False)
        sim.update(190)
        sim.sync()
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter()
        sim.update(10)
        sim.sync()
        dt = (time.perf_counter() - t) / 10 * 1e3
        done += 200
        closed, levels = (int(v) for v in sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2))
        cnt = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
        st = sim.stage_ms()
        print("step %5d: %.3f ms/step  closed level %d, keys %d levels, %d nodes  %s" % (
            done, dt, closed, levels, int(cnt[2]), " ".join("%s=%.3f" % kv for kv in st.items() if kv[1] > 0)), flush=True)
    out = sim.read_bodies()
    print("finite:", bool(np.all(np.isfinite(out["pos"])) and np.all(np.isfinite(out["vel"]))),
          "extent %.3g m" % float(np.abs(out["pos"]).max()), flush=True)
