"""Times the 3-D Barnes-Hut step of the library named by NBODY_HIP_LIB (tuning aid, not a test)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
scen = {"collision": nb.COLLISION, "spiral": nb.SPIRAL, "random": nb.RANDOM}[sys.argv[2] if len(sys.argv) > 2 else "collision"]
thick = float(sys.argv[3]) if len(sys.argv) > 3 else 2.0e10
b = nb.init_scenario3(scen, n, 12345, thick)
with nb.NBodySim3(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(3)
    sim.set_stage_timing(True, counters=False)
    sim.update(10)
    st = sim.stage_ms()
    sim.set_stage_timing(False, counters=True)
    sim.update(1)
    c = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
    kinds = sim.debug(nb.DBG_VISIT_KINDS, np.uint64, 8)
print(os.path.basename(os.environ.get("NBODY_HIP_LIB", "default")), "N=%d" % n,
      " ".join("%s %.3f" % kv for kv in st.items()), "total %.3f" % sum(st.values()),
      "| evals %.3g acc %.3g nodes %d redo %d visits %.3g batches %.3g (%.1f/batch) stack %d"
      % (c[0], c[1], c[2], c[4], c[5], c[6], c[5] / max(1, c[6]), c[7]),
      "| kinds(M) open %.1f none %.1f full %.1f sure %.1f farc %.1f near %.1f body %.1f drop %.1f" % tuple(float(k) / 1e6 for k in kinds),
      flush=True)
