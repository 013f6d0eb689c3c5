import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from conftest import load_package
nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
scen = {"collision": nb.COLLISION, "spiral": nb.SPIRAL, "random": nb.RANDOM}[sys.argv[2] if len(sys.argv) > 2 else "collision"]
b = nb.init_scenario(scen, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(2)
    sim.set_stage_timing(False, counters=True)
    sim.update(1)
    sim.sync()
    c = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
    print(os.path.basename(os.environ.get("NBODY_HIP_LIB", "default")), list(map(int, c)))
