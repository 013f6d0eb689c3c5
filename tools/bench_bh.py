"""Times the Barnes-Hut step of the library named by NBODY_HIP_LIB (tuning aid, not a test)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
scen = {"collision": nb.COLLISION, "spiral": nb.SPIRAL, "random": nb.RANDOM}[sys.argv[2] if len(sys.argv) > 2 else "collision"]
b = nb.init_scenario(scen, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(3)
    sim.set_stage_timing(True, counters=False)
    acc = {}
    steps = 10
    for _ in range(steps):
        sim.update(1)
        for k, v in sim.stage_ms().items():
            acc[k] = acc.get(k, 0.0) + v / steps
print(os.path.basename(os.environ.get("NBODY_HIP_LIB", "default")), "N=%d" % n,
      " ".join("%s %.3f" % kv for kv in acc.items()), "total %.3f" % sum(acc.values()), flush=True)
