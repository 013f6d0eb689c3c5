"""A few Barnes-Hut steps of one scenario for a kernel trace (tuning aid): python tools/bh_steps.py <n> <scenario> <steps>."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000000
scen = {"collision": nb.COLLISION, "spiral": nb.SPIRAL, "random": nb.RANDOM}[sys.argv[2] if len(sys.argv) > 2 else "spiral"]
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
b = nb.init_scenario(scen, n, 12345)
print("bodies made", flush=True)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    print("bodies set", flush=True)
    sim.update(steps)
    sim.sync()
print("done", flush=True)
