"""Direct-sum rate at small N over a longer run, in chunks of steps (tuning aid, not a test)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
for n in (16384, 65536, 131072):
    b = nb.init_scenario(nb.SPIRAL, n, 12345)
    with nb.NBodySim(n, nb.DIRECT) as sim:
        sim.set_bodies(b)
        sim.update(2)
        sim.sync()
        out = []
        for chunk in (10, 10, 20, 20, 40, 40):
            t = time.perf_counter()
            sim.update(chunk)
            sim.sync()
            dt = (time.perf_counter() - t) / chunk
            out.append("%d: %.0f" % (chunk, n * n / dt / 1e9))
    print("N=%d G/s per chunk of steps  " % n + " | ".join(out), flush=True)
