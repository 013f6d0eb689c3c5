"""Times the 3-D direct-sum step of the library named by NBODY_HIP_LIB (tuning aid, not a test)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
out = []
for n in [int(x) for x in sys.argv[1:]] or [65536, 1000000]:
    b = nb.init_scenario3(nb.SPIRAL, n, 12345, 5.0e10)
    with nb.NBodySim3(n, nb.DIRECT) as sim:
        sim.set_bodies(b)
        sim.update(20 if n <= 65536 else 1)
        sim.sync()
        steps = 100 if n <= 65536 else 3
        t = time.perf_counter()
        sim.update(steps)
        sim.sync()
        dt = (time.perf_counter() - t) / steps
    out.append("N=%d %.3f ms %.0f G/s" % (n, dt * 1e3, n * n / dt / 1e9))
print(os.path.basename(os.environ.get("NBODY_HIP_LIB", "default")), "3-D direct", " | ".join(out), flush=True)
