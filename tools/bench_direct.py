"""Times the direct-sum step of the library named by NBODY_HIP_LIB (tuning aid, not a test)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
sizes = [int(x) for x in sys.argv[1:]] or [65536, 262144, 1000000]
out = []
for n in sizes:
    b = nb.init_scenario(nb.SPIRAL, n, 12345)
    with nb.NBodySim(n, nb.DIRECT) as sim:
        sim.set_bodies(b)
        sim.update(2)
        sim.sync()
        steps = 20 if n <= 65536 else (6 if n <= 262144 else 3)
        t = time.perf_counter()
        sim.update(steps)
        sim.sync()
        dt = (time.perf_counter() - t) / steps
    out.append("N=%d %.3f ms %.0f G/s" % (n, dt * 1e3, n * n / dt / 1e9))
print(os.environ.get("NBODY_HIP_LIB", "default"), " | ".join(out), flush=True)
