"""Direct sum at N = 65 536 (BASELINE.json configs[1]) for the library named by NBODY_HIP_LIB: steady-state
rate over 400 steps after 100 warm-up steps, with the stage timer's split (tuning aid, not a test)."""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
b = nb.init_scenario(nb.SPIRAL, n, 12345)
with nb.NBodySim(n, nb.DIRECT) as sim:
    sim.set_bodies(b)
    sim.update(100)
    sim.sync()
    sim.set_stage_timing(True, counters=False)
    t = time.perf_counter()
    sim.update(400)
    sim.sync()
    dt = (time.perf_counter() - t) / 400
    st = sim.stage_ms()
print(os.path.basename(os.environ.get("NBODY_HIP_LIB", "default")), "N=%d %.4f ms/step %.0f G/s (%.1f %% of 157.3 TFLOP/s at 20 FLOP); force stage %.4f ms"
      % (n, dt * 1e3, n * n / dt / 1e9, n * n / dt * 20 / 157.3e12 * 100, st["force"]), flush=True)
