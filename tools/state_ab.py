"""A/B aid for changes that must not alter a Barnes-Hut run at all (run by hand on a GPU box, one process
per library build):  NBODY_HIP_LIB=<variant .so> python tools/state_ab.py <tag> [n] [steps]
steps the collision workload, prints the step time over the last 10 steps and stores every 16th body's
position, velocity and acceleration in gpurun_out/sab_<tag>.npz;  python tools/state_ab.py diff <a> <b>
counts the entries that differ."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))

if sys.argv[1] == "diff":
    a = np.load("gpurun_out/sab_%s.npz" % sys.argv[2])
    b = np.load("gpurun_out/sab_%s.npz" % sys.argv[3])
    for k in a.files:
        print("%-6s differing entries: %d of %d" % (k, int((a[k] != b[k]).sum()), a[k].size))
    sys.exit(0)

from conftest import load_package  # noqa: E402

nb = load_package()
tag = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(steps - 10)
    sim.compute_accel()  # (a force evaluation in between: the step after it finds its box by a pass of its own)
    sim.sync()
    sim.set_stage_timing(True, counters=False)
    t = time.perf_counter()
    sim.update(10)
    sim.sync()
    dt = (time.perf_counter() - t) / 10 * 1e3
    print("%-10s n=%d steps %d-%d: %.4f ms/step  %s" % (tag, n, steps - 10, steps, dt,
          " ".join("%s=%.3f" % kv for kv in sim.stage_ms().items() if kv[1] > 0)), flush=True)
    r = sim.read_bodies()
    np.savez("gpurun_out/sab_%s.npz" % tag, pos=r["pos"][::16], vel=r["vel"][::16], acc=r["acc"][::16])
