"""A/B aid for traversal variants (run by hand on a GPU box, one process per library build):
    NBODY_HIP_LIB=<variant .so> python tools/traverse_ab.py [late_step] [n] [tag]
prints the stage times of the collision workload at steps 3-13 and at `late_step`, the traversal
counters of one step at each state, and stores every 16th body's acceleration at the initial state in
gpurun_out/ab_<tag>.npy so that two builds can be compared (tools/traverse_ab.py diff a b)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

if len(sys.argv) > 1 and sys.argv[1] == "diff":
    a = np.load("gpurun_out/ab_%s.npy" % sys.argv[2])
    b = np.load("gpurun_out/ab_%s.npy" % sys.argv[3])
    d = np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(a, axis=1), 1e-300)
    print("diff %s %s: max rel %.3e, median %.3e, identical %d of %d" % (sys.argv[2], sys.argv[3], d.max(), np.median(d),
                                                                        int((d == 0).sum()), len(d)))
    sys.exit(0)

nb = load_package()
late = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5000000
tag = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(os.environ.get("NBODY_HIP_LIB", "default")).replace(".so", "")
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    np.save("gpurun_out/ab_%s.npy" % tag, sim.compute_accel()[::16])  # (a sample: what comes back from a GPU box is capped)

    def timed(label, k=10):
        sim.sync()
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter()
        sim.update(k)
        sim.sync()
        dt = (time.perf_counter() - t) / k * 1e3
        st = sim.stage_ms()
        sim.set_stage_timing(True, counters=True)
        sim.update(1)
        sim.sync()
        c = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
        kinds = sim.debug(nb.DBG_VISIT_KINDS, np.uint64, 8)
        sim.set_stage_timing(False, counters=False)
        print("%-10s %-14s %.3f ms/step  %s | evals %.3g accepted %.3g visits %.3g batches %.3g stack %d | kinds(M) open %.2f none %.2f full %.2f sure %.2f farc %.2f near %.2f body %.2f drop %.2f" % (
            tag, label, dt, " ".join("%s=%.3f" % kv for kv in st.items() if kv[1] > 0), c[0], c[1], c[5], c[6], c[7],
            *(float(k) / 1e6 for k in kinds[:8])), flush=True)

    sim.update(3)
    timed("steps 3-13")
    done = 14
    while done < late:
        k = min(100, late - done)
        sim.update(k)
        sim.sync()
        done += k
    if late > 14:
        timed("steps %d-%d" % (late, late + 10))
