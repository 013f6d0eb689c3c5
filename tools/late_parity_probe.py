"""Error of the Barnes-Hut accelerations against the oracle's full tree on sampled bodies at a late state
of the collision workload (run by hand on a GPU box): python tools/late_parity_probe.py [n] [step] [targets]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package, rel_err, sampled_targets  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

nb = load_package()
orc = Oracle()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
step = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 256
b = nb.init_scenario(nb.COLLISION, n, 12345)
if os.environ.get("PERTURB"):
    b["vel"] *= 1.0 + float(os.environ["PERTURB"])
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(step)
    state = sim.read_bodies()
    acc = sim.compute_accel()
targets = sampled_targets(state, nt, step)
ref = orc.bh_accel_list(state, targets)
err = rel_err(acc[targets], ref)
print("n=%d step %d: max err %.3e, median %.3e, above 2e-6: %d of %d  (NB_KEEP_SLOTS=%s)" % (
    n, step, err.max(), np.median(err), int((err > 2e-6).sum()), len(err), os.environ.get("NB_KEEP_SLOTS")), flush=True)
print("worst targets:", targets[np.argsort(err)[-5:]], np.sort(err)[-5:])
