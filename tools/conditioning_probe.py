"""How well conditioned the acceleration of the worst-matching bodies is (run by hand on a GPU box): for
the sampled bodies of a late collision state with the largest error against the oracle's tree, the ratio
sum |term| / |sum of terms| of the FP64 direct sum over all bodies outside the cutoff.  A float evaluation of a sum with
ratio c is good to about c * 1e-7.  python tools/conditioning_probe.py [n] [step] [targets]"""
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
nt = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
b = nb.init_scenario(nb.COLLISION, n, 12345)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.update(step)
    state = sim.read_bodies()
    acc = sim.compute_accel()
targets = sampled_targets(state, nt, step)
ref = orc.bh_accel_list(state, targets)
err = rel_err(acc[targets], ref)
order = np.argsort(err)
pos, mass, radius = state["pos"], state["mass"], state["radius"]
cth = nb.default_params().collision_th


def conditioning(t):
    d = pos - pos[t]
    r2 = (d * d).sum(axis=1)
    r2[t] = np.inf
    # the reference's cutoff (directSum.cpp:69-72): pairs closer than r_i + r_j + collision_th do not interact
    r2[np.sqrt(r2) <= radius + radius[t] + cth] = np.inf
    w = mass / (r2 * np.sqrt(r2))
    terms = d * w[:, None]
    return np.linalg.norm(terms, axis=1).sum() / np.linalg.norm(terms.sum(axis=0))


print("n=%d step %d: max err %.3e, median %.3e" % (n, step, err.max(), np.median(err)))
for name, sel in (("worst", order[-6:]), ("median", order[len(order) // 2 - 3: len(order) // 2 + 3])):
    for i in sel:
        c = conditioning(targets[i])
        print("  %-6s body %8d: err %.3e  conditioning %8.1f  err / (conditioning * 6e-8) = %.2f" % (name, targets[i], err[i], c, err[i] / (c * 6e-8)))
