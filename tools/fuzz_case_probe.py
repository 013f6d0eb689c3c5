"""One case of tools/fuzz_parity.py looked at body by body (the 3-D tree case of seed 273587890: theta 1.5, cutoff
3e11 m, n 3001, which misses the 1e-4 bar at one body): the worst body of the whole system and of three ranks of
partitioned runs, that body's error in the whole-system run, and how its tree acceleration compares with the
direct sum's -- to tell an error of the target lists from an ill-conditioned sum."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from conftest import load_package, rel_err
from oracle_lib import Oracle
nb = load_package(); orc = Oracle()
theta, cth, eps, seed, n = 1.5, 3.0e11, 0.5, 273587890, 3001
po = orc.params(theta=theta, collision_th=cth, eps=eps)
pn = nb.default_params(theta=theta, collision_th=cth, eps=eps)
b = orc.scenario3(1, n, seed, thickness=0.0)
ref = orc.bh_accel3(b, po)
dref = orc.direct_accel3(b, po)
dyn = b["is_dynamic"] != 0
with nb.NBodySim3(n, 1, pn) as s:
    s.set_bodies(b)
    whole = s.compute_accel()
    order = s.order()
errs_whole = rel_err(whole, ref)
print("whole: max err %.3e at body %d" % (errs_whole[dyn].max(), int(np.argmax(np.where(dyn, errs_whole, 0)))))
for world, rank in ((16, 7), (5, 2), (2, 1)):
    with nb.NBodySim3(n, 1, pn) as s:
        s.set_bodies(b)
        s.set_partition(rank, world)
        acc = s.compute_accel()
        chunk = (n + world - 1) // world
        own = np.zeros(n, bool); own[s.order()[rank*chunk:min(n,(rank+1)*chunk)]] = True
    e = rel_err(acc, ref)
    m = own & dyn
    i = int(np.argmax(np.where(m, e, 0)))
    print("rank %d of %d: max err %.3e at body %d; that body's error in the whole-system run %.3e; |acc| %.3e vs |direct| %.3e; tree-vs-direct %.3e" %
          (rank, world, e[m].max(), i, errs_whole[i], np.linalg.norm(ref[i]), np.linalg.norm(dref[i]), rel_err(ref[i:i+1], dref[i:i+1])[0]))
    # difference between the partitioned and the whole-system result for that body, relative
    print("    partitioned vs whole for that body: %.3e" % (np.linalg.norm(acc[i]-whole[i])/np.linalg.norm(whole[i])))
