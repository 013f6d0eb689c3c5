"""The same collision run (N = 1 M, Barnes-Hut) twice, once with the step that leaves the bodies in the order of its
own sort and once with NB_KEEP_SLOTS=1 (the bodies keep their slots), each in a process of its own: how many values
of positions, velocities, accelerations and the per-body constants differ after 1, 2, 5 and 50 steps, and by how
much (the constants must not differ at all; the rest differs by FP32 summation order only)."""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
if len(sys.argv) > 1:
    from conftest import load_package
    nb = load_package()
    n = 1000000
    b = nb.init_scenario(nb.COLLISION, n, 12345)
    out = {}
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        done = 0
        for upto in (1, 2, 5, 50):
            sim.update(upto - done); done = upto
            r = sim.read_bodies()
            out["pos%d" % upto] = r["pos"]; out["vel%d" % upto] = r["vel"]; out["acc%d" % upto] = r["acc"]
            out["mass%d" % upto] = r["mass"]; out["radius%d" % upto] = r["radius"]
    np.savez("/tmp/rc_%s.npz" % sys.argv[1], **out)
    sys.exit(0)
env = dict(os.environ)
subprocess.check_call([sys.executable, __file__, "resort"], env=env)
env["NB_KEEP_SLOTS"] = "1"
subprocess.check_call([sys.executable, __file__, "keep"], env=env)
a = np.load("/tmp/rc_resort.npz"); b = np.load("/tmp/rc_keep.npz")
for k in a.files:
    x, y = a[k], b[k]
    d = np.abs(x - y)
    print("%-9s differing %8d of %d   max |diff|/max|v| %.3e" % (k, int((x != y).sum()), x.size, d.max() / np.abs(y).max()))
