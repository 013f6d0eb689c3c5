import os, sys, subprocess
import numpy as np
sys.path.insert(0, "/root/repo/tests")
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
