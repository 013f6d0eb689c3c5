"""Exploratory GPU run (not collected by pytest): prints parity numbers of the HIP path
against the golden vectors and the oracle, and rough timings."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import bodies_from_golden, golden, load_package, rel_err  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

nb = load_package()
orc = Oracle()


def stats(name, e):
    e = np.asarray(e)
    print("%-40s median %.2e  p99 %.2e  max %.2e  (>1e-5: %d, >1e-4: %d of %d)" % (
        name, np.median(e), np.percentile(e, 99), e.max(), (e > 1e-5).sum(), (e > 1e-4).sum(), len(e)), flush=True)


for name in ("solar_n5", "spiral_n256", "random_n256", "collision_n256", "spiral_n4096", "random_n4096", "collision_n4096"):
    g = golden(name)
    b = bodies_from_golden(g)
    dyn = b["is_dynamic"] != 0
    for alg, key in ((nb.DIRECT, "direct"), (nb.BARNES_HUT, "bh")):
        with nb.NBodySim(len(b), alg) as sim:
            sim.set_bodies(b)
            acc = sim.compute_accel()
            stats("%s %s accel" % (name, key), rel_err(acc[dyn], g["acc_" + key][dyn]))
            sim.update(100)
            out = sim.read_bodies()
            scale = np.linalg.norm(g["pos_%s_100" % key], axis=1).max()
            dp = np.linalg.norm(out["pos"] - g["pos_%s_100" % key], axis=1) / scale
            print("    100 steps: max |dpos|/extent %.2e   max rel dvel %.2e" % (
                dp.max(), rel_err(out["vel"][dyn], g["vel_%s_100" % key][dyn]).max()), flush=True)

for n, sim_id in ((65536, nb.SPIRAL), (1000000, nb.SPIRAL)):
    b = nb.init_scenario(sim_id, n)
    rng = np.random.default_rng(1)
    targets = np.sort(rng.choice(n - 1, 512, replace=False)).astype(np.int32)
    t = time.time()
    ref = orc.direct_accel_list(b, targets)
    print("oracle direct %d targets x %d: %.1fs" % (len(targets), n, time.time() - t), flush=True)
    with nb.NBodySim(n, nb.DIRECT) as sim:
        sim.set_bodies(b)
        t = time.time()
        acc = sim.compute_accel()
        t1 = time.time() - t
        stats("direct N=%d accel (512 targets)" % n, rel_err(acc[targets], ref))
        sim.sync()
        t = time.time()
        sim.update(3)
        sim.sync()
        dt = (time.time() - t) / 3
        print("    direct N=%d: %.3f ms/step, %.1f G interactions/s (first call %.3fs)" % (n, dt * 1e3, n * n / dt / 1e9, t1), flush=True)
    t = time.time()
    refbh, st = orc.bh_accel(b, with_stats=True)
    print("oracle BH N=%d: %.1fs; visits/body %.0f accepted/body %.0f depth %d nodes %d" % (
        n, time.time() - t, st[0] / n, st[1] / n, st[2], st[3]), flush=True)
    with nb.NBodySim(n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        sim.set_stage_timing(True)
        acc = sim.compute_accel()
        dyn = b["is_dynamic"] != 0
        stats("BH N=%d accel vs oracle BH" % n, rel_err(acc[dyn], refbh[dyn]))
        print("    stage ms", sim.stage_ms(), flush=True)
        cnt = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
        print("    evals/body %.0f accepted/body %.0f nodes %d levels %d" % (cnt[0] / n, cnt[1] / n, cnt[2], cnt[3]), flush=True)
        sim.set_stage_timing(False)
        sim.update(2)
        sim.sync()
        t = time.time()
        sim.update(5)
        sim.sync()
        print("    BH N=%d: %.3f ms/step" % (n, (time.time() - t) / 5 * 1e3), flush=True)

n = 5000000
b = nb.init_scenario(nb.COLLISION, n)
with nb.NBodySim(n, nb.BARNES_HUT) as sim:
    sim.set_bodies(b)
    sim.set_stage_timing(True)
    sim.update(1)
    print("BH N=5M collision stage ms", sim.stage_ms(), flush=True)
    cnt = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
    print("    evals/body %.0f accepted/body %.0f nodes %d levels %d" % (cnt[0] / n, cnt[1] / n, cnt[2], cnt[3]), flush=True)
    sim.set_stage_timing(False)
    sim.update(2)
    sim.sync()
    t = time.time()
    sim.update(5)
    sim.sync()
    print("    BH N=5M: %.3f ms/step" % ((time.time() - t) / 5 * 1e3), flush=True)
