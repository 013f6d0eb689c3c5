"""Randomised parity sweep against the CPU oracle (not collected by pytest; run by hand on a GPU box):

    python tools/fuzz_parity.py [cases] [seed] [big]

("big": sizes 8 000 .. 50 000 instead of 2 .. 3001; the oracle's direct sum then takes seconds per case.)

Every case draws a scenario, a size, an opening angle, a cutoff, a softening length, the number of
dimensions, the algorithm and sometimes a rank of a partitioned run, and compares the library's accelerations
(and, for a whole system, a few steps) with the oracle's.  Prints the worst case per kind and exits non-zero when a bar (1e-5 direct, 1e-4 tree)
is missed.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package, rel_err  # noqa: E402
from oracle_lib import Oracle  # noqa: E402

nb = load_package()
orc = Oracle()
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
worst = {}
bad = 0
for case in range(cases):
    dims = int(rng.choice([2, 3]))
    alg = int(rng.choice([0, 1]))
    sim = int(rng.choice([0, 1, 2]))
    n = int(rng.choice([8000, 20000, 50000] if "big" in sys.argv[3:] else [2, 3, 17, 64, 65, 200, 777, 1500, 3001]))
    if sim == 2 and n < 4:
        n = 4
    theta = float(rng.choice([0.0, 0.2, 0.5, 0.5, 0.9, 1.5]))
    cth = float(rng.choice([0.0, 1.0e9, 1.0e10, 1.0e10, 5.0e10, 3.0e11]))
    eps = float(rng.choice([0.0, 0.5, 0.5, 1.0e7, 3.0e9]))
    if cth == 0.0:
        # without a cutoff a pair may be closer than the float coordinates resolve (1e-7 of the extent);
        # a softening length well above that keeps such a pair from dominating a body's acceleration
        eps = 3.0e9
    seed = int(rng.integers(1, 1 << 30))
    thick = float(rng.choice([0.0, 1.0e10, 1.0e11, 4.0e11]))
    po = orc.params(theta=theta, collision_th=cth, eps=eps)
    pn = nb.default_params(theta=theta, collision_th=cth, eps=eps)
    if dims == 2:
        b = orc.scenario(sim, n, seed)
        Sim = nb.NBodySim
        accel = (lambda x: orc.direct_accel(x, po)) if alg == 0 else (lambda x: orc.bh_accel(x, po))
        step = (lambda x, k: orc.direct_step(x, k, po)) if alg == 0 else (lambda x, k: orc.bh_step(x, k, po))
    else:
        b = orc.scenario3(sim, n, seed, thickness=thick)
        Sim = nb.NBodySim3
        accel = (lambda x: orc.direct_accel3(x, po)) if alg == 0 else (lambda x: orc.bh_accel3(x, po))
        step = (lambda x, k: orc.direct_step3(x, k, po)) if alg == 0 else (lambda x, k: orc.bh_step3(x, k, po))
    shift = rng.choice([0.0, 0.0, 7.0e12]) * rng.uniform(-1, 1, size=dims)
    b["pos"] += shift
    try:
        ref = accel(b)
    except ValueError:
        continue  # coincident bodies: the reference's tree does not terminate
    # one case in three: a rank of a partitioned run (its own slots' accelerations; a rank alone cannot step)
    world = int(rng.choice([1, 1, 2, 5, 16]))
    rank = int(rng.integers(0, world))
    dyn = b["is_dynamic"] != 0
    with Sim(len(b), alg, pn) as s:
        s.set_bodies(b)
        if world > 1:
            s.set_partition(rank, world)
            chunk = (len(b) + world - 1) // world
            own = np.zeros(len(b), bool)
            own[s.order()[min(len(b), rank * chunk):min(len(b), (rank + 1) * chunk)]] = True
            dyn = dyn & own
            acc = s.compute_accel()
            out = None
        else:
            acc = s.compute_accel()
            s.update(3)
            out = s.read_bodies()
    err = rel_err(acc[dyn], ref[dyn]).max() if dyn.any() else 0.0
    dpos = 0.0
    if out is not None:
        want = step(b, 3)
        dpos = np.abs(out["pos"] - want["pos"]).max() / 5.0e11
    tol = 1e-5 if alg == 0 else 1e-4
    key = (dims, alg)
    desc = "dims %d alg %d sim %d n %d theta %.2g cth %.2g eps %.2g thick %.2g shift %s seed %d rank %d of %d" % (
        dims, alg, sim, len(b), theta, cth, eps, thick if dims == 3 else 0.0, np.array2string(shift, precision=2), seed,
        rank, world)
    if err > worst.get(key, (0.0, ""))[0]:
        worst[key] = (err, desc)
    if "big" in sys.argv[3:]:
        print("case %d: acc %.2e dpos %.2e : %s" % (case, err, dpos, desc), flush=True)  # the oracle takes a while
    if not (err <= tol) or not (dpos <= 1e-6):
        bad += 1
        print("MISS acc %.3e dpos %.3e : %s" % (err, dpos, desc), flush=True)
    if case % 100 == 99:
        print("... %d cases, %d misses so far" % (case + 1, bad), flush=True)
for key in sorted(worst):
    print("worst dims %d alg %d: %.3e  (%s)" % (key[0], key[1], worst[key][0], worst[key][1]))
print("%d cases, %d misses" % (cases, bad))
sys.exit(1 if bad else 0)
