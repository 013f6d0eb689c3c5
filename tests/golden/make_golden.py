"""Generates the golden vectors in this directory from the REFERENCE's own CPU classes.

Run in the build container (needs oracle/_ref/libnbody_ref.so, which oracle/Makefile
compiles from the reference checkout):

    python tests/golden/make_golden.py

Each .npz holds plain float64/uint8 arrays (no pickles):
    is_dynamic, mass, radius, pos0, vel0      the initial state
    acc_direct, acc_bh                        accelerations of that state (DirectSum / BarnesHut)
    pos_direct_1, vel_direct_1, pos_bh_1, vel_bh_1   (solar only) state after 1 step
    pos_direct_100, vel_direct_100            state after 100 DirectSum::update()
    pos_bh_100, vel_bh_100                    state after 100 BarnesHut::update()

The initial states of spiral/random/solar come from the reference's NBody constructor with
srand pinned to the seed.  The CPU program has no collision scenario; its initial state is
written by the restatement of BarnesHutCuda::initCollideGalaxy in oracle/nbody_oracle.c
and then stepped by the reference's classes like the others.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle_lib import Oracle, Reference  # noqa: E402

SEED = 12345


def record(ref, b0, with_step1=False):
    out = {
        "is_dynamic": b0["is_dynamic"].copy(),
        "mass": b0["mass"].copy(),
        "radius": b0["radius"].copy(),
        "pos0": b0["pos"].copy(),
        "vel0": b0["vel"].copy(),
    }
    for alg, name in ((0, "direct"), (1, "bh")):
        s1 = ref.step(alg, b0, 1)
        out["acc_" + name] = s1["acc"].copy()
        if with_step1:
            out["pos_%s_1" % name] = s1["pos"].copy()
            out["vel_%s_1" % name] = s1["vel"].copy()
        s100 = ref.step(alg, b0, 100)
        out["pos_%s_100" % name] = s100["pos"].copy()
        out["vel_%s_100" % name] = s100["vel"].copy()
    return out


def main():
    ref = Reference()
    orc = Oracle()
    np.savez(os.path.join(HERE, "solar_n5.npz"), **record(ref, ref.scenario(2, 5, SEED), with_step1=True))
    for n in (256, 4096):
        for sim_cpu, name in ((0, "spiral"), (1, "random")):
            b0 = ref.scenario(sim_cpu, n, SEED)
            np.savez(os.path.join(HERE, "%s_n%d.npz" % (name, n)), **record(ref, b0))
            print("wrote", name, n, flush=True)
        b0 = orc.scenario(2, n, SEED)
        np.savez(os.path.join(HERE, "collision_n%d.npz" % n), **record(ref, b0))
        print("wrote collision", n, flush=True)


if __name__ == "__main__":
    main()
