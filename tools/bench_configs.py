"""Times the reference benchmark configurations (N = 65536 direct, 1M and 5M Barnes-Hut, 1M direct, 50M Barnes-Hut) on one GPU; `d3` = the same shapes with three-dimensional bodies (tuning/record aid, not a test)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import load_package  # noqa: E402

nb = load_package()


def run(alg, scenario, n, steps, warm=2):
    t = time.perf_counter()
    b = nb.init_scenario(scenario, n, 12345)
    tgen = time.perf_counter() - t
    with nb.NBodySim(n, alg) as sim:
        t = time.perf_counter()
        sim.set_bodies(b)
        tset = time.perf_counter() - t
        sim.update(warm)
        sim.sync()
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter()
        sim.update(steps)
        sim.sync()
        dt = (time.perf_counter() - t) / steps
        st = sim.stage_ms()
        t = time.perf_counter()
        out = sim.read_bodies()
        tread = time.perf_counter() - t
    assert np.all(np.isfinite(out["pos"]))
    name = {0: "direct", 1: "BH"}[alg] + " " + {0: "spiral", 1: "random", 2: "collision", 3: "solar"}[scenario]
    extra = "%.0f Ginter/s" % (n * n / dt / 1e9) if alg == 0 else " ".join("%s=%.2f" % kv for kv in st.items() if kv[1] > 0)
    print("%-20s N=%-9d %9.3f ms/step  %s  (gen %.1fs set %.1fs read %.2fs)" % (name, n, dt * 1e3, extra, tgen, tset, tread), flush=True)


def run3(alg, scenario, n, steps, thickness, warm=2):
    """The same configurations for three-dimensional bodies (discs of the given half-thickness)."""
    b = nb.init_scenario3(scenario, n, 12345, thickness)
    with nb.NBodySim3(n, alg) as sim:
        sim.set_bodies(b)
        sim.update(warm)
        sim.sync()
        sim.set_stage_timing(True, counters=False)
        t = time.perf_counter()
        sim.update(steps)
        sim.sync()
        dt = (time.perf_counter() - t) / steps
        st = sim.stage_ms()
        if alg == 1:
            sim.set_stage_timing(False, counters=True)
            sim.update(1)
            c = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
            cut = sim.debug(nb.DBG_KEY_LEVELS, np.uint32, 2)
        out = sim.read_bodies()
    assert np.all(np.isfinite(out["pos"]))
    name = "3-D " + {0: "direct", 1: "BH"}[alg] + " " + {0: "spiral", 1: "random", 2: "collision"}[scenario]
    if alg == 0:
        extra = "%.0f Ginter/s" % (n * n / dt / 1e9)
    else:
        extra = " ".join("%s=%.2f" % kv for kv in st.items() if kv[1] > 0)
        extra += " | evals %.3g accepted %.3g nodes %d levels %d closed %d stack %d" % (c[0], c[1], c[2], c[3], cut[0], c[7])
    print("%-22s N=%-9d thick %.0e %9.3f ms/step  %s" % (name, n, thickness, dt * 1e3, extra), flush=True)


def ranks(world, n, steps, cells, scenario=2, alg=None):
    """Per-rank GPU time of a run cut over `world` ranks, measured with `world` handles on this one
    GPU (the collectives are done through the host between the parts of a step and are not timed):
    what each rank's kernels cost, stage by stage, rank 0 and the slowest rank."""
    import test_gpu_partition_edge as pe

    hip = pe._hip()
    b = nb.init_scenario(scenario, n, 12345)
    alg = nb.BARNES_HUT if alg is None else alg
    sims = [nb.NBodySim(n, alg) for _ in range(world)]
    try:
        for r, s in enumerate(sims):
            s.set_bodies(b)
            s.set_partition(r, world)
            if cells:
                s.set_exchange(cells)

        def gather_positions():
            bufs = [s.exchange_buffer() for s in sims]
            chunk, rb = bufs[0][1], bufs[0][2]
            for src in range(world):
                lo, hi = min(n, src * chunk), min(n, (src + 1) * chunk)
                for dst in range(world):
                    if dst != src and hi > lo:
                        assert hip.hipMemcpy(bufs[dst][0] + lo * rb, bufs[src][0] + lo * rb, (hi - lo) * rb, 3) == 0
            assert hip.hipDeviceSynchronize() == 0

        def step():
            level = -1
            if cells:
                said = [s.cells_level() for s in sims]
                assert all(x == said[0] for x in said), said
                level, gather = said[0]
                if gather:
                    gather_positions()
                if level != step.level:
                    print("  step %d: cell level %d -> %d" % (step.count, step.level, level), flush=True)
                    step.level = level
            step.count += 1
            if level >= 0:
                for part, op in ((0, np.maximum), (1, np.add)):
                    for s in sims:
                        s.step_cells(part)
                        s.sync()  # one rank's kernels at a time: they would share the GPU otherwise
                    pe.reduce_between_ranks(hip, sims, part, op)
                for s in sims:
                    s.step_cells(2)
                    s.sync()
            else:
                for s in sims:
                    s.step_local()
                    s.sync()
                gather_positions()
                for s in sims:
                    s.step_finish()

        step.level, step.count = -2, 0

        for _ in range(3):
            step()
        for s in sims:
            s.sync()
            s.set_stage_timing(True, counters=False)
        for _ in range(steps):
            step()
        for s in sims:
            s.sync()
        per = [s.stage_ms() for s in sims]
        if alg == nb.BARNES_HUT:
            # what the waves of the last traversal of rank 0 and of the last rank took
            for r in (0, world - 1):
                nw = ((n + world - 1) // world + 63) // 64
                c = sims[r].debug(nb.DBG_WAVE_COST, np.uint32, nw).astype(np.float64) * 0.01
                c = c[c > 0]
                print("  rank %d: %d waves, mean %.1f us, median %.1f, p99 %.1f, max %.1f; sum / 8192 slots %.1f us" %
                      (r, len(c), c.mean(), np.median(c), np.percentile(c, 99), c.max(), c.sum() / 8192))
                print("    slowest (index: us): " + ", ".join("%d: %.0f" % (i, c[i]) for i in np.argsort(-c)[:8]))
    finally:
        for s in sims:
            s.close()
    tot = [sum(v for k, v in st.items() if k != "exchange") for st in per]
    worst = int(np.argmax(tot))
    fmt = lambda st: " ".join("%s=%.3f" % kv for kv in st.items() if kv[1] > 0)  # noqa: E731
    print(("BH" if alg == nb.BARNES_HUT else "direct sum") + " %s N=%d world=%d %s: rank 0 %.3f ms [%s]; slowest rank %d %.3f ms [%s]" % (
        {0: "spiral", 2: "collision"}[scenario], n, world,
        {0: "position exchange", 1: "cell exchange", 2: "cell exchange (auto)"}[int(cells)],
        tot[0], fmt(per[0]), worst, tot[worst], fmt(per[worst])), flush=True)


which = sys.argv[1:] or ["c2", "c3", "c4", "c5"]
if "cells_long" in which:
    # the cell exchange over the stretch bench.py's late-state leg covers: the level of the exchanged
    # cells must keep up with the box (an error from nb_step_cells otherwise)
    ranks(2, 5000000, 330, nb.EXCHANGE_CELLS)
    ranks(2, 5000000, 330, nb.EXCHANGE_AUTO)
if "ranks8" in which:
    ranks(8, 5000000, 5, nb.EXCHANGE_CELLS)
    ranks(4, 5000000, 5, nb.EXCHANGE_CELLS)
if "ranks" in which:
    for world in (2, 4, 8):
        ranks(world, 5000000, 5, nb.EXCHANGE_CELLS)
        ranks(world, 5000000, 5, nb.EXCHANGE_POSITIONS)
if "ranks_direct" in which:
    # per-rank kernels of the direct sum cut over 2, 4, 8 ranks: force = k_run_boxes + k_direct + k_direct_finish
    # (the source split is what direct_set_split chooses for the slice), build = re-packing the other ranks' records
    for world in (2, 4, 8):
        ranks(world, 1000000, 3, 0, scenario=0, alg=nb.DIRECT)
if "d3" in which:
    run3(nb.DIRECT, nb.SPIRAL, 65536, 20, 5.0e10)
    run3(nb.BARNES_HUT, nb.SPIRAL, 1000000, 10, 5.0e10)
    run3(nb.BARNES_HUT, nb.COLLISION, 5000000, 10, 2.0e10)
    run3(nb.BARNES_HUT, nb.COLLISION, 5000000, 10, 1.0e11)
    run3(nb.DIRECT, nb.SPIRAL, 1000000, 3, 5.0e10, 1)
if "d3c" in which:
    run3(nb.BARNES_HUT, nb.COLLISION, 5000000, 10, 2.0e10)
if "c2long" in which:
    for steps in (200, 1000, 3000):
        run(nb.DIRECT, nb.SPIRAL, 65536, steps, 5)
if "c2" in which:
    run(nb.DIRECT, nb.SPIRAL, 65536, 20)
if "c3" in which:
    run(nb.BARNES_HUT, nb.SPIRAL, 1000000, 10)
if "c4" in which:
    run(nb.BARNES_HUT, nb.COLLISION, 5000000, 10)
if "c5" in which:
    run(nb.DIRECT, nb.SPIRAL, 1000000, 3, 1)
    run(nb.BARNES_HUT, nb.SPIRAL, 50000000, 3, 1)
