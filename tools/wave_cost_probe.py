#!/usr/bin/env python3
"""What the waves of one traversal launch cost (NB_DBG_WAVE_COST: 10 ns ticks per group of 64 consecutive sorted
bodies), how unevenly, and what the order they are dispatched in makes of it: a list-scheduling model of the
launch (8 XCDs of 1024 wave slots, blocks dealt round-robin as the hardware does) for the order the kernel uses
now, for the heaviest waves first, and for two tiers (heavy ones first, the rest as now).
Usage: python tools/wave_cost_probe.py [N] [steps ...]"""
import heapq
import importlib
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
nb = importlib.import_module("n-body-simulation_amd")
CHUNK = 32
SLOTS = 1024  # wave slots per XCD: 32 CUs x 4 SIMDs x 8


def logical(orig, nblocks):
    if orig >= nblocks - nblocks % (8 * CHUNK):
        return orig
    x, q = orig & 7, orig >> 3
    return ((q // CHUNK) * 8 + x) * CHUNK + q % CHUNK


def makespan(cost, wave_of_block):
    """Every XCD takes its blocks (blockIdx % 8) in blockIdx order, a block starts when a slot is free."""
    n = len(wave_of_block)
    worst = 0.0
    for x in range(8):
        free = [0.0] * SLOTS
        heapq.heapify(free)
        for b in range(x, n, 8):
            t = heapq.heappop(free)
            heapq.heappush(free, t + cost[wave_of_block[b]])
        worst = max(worst, max(free))
    return worst


def main():  # noqa: C901
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 5_000_000
    at = [int(v) for v in sys.argv[2:]] or [10, 300]
    sim = nb.NBodySim(n, nb.BARNES_HUT)
    sim.set_bodies(nb.init_scenario(nb.COLLISION, n, 12345))
    done = 0
    for step in at:
        sim.update(step - done)
        done = step
        sim.sync()
        nw = (n + 63) // 64
        c = sim.debug(nb.DBG_WAVE_COST, np.uint32, nw).astype(np.float64) * 0.01  # microseconds
        mean = c.mean()
        print("step %d: %d waves, mean %.1f us, median %.1f, p90 %.1f, p99 %.1f, p99.9 %.1f, max %.1f; sum / 8192 slots = %.1f us"
              % (step, nw, mean, np.median(c), np.percentile(c, 90), np.percentile(c, 99), np.percentile(c, 99.9), c.max(),
                 c.sum() / 8192), flush=True)
        for f in (1.5, 2, 3, 4):
            print("   above %.1f x mean: %.2f %% of the waves, %.1f %% of the time" %
                  (f, 100.0 * (c > f * mean).mean(), 100.0 * c[c > f * mean].sum() / c.sum()))
        now = [logical(b, nw) for b in range(nw)]
        by_cost = list(np.argsort(-c))
        print("   model: as dispatched now %.1f us; heaviest first %.1f us" % (makespan(c, now), makespan(c, by_cost)), flush=True)
        for f in (1.5, 2, 3):
            heavy = np.flatnonzero(c > f * mean)
            flag = np.zeros(nw, bool)
            flag[heavy] = True
            order = list(heavy) + [w for w in now if not flag[w]]
            print("   model: waves above %.1f x mean first (in index order), the rest as now: %.1f us" % (f, makespan(c, order)))
            order = list(heavy[np.argsort(-c[heavy])]) + [w for w in now if not flag[w]]
            print("          the same, the heavy ones by cost: %.1f us" % makespan(c, order), flush=True)
        # cheaper orders than a full sort: Q tiers of equal cost range, heaviest tier first, each tier in the order in
        # use now; and runs of 32 consecutive waves (what one XCD is dealt at a time) sorted by their mean cost
        for q in (4, 16, 64):
            lo, hi = c.min(), np.percentile(c, 99.9)
            tier = np.clip(((hi - c) / (hi - lo) * q).astype(int), 0, q - 1)  # 0 = heaviest
            nowa = np.array(now)
            order = list(nowa[np.argsort(tier[nowa], kind="stable")])
            print("   model: %d tiers by cost, heaviest first: %.1f us" % (q, makespan(c, order)))
        nch = nw // CHUNK
        chunk_cost = c[:nch * CHUNK].reshape(nch, CHUNK).mean(axis=1)
        chunks = np.argsort(-chunk_cost)
        # chunk j of the sorted list goes where logical chunk j goes now: dispatch position -> logical wave -> sorted chunk
        order = []
        for w in now:
            order.append(int(chunks[w // CHUNK]) * CHUNK + w % CHUNK if w < nch * CHUNK else w)
        print("   model: runs of %d waves sorted by their mean cost: %.1f us" % (CHUNK, makespan(c, order)), flush=True)
        # where along the order the heavy waves lie
        q = np.add.reduceat(c, np.arange(0, nw, max(1, nw // 16))) / c.sum()
        print("   share of the time per sixteenth of the order: " + " ".join("%.3f" % v for v in q[:16]))
    sim.close()


if __name__ == "__main__":
    main()
