"""Partitions in which some ranks own no body at all.

chunk = ceil(n / world), so a rank's slice [rank * chunk, (rank + 1) * chunk) is empty whenever
(world - 1) * chunk >= n: the reference's 5-body solar system on 4 ranks gives slices of
2, 2, 1 and 0 bodies.  Every handle, the empty one included, must step without error and the
ranks together must reproduce the unpartitioned run (bar: positions within 1e-9 of the system's
extent after 5 steps; the runs differ only by FP32 summation order).
"""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _hip():
    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    return hip


def step_ranks_together(hip, ranks, n):
    """One step of `ranks` (handles partitioned 0..world-1 on one GPU) with the all-gather of
    positions done by device-to-device copies between their exchange buffers."""
    world = len(ranks)
    for s in ranks:
        s.step_local()
    for s in ranks:
        s.sync()
    bufs = [s.exchange_buffer() for s in ranks]
    chunk, rb = bufs[0][1], bufs[0][2]
    for src in range(world):
        lo, hi = min(n, src * chunk), min(n, (src + 1) * chunk)
        if hi <= lo:
            continue
        for dst in range(world):
            if dst != src:
                assert hip.hipMemcpy(bufs[dst][0] + lo * rb, bufs[src][0] + lo * rb, (hi - lo) * rb, 3) == 0
    assert hip.hipDeviceSynchronize() == 0
    for s in ranks:
        s.step_finish()


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("alg", ["direct", "bh"])
@pytest.mark.parametrize("n,world", [(5, 4), (49, 8)])
def test_ranks_with_an_empty_slice_step_like_the_others(nb, oracle, dims, alg, n, world):
    algc = nb.DIRECT if alg == "direct" else nb.BARNES_HUT
    if dims == 2:
        b = nb.init_scenario(nb.SOLAR if n == 5 else nb.SPIRAL, n, 3)
        make = lambda: nb.NBodySim(n, algc)  # noqa: E731
    else:
        b = nb.init_scenario3(nb.SOLAR if n == 5 else nb.SPIRAL, n, 3, 1.0e10)
        make = lambda: nb.NBodySim3(n, algc)  # noqa: E731
    chunk = (n + world - 1) // world
    assert (world - 1) * chunk >= n, "the case must hold a rank without bodies"
    hip = _hip()
    one = make()
    ranks = [make() for _ in range(world)]
    try:
        one.set_bodies(b)
        for r, s in enumerate(ranks):
            s.set_bodies(b)
            s.set_partition(r, world)
        for _ in range(5):
            one.update()
            step_ranks_together(hip, ranks, n)
        ref = one.read_bodies()
        outs = [s.read_bodies() for s in ranks]
        order = ranks[0].order()
    finally:
        one.close()
        for s in ranks:
            s.close()
    extent = np.abs(ref["pos"]).max()
    for o in outs:
        assert np.array_equal(o["pos"], outs[0]["pos"])
        assert np.linalg.norm(o["pos"] - ref["pos"], axis=1).max() <= 1e-9 * extent
    # every rank keeps the velocities of its own slots up to date
    for r, o in enumerate(outs):
        own = order[min(n, r * chunk):min(n, (r + 1) * chunk)]
        # (FP32 sums taken in another order: small components of a velocity differ by 1e-6 of the speed)
        assert np.allclose(o["vel"][own], ref["vel"][own], rtol=1e-6, atol=1e-6 * np.abs(ref["vel"]).max())


@pytest.mark.parametrize("dims", [2, 3])
def test_targets_that_left_their_ranks_run_of_the_sorted_order(nb, dims):
    """Seven ranks over ten tiles of the sorted order, thirty steps of two discs that pass through each other: a
    rank's targets are then one long run of the sorted order plus strays anywhere, and the list of targets gives
    every tile's targets groups of 64 of their own, padded (tiles shared by ranks, tiles with a single stray).
    The accelerations the ranks then compute must be those of a handle that is given the same state afresh."""
    n, world, steps = 20000, 7, 30
    if dims == 2:
        b = nb.init_scenario(nb.COLLISION, n, 11)
        make = lambda: nb.NBodySim(n, nb.BARNES_HUT)  # noqa: E731
    else:
        b = nb.init_scenario3(nb.COLLISION, n, 11, 2.0e10)
        make = lambda: nb.NBodySim3(n, nb.BARNES_HUT)  # noqa: E731
    hip = _hip()
    chunk = (n + world - 1) // world
    ranks = [make() for _ in range(world)]
    fresh = make()
    try:
        for r, s in enumerate(ranks):
            s.set_bodies(b)
            s.set_partition(r, world)
        for _ in range(steps):
            step_ranks_together(hip, ranks, n)
        order = ranks[0].order()
        own = [order[min(n, r * chunk):min(n, (r + 1) * chunk)] for r in range(world)]
        # the state the ranks hold together: positions from any of them, velocities from their owners
        state = ranks[0].read_bodies()
        for r, s in enumerate(ranks):
            o = s.read_bodies()
            assert np.array_equal(o["pos"], state["pos"])
            state["vel"][own[r]] = o["vel"][own[r]]
        # the accelerations of that state: by a handle of its own, which deals the slots anew (no strays, no list
        # of targets), and by the ranks as they are
        fresh.set_bodies(state)
        fresh.compute_accel()
        ref = fresh.read_bodies()["acc"]
        acc = np.zeros_like(ref)
        for r, s in enumerate(ranks):
            s.compute_accel()
            acc[own[r]] = s.read_bodies()["acc"][own[r]]
        # how far the bodies have wandered: some rank must have targets outside its run of the sorted order
        sorted_slot = ranks[0].debug(nb.DBG_SORTED_IDX, np.uint32, n)
    finally:
        fresh.close()
        for s in ranks:
            s.close()
    rank_of = np.minimum(sorted_slot // chunk, world - 1)
    assert np.count_nonzero(np.diff(rank_of.astype(np.int64)) != 0) > 4 * world, "no body has left its rank's run"
    size = np.linalg.norm(ref, axis=1)
    err = np.linalg.norm(acc - ref, axis=1) / np.maximum(size, 1e-3 * np.median(size))
    # (the same interactions summed in FP32 by other groups of 64: 1e-6 of the acceleration, more where it is a
    # small difference of large terms)
    assert np.median(err) <= 2e-6
    assert err.max() <= 3e-4


def test_partition_is_refused_once_stepped(nb):
    with nb.NBodySim(64, nb.DIRECT) as sim:
        sim.setup(nb.RANDOM, 1)
        sim.update()
        with pytest.raises(nb.NBodyError, match="stepped already"):
            sim.set_partition(0, 2)
        sim.setup(nb.RANDOM, 1)
        sim.set_partition(0, 2)
        sim.step_local()
        with pytest.raises(nb.NBodyError, match="in flight"):
            sim.set_partition(1, 2)


def gather_between_ranks(hip, ranks, n, which):
    """All-gather of the exchange (which = "pos") or velocity ("vel") buffers of handles that share one GPU."""
    bufs = [(s.exchange_buffer() if which == "pos" else s.velocity_buffer()) for s in ranks]
    chunk, rb = bufs[0][1], bufs[0][2]
    for s in ranks:
        s.sync()
    for src in range(len(ranks)):
        lo, hi = min(n, src * chunk), min(n, (src + 1) * chunk)
        for dst in range(len(ranks)):
            if dst != src and hi > lo:
                assert hip.hipMemcpy(bufs[dst][0] + lo * rb, bufs[src][0] + lo * rb, (hi - lo) * rb, 3) == 0
    assert hip.hipDeviceSynchronize() == 0


@pytest.mark.parametrize("dims", [2, 3])
@pytest.mark.parametrize("alg", ["direct", "bh"])
def test_partitioned_ranks_renew_their_order_together(nb, oracle, dims, alg):
    """Three ranks step a system, all-gather their velocities, renew the exchange order (every rank
    derives the same one from the same positions, on the device) and go on: the slots a rank owns
    afterwards hold other bodies than before.  The run must stay on the unpartitioned one's path."""
    n, world = 2500, 3
    algc = nb.DIRECT if alg == "direct" else nb.BARNES_HUT
    if dims == 2:
        b = oracle.scenario(0, n, 14)
        make = lambda: nb.NBodySim(n, algc)  # noqa: E731
    else:
        b = oracle.scenario3(0, n, 14, thickness=4.0e10)
        make = lambda: nb.NBodySim3(n, algc)  # noqa: E731
    hip = _hip()
    one = make()
    ranks = [make() for _ in range(world)]
    try:
        one.set_bodies(b)
        for r, s in enumerate(ranks):
            s.set_bodies(b)
            s.set_partition(r, world)
        for _ in range(30):
            one.update()
            step_ranks_together(hip, ranks, n)
        before = ranks[0].order().copy()
        gather_between_ranks(hip, ranks, n, "vel")
        for s in ranks:
            s.refresh_order()
        orders = [s.order() for s in ranks]
        for _ in range(10):
            one.update()
            step_ranks_together(hip, ranks, n)
        ref = one.read_bodies()
        outs = [s.read_bodies() for s in ranks]
    finally:
        one.close()
        for s in ranks:
            s.close()
    assert not np.array_equal(before, orders[0])
    for o in orders:
        assert np.array_equal(o, orders[0]) and np.array_equal(np.sort(o), np.arange(n))
    extent = np.abs(ref["pos"]).max()
    chunk = (n + world - 1) // world
    # 40 steps of FP32 sums taken in different orders (other 64-body groups, other run choices):
    # observed 1.4e-8 of the extent; a rank that lost a velocity would be off by 1e-3 and more
    for r, o in enumerate(outs):
        assert np.linalg.norm(o["pos"] - ref["pos"], axis=1).max() <= 1e-7 * extent
        own = orders[0][r * chunk:min(n, (r + 1) * chunk)]
        assert np.linalg.norm(o["vel"][own] - ref["vel"][own], axis=1).max() <= 0.1


def reduce_between_ranks(hip, ranks, which, op):
    """In-place reduction (op = np.maximum or np.add) of the cell-exchange buffer `which` of handles
    that share one GPU, through the host."""
    for s in ranks:
        s.sync()
    bufs = [s.cells_buffer(which) for s in ranks]
    count = bufs[0][1]
    assert all(b[1] == count for b in bufs)
    total = None
    for ptr, _ in bufs:
        h = np.empty(count, dtype=np.float64)
        assert hip.hipMemcpy(h.ctypes.data, ptr, 8 * count, 2) == 0
        total = h if total is None else op(total, h)
    for ptr, _ in bufs:
        assert hip.hipMemcpy(ptr, total.ctypes.data, 8 * count, 1) == 0
    assert hip.hipDeviceSynchronize() == 0


@pytest.mark.parametrize("scenario,n,world", [(0, 3001, 2), (2, 20000, 3), (0, 5, 4), (0, 200000, 4)])
def test_cell_exchange_reproduces_the_single_gpu_run(nb, oracle, scenario, n, world):
    """NB_EXCHANGE_CELLS: every rank sorts only its own bodies; the ranks add up one record per cell
    of the level at which the tree closes and build the same tree from the totals.  The accelerations
    of every rank's bodies must be the reference's (bar 5e-6, as on one GPU) and the run must stay on
    the single-GPU run's path."""
    b = nb.init_scenario(nb.SOLAR if n == 5 else scenario, n, 17)
    hip = _hip()
    one = nb.NBodySim(n, nb.BARNES_HUT)
    ranks = [nb.NBodySim(n, nb.BARNES_HUT) for _ in range(world)]
    steps = 6
    try:
        one.set_bodies(b)
        for r, s in enumerate(ranks):
            s.set_bodies(b)
            s.set_partition(r, world)
            s.set_exchange(nb.EXCHANGE_CELLS)
        order = ranks[0].order()
        chunk = (n + world - 1) // world
        for step in range(steps):
            state = one.read_bodies()
            one.update()
            for s in ranks:
                s.step_cells(0)
            reduce_between_ranks(hip, ranks, 0, np.maximum)
            for s in ranks:
                s.step_cells(1)
            reduce_between_ranks(hip, ranks, 1, np.add)
            for s in ranks:
                s.step_cells(2)
            if step == 0:
                # the accelerations the first step used, rank by rank, against the oracle's tree
                ref = oracle.bh_accel(state)
                for r, s in enumerate(ranks):
                    own = order[min(n, r * chunk):min(n, (r + 1) * chunk)]
                    acc = s.read_bodies()["acc"][own]
                    dyn = state["is_dynamic"][own] != 0
                    if dyn.any():
                        err = np.linalg.norm(acc[dyn] - ref[own][dyn], axis=1) / np.linalg.norm(ref[own][dyn], axis=1)
                        assert err.max() <= 5e-6, (r, err.max())
        ref = one.read_bodies()
        outs = [s.read_bodies() for s in ranks]
    finally:
        one.close()
        for s in ranks:
            s.close()
    extent = np.abs(ref["pos"]).max()
    for r, o in enumerate(outs):
        own = order[min(n, r * chunk):min(n, (r + 1) * chunk)]
        if len(own):  # (FP32 sums in another order over six steps: 2e-9 of the extent observed)
            assert np.linalg.norm(o["pos"][own] - ref["pos"][own], axis=1).max() <= 1e-8 * extent
            assert np.linalg.norm(o["vel"][own] - ref["vel"][own], axis=1).max() <= 0.1  # speeds are 1e4..1e5 m/s


def step_ranks_following_the_box(nb, hip, ranks, n, levels):
    """One step of handles that share one GPU by the protocol of nb_cells_level: the cell exchange
    while the library says it applies, the position exchange while it is suspended."""
    said = [s.cells_level() for s in ranks]
    assert all(x == said[0] for x in said), said
    level, gather = said[0]
    levels.append(level)
    if gather:
        gather_between_ranks(hip, ranks, n, "pos")
    if level < 0:
        step_ranks_together(hip, ranks, n)
        return
    for part, op in ((0, np.maximum), (1, np.add)):
        for s in ranks:
            s.step_cells(part)
        reduce_between_ranks(hip, ranks, part, op)
    for s in ranks:
        s.step_cells(2)


def test_cell_exchange_is_suspended_when_the_box_outgrows_it(nb, oracle):
    """A body leaving the system at speed makes the bounding box grow until the tree closes below the
    last level the cell records can be exchanged at: the library then has the ranks gather their
    positions once and exchange positions from there on.  The run must stay on the single-GPU run's
    path across the switch."""
    n, world, steps = 3001, 3, 200
    b = nb.init_scenario(nb.SPIRAL, n, 21)
    p = nb.default_params()
    extent = np.abs(b["pos"]).max()
    k = int(np.argmax(np.abs(b["pos"][:, 0])))
    b["vel"][k] = (np.sign(b["pos"][k, 0]) * 0.04 * extent / p.dt, 0.0)  # 2 % of the box width per step
    hip = _hip()
    one = nb.NBodySim(n, nb.BARNES_HUT)
    ranks = [nb.NBodySim(n, nb.BARNES_HUT) for _ in range(world)]
    levels = []
    try:
        one.set_bodies(b)
        for r, s in enumerate(ranks):
            s.set_bodies(b)
            s.set_partition(r, world)
            s.set_exchange(nb.EXCHANGE_CELLS)
        order = ranks[0].order()
        for _ in range(steps):
            one.update()
            step_ranks_following_the_box(nb, hip, ranks, n, levels)
        ref = one.read_bodies()
        outs = [s.read_bodies() for s in ranks]
    finally:
        one.close()
        for s in ranks:
            s.close()
    assert levels[0] >= 0 and levels[-1] == -1, levels[::16]
    assert sorted(set(levels), key=levels.index)[:-1] == sorted(x for x in set(levels) if x >= 0)  # level only rises
    chunk = (n + world - 1) // world
    extent = np.abs(ref["pos"]).max()
    for r, o in enumerate(outs):
        own = order[min(n, r * chunk):min(n, (r + 1) * chunk)]
        # after the switch every rank holds every position again.  200 steps of FP32 sums taken in
        # another order: 3e-7 of the extent observed (1.4e-8 after 40 steps); positions that missed
        # an exchange would be off by 1e-3 and more
        assert np.linalg.norm(o["pos"] - ref["pos"], axis=1).max() <= 3e-6 * extent
        # (speeds are 1e4..1e5 m/s; the same summation noise, 3 to 10 m/s observed depending on the order of the sums)
        assert np.linalg.norm(o["vel"][own] - ref["vel"][own], axis=1).max() <= 25.0


def test_cell_exchange_reports_a_box_that_outgrows_its_margin(nb):
    """The level of the exchanged cells allows the box to grow by a tenth within four steps.  A body
    leaving ten times faster than that makes some step's tree close below the exchanged level: that
    step's forces are not the reference's, and every rank is told so at the same later step."""
    n, world = 3001, 2
    b = nb.init_scenario(nb.SPIRAL, n, 23)
    extent = np.abs(b["pos"]).max()
    k = int(np.argmax(np.abs(b["pos"][:, 0])))
    b["vel"][k] = (np.sign(b["pos"][k, 0]) * 1.5 * extent / nb.default_params().dt, 0.0)
    hip = _hip()
    ranks = [nb.NBodySim(n, nb.BARNES_HUT) for _ in range(world)]
    failed_at = []
    try:
        for r, s in enumerate(ranks):
            s.set_bodies(b)
            s.set_partition(r, world)
            s.set_exchange(nb.EXCHANGE_CELLS)
        for step in range(12):
            try:
                said = [s.cells_level() for s in ranks]
            except nb.NBodyError as e:
                assert "grew by more than a tenth" in str(e)
                # the other rank fails at the same step
                for s in ranks[1:]:
                    with pytest.raises(nb.NBodyError, match="grew by more than a tenth"):
                        s.cells_level()
                failed_at.append(step)
                break
            assert said[0] == said[1] and said[0][0] >= 0
            for part, op in ((0, np.maximum), (1, np.add)):
                for s in ranks:
                    s.step_cells(part)
                reduce_between_ranks(hip, ranks, part, op)
            for s in ranks:
                s.step_cells(2)
    finally:
        for s in ranks:
            s.close()
    assert failed_at and failed_at[0] <= 8, failed_at


def test_auto_exchange_prefers_positions_for_a_small_system(nb):
    """NB_EXCHANGE_AUTO: 3000 positions are far less data than the records of 65 536 cells, so the
    cell exchange starts suspended; nb_step_cells refuses, nb_step_local works."""
    with nb.NBodySim(3000, nb.BARNES_HUT) as sim:
        sim.setup(nb.SPIRAL, 2)
        sim.set_partition(0, 2)
        sim.set_exchange(nb.EXCHANGE_AUTO)
        assert sim.cells_level() == (-1, False)
        with pytest.raises(nb.NBodyError, match="suspended"):
            sim.step_cells(0)
        sim.step_local()
        sim.step_finish()


def test_cell_exchange_call_order(nb):
    with nb.NBodySim(100, nb.BARNES_HUT) as sim:
        sim.setup(nb.SPIRAL, 1)
        with pytest.raises(nb.NBodyError, match="not switched on"):
            sim.step_cells(0)
        sim.set_partition(0, 2)
        sim.set_exchange(nb.EXCHANGE_CELLS)
        with pytest.raises(nb.NBodyError, match="0, 1, 2"):
            sim.step_cells(1)
        with pytest.raises(nb.NBodyError, match="nb_step_cells"):
            sim.step_local()
    with nb.NBodySim(100, nb.DIRECT) as sim:
        sim.setup(nb.SPIRAL, 1)
        with pytest.raises(nb.NBodyError, match="Barnes-Hut"):
            sim.set_exchange(nb.EXCHANGE_CELLS)
    with nb.NBodySim(100, nb.BARNES_HUT, nb.default_params(collision_th=1.0e7)) as sim:
        sim.setup(nb.SPIRAL, 1)
        with pytest.raises(nb.NBodyError, match="closes below level"):
            sim.set_exchange(nb.EXCHANGE_CELLS)
