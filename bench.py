#!/usr/bin/env python3
"""Benchmark of the N-body hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

Headline (BASELINE.json's metric, the configuration the reference publishes a number for):
Barnes-Hut ms/step, galaxy collision, N = 5,000,000, theta = 0.5 -- the reference's README
quotes 1509 ms/iteration for it on an unnamed NVIDIA GPU.  A "step" is one full update of all
bodies: bounding box, keys, sort, tree build, centre of mass, traversal + integration.  Bodies
are resident in HBM before the timed region.  The second half of the metric, direct-sum
G interactions/s, is measured in the same run (N = 65,536 and N = 1,000,000) and reported in
"direct_sum".  "three_d" holds the same two measurements for three-dimensional bodies.

With N > 1 the driver starts one rank per GPU (torch.distributed.run); every rank holds all
bodies, integrates its slice and all-gathers positions over RCCL each step (strong scaling:
the total work is fixed).

Rank 0 prints ONE JSON line.
"""
import argparse
import importlib.util
import json
import math
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "n-body-simulation_amd")

PUBLISHED_MS = 1509.0           # reference README.md:18, GPU Barnes-Hut N = 5M per iteration
REHEARSAL = os.environ.get("NBODY_BENCH_REHEARSAL") == "1"
TRANSPORT = {"used": None}      # who issued the collectives of the multi-GPU legs
HBM_PEAK_GBS = 8000.0           # MI355X HBM3E, spec
VALU_PEAK_TFLOPS = 157.3        # MI355X FP32 vector peak, spec (256 CU x 128 lanes x 2 x 2.4 GHz)
FLOP_PER_INTERACTION = 20       # GPU Gems 3 ch. 31 convention
BH_FORCE_BYTES_PER_BODY = 128   # traversal + integrate: own state 32 r + 32 w, each of ~2N 32-B nodes once
# A one-GPU step also leaves the bodies in the order of its sort (the traversal's epilogue copies every
# per-body constant to the body's new slot and writes the acceleration there): bytes beyond the
# algorithmic ones, moved on purpose, that show in the measured traffic
BH_REORDER_BYTES_PER_BODY = 65 + 16  # constants 28 r + 37 w; acceleration 16 w


def load_package():
    path = os.path.join(PKG, "__init__.py")
    spec = importlib.util.spec_from_file_location("nbody_amd", path, submodule_search_locations=[PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["nbody_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


PREWARM_S = 0.1


def prewarm(sim):
    """Brings the GPU's clocks to their steady state before a measurement: force evaluations of the
    simulation's own state (nb_compute_accel: nothing is advanced, the steps that follow are the same steps)
    for PREWARM_S seconds.  Setting the bodies leaves the GPU idle for a second or more while the host orders
    them; the first thirty steps after that run 5-15 % slower than the thirtieth (tools/clock_ramp_probe.py),
    and K timed steps after W warm-up steps would measure that transient instead of the step."""
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < PREWARM_S:
        for _ in range(4):
            sim.evaluate()
        sim.sync()


_SPARE = {}


def prewarm_beside(nb):
    """The same for a rank of a multi-GPU run, whose simulation cannot evaluate forces by itself (with the
    cell exchange it does not hold the other ranks' positions, and a loop bounded by time would issue a
    different number of collectives on every rank): PREWARM_S seconds of steps of a small direct-sum
    system of this rank's own, which touches neither the simulation nor the communicator."""
    if "sim" not in _SPARE:
        _SPARE["sim"] = nb.NBodySim(65536, nb.DIRECT)
        _SPARE["sim"].set_bodies(nb.init_scenario(nb.SPIRAL, 65536, 1))
    spare = _SPARE["sim"]
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < PREWARM_S:
        spare.update(8)
        spare.sync()


def barrier_sync(world):
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def timed_run(nb, alg, bodies, steps, warmup, rank, world, device, stage_timing=True, params=None, late_at=0,
              cells=False, warm_clocks=True):
    """Returns (seconds for `steps` steps, max over ranks; mean per-stage ms of this rank; traversal
    counters of rank 0; and, with late_at > 0 (one step number or several), the same timing repeated
    once the run has reached that step: {"ms_per_step", "stage_ms", "at_step"}, further steps as a
    list of such entries under "later")."""
    from nbody_amd.distributed import DistributedSim, GpuEngine

    sim = nb.NBodySim(len(bodies), alg, params)
    sim.set_bodies(bodies)
    if world > 1 and REHEARSAL:
        # ranks sharing one GPU cannot form an RCCL group: the Python stepping layer over gloo
        eng = GpuEngine(sim, device)
        stepper = DistributedSim(eng)
        if cells:
            eng.use_cell_exchange(nb.EXCHANGE_AUTO)
        step = lambda: stepper.update(1)  # noqa: E731
    elif world > 1:
        # the library's own communicator: the step's collectives (one in-place ncclAllGather of the
        # positions, or with the cell exchange two ncclAllReduce) are enqueued on the simulation's
        # stream between its kernels, no host synchronisation inside a step
        ok = 1
        try:
            cid = [nb.comm_unique_id() if rank == 0 else None]
        except nb.NBodyError as e:  # librccl.so.1 not loadable: the same on every rank of this node
            cid, ok = [None], 0
            print("bench.py: rank %d: %s" % (rank, e), file=sys.stderr)
        dist.broadcast_object_list(cid, src=0)
        if cid[0] is None:
            ok = 0
        if ok:
            try:
                sim.comm_init(rank, world, cid[0])
                if cells:
                    sim.set_exchange(nb.EXCHANGE_AUTO)
            except nb.NBodyError as e:
                ok = 0
                print("bench.py: rank %d: %s" % (rank, e), file=sys.stderr)
        flag = torch.tensor([ok], dtype=torch.int32, device=device)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()):
            TRANSPORT["used"] = "library (librccl.so.1 loaded by libnbody_hip.so)"
            step = lambda: sim.step_distributed(1)  # noqa: E731
        else:
            # some rank could not set the library's communicator up: the same protocol with the
            # collectives issued by torch.distributed (which is RCCL as well), said so in the line
            TRANSPORT["used"] = "torch.distributed (the library's own communicator failed, see stderr)"
            sim.close()
            sim = nb.NBodySim(len(bodies), alg, params)
            sim.set_bodies(bodies)
            eng = GpuEngine(sim, device)
            stepper = DistributedSim(eng)
            if cells:
                eng.use_cell_exchange(nb.EXCHANGE_AUTO)
            step = lambda: stepper.update(1)  # noqa: E731
    else:
        step = lambda: sim.update(1)  # noqa: E731
    if warm_clocks and PREWARM_S > 0:
        if world == 1:
            prewarm(sim)
        else:
            prewarm_beside(nb)
    for _ in range(warmup):
        step()

    def timed_block():
        # the K timed steps run without the stage timer (its seven event records per step cost about 1 % of a
        # 1 ms step); the per-stage times come from K further steps with the timer on
        sim.set_stage_timing(False, counters=False)
        barrier_sync(world)
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        barrier_sync(world)
        took = time.perf_counter() - t0
        per_stage = {}
        if stage_timing:
            sim.set_stage_timing(True, counters=False)
            for _ in range(steps):
                step()
            barrier_sync(world)
            per_stage = sim.stage_ms()  # HIP events recorded over these steps, read only now: mean per step
            sim.set_stage_timing(False, counters=False)
        return took, per_stage, steps * (2 if stage_timing else 1)

    dt, stages, ran = timed_block()
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    counters = None
    if alg == nb.BARNES_HUT and rank == 0:
        sim.set_stage_timing(False, counters=True)
        step()
        torch.cuda.synchronize()
        c = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
        counters = {"node_evaluations": int(c[0]), "accepted_interactions": int(c[1]), "nodes": int(c[2]),
                    "levels": int(c[3]), "fp64_redecisions": int(c[4]), "group_node_visits": int(c[5]),
                    "batches": int(c[6]), "max_stack": int(c[7])}
    elif alg == nb.BARNES_HUT and world > 1:
        step()  # keep the ranks' collectives matched with rank 0's extra step
    late = None
    done = warmup + ran + (1 if alg == nb.BARNES_HUT else 0)
    for at in sorted(a for a in (late_at if isinstance(late_at, (tuple, list)) else (late_at,)) if a > 0):
        if at <= done:
            continue
        sim.set_stage_timing(False, counters=False)
        for _ in range(at - done):
            step()
        ldt, lstages, lran = timed_block()
        if world > 1:
            t = torch.tensor([ldt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            ldt = float(t.item())
        done = at + lran
        entry = {"at_step": at, "ms_per_step": ldt * 1e3 / steps, "stage_ms": lstages}
        if late is None:
            late = entry
        else:
            late.setdefault("later", []).append(entry)
    barrier_sync(world)  # every rank's last collectives are done before the communicator goes
    sim.close()
    return dt, stages, counters, late


def three_d_leg(nb, bh_n, direct_n, steps, warmup):
    """The same two measurements for three-dimensional bodies (the nb3_ entry points; one GPU): the
    collision discs with a half-thickness of 2e10 m, and the spiral disc for the direct sum."""
    out = {"thickness_m": 2.0e10,
           "note": "the reference is two-dimensional: no published number and no reference run exists for these"}
    b = nb.init_scenario3(nb.COLLISION, bh_n, 12345, 2.0e10)
    with nb.NBodySim3(bh_n, nb.BARNES_HUT) as sim:
        sim.set_bodies(b)
        if PREWARM_S > 0:
            prewarm(sim)
        sim.update(warmup)
        torch.cuda.synchronize()
        sim.sync()
        t0 = time.perf_counter()
        sim.update(steps)
        sim.sync()
        dt = time.perf_counter() - t0
        sim.set_stage_timing(True, counters=False)  # (the stage times: from further steps, as in timed_run)
        sim.update(steps)
        sim.sync()
        stages = sim.stage_ms()
        sim.set_stage_timing(False, counters=True)
        sim.update(1)
        c = sim.debug(nb.DBG_COUNTERS, np.uint64, 8)
    out["barnes_hut"] = {"n": bh_n, "scenario": "collision", "ms_per_step": dt * 1e3 / steps, "stage_ms": stages,
                         "tree": {"node_evaluations": int(c[0]), "accepted_interactions": int(c[1]), "nodes": int(c[2]),
                                  "levels": int(c[3]), "group_node_visits": int(c[5]), "batches": int(c[6])}}
    del b
    b = nb.init_scenario3(nb.SPIRAL, direct_n, 12345, 2.0e10)
    with nb.NBodySim3(direct_n, nb.DIRECT) as sim:
        sim.set_bodies(b)
        if PREWARM_S > 0:
            prewarm(sim)
        sim.update(1)
        sim.sync()
        t0 = time.perf_counter()
        sim.update(3)
        sim.sync()
        dt = time.perf_counter() - t0
    out["direct_sum"] = {"n": direct_n, "scenario": "spiral", "ms_per_step": dt * 1e3 / 3,
                         "ginteractions_per_s": float(direct_n) * direct_n * 3 / dt / 1e9}
    return out


XGMI_LINK_GBS = 153.0   # one xGMI link, one direction (MI355X: 7 links per GPU, one to each peer of an 8-GPU node)
COLLECTIVE_LATENCY_MS = 0.03  # assumed fixed cost of one RCCL collective enqueued on a busy stream


def per_rank_kernels(nb, alg, bodies, ranks, steps, warm):
    """GPU time of ONE rank's kernels per step when the run is cut over `ranks` ranks, measured on this
    GPU playing rank 0 (nb_set_partition(0, ranks); nb_step_local / nb_step_finish): its slice of the
    targets against all sources, and for Barnes-Hut the whole tree (position exchange).  The other
    ranks' bodies stand still here -- nobody advances them -- which leaves the work per step what it
    is in a real run.  No collective runs: returns (ms per step summed over the stages, stage ms)."""
    import ctypes as C

    hip = C.CDLL("libamdhip64.so")
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    n = len(bodies)
    with nb.NBodySim(n, alg) as sim:
        sim.set_bodies(bodies)
        sim.set_partition(0, ranks)

        def step():
            cur, chunk, rb = sim.exchange_buffer()
            sim.step_local()
            nxt = sim.exchange_buffer()[0]
            if nxt != cur and n > chunk:
                # direct sum: the new positions go to the other buffer; the slices of the other ranks
                # (unchanged here) are what the all-gather would have delivered
                sim.sync()
                assert hip.hipMemcpy(nxt + chunk * rb, cur + chunk * rb, (n - chunk) * rb, 3) == 0
            sim.step_finish()

        for _ in range(warm):
            step()
        sim.set_stage_timing(True, counters=False)
        for _ in range(steps):
            step()
        st = sim.stage_ms()
    return sum(v for k, v in st.items() if k != "exchange"), {k: v for k, v in st.items() if v > 0 and k != "exchange"}


def projected_scaling(nb, legs):
    """What the first real multi-GPU run can be read against: per-rank kernel time MEASURED on this one
    GPU (per_rank_kernels), the step's all-gather ESTIMATED from the link speed, and the speed-up over
    the one-GPU step both give together.  legs: (name, alg, bodies, one-GPU ms/step, rank counts)."""
    out = {"note": "kernel_ms: measured here, one rank's kernels of a run cut over that many ranks (this GPU as rank 0, "
                   "position exchange: targets split, sources / tree replicated); allgather_ms: ESTIMATED, 16 B per body "
                   "in place over xGMI -- 'direct' = every peer's slice over its own link at %.0f GB/s, 'ring' = ranks - 1 "
                   "slices one after the other over one link -- plus %.2f ms per collective; no multi-GPU run stands "
                   "behind these lines" % (XGMI_LINK_GBS, COLLECTIVE_LATENCY_MS)}
    for name, alg, bodies, one_gpu_ms, rank_counts in legs:
        n = len(bodies)
        entry = {"n": n, "one_gpu_ms_per_step": one_gpu_ms, "ranks": {}}
        for p in rank_counts:
            steps, warm = (3, 1) if (alg == nb.DIRECT or n > 10000000) else (8, 3)
            ms, st = per_rank_kernels(nb, alg, bodies, p, steps, warm)
            slice_ms = 16.0 * n / p / (XGMI_LINK_GBS * 1e9) * 1e3
            direct_ms = slice_ms + COLLECTIVE_LATENCY_MS
            ring_ms = (p - 1) * slice_ms + COLLECTIVE_LATENCY_MS
            entry["ranks"][str(p)] = {"kernel_ms": ms, "stage_ms": st,
                                      "allgather_ms": {"direct": direct_ms, "ring": ring_ms},
                                      "projected_speedup": {"direct": one_gpu_ms / (ms + direct_ms),
                                                            "ring": one_gpu_ms / (ms + ring_ms)}}
        out[name] = entry
    return out


def cpu_baseline(nb, bh_n):
    """The reference's own CPU Barnes-Hut (its classes, compiled unmodified into oracle/_ref; the
    plain-C port when that library is absent) timed on this host: ONE real step of the workload's
    scenario at N = 1,000,000, single thread (the reference has no threading) -- nothing
    extrapolated in `value`.  Beside it: the port on all the host cores this job may use (OpenMP over
    the targets of the traversal), the N log N projection to the workload's N, and the CPU direct
    sum."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from oracle_lib import Oracle, Reference

    sample_n = min(1000000, bh_n)
    b = nb.init_scenario(nb.COLLISION, sample_n, 12345)
    orc = Oracle()
    if Reference.available():
        ref = Reference()
        kind = "reference"
        t0 = time.perf_counter()
        ref.step(1, b, 1)
        dt = time.perf_counter() - t0
    else:
        kind = "port"
        t0 = time.perf_counter()
        orc.bh_step(b, 1)
        dt = time.perf_counter() - t0
    # all cores: the share of the machine this job has (OMP_NUM_THREADS on the GPU box), not nproc
    cores = int(os.environ.get("OMP_NUM_THREADS", "0") or 0) or len(os.sched_getaffinity(0))
    cores = max(1, min(cores, len(os.sched_getaffinity(0))))
    orc.set_threads(cores)
    t0 = time.perf_counter()
    orc.bh_step(b, 1)
    dt_all = time.perf_counter() - t0
    bd = nb.init_scenario(nb.SPIRAL, 16384, 12345)
    t0 = time.perf_counter()
    orc.direct_step(bd, 1)
    td_all = time.perf_counter() - t0
    orc.set_threads(1)
    t0 = time.perf_counter()
    (ref.step(0, bd, 1) if kind == "reference" else orc.direct_step(bd, 1))
    td = time.perf_counter() - t0
    scale = (bh_n * math.log2(bh_n)) / (sample_n * math.log2(sample_n))
    return {
        "value": dt * 1e3,
        "unit": "ms/step at N=%d" % sample_n,
        "cores": 1,
        "kind": kind,
        "sample": "1 step of the reference CPU BarnesHut::update, galaxy collision N=%d (the workload's generator and "
                  "seed), single thread; measured, not extrapolated" % sample_n,
        "projected_to_workload_ms": dt * 1e3 * scale,
        "projection_rule": "N log2 N from N=%d to N=%d" % (sample_n, bh_n),
        "all_cores": {"value": dt_all * 1e3, "unit": "ms/step at N=%d" % sample_n, "cores": cores, "kind": "port",
                      "note": "plain-C port of the reference (oracle/), OpenMP over the traversal's targets; the tree "
                              "build stays on one thread as in the reference"},
        "direct_sum_ginteractions_per_s": 16384.0 * 16384.0 / td / 1e9,
        "direct_sum_all_cores_ginteractions_per_s": 16384.0 * 16384.0 / td_all / 1e9,
        "direct_sum_sample": "1 step of the CPU DirectSum::update, spiral N=16384",
        "host_cpus": os.cpu_count(),
    }


def launch_ranks(n):
    """Runs this script as n ranks of torch.distributed.run on this node; returns its exit code.
    The ranks' output (rank 0's JSON line) passes straight through."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--bh-n", type=int, default=5000000)
    ap.add_argument("--direct-n", type=int, default=1000000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-direct", action="store_true", help="skip the direct-sum half of the metric")
    ap.add_argument("--no-50m", action="store_true", help="skip the Barnes-Hut N = 50M leg (BASELINE.json configs[4])")
    ap.add_argument("--big-n", type=int, default=50000000)
    ap.add_argument("--headline-only", action="store_true",
                    help="only the headline run (and its late states): no GPU-tree mode, N = 50M, direct sum, 3-D, CPU or "
                         "projection legs -- what the counter passes of profiles/collect.sh run")
    ap.add_argument("--no-prewarm", action="store_true",
                    help="do not warm the clocks before the warm-up steps (the counter passes of profiles/collect.sh: "
                         "their first launches of the traversal are then the first steps)")
    ap.add_argument("--no-projection", action="store_true",
                    help="one GPU: skip the per-rank kernel times of a run cut over 2/4/8 ranks (projected_scaling)")
    ap.add_argument("--position-exchange", action="store_true",
                    help="N > 1: all-gather the positions every step instead of the cell exchange")
    ap.add_argument("--late-step", type=int, default=300, help="also time the headline run from this step on (0: no)")
    ap.add_argument("--later-step", type=int, default=620,
                    help="and once more from this step on (ejected bodies have stretched the box: the tree closes at level 11); "
                         "0 or --late-step 0: no")
    args = ap.parse_args()
    if args.no_prewarm:
        global PREWARM_S
        PREWARM_S = 0.0
    if args.headline_only:
        args.no_cpu = args.no_direct = args.no_50m = args.no_projection = True

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # Started without a launcher: start one rank per GPU ourselves.  Nothing in this process
        # has touched the GPU yet, and the ranks are fresh child processes.
        sys.exit(launch_ranks(args.gpus))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but the launcher started %d rank(s)" % (args.gpus, world))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP library has no CPU path")
    # NBODY_BENCH_REHEARSAL=1: every rank on GPU 0 and gloo instead of RCCL -- runs the multi-rank
    # control flow on a one-GPU box (RCCL wants one GPU per rank); its numbers mean nothing
    rehearsal = REHEARSAL
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=device)
    nb = load_package()

    # the ranks RCCL really connected: every rank contributes its number to an all-gather
    rccl_ranks = 1
    if world > 1:
        mine = torch.tensor([rank], dtype=torch.int64, device=device if not rehearsal else "cpu")
        seen = torch.empty(world, dtype=torch.int64, device=mine.device)
        dist.all_gather_into_tensor(seen, mine)
        rccl_ranks = int(torch.unique(seen).numel())
        assert rccl_ranks == dist.get_world_size() == world

    # ---- headline: Barnes-Hut, galaxy collision N = 5M ----
    bodies = nb.init_scenario(nb.COLLISION, args.bh_n, 12345)
    # More than one GPU: the ranks exchange one record per cell of the level at which the tree
    # closes (NB_EXCHANGE_CELLS) and each sorts only its own bodies; the mode BASELINE.json words
    # (all-gather of the positions, whole tree rebuilt on every GPU) is timed beside it.
    cells = world > 1 and not args.position_exchange
    dt, stages, counters, late = timed_run(nb, nb.BARNES_HUT, bodies, args.steps, args.warmup, rank, world, device,
                                           late_at=(args.late_step, args.later_step if args.late_step > 0 else 0),
                                           cells=cells)
    ms_per_step = dt * 1e3 / args.steps
    # the same K steps after the same W warm-up steps without warming the clocks first: what a run's first
    # steps take right after the bodies were uploaded
    cold = None
    if world == 1 and not args.headline_only:
        cdt, _, _, _ = timed_run(nb, nb.BARNES_HUT, bodies, args.steps, args.warmup, rank, world, device, stage_timing=False,
                                 warm_clocks=False)
        cold = {"ms_per_step": cdt * 1e3 / args.steps,
                "note": "steps W+1..W+K of a fresh handle without clock_prewarm: the GPU's clocks are still in the transient "
                        "that follows the idle upload of the bodies (tools/clock_ramp_probe.py)"}
    gathered = None
    if cells:
        pdt, pstages, _, _ = timed_run(nb, nb.BARNES_HUT, bodies, args.steps, args.warmup, rank, world, device)
        gathered = {"ms_per_step": pdt * 1e3 / args.steps, "stage_ms": pstages,
                    "note": "the same workload with NB_EXCHANGE_POSITIONS: one all-gather of all positions per step, "
                            "sort + tree + centres of mass repeated on every GPU"}
    projection_legs = []
    project = world == 1 and not args.no_projection
    if project:
        projection_legs.append(("barnes_hut_n%d" % args.bh_n, nb.BARNES_HUT, bodies, ms_per_step, (2, 4, 8)))
    proj = projected_scaling(nb, projection_legs) if project else None
    # the same workload on the tree of the reference's GPU program (depth-9 cap, leaves as point
    # masses): the configuration the published 1509 ms was measured with
    gdt, gstages, gcounters = 0.0, {}, None
    if not args.headline_only:
        gdt, gstages, gcounters, _ = timed_run(nb, nb.BARNES_HUT, bodies, args.steps, args.warmup, rank, world, device,
                                               params=nb.default_params(bucket_depth=9))
    del bodies

    # ---- BASELINE.json configs[4]: Barnes-Hut, spiral galaxy N = 50M (on 1..8 GPUs) ----
    big = None
    if not args.no_50m:
        bodies = nb.init_scenario(nb.SPIRAL, args.big_n, 12345)
        bdt, bstages, bcounters, _ = timed_run(nb, nb.BARNES_HUT, bodies, max(3, args.steps // 2), 2, rank, world, device,
                                               cells=cells)
        if project:
            proj.update({k: v for k, v in projected_scaling(
                nb, [("barnes_hut_n%d" % args.big_n, nb.BARNES_HUT, bodies, bdt * 1e3 / max(3, args.steps // 2), (8,))]).items()
                if k != "note"})
        del bodies
        big = {"n": args.big_n, "scenario": "spiral", "ms_per_step": bdt * 1e3 / max(3, args.steps // 2),
               "stage_ms": bstages, "tree": bcounters}

    # ---- second half of the metric: direct-sum G interactions/s ----
    direct = None
    if not args.no_direct:
        direct = {}
        # N = 65,536: a step takes 0.9 ms; after the idle stretch in which the bodies are made and
        # uploaded the GPU needs some 20 ms of work to be back at its steady clocks, so this leg
        # times 200 steps (5 warm-up steps): the ramp is inside the timed region and weighs 1/10
        for n, steps, warm in ((65536, max(args.steps, 200), 5), (args.direct_n, 3, 1)):
            b = nb.init_scenario(nb.SPIRAL, n, 12345)
            ddt, dstages, _, _ = timed_run(nb, nb.DIRECT, b, steps, warm, rank, world, device)
            gi = float(n) * n * steps / ddt / 1e9
            kern_ms = dstages.get("force", 0.0)
            entry = {"n": n, "steps": steps, "ms_per_step": ddt * 1e3 / steps, "ginteractions_per_s": gi,
                     "scenario": "spiral"}
            if world == 1 and kern_ms > 0:
                tf = float(n) * n * FLOP_PER_INTERACTION / (kern_ms * 1e-3) / 1e12
                entry["roofline"] = {"bound": "valu_fp32", "kernel": "k_direct", "achieved": tf,
                                     "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / VALU_PEAK_TFLOPS,
                                     "kernel_ms": kern_ms, "flop_per_interaction": FLOP_PER_INTERACTION,
                                     "traffic": None}
            direct["n%d" % n] = entry
            if project and n == args.direct_n:
                proj.update({k: v for k, v in projected_scaling(
                    nb, [("direct_sum_n%d" % n, nb.DIRECT, b, ddt * 1e3 / steps, (2, 4, 8))]).items() if k != "note"})

    if "sim" in _SPARE:
        _SPARE.pop("sim").close()
    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel of the headline step (the traversal) ----
    force_ms = stages.get("force", 0.0)
    per_rank_bodies = args.bh_n / world
    alg_bytes = BH_FORCE_BYTES_PER_BODY * per_rank_bodies
    achieved = alg_bytes / (force_ms * 1e-3) / 1e9 if force_ms > 0 else 0.0
    traffic = None
    l2_hit = None
    profiled = {}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and world == 1:
        try:
            profiled = json.load(open(tpath))
            measured = profiled.get("k_bh_force", {})
            traffic = measured.get("hbm_bytes_per_launch")
            l2_hit = measured.get("l2_hit_rate")
        except Exception:
            traffic = None

    def issue_roofline(counters_of, kernel_ms):
        """The kernel against the rate at which a SIMD issues instructions: the vector and LDS instructions
        the counter passes recorded per launch (profiles/traffic.json), priced in issue slots by the
        static mix of the kernel's hot loop (profiles/issue_census.py), over this run's kernel time."""
        census = profiled.get("issue_census") or {}
        got = profiled.get(counters_of) or {}
        if "slots_per_vector_instruction" not in census or "valu_wave_insts" not in got or not kernel_ms:
            return None
        # every LDS instruction takes one issue slot of its SIMD; what it costs beyond that (the LDS pipe's
        # own time) overlaps with the other waves' vector work to a degree that depends on the state
        vslots = got["valu_wave_insts"] * census["slots_per_vector_instruction"]
        lds = got.get("lds_wave_insts", 0.0)
        slots = vslots + lds
        # peak: a SIMD issues one wave64 vector instruction every two cycles (32 lanes a cycle: 157.3 TFLOP/s of FMA at
        # 2.4 GHz): 256 CUs x 4 SIMDs x 2.4e9 / 2.  The clock under this load cannot be read here; an FMA-only kernel
        # sustains 62 T lane-ops/s (bin/microbench, 8 waves per SIMD), 79 % of that rate, power-limited.
        peak = 256 * 4 * 2.4e9 / 2
        fma_rate = census["issue_rate_wave_slots_per_s"]
        rate = slots / (kernel_ms * 1e-3)
        marginal = (vslots + lds * census["slots_per_lds_instruction"]) / (kernel_ms * 1e-3)
        return {"bound": "vector issue", "unit": "wave-instruction slots/s", "achieved": rate,
                "peak": peak, "frac": rate / peak,
                "slots_per_launch": slots, "valu_wave_insts": got["valu_wave_insts"],
                "salu_wave_insts": got.get("salu_wave_insts"), "lds_wave_insts": got.get("lds_wave_insts"),
                "slots_per_vector_instruction": census["slots_per_vector_instruction"],
                "against_sustained_fma_rate": {"rate": fma_rate, "frac": rate / fma_rate,
                                               "note": "the rate an FMA-only kernel sustains (power-limited clock); this kernel "
                                                       "draws less and may clock higher, so this figure can pass 1"},
                "lds_at_marginal_cost": {"slots_per_lds_instruction": census["slots_per_lds_instruction"], "frac": marginal / peak,
                                         "note": "an LDS read priced at what adding one per visit cost (3.3 vector slots in the "
                                                 "first steps after an upload) instead of its one issue slot: an upper estimate"},
                "source": "SQ_INSTS_VALU / SQ_INSTS_LDS per launch from the rocprofv3 --pmc pass recorded in profiles/traffic.json "
                          "(%s); the vector instructions' slot weights (static mix of the hot loop) from profiles/issue_census.py; "
                          "not counters of this run" % got.get("workload", "")}
    roofline = {"bound": "hbm", "kernel": "k_bh_force", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "l2_hit_rate": l2_hit, "kernel_ms": force_ms,
                "algorithmic_bytes_per_launch": alg_bytes,
                "reorder_bytes_per_launch": BH_REORDER_BYTES_PER_BODY * per_rank_bodies if world == 1 else 0,
                "traffic_note": "traffic exceeds the algorithmic bytes by what the step's re-ordering of the bodies moves "
                                "(reorder_bytes_per_launch: every per-body array is written in the order of the step's sort, "
                                "so that the next step's gathers are coalesced); nothing is read twice",
                "traffic_source": "profiles/traffic.json: 2 x FETCH_SIZE + WRITE_SIZE (and TCC hits / (hits + misses)) of this kernel from separate "
                                  "rocprofv3 --pmc passes over `bench.py --steps 2 --warmup 1 --headline-only` (steps 2-3 of "
                                  "this workload), not measured in this run"}
    roofline["issue"] = issue_roofline("k_bh_force", force_ms)
    if late and world == 1:
        # the late state has its own counters (steps 300-301 of the same workload)
        lforce = (late.get("stage_ms") or {}).get("force", 0.0)
        lt = profiled.get("k_bh_force_late") or {}
        if lforce > 0:
            late["roofline"] = {"bound": "hbm", "kernel": "k_bh_force", "kernel_ms": lforce,
                                "achieved": alg_bytes / (lforce * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                "frac": alg_bytes / (lforce * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                "traffic": lt.get("hbm_bytes_per_launch"), "l2_hit_rate": lt.get("l2_hit_rate"),
                                "traffic_source": "profiles/traffic.json k_bh_force_late (%s)" % lt.get("workload", "not collected"),
                                "issue": issue_roofline("k_bh_force_late", lforce)}
    if counters and force_ms > 0:
        roofline["body_node_interactions_per_s"] = counters["accepted_interactions"] / (force_ms * 1e-3)
        roofline["node_evaluations_per_s"] = counters["node_evaluations"] / (force_ms * 1e-3)
        # the traversal is bound by vector-ALU issue, not by HBM: the same kernel priced like the
        # direct sum (accepted body-node interactions x 20 FLOP against the FP32 vector peak)
        tf = counters["accepted_interactions"] * FLOP_PER_INTERACTION / (force_ms * 1e-3) / 1e12
        roofline["valu_fp32"] = {"achieved": tf, "peak": VALU_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": tf / VALU_PEAK_TFLOPS,
                                 "flop_per_interaction": FLOP_PER_INTERACTION,
                                 "note": "opening tests, stack and batch handling are not counted as FLOP"}

    out = {
        "metric": "Barnes-Hut ms/step at N=5M",
        "value": ms_per_step,
        "unit": "ms/step",
        "n_gpus": world,
        "rccl_ranks": rccl_ranks,
        "collectives_by": TRANSPORT["used"] if not rehearsal else "torch.distributed over gloo (rehearsal)",
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": False,
        "scaling": "strong",
        "vs_baseline": ms_per_step / PUBLISHED_MS,
        "dtype": "f32",
        "data": "synthetic" if not rehearsal else "synthetic; REHEARSAL (all ranks on one GPU, gloo): not a measurement",
        "config": {
            "workload": "galaxy collision N=%d Barnes-Hut theta=0.5, 2-D (BASELINE.json configs[3]); forces FP32, "
                        "state and integration FP64" % args.bh_n,
            "scenario_seed": 12345,
            "parallelism": ("one GPU" if world == 1 else
                            "targets split over %d GPUs; %s, issued by the library on its own stream" % (
                                world, "each rank sorts its own bodies, two ncclAllReduce per step (bounding box; one "
                                       "record per cell of the level the tree closes at), same tree built on every GPU"
                                if cells else "one in-place ncclAllGather of all positions per step, whole tree "
                                              "rebuilt on every GPU")),
        },
        "speedup_vs_published": PUBLISHED_MS / ms_per_step,
        "stage_ms": stages,
        "cold_start": cold,
        "clock_prewarm": None if PREWARM_S <= 0 else (
            ("one GPU: %.1f s of force evaluations of the unchanged initial state (nb_compute_accel)" if world == 1 else
             "every rank: %.1f s of steps of a 65,536-body direct-sum system of its own (the simulation and its communicator "
             "are not touched)") % PREWARM_S +
            " before the W warm-up steps, so that the K timed steps run at the clocks a run settles to and not in the "
            "transient after the idle upload"),
        "stage_ms_note": "HIP-event times of the K steps that follow the K timed ones: the timed steps run without the stage "
                         "timer, whose event records cost about 1 % of a step",
        "tree": counters,
        "late_state": late,
        "position_exchange": gathered,
        "barnes_hut_50m": big,
        "gpu_tree_mode": None if args.headline_only else {
            "ms_per_step": gdt * 1e3 / args.steps, "bucket_depth": 9, "stage_ms": gstages, "tree": gcounters,
            "speedup_vs_published": PUBLISHED_MS / (gdt * 1e3 / args.steps),
            "note": "nb_params.bucket_depth = 9: the reference GPU program's tree (cells not split below "
                    "level 9, leaves as point masses); checked against the oracle's restatement of that "
                    "program, which itself is unpinned"},
        "roofline": roofline,
        "direct_sum": direct,
        "projected_scaling": proj,
    }
    if world == 1 and not args.no_direct:
        out["three_d"] = three_d_leg(nb, args.bh_n, args.direct_n, args.steps, args.warmup)
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(nb, args.bh_n)
    print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
