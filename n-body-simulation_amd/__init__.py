"""Python host side of the MI355X N-body engine: a ctypes binding of the C ABI in
include/nbody.h (libnbody_hip.so, hand-written HIP for gfx950).

The classes mirror the reference's simulation objects so that tests read like a user of
the reference would write them:

    reference (src/gpu/BH/barnesHutCuda.cuh:49-80)      here
    BarnesHutCuda(n)                                    NBodySim(n, alg=BARNES_HUT)
    setup(sim)                                          .setup(sim, seed)
    update()                                            .update()
    readDeviceBodies(); getBodies()                     .read_bodies()

    reference (src/cpu/nBody.h:25-37)                   NBody(n, alg, sim)  (same argument order)

There is no CPU fallback: importing works anywhere, but creating a simulation needs the
shared library and a HIP device, and raises otherwise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# NBODY_HIP_LIB selects another build of the same library (kernel tuning experiments)
LIB_PATH = os.environ.get("NBODY_HIP_LIB") or os.path.join(_HERE, "libnbody_hip.so")

DIRECT, BARNES_HUT = 0, 1
SPIRAL, RANDOM, COLLISION, SOLAR = 0, 1, 2, 3
EULER, LEAPFROG = 0, 1
EXCHANGE_POSITIONS, EXCHANGE_CELLS, EXCHANGE_AUTO = 0, 1, 2

# Same 72-byte layout as the reference's Body.
BODY_DTYPE = np.dtype(
    {
        "names": ["is_dynamic", "mass", "radius", "pos", "vel", "acc"],
        "formats": [np.uint8, np.float64, np.float64, (np.float64, 2), (np.float64, 2), (np.float64, 2)],
        "offsets": [0, 8, 16, 24, 40, 56],
        "itemsize": 72,
    }
)

# nb_body3: the same fields with three-component vectors, 96 bytes.
BODY3_DTYPE = np.dtype(
    {
        "names": ["is_dynamic", "mass", "radius", "pos", "vel", "acc"],
        "formats": [np.uint8, np.float64, np.float64, (np.float64, 3), (np.float64, 3), (np.float64, 3)],
        "offsets": [0, 8, 16, 24, 48, 72],
        "itemsize": 96,
    }
)

DBG_BBOX, DBG_KEYS, DBG_SORTED_KEYS, DBG_SORTED_IDX = 0, 1, 2, 3
DBG_NODE_RANGE, DBG_NODE_CHILD, DBG_NODE_COM, DBG_LEVEL_START, DBG_ORDER, DBG_COUNTERS = 4, 5, 6, 7, 8, 9
DBG_KEY_LEVELS = 10
DBG_VISIT_KINDS = 11
DBG_WAVE_COST = 12

STATIC_MAX_BODIES = 131072  # NB_STATIC_MAX_BODIES of include/nbody.h

# Every symbol include/nbody.h declares.
ABI_SYMBOLS = [
    "nb_default_params", "nb_init_scenario", "nb_create", "nb_destroy", "nb_setup", "nb_set_bodies",
    "nb_step", "nb_compute_accel", "nb_read_bodies", "nb_sync", "nb_set_stream", "nb_set_partition",
    "nb_exchange_buffer", "nb_step_local", "nb_step_finish", "nb_debug_read", "nb_set_stage_timing",
    "nb_stage_ms", "nb_diagnostics", "nb_refresh_order", "nb_compute_accel_static", "nb_comm_unique_id", "nb_comm_init", "nb_step_distributed",
    "nb_velocity_buffer", "nb_set_exchange", "nb_cells_level", "nb_step_cells", "nb_cells_buffer", "nb3_comm_init", "nb3_step_distributed", "nb3_refresh_order", "nb3_velocity_buffer", "nb_last_error", "nb_version",
    "nb3_init_scenario", "nb3_create", "nb3_destroy", "nb3_setup", "nb3_set_bodies", "nb3_step",
    "nb3_compute_accel", "nb3_read_bodies", "nb3_sync", "nb3_debug_read", "nb3_set_stage_timing", "nb3_stage_ms",
    "nb3_diagnostics", "nb3_set_stream", "nb3_set_partition", "nb3_exchange_buffer", "nb3_step_local", "nb3_step_finish",
]


class Params(C.Structure):
    _fields_ = [
        ("gravity", C.c_double),
        ("eps", C.c_double),
        ("dt", C.c_double),
        ("theta", C.c_double),
        ("collision_th", C.c_double),
        ("bbox_margin", C.c_double),
        ("integrator", C.c_int),
        ("bucket_depth", C.c_int),
    ]


class NBodyError(RuntimeError):
    pass


_lib = None


def load_library(path=None):
    """Loads libnbody_hip.so and declares the prototypes; raises if it is missing."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    path = path or LIB_PATH
    if not os.path.exists(path):
        raise NBodyError(
            "%s is missing: build it with `make -C n-body-simulation_amd` (hipcc, gfx950); "
            "there is no CPU fallback" % path
        )
    lib = C.CDLL(path)
    vp, i, u = C.c_void_p, C.c_int, C.c_uint
    lib.nb_default_params.argtypes = [C.POINTER(Params)]
    lib.nb_default_params.restype = None
    lib.nb_init_scenario.argtypes = [i, i, u, vp]
    lib.nb_create.argtypes = [C.POINTER(Params), i, i, C.POINTER(vp)]
    lib.nb_destroy.argtypes = [vp]
    lib.nb_setup.argtypes = [vp, i, u]
    lib.nb_set_bodies.argtypes = [vp, vp, i]
    lib.nb_step.argtypes = [vp, i]
    lib.nb_compute_accel.argtypes = [vp]
    lib.nb_read_bodies.argtypes = [vp, vp]
    lib.nb_sync.argtypes = [vp]
    lib.nb_set_stream.argtypes = [vp, vp]
    lib.nb_set_partition.argtypes = [vp, i, i]
    lib.nb_exchange_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.nb_step_local.argtypes = [vp]
    lib.nb_step_finish.argtypes = [vp]
    lib.nb_debug_read.argtypes = [vp, i, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.nb_set_stage_timing.argtypes = [vp, i]
    lib.nb_stage_ms.argtypes = [vp, C.POINTER(C.c_float), i]
    lib.nb_diagnostics.argtypes = [vp, C.POINTER(C.c_double), i]
    lib.nb_refresh_order.argtypes = [vp]
    lib.nb_compute_accel_static.argtypes = [vp, vp, vp, vp]
    lib.nb_comm_unique_id.argtypes = [vp, C.c_size_t]
    lib.nb_comm_init.argtypes = [vp, i, i, vp]
    lib.nb_step_distributed.argtypes = [vp, i]
    lib.nb_velocity_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.nb3_velocity_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.nb3_refresh_order.argtypes = [vp]
    lib.nb_set_exchange.argtypes = [vp, i]
    lib.nb_step_cells.argtypes = [vp, i]
    lib.nb_cells_level.argtypes = [vp, C.POINTER(i), C.POINTER(i)]
    lib.nb_cells_buffer.argtypes = [vp, i, C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.nb3_comm_init.argtypes = [vp, i, i, vp]
    lib.nb3_step_distributed.argtypes = [vp, i]
    lib.nb3_init_scenario.argtypes = [i, i, u, C.c_double, vp]
    lib.nb3_create.argtypes = [C.POINTER(Params), i, i, C.POINTER(vp)]
    lib.nb3_destroy.argtypes = [vp]
    lib.nb3_setup.argtypes = [vp, i, u, C.c_double]
    lib.nb3_set_bodies.argtypes = [vp, vp, i]
    lib.nb3_step.argtypes = [vp, i]
    lib.nb3_compute_accel.argtypes = [vp]
    lib.nb3_read_bodies.argtypes = [vp, vp]
    lib.nb3_sync.argtypes = [vp]
    lib.nb3_debug_read.argtypes = [vp, i, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    lib.nb3_diagnostics.argtypes = [vp, C.POINTER(C.c_double), i]
    lib.nb3_set_stream.argtypes = [vp, vp]
    lib.nb3_set_partition.argtypes = [vp, i, i]
    lib.nb3_exchange_buffer.argtypes = [vp, C.POINTER(vp), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]
    lib.nb3_step_local.argtypes = [vp]
    lib.nb3_step_finish.argtypes = [vp]
    lib.nb3_set_stage_timing.argtypes = [vp, i]
    lib.nb3_stage_ms.argtypes = [vp, C.POINTER(C.c_float), i]
    lib.nb_last_error.restype = C.c_char_p
    lib.nb_version.restype = C.c_char_p
    for name in ABI_SYMBOLS:
        f = getattr(lib, name)
        if name not in ("nb_default_params", "nb_last_error", "nb_version"):
            f.restype = i
    _lib = lib
    return lib


def _check(rc):
    if rc != 0:
        raise NBodyError("nbody error %d: %s" % (rc, load_library().nb_last_error().decode()))


COMM_ID_BYTES = 128  # NB_COMM_ID_BYTES


def comm_unique_id():
    """The bytes rank 0 hands to every rank for comm_init (ncclGetUniqueId)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(load_library().nb_comm_unique_id(buf, COMM_ID_BYTES))
    return buf.raw


def default_params(**overrides):
    p = Params()
    load_library().nb_default_params(C.byref(p))
    for k, v in overrides.items():
        if not hasattr(p, k):
            raise AttributeError(k)
        setattr(p, k, v)
    return p


def make_bodies(n):
    return np.zeros(n, dtype=BODY_DTYPE)


def init_scenario(sim, n, seed=12345):
    """Host-only initial conditions (BarnesHutCuda::init*Bodies); SOLAR always has 5 bodies."""
    m = 5 if sim == SOLAR else n
    b = make_bodies(m)
    _check(load_library().nb_init_scenario(sim, m, seed, b.ctypes.data_as(C.c_void_p)))
    return b


class NBodySim:
    """One simulation on the current HIP device (the role of the reference's BarnesHutCuda)."""

    def __init__(self, n, alg=BARNES_HUT, params=None, stream=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.n = int(n)
        self.alg = alg
        self.params = params or default_params()
        _check(self._lib.nb_create(C.byref(self.params), self.n, alg, C.byref(self._h)))
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        _check(self._lib.nb_set_stream(self._h, C.c_void_p(stream)))

    # -- reference surface --
    def setup(self, sim, seed=12345):
        _check(self._lib.nb_setup(self._h, sim, seed))

    def update(self, nsteps=1):
        _check(self._lib.nb_step(self._h, nsteps))

    def read_bodies(self):
        out = make_bodies(self.n)
        _check(self._lib.nb_read_bodies(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    # -- additions --
    def set_bodies(self, bodies):
        b = np.ascontiguousarray(bodies, dtype=BODY_DTYPE)
        _check(self._lib.nb_set_bodies(self._h, b.ctypes.data_as(C.c_void_p), len(b)))

    def evaluate(self):
        """Enqueues one force evaluation (nb_compute_accel) and returns at once: the state is left as it is,
        the stored accelerations are those of the current positions."""
        _check(self._lib.nb_compute_accel(self._h))

    def compute_accel(self):
        _check(self._lib.nb_compute_accel(self._h))
        return self.read_bodies()["acc"]

    def sync(self):
        _check(self._lib.nb_sync(self._h))

    def set_partition(self, rank, world):
        _check(self._lib.nb_set_partition(self._h, rank, world))

    def exchange_buffer(self):
        """(device pointer, records per rank, bytes per record) of the position exchange buffer."""
        p, c, r = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _check(self._lib.nb_exchange_buffer(self._h, C.byref(p), C.byref(c), C.byref(r)))
        return p.value, c.value, r.value

    def set_exchange(self, mode):
        """What the ranks exchange each step: EXCHANGE_POSITIONS, EXCHANGE_CELLS or EXCHANGE_AUTO (nb_set_exchange)."""
        _check(self._lib.nb_set_exchange(self._h, mode))

    def cells_level(self):
        """Before a step with the cell exchange: (level of the cells to be exchanged, or -1 while it is
        suspended and the step is a position-exchange step; whether the ranks must all-gather their
        positions first because it was suspended just now)  (nb_cells_level)."""
        lvl, gather = C.c_int(-1), C.c_int(0)
        _check(self._lib.nb_cells_level(self._h, C.byref(lvl), C.byref(gather)))
        return lvl.value, bool(gather.value)

    def step_cells(self, part):
        _check(self._lib.nb_step_cells(self._h, part))

    def cells_buffer(self, which):
        """(device pointer, number of doubles) of the extent (0: reduce with MAX) or the cell records (1: SUM)."""
        p, c = C.c_void_p(), C.c_size_t()
        _check(self._lib.nb_cells_buffer(self._h, which, C.byref(p), C.byref(c)))
        return p.value, c.value

    def velocity_buffer(self):
        """(device pointer, records per rank, bytes per record) of the velocity array, laid out like the
        exchange buffer (all-gathered before a partitioned simulation renews its order)."""
        p, c, r = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _check(self._lib.nb_velocity_buffer(self._h, C.byref(p), C.byref(c), C.byref(r)))
        return p.value, c.value, r.value

    def step_local(self):
        _check(self._lib.nb_step_local(self._h))

    def comm_init(self, rank, world, comm_id):
        """Joins the RCCL communicator (collective over the ranks); sets the partition."""
        _check(self._lib.nb_comm_init(self._h, rank, world, C.c_char_p(comm_id)))

    def step_distributed(self, nsteps=1):
        _check(self._lib.nb_step_distributed(self._h, nsteps))

    def step_finish(self):
        _check(self._lib.nb_step_finish(self._h))

    def set_stage_timing(self, on=True, counters=None):
        """on: per-stage hipEvent timing; counters (default: same as on): traversal statistics."""
        counters = on if counters is None else counters
        _check(self._lib.nb_set_stage_timing(self._h, (1 if on else 0) | (2 if counters else 0)))

    def order(self):
        """Caller index of the body held in each exchange-order slot."""
        return self.debug(DBG_ORDER, np.uint32, self.n)

    def stage_ms(self):
        out = (C.c_float * 8)()
        _check(self._lib.nb_stage_ms(self._h, out, 8))
        names = ["bbox", "keys", "sort", "build", "com", "force", "exchange"]
        return {k: out[i] for i, k in enumerate(names)}

    def refresh_order(self):
        """Re-sorts the device state by the bodies' current positions (see nb_refresh_order)."""
        _check(self._lib.nb_refresh_order(self._h))

    def compute_accel_static(self, pos_ptr, acc_ptr, stream=None):
        """nb_compute_accel_static: accelerations at the positions in device memory at pos_ptr
        (n x 2 doubles, caller order) into device memory at acc_ptr; allocates nothing."""
        _check(self._lib.nb_compute_accel_static(self._h, C.c_void_p(pos_ptr), C.c_void_p(acc_ptr), C.c_void_p(stream)))

    def diagnostics(self):
        """Conserved quantities of the current state (FP64); potential is NaN above 131072 bodies."""
        out = (C.c_double * 8)()
        _check(self._lib.nb_diagnostics(self._h, out, 8))
        return {"mass": out[0], "com": (out[1], out[2]), "momentum": (out[3], out[4]), "angular_momentum": out[5],
                "kinetic": out[6], "potential": out[7]}

    def debug(self, which, dtype, count):
        out = np.zeros(count, dtype=dtype)
        w = C.c_size_t()
        _check(self._lib.nb_debug_read(self._h, which, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(w)))
        return out[: w.value // out.itemsize]

    def close(self):
        if self._h:
            self._lib.nb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


def init_scenario3(sim, n, seed=12345, thickness=0.0):
    """The 2-D scenario with a height per body drawn from [-thickness, thickness] (suns stay at z = 0)."""
    m = 5 if sim == SOLAR else n
    b = np.zeros(m, dtype=BODY3_DTYPE)
    _check(load_library().nb3_init_scenario(sim, m, seed, thickness, b.ctypes.data_as(C.c_void_p)))
    return b


class NBodySim3:
    """NBodySim for three-dimensional bodies (the nb3_ entry points)."""

    def __init__(self, n, alg=BARNES_HUT, params=None, stream=None):
        self._lib = load_library()
        self._h = C.c_void_p()
        self.n = int(n)
        self.alg = alg
        self.params = params or default_params()
        _check(self._lib.nb3_create(C.byref(self.params), self.n, alg, C.byref(self._h)))
        if stream is not None:
            self.set_stream(stream)

    def set_stream(self, stream):
        _check(self._lib.nb3_set_stream(self._h, C.c_void_p(stream)))

    def set_partition(self, rank, world):
        _check(self._lib.nb3_set_partition(self._h, rank, world))

    def exchange_buffer(self):
        """(device pointer, records per rank, bytes per record) of the position exchange buffer."""
        p, c, r = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _check(self._lib.nb3_exchange_buffer(self._h, C.byref(p), C.byref(c), C.byref(r)))
        return p.value, c.value, r.value

    def velocity_buffer(self):
        p, c, r = C.c_void_p(), C.c_size_t(), C.c_size_t()
        _check(self._lib.nb3_velocity_buffer(self._h, C.byref(p), C.byref(c), C.byref(r)))
        return p.value, c.value, r.value

    def refresh_order(self):
        _check(self._lib.nb3_refresh_order(self._h))

    def step_local(self):
        _check(self._lib.nb3_step_local(self._h))

    def comm_init(self, rank, world, comm_id):
        _check(self._lib.nb3_comm_init(self._h, rank, world, C.c_char_p(comm_id)))

    def step_distributed(self, nsteps=1):
        _check(self._lib.nb3_step_distributed(self._h, nsteps))

    def step_finish(self):
        _check(self._lib.nb3_step_finish(self._h))

    def setup(self, sim, seed=12345, thickness=0.0):
        _check(self._lib.nb3_setup(self._h, sim, seed, thickness))

    def update(self, nsteps=1):
        _check(self._lib.nb3_step(self._h, nsteps))

    def read_bodies(self):
        out = np.zeros(self.n, dtype=BODY3_DTYPE)
        _check(self._lib.nb3_read_bodies(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def set_bodies(self, bodies):
        b = np.ascontiguousarray(bodies, dtype=BODY3_DTYPE)
        _check(self._lib.nb3_set_bodies(self._h, b.ctypes.data_as(C.c_void_p), len(b)))

    def evaluate(self):
        """Enqueues one force evaluation (nb3_compute_accel) and returns at once (see NBodySim.evaluate)."""
        _check(self._lib.nb3_compute_accel(self._h))

    def compute_accel(self):
        _check(self._lib.nb3_compute_accel(self._h))
        return self.read_bodies()["acc"]

    def sync(self):
        _check(self._lib.nb3_sync(self._h))

    def set_stage_timing(self, on=True, counters=None):
        counters = on if counters is None else counters
        _check(self._lib.nb3_set_stage_timing(self._h, (1 if on else 0) | (2 if counters else 0)))

    def stage_ms(self):
        out = (C.c_float * 8)()
        _check(self._lib.nb3_stage_ms(self._h, out, 8))
        names = ["bbox", "keys", "sort", "build", "com", "force", "exchange"]
        return {k: out[i] for i, k in enumerate(names)}

    def order(self):
        return self.debug(DBG_ORDER, np.uint32, self.n)

    def diagnostics(self):
        """Conserved quantities of the current state (FP64); potential is NaN above 131072 bodies."""
        out = (C.c_double * 12)()
        _check(self._lib.nb3_diagnostics(self._h, out, 12))
        return {"mass": out[0], "com": tuple(out[1:4]), "momentum": tuple(out[4:7]), "angular_momentum": tuple(out[7:10]),
                "kinetic": out[10], "potential": out[11]}

    def debug(self, which, dtype, count):
        out = np.zeros(count, dtype=dtype)
        w = C.c_size_t()
        _check(self._lib.nb3_debug_read(self._h, which, out.ctypes.data_as(C.c_void_p), out.nbytes, C.byref(w)))
        return out[: w.value // out.itemsize]

    def close(self):
        if self._h:
            self._lib.nb3_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()


class NBody(NBodySim):
    """The CPU program's object, same argument order: NBody(n, alg, sim) (src/cpu/nBody.cpp:26-51);
    sim uses the CPU numbering 0 spiral, 1 random, anything else solar (n forced to 5)."""

    def __init__(self, n, alg, sim, seed=12345, params=None):
        scenario = {0: SPIRAL, 1: RANDOM}.get(sim, SOLAR)
        if scenario == SOLAR:
            n = 5
        super().__init__(n, alg, params)
        self.setup(scenario, seed)
