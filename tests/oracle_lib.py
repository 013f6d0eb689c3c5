"""ctypes access to the CPU checkers under oracle/ (test infrastructure only).

`Oracle` wraps oracle/liboracle.so, the plain-C restatement of the reference's CPU
algorithms.  `Reference` wraps oracle/_ref/libnbody_ref.so, the reference's own CPU
classes compiled unmodified (present only where oracle/Makefile could build it, or where
the prebuilt file travelled with the tree).
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# Layout of the reference's Body: isDynamic@0, mass@8, radius@16, position@24,
# velocity@40, acceleration@56; 72 bytes (src/cpu/body.h:22-34).
BODY_DTYPE = np.dtype(
    {
        "names": ["is_dynamic", "mass", "radius", "pos", "vel", "acc"],
        "formats": [np.uint8, np.float64, np.float64, (np.float64, 2), (np.float64, 2), (np.float64, 2)],
        "offsets": [0, 8, 16, 24, 40, 56],
        "itemsize": 72,
    }
)


# The 3-D body of the library's nb_body3 / the oracle's orc3_body: 96 bytes.
BODY3_DTYPE = np.dtype(
    {
        "names": ["is_dynamic", "mass", "radius", "pos", "vel", "acc"],
        "formats": [np.uint8, np.float64, np.float64, (np.float64, 3), (np.float64, 3), (np.float64, 3)],
        "offsets": [0, 8, 16, 24, 48, 72],
        "itemsize": 96,
    }
)


def lift3(b, z=None):
    """2-D bodies as 3-D bodies in the plane z = 0 (or at the given heights)."""
    out = np.zeros(len(b), dtype=BODY3_DTYPE)
    for f in ("is_dynamic", "mass", "radius"):
        out[f] = b[f]
    for f in ("pos", "vel", "acc"):
        out[f][:, :2] = b[f]
    if z is not None:
        out["pos"][:, 2] = z
    return out


class Params(C.Structure):
    _fields_ = [
        ("gravity", C.c_double),
        ("eps", C.c_double),
        ("dt", C.c_double),
        ("theta", C.c_double),
        ("collision_th", C.c_double),
        ("bbox_margin", C.c_double),
    ]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def make_bodies(n):
    return np.zeros(n, dtype=BODY_DTYPE)


class Oracle:
    def __init__(self, path=None):
        path = path or os.path.join(ROOT, "oracle", "liboracle.so")
        self.lib = C.CDLL(path)
        L = self.lib
        L.orc_default_params.argtypes = [C.POINTER(Params)]
        L.orc_init_scenario.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_void_p]
        L.orc_init_scenario.restype = C.c_int
        L.orc_direct_accel.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc_direct_accel_list.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        L.orc_direct_step.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int]
        L.orc_bh_bbox.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p]
        L.orc_bh_accel.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bh_accel.restype = C.c_int
        L.orc_bh_step.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int]
        L.orc_bh_step.restype = C.c_int
        L.orc_bh_accel_list.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p,
                                        C.c_void_p, C.c_void_p]
        L.orc_bh_accel_list.restype = C.c_int
        L.orc_set_threads.argtypes = [C.c_int]
        L.orc_morton_keys.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_bh_gpu_bbox.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p]
        L.orc_bh_gpu_accel.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        L.orc_bh_gpu_step.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int, C.c_int]
        L.orc_bh_gpu_accel_list.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        L.orc3_bh_accel_list.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                         C.c_void_p]
        L.orc3_bh_accel_list.restype = C.c_int
        L.orc3_init_scenario.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_double, C.c_void_p]
        L.orc3_init_scenario.restype = C.c_int
        L.orc3_direct_accel.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        L.orc3_direct_step.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int]
        L.orc3_bh_bbox.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p]
        L.orc3_bh_accel.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc3_bh_accel.restype = C.c_int
        L.orc3_bh_step.argtypes = [C.POINTER(Params), C.c_int, C.c_void_p, C.c_int]
        L.orc3_bh_step.restype = C.c_int
        L.orc3_morton_keys.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]

    def params(self, **kw):
        p = Params()
        self.lib.orc_default_params(C.byref(p))
        for k, v in kw.items():
            setattr(p, k, v)
        return p

    def scenario(self, sim, n, seed=12345):
        m = 5 if sim == 3 else n
        b = make_bodies(m)
        rc = self.lib.orc_init_scenario(sim, m, seed, _ptr(b))
        if rc != 0:
            raise ValueError("orc_init_scenario failed")
        return b

    def direct_accel(self, b, p=None, t0=0, t1=None):
        p = p or self.params()
        n = len(b)
        t1 = n if t1 is None else t1
        acc = np.zeros((t1 - t0, 2))
        self.lib.orc_direct_accel(C.byref(p), n, _ptr(b), t0, t1, _ptr(acc))
        return acc

    def direct_accel_list(self, b, targets, p=None):
        p = p or self.params()
        t = np.ascontiguousarray(targets, dtype=np.int32)
        acc = np.zeros((len(t), 2))
        self.lib.orc_direct_accel_list(C.byref(p), len(b), _ptr(b), len(t), _ptr(t), _ptr(acc))
        return acc

    def direct_step(self, b, nsteps, p=None):
        p = p or self.params()
        b = b.copy()
        self.lib.orc_direct_step(C.byref(p), len(b), _ptr(b), nsteps)
        return b

    def bh_bbox(self, b, p=None):
        p = p or self.params()
        box = np.zeros(4)
        self.lib.orc_bh_bbox(C.byref(p), len(b), _ptr(b), _ptr(box))
        return box

    def bh_accel(self, b, p=None, with_stats=False):
        p = p or self.params()
        acc = np.zeros((len(b), 2))
        stats = np.zeros(4)
        rc = self.lib.orc_bh_accel(C.byref(p), len(b), _ptr(b), _ptr(acc), _ptr(stats))
        if rc != 0:
            raise ValueError("coincident bodies")
        return (acc, stats) if with_stats else acc

    def set_threads(self, threads):
        """Threads of the per-target loops (1 = the reference's single thread)."""
        self.lib.orc_set_threads(int(threads))

    def bh_accel_list(self, b, targets, p=None, sorted_insert=True, with_stats=False):
        """Barnes-Hut accelerations of the listed bodies only, from the tree of all bodies.
        sorted_insert: insert the bodies in the order of their quadrant paths (same tree, built
        several times faster at millions of bodies)."""
        p = p or self.params()
        t = np.ascontiguousarray(targets, dtype=np.int32)
        order = None
        if sorted_insert:
            keys = self.morton_keys(b, self.bh_bbox(b, p))
            order = np.ascontiguousarray(np.argsort(keys, kind="stable"), dtype=np.int32)
            del keys
        acc = np.zeros((len(t), 2))
        stats = np.zeros(4)
        rc = self.lib.orc_bh_accel_list(C.byref(p), len(b), _ptr(b), _ptr(order) if order is not None else None,
                                        len(t), _ptr(t), _ptr(acc), _ptr(stats))
        if rc != 0:
            raise ValueError("coincident bodies")
        return (acc, stats) if with_stats else acc

    def bh_step(self, b, nsteps, p=None):
        p = p or self.params()
        b = b.copy()
        rc = self.lib.orc_bh_step(C.byref(p), len(b), _ptr(b), nsteps)
        if rc != 0:
            raise ValueError("coincident bodies")
        return b

    def gpu_bbox(self, b, p=None):
        p = p or self.params()
        box = np.zeros(4)
        self.lib.orc_bh_gpu_bbox(C.byref(p), len(b), _ptr(b), _ptr(box))
        return box

    def gpu_accel(self, b, depth=9, p=None):
        p = p or self.params()
        acc = np.zeros((len(b), 2))
        self.lib.orc_bh_gpu_accel(C.byref(p), len(b), _ptr(b), depth, _ptr(acc))
        return acc

    def gpu_accel_list(self, b, targets, depth=9, p=None):
        """The GPU program's tree of all bodies, walked for the listed bodies only."""
        p = p or self.params()
        t = np.ascontiguousarray(targets, dtype=np.int32)
        acc = np.zeros((len(t), 2))
        self.lib.orc_bh_gpu_accel_list(C.byref(p), len(b), _ptr(b), depth, len(t), _ptr(t), _ptr(acc))
        return acc

    def gpu_step(self, b, nsteps, depth=9, p=None):
        p = p or self.params()
        b = b.copy()
        self.lib.orc_bh_gpu_step(C.byref(p), len(b), _ptr(b), depth, nsteps)
        return b

    # ---- three dimensions (bodies of BODY3_DTYPE) ----

    def scenario3(self, sim, n, seed=12345, thickness=0.0):
        m = 5 if sim == 3 else n
        b = np.zeros(m, dtype=BODY3_DTYPE)
        if self.lib.orc3_init_scenario(sim, m, seed, thickness, _ptr(b)) != 0:
            raise ValueError("orc3_init_scenario failed")
        return b

    def direct_accel3(self, b, p=None, t0=0, t1=None):
        p = p or self.params()
        t1 = len(b) if t1 is None else t1
        acc = np.zeros((t1 - t0, 3))
        self.lib.orc3_direct_accel(C.byref(p), len(b), _ptr(b), t0, t1, _ptr(acc))
        return acc

    def direct_step3(self, b, nsteps, p=None):
        p = p or self.params()
        b = b.copy()
        self.lib.orc3_direct_step(C.byref(p), len(b), _ptr(b), nsteps)
        return b

    def bh_bbox3(self, b, p=None):
        p = p or self.params()
        box = np.zeros(6)
        self.lib.orc3_bh_bbox(C.byref(p), len(b), _ptr(b), _ptr(box))
        return box

    def bh_accel3(self, b, p=None, with_stats=False):
        p = p or self.params()
        acc = np.zeros((len(b), 3))
        stats = np.zeros(4)
        if self.lib.orc3_bh_accel(C.byref(p), len(b), _ptr(b), _ptr(acc), _ptr(stats)) != 0:
            raise ValueError("coincident bodies")
        return (acc, stats) if with_stats else acc

    def bh_accel_list3(self, b, targets, p=None, sorted_insert=True, with_stats=False):
        """Octree accelerations of the listed bodies only, from the tree of all bodies (inserted in
        the order of their octant paths: same tree, built faster)."""
        p = p or self.params()
        t = np.ascontiguousarray(targets, dtype=np.int32)
        order = None
        if sorted_insert:
            keys = self.morton_keys3(b, self.bh_bbox3(b, p))
            order = np.ascontiguousarray(np.argsort(keys, kind="stable"), dtype=np.int32)
            del keys
        acc = np.zeros((len(t), 3))
        stats = np.zeros(4)
        rc = self.lib.orc3_bh_accel_list(C.byref(p), len(b), _ptr(b), _ptr(order) if order is not None else None,
                                         len(t), _ptr(t), _ptr(acc), _ptr(stats))
        if rc != 0:
            raise ValueError("coincident bodies")
        return (acc, stats) if with_stats else acc

    def bh_step3(self, b, nsteps, p=None):
        p = p or self.params()
        b = b.copy()
        if self.lib.orc3_bh_step(C.byref(p), len(b), _ptr(b), nsteps) != 0:
            raise ValueError("coincident bodies")
        return b

    def tree_keys3(self, b, box):
        """The keys the octree pipeline sorts by: the octant path of every body (morton_keys3) with the
        octants' positions along a Hilbert curve as digits -- Skilling's transform of the three bit
        strings, as in the library's k_keys3.  A bijection on paths whose every digit depends on the
        path down to its own level only: cut at any level it is that level's key."""
        m = self.morton_keys3(b, box)
        levels = 20
        c = [np.zeros(len(m), dtype=np.uint64) for _ in range(3)]
        for level in range(levels):
            digit = (m >> np.uint64(62 - 3 * (level + 1))) & np.uint64(7)
            for d in range(3):  # digit = z << 2 | y << 1 | x
                c[d] = (c[d] << np.uint64(1)) | ((digit >> np.uint64(d)) & np.uint64(1))
        top = 1 << (levels - 1)
        q = top
        while q > 1:
            low = np.uint64(q - 1)
            for d in range(3):
                hit = (c[d] & np.uint64(q)) != 0
                t = np.where(hit, np.uint64(0), (c[0] ^ c[d]) & low)
                c[0] = np.where(hit, c[0] ^ low, c[0] ^ t)
                c[d] = np.where(hit, c[d], c[d] ^ t) if d else c[0]
            q >>= 1
        c[1] = c[1] ^ c[0]
        c[2] = c[2] ^ c[1]
        t = np.zeros(len(m), dtype=np.uint64)
        q = top
        while q > 1:
            t = np.where((c[2] & np.uint64(q)) != 0, t ^ np.uint64(q - 1), t)
            q >>= 1
        c = [v ^ t for v in c]
        k = np.zeros(len(m), dtype=np.uint64)
        for level in range(levels - 1, -1, -1):
            sh = np.uint64(level)
            k = (k << np.uint64(3)) | (((c[0] >> sh) & np.uint64(1)) << np.uint64(2)) | (((c[1] >> sh) & np.uint64(1)) << np.uint64(1)) | ((c[2] >> sh) & np.uint64(1))
        return k << np.uint64(2)

    def morton_keys3(self, b, box):
        keys = np.zeros(len(b), dtype=np.uint64)
        box = np.ascontiguousarray(box, dtype=np.float64)
        self.lib.orc3_morton_keys(len(b), _ptr(b), _ptr(box), _ptr(keys))
        return keys

    def tree_keys(self, b, box):
        """The keys the library sorts by: the reference's quadrant path of every body (morton_keys), each
        digit replaced by the quadrant's position along a Hilbert curve through its cell (the same
        two tables as the library's k_keys: entry 4 * orientation + quadrant).  A bijection on paths:
        equal keys mean equal quadrant decisions at every level."""
        m = self.morton_keys(b, box)
        digit_of, next_of = 0x1E36B49C, 0xB3DA640D
        h = np.zeros(len(m), dtype=np.uint64)
        curve = np.zeros(len(m), dtype=np.uint64)
        for level in range(31):
            q = (m >> np.uint64(60 - 2 * level)) & np.uint64(3)
            at = np.uint64(2) * (np.uint64(4) * curve + q)
            h = (h << np.uint64(2)) | ((np.uint64(digit_of) >> at) & np.uint64(3))
            curve = (np.uint64(next_of) >> at) & np.uint64(3)
        return h

    def morton_keys(self, b, box):
        keys = np.zeros(len(b), dtype=np.uint64)
        box = np.ascontiguousarray(box, dtype=np.float64)
        self.lib.orc_morton_keys(len(b), _ptr(b), _ptr(box), _ptr(keys))
        return keys


class Reference:
    """The reference's own CPU classes (src/cpu), compiled unmodified."""

    @staticmethod
    def path():
        return os.path.join(ROOT, "oracle", "_ref", "libnbody_ref.so")

    @classmethod
    def available(cls):
        return os.path.exists(cls.path())

    def __init__(self):
        self.lib = C.CDLL(self.path())
        self.lib.ref_init_scenario.argtypes = [C.c_int, C.c_int, C.c_uint, C.c_void_p]
        self.lib.ref_init_scenario.restype = C.c_int
        self.lib.ref_step.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_int]
        self.lib.ref_step.restype = C.c_int

    def scenario(self, sim_cpu, n, seed=12345):
        """sim_cpu: 0 spiral, 1 random, 2 solar (the CPU program's numbering)."""
        b = make_bodies(max(n, 5))
        m = self.lib.ref_init_scenario(sim_cpu, n, seed, _ptr(b))
        return b[:m].copy()

    def step(self, alg, b, nsteps):
        b = b.copy()
        self.lib.ref_step(alg, len(b), _ptr(b), nsteps)
        return b
