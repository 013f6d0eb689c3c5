"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/nbody.h
declares, its host-only entry points (scenario generators, defaults, argument checks)
agree with the oracle, and it refuses to run without a GPU instead of falling back."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from conftest import ROOT, golden


def header_functions():
    text = open(os.path.join(ROOT, "include", "nbody.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(nb3?_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(nb):
    lib = nb.load_library()
    names = header_functions()
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(nb.ABI_SYMBOLS) == names
    assert lib.nb_version().startswith(b"nbody_hip")


def test_body_layout_matches_reference_struct(nb):
    assert nb.BODY_DTYPE.itemsize == 72
    assert [nb.BODY_DTYPE.fields[k][1] for k in ("is_dynamic", "mass", "radius", "pos", "vel", "acc")] == [0, 8, 16, 24, 40, 56]


def test_default_params_are_the_reference_constants(nb):
    p = nb.default_params()
    assert (p.gravity, p.eps, p.dt, p.theta, p.collision_th, p.bbox_margin) == (6.67e-11, 0.5, 25000.0, 0.5, 1.0e10, 1.0e10)


@pytest.mark.parametrize("sim", [0, 1, 2, 3])
@pytest.mark.parametrize("n,seed", [(5, 1), (300, 777), (4096, 12345)])
def test_scenarios_match_oracle_bit_for_bit(nb, oracle, sim, n, seed):
    a = nb.init_scenario(sim, n, seed)
    b = oracle.scenario(sim, n, seed)
    assert len(a) == len(b)
    for f in a.dtype.names:
        assert np.array_equal(a[f], b[f]), f


@pytest.mark.parametrize("sim,name", [(0, "spiral"), (1, "random"), (2, "collision")])
def test_scenarios_match_reference_golden(nb, sim, name):
    g = golden("%s_n4096" % name)
    a = nb.init_scenario(sim, 4096, 12345)
    assert np.array_equal(a["pos"], g["pos0"])
    assert np.array_equal(a["vel"], g["vel0"])
    assert np.array_equal(a["mass"], g["mass"])
    assert np.array_equal(a["is_dynamic"], g["is_dynamic"])


def test_collision_scenario_structure(nb):
    n = 1001
    b = nb.init_scenario(nb.COLLISION, n, 3)
    g1 = n // 2
    suns = np.flatnonzero(b["mass"] > 1e29)
    assert list(suns) == [g1 - 1, n - 1]
    assert np.all(b["is_dynamic"] == 1)  # both suns move (barnesHutCuda.cu:184,220)
    assert b["pos"][g1 - 1, 0] == -10.0e11 / 6.0 and b["pos"][n - 1, 0] == 10.0e11 / 6.0
    r1 = np.linalg.norm(b["pos"][: g1 - 1] - b["pos"][g1 - 1], axis=1)
    assert r1.min() >= 2.0e10 * (1 - 1e-12) and r1.max() <= 1.25e11 * (1 + 1e-12)


def test_argument_errors_carry_the_reference_messages(nb):
    lib = nb.load_library()
    buf = nb.make_bodies(8)
    assert lib.nb_init_scenario(0, 0, 1, buf.ctypes.data_as(C.c_void_p)) == 1
    assert b"at least 1 body" in lib.nb_last_error()  # checkArgs, src/gpu/BH/main.cu:65-69
    assert lib.nb_init_scenario(7, 8, 1, buf.ctypes.data_as(C.c_void_p)) == 1
    assert b"simulation doesn't exist" in lib.nb_last_error()  # main.cu:71-75
    assert lib.nb_init_scenario(3, 4, 1, buf.ctypes.data_as(C.c_void_p)) == 1
    h = C.c_void_p()
    assert lib.nb_create(None, 0, 0, C.byref(h)) == 1
    assert lib.nb_create(None, 10, 5, C.byref(h)) == 1
    assert lib.nb_step(None, 1) == 1


def test_no_silent_cpu_fallback(nb):
    """Without a HIP device the library must fail loudly, not compute on the CPU."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(nb.NBodyError, match="no HIP device"):
        nb.NBodySim(16, nb.DIRECT)
    with pytest.raises(nb.NBodyError):
        nb.load_library(os.path.join(ROOT, "n-body-simulation_amd", "does_not_exist.so"))
