import importlib.util
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_package():
    """Imports the package directory n-body-simulation_amd/ under the name nbody_amd."""
    if "nbody_amd" in sys.modules:
        return sys.modules["nbody_amd"]
    path = os.path.join(ROOT, "n-body-simulation_amd", "__init__.py")
    spec = importlib.util.spec_from_file_location(
        "nbody_amd", path, submodule_search_locations=[os.path.dirname(path)]
    )
    mod = importlib.util.module_from_spec(spec)
    sys.modules["nbody_amd"] = mod
    spec.loader.exec_module(mod)
    return mod


@pytest.fixture(scope="session")
def oracle():
    so = os.path.join(ROOT, "oracle", "liboracle.so")
    if not os.path.exists(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "liboracle.so"])
    from oracle_lib import Oracle

    return Oracle(so)


@pytest.fixture(scope="session")
def nb():
    return load_package()


def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def bodies_from_golden(g):
    from oracle_lib import make_bodies

    b = make_bodies(len(g["mass"]))
    b["is_dynamic"] = g["is_dynamic"]
    b["mass"] = g["mass"]
    b["radius"] = g["radius"]
    b["pos"] = g["pos0"]
    b["vel"] = g["vel0"]
    return b


def rel_err(a, ref):
    """Per-body |a - ref| / |ref| over 2-vectors (0 where both vanish)."""
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    num = np.linalg.norm(a - ref, axis=-1)
    den = np.linalg.norm(ref, axis=-1)
    out = np.zeros_like(num)
    nz = den > 0
    out[nz] = num[nz] / den[nz]
    out[~nz & (num > 0)] = np.inf
    return out
