import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_golden(name):
    # plain arrays only: numpy.load's default allow_pickle=False
    return dict(np.load(os.path.join(GOLDEN, name + ".npz")))


@pytest.fixture(scope="session")
def oracle():
    from oracle.pyoracle import Oracle, build_oracle
    build_oracle()
    return Oracle()


@pytest.fixture(scope="session")
def ref():
    from oracle.pyoracle import Ref
    if not Ref.available():
        pytest.skip("oracle/_ref/libvpiref.so not built here (needs /root/reference)")
    return Ref()


def system_from_golden(d, **over):
    """oracle.pyoracle.System described by a fixture's metadata."""
    from oracle.pyoracle import System
    kw = dict(dim=int(d["dim"]), Np=int(d["Np"]), Nb=int(d["Nb"]), Nmax=int(d["Nmax"]),
              density=float(d["density"]), Rm=float(d["Rm"]), dt=float(d["dt"]),
              trap=bool(int(d["trap"])), a_ho=d["a_ho"], Lbox=d["Lbox"], rcut=float(d["rcut"]))
    kw.update(over)
    S = System(**kw)
    assert S.dr == float(d["dr"])
    return S


def config_from_golden(d, **over):
    """pathintegralgroundstate_amd.SystemConfig described by a fixture's metadata."""
    from pathintegralgroundstate_amd import SystemConfig
    kw = dict(dim=int(d["dim"]), Np=int(d["Np"]), Nb=int(d["Nb"]), Nmax=int(d["Nmax"]),
              density=float(d["density"]), Rm=float(d["Rm"]), dt=float(d["dt"]),
              trap=bool(int(d["trap"])), a_ho=list(d["a_ho"]), Lbox=list(d["Lbox"]),
              rcut=float(d["rcut"]))
    kw.update(over)
    c = SystemConfig(**kw)
    assert c.dr == float(d["dr"])
    return c


@pytest.fixture(scope="session")
def gpu_lib():
    """The built HIP library on a machine with a GPU -- fails (not skips) if it is missing."""
    from pathintegralgroundstate_amd import api
    api.load_library()
    assert api.device_count() >= 1, "no HIP device visible: -m gpu tests need an MI355X"
    return api
