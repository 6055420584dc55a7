"""pathintegralgroundstate_amd -- MI355X (gfx950) drop-in for the PIGS action / energy hot path.

The product is ``libpigs_hip.so`` (hand-written HIP kernels behind the C ABI of
``include/pigs_hip.h``).  This package is only the host-side mirror of the reference's
procedure interface (``UpdateAction``, ``ThermEnergy``, ``LocalEnergy``, ``PotentialEnergy``)
over that C ABI, plus the namelist front end and walker sharding.  There is no CPU fallback:
importing :mod:`pathintegralgroundstate_amd.api` without the built library raises.
"""
from .system import SystemConfig, read_namelists  # noqa: F401

__all__ = ["SystemConfig", "read_namelists"]
__version__ = "0.1.0"
