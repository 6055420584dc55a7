"""Host-side set-up of the hot path: the six namelists of ``vpi.in`` and the derived box,
cutoff and table-grid quantities.

Restates what the reference driver computes before its first MC step
(reference vpi.f90:76-128, vpi_mod.f90:39-75,94; system_mod.f90:15-34), including its
single-precision ``real()`` conversions, so that ``Lbox``, ``rcut`` and ``dr`` are
bit-identical to the reference's.
"""
from __future__ import annotations

import math
import re
from dataclasses import dataclass, field

import numpy as np

# reference defaults: vpi_mod.f90:39-60
_DEFAULTS = {
    "system": {"crystal": False, "trap": False},
    "samp": {"resume": False, "seed": 1982, "lstag": 2, "nlev": 1},
    "obdm": {"swapping": False, "cworm": 0.0, "nobdm": 0, "npw": 0},
    "wavefun": {"nmax": 10000, "wf_table": False, "v_table": False},
    "jastrow": {},
    "extpot": {},
}


def _value(tok):
    t = tok.strip().strip(",")
    tl = t.lower()
    if tl in ("t", ".true.", "true", ".t."):
        return True
    if tl in ("f", ".false.", "false", ".f."):
        return False
    if (t.startswith("'") and t.endswith("'")) or (t.startswith('"') and t.endswith('"')):
        return t[1:-1]
    try:
        return int(t)
    except ValueError:
        return float(tl.replace("d", "e"))


def read_namelists(text):
    """Parse the Fortran namelist groups of a ``vpi.in`` (comments after ``!`` dropped)."""
    groups = {k: dict(v) for k, v in _DEFAULTS.items()}
    body = "\n".join(line.split("!", 1)[0] for line in text.splitlines())
    for m in re.finditer(r"&(\w+)(.*?)(?:^|\s)/", body, flags=re.S | re.M):
        name = m.group(1).lower()
        g = groups.setdefault(name, {})
        for key, val in re.findall(r"(\w+)\s*=\s*([^=]+?)(?=\s*(?:\w+\s*=|$))", m.group(2).strip(), flags=re.S):
            vals = [v for v in re.split(r"[,\s]+", val.strip()) if v]
            parsed = [_value(v) for v in vals]
            g[key.lower()] = parsed[0] if len(parsed) == 1 else parsed
    return groups


def _f32(x):
    return float(np.float32(x))


@dataclass
class SystemConfig:
    """Everything the hot path needs (``pigs_params`` of include/pigs_hip.h)."""
    dim: int = 3
    Np: int = 64
    Nb: int = 32
    Nmax: int = 10000
    density: float = 0.365
    dt: float = 5e-3
    Rm: float = 1.2
    trap: bool = False
    wf_table: bool = True
    v_table: bool = True
    a_ho: list = field(default_factory=lambda: [1.0, 1.0, 1.0])
    Lbox: list = None
    rcut: float = None
    delta_cm: float = 0.12
    seed: int = 1982
    CMFreq: int = 1
    sampling: str = "bis"
    Lstag: int = 2
    Nlev: int = 1
    Nstag: int = 1
    Nblock: int = 1
    Nstep: int = 1
    Nbin: int = 100
    Nk: int = 50
    swapping: bool = False
    CWorm: float = 0.0
    Nobdm: int = 0
    Npw: int = 0

    def __post_init__(self):
        self.a_ho = (list(np.atleast_1d(self.a_ho).astype(float)) + [1.0, 1.0, 1.0])[:3]
        if self.trap:
            # vpi.f90:82-93
            rc = 1.0
            for k in range(self.dim):
                rc = 3.0 * rc * self.a_ho[k]
            vol = math.pi ** (0.5 * self.dim) * rc / math.gamma(0.5 * self.dim + 1.0)
            self.density = _f32(self.Np) / vol
            rc = rc ** (1.0 / _f32(self.dim))
            if self.rcut is None:
                self.rcut = 10.0 * rc
            self.delta_cm_eff = self.delta_cm * min(self.a_ho[:self.dim])
            if self.Lbox is None:
                self.Lbox = [1.0, 1.0, 1.0]
        else:
            if self.Lbox is None:
                L = (_f32(self.Np) / self.density) ** (1.0 / _f32(self.dim))   # vpi.f90:112
                self.Lbox = [L, L, L]
            if self.rcut is None:
                self.rcut = min(0.5 * l for l in self.Lbox[:self.dim])        # vpi.f90:122
            self.delta_cm_eff = self.delta_cm / self.density ** (1.0 / _f32(self.dim))  # :123
        self.Lbox = (list(np.atleast_1d(self.Lbox).astype(float)) + [1.0, 1.0, 1.0])[:3]
        self.rcut2 = self.rcut * self.rcut                                      # vpi.f90:127
        self.dr = self.rcut / _f32(self.Nmax - 1)                               # vpi_mod.f90:94
        self.rbin = self.rcut / _f32(self.Nbin)                                 # vpi.f90:128

    @property
    def M(self):
        """Number of beads (the reference indexes them 0:2*Nb)."""
        return 2 * self.Nb + 1

    @property
    def path_shape(self):
        """numpy C-order shape of one worldline == Fortran Path(dim,Np,0:2*Nb)."""
        return (self.M, self.Np, self.dim)

    @classmethod
    def from_namelists(cls, text, **override):
        g = read_namelists(text)
        s, p, o, w = g["system"], g["samp"], g["obdm"], g["wavefun"]
        kw = dict(dim=s["dim"], Np=s["np"], density=s.get("density", 0.365), trap=s["trap"],
                  dt=p["dt"], Nb=p["nb"], seed=p["seed"], delta_cm=p.get("delta_cm", 0.12),
                  CMFreq=p.get("cmfreq", 1), sampling=p.get("sampling", "bis"), Lstag=p["lstag"],
                  Nlev=p["nlev"], Nstag=p.get("nstag", 1), Nblock=p.get("nblock", 1),
                  Nstep=p.get("nstep", 1), Nbin=p.get("nbin", 100), Nk=p.get("nk", 50),
                  swapping=o["swapping"], CWorm=o["cworm"], Nobdm=o["nobdm"], Npw=o["npw"],
                  Nmax=w["nmax"], wf_table=w["wf_table"], v_table=w["v_table"],
                  Rm=g["jastrow"].get("rm", 1.2))
        if s["trap"]:
            kw["a_ho"] = g["extpot"].get("a_ho", [1.0])
        kw.update(override)
        return cls(**kw)
