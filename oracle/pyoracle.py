"""ctypes bindings to the CPU oracle -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Two libraries:
  * ``Oracle`` : oracle/libpigs_oracle.so, our scalar C restatement (pigs_oracle.c).
  * ``Ref``    : oracle/_ref/libvpiref.so, the unmodified reference Fortran compiled by
                 oracle/Makefile plus our bind(C) probe (ref_probe.f90).  Exists only
                 where it was built (the build container: .gpurunignore keeps it off the
                 GPU box -- the fixtures it generated travel instead).  ``Ref.available()``
                 says whether it is there.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
Array conventions are the reference's: Fortran column-major, ip 1-based, ib 0-based.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libpigs_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libvpiref.so")
REF_VPI = os.path.join(HERE, "_ref", "vpi")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def build_oracle(force=False):
    """Compile oracle/libpigs_oracle.so (and oracle/_ref when /root/reference exists)."""
    if force or not os.path.exists(ORACLE_SO) or \
            os.path.getmtime(ORACLE_SO) < os.path.getmtime(os.path.join(HERE, "pigs_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", HERE, "libpigs_oracle.so"])
    if os.path.exists("/root/reference/vpi_mod.f90"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref"])


class PoSys(C.Structure):
    _fields_ = [("dim", C.c_int32), ("Np", C.c_int32), ("Nb", C.c_int32), ("Nmax", C.c_int32),
                ("trap", C.c_int32), ("wf_table", C.c_int32), ("v_table", C.c_int32),
                ("pad_", C.c_int32),
                ("dr", C.c_double), ("rcut2", C.c_double), ("Rm", C.c_double),
                ("Lbox", C.c_double * 3), ("LboxHalf", C.c_double * 3), ("a_ho", C.c_double * 3)]


class PoRng(C.Structure):
    _fields_ = [("mti", C.c_int32), ("mt", C.c_uint32 * 624)]


class System:
    """Plain description of one physical set-up (what vpi.f90:76-153 derives from vpi.in)."""

    def __init__(self, dim=3, Np=64, Nb=40, Nmax=10000, density=0.365, Rm=1.2, dt=5e-3,
                 trap=False, a_ho=None, Lbox=None, rcut=None, wf_table=True, v_table=True,
                 Nbin=100, Npw=0, CWorm=0.5):
        self.dim, self.Np, self.Nb, self.Nmax = int(dim), int(Np), int(Nb), int(Nmax)
        self.density, self.Rm, self.dt = float(density), float(Rm), float(dt)
        self.trap = bool(trap)
        self.wf_table, self.v_table = bool(wf_table), bool(v_table)
        self.Nbin, self.Npw, self.CWorm = int(Nbin), int(Npw), float(CWorm)
        self.a_ho = np.ones(3) if a_ho is None else np.resize(f64(a_ho), 3).copy()
        if Lbox is None:
            # vpi.f90:112 (single-precision real(Np), real(dim))
            L = (float(np.float32(self.Np)) / self.density) ** (1.0 / float(np.float32(self.dim)))
            Lbox = [L] * 3
        self.Lbox = np.resize(f64(Lbox), 3).copy()
        if rcut is None:
            if self.trap:
                # vpi.f90:84-92
                rc = 1.0
                for k in range(self.dim):
                    rc = 3.0 * rc * self.a_ho[k]
                rc = rc ** (1.0 / float(np.float32(self.dim)))
                rcut = 10.0 * rc
            else:
                rcut = float(np.min(0.5 * self.Lbox[:self.dim]))      # vpi.f90:122
        self.rcut = float(rcut)
        self.rcut2 = self.rcut * self.rcut
        self.dr = self.rcut / float(np.float32(self.Nmax - 1))        # vpi_mod.f90:94
        self.rbin = self.rcut / float(np.float32(self.Nbin))          # vpi.f90:128

    @property
    def M(self):
        return 2 * self.Nb + 1

    def posys(self):
        s = PoSys()
        s.dim, s.Np, s.Nb, s.Nmax = self.dim, self.Np, self.Nb, self.Nmax
        s.trap, s.wf_table, s.v_table = int(self.trap), int(self.wf_table), int(self.v_table)
        s.dr, s.rcut2, s.Rm = self.dr, self.rcut2, self.Rm
        for k in range(3):
            s.Lbox[k] = self.Lbox[k]
            s.LboxHalf[k] = 0.5 * self.Lbox[k]
            s.a_ho[k] = self.a_ho[k]
        return s


class Oracle:
    """Our C restatement."""

    def __init__(self, path=ORACLE_SO):
        if not os.path.exists(path):
            build_oracle()
        L = self.L = C.CDLL(path)
        L.po_interpolate.restype = C.c_double
        L.po_interpolate.argtypes = [C.c_int, C.c_int, C.c_double, _dp, C.c_double]
        L.po_green_function.restype = C.c_double
        L.po_green_function.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.po_potential.restype = C.c_double
        L.po_potential.argtypes = [C.c_double]
        L.po_logpsi.restype = C.c_double
        L.po_logpsi.argtypes = [C.c_int, C.c_double, C.c_double]
        L.po_table_dr.restype = C.c_double
        L.po_table_dr.argtypes = [C.c_double, C.c_int]
        L.po_box_length.restype = C.c_double
        L.po_box_length.argtypes = [C.c_int, C.c_int, C.c_double]
        L.po_potential_table.argtypes = [C.c_int, C.c_double, _dp]
        L.po_jastrow_table.argtypes = [C.c_int, C.c_double, C.c_double, _dp]
        L.po_minimum_image.argtypes = [C.POINTER(PoSys), _dp, _dp]
        L.po_update_pot.argtypes = [C.POINTER(PoSys), _dp, C.c_int, _dp, _dp, _dp, _dp, _dp]
        L.po_update_wf.argtypes = [C.POINTER(PoSys), _dp, C.c_int, _dp, _dp, _dp, _dp]
        L.po_update_action.argtypes = [C.POINTER(PoSys), _dp, _dp, _dp, C.c_int, C.c_int,
                                       _dp, _dp, C.c_double, _dp]
        L.po_potential_energy.argtypes = [C.POINTER(PoSys), _dp, _dp, _dp, _dp]
        L.po_local_energy.argtypes = [C.POINTER(PoSys), _dp, _dp, _dp, _dp, _dp, _dp]
        L.po_therm_energy.argtypes = [C.POINTER(PoSys), _dp, _dp, C.c_double, _dp, _dp, _dp]
        L.po_pair_correlation.argtypes = [C.POINTER(PoSys), C.c_int, C.c_double, _dp, _dp]
        L.po_structure_factor.argtypes = [C.POINTER(PoSys), C.c_int, _dp, _dp]
        L.po_obdm.argtypes = [C.POINTER(PoSys), C.c_int, C.c_int, C.c_double, _dp, _dp]
        L.po_delta_action_batch.restype = C.c_int64
        L.po_delta_action_batch.argtypes = [C.POINTER(PoSys), _dp, _dp, _dp, C.c_int64,
                                            _ip, _ip, _ip, _dp, _dp, C.c_double, _dp]
        L.po_sgrnd.argtypes = [C.POINTER(PoRng), C.c_int32]
        L.po_grnd.restype = C.c_double
        L.po_grnd.argtypes = [C.POINTER(PoRng)]
        L.po_rangauss.argtypes = [C.POINTER(PoRng), C.c_double, C.c_double, _dp, _dp]

    # -- primitives
    def interpolate(self, opt, N, dx, F, x):
        F = f64(F)
        return self.L.po_interpolate(opt, N, dx, _d(F), x)

    def green_function(self, opt, ib, Nb, dt, Pot, F2):
        return self.L.po_green_function(opt, ib, Nb, dt, Pot, F2)

    def minimum_image(self, sys, xij):
        x = f64(xij).copy()
        r2 = C.c_double()
        s = sys.posys()
        self.L.po_minimum_image(C.byref(s), _d(x), C.byref(r2))
        return x, r2.value

    def tables(self, sys):
        VT = np.zeros(sys.Nmax + 2)
        WF = np.zeros(sys.Nmax + 2)
        with np.errstate(all="ignore"):
            self.L.po_potential_table(sys.Nmax, sys.rcut, _d(VT))
            self.L.po_jastrow_table(sys.Nmax, sys.Rm, sys.rcut, _d(WF))
        return VT, WF

    # -- hot path
    def update_pot(self, sys, VT, ip, R, xnew, xold, want_f2):
        s = sys.posys()
        dp, df = C.c_double(), C.c_double()
        R, xnew, xold, VT = f64(R), f64(xnew), f64(xold), f64(VT)
        self.L.po_update_pot(C.byref(s), _d(VT), ip, _d(R), _d(xnew), _d(xold), C.byref(dp),
                             C.byref(df) if want_f2 else None)
        return dp.value, (df.value if want_f2 else 0.0)

    def update_wf(self, sys, WF, ip, R, xnew, xold):
        s = sys.posys()
        d = C.c_double()
        R, xnew, xold, WF = f64(R), f64(xnew), f64(xold), f64(WF)
        self.L.po_update_wf(C.byref(s), _d(WF), ip, _d(R), _d(xnew), _d(xold), C.byref(d))
        return d.value

    def update_action(self, sys, WF, VT, Path, ip, ib, xnew, xold, dt=None):
        s = sys.posys()
        d = C.c_double()
        Path, xnew, xold, VT, WF = f64(Path), f64(xnew), f64(xold), f64(VT), f64(WF)
        self.L.po_update_action(C.byref(s), _d(WF), _d(VT), _d(Path), ip, ib, _d(xnew), _d(xold),
                                sys.dt if dt is None else dt, C.byref(d))
        return d.value

    def delta_action_batch(self, sys, WF, VT, Paths, walker, ip, ib, xnew, xold, dt=None):
        """Paths: (W, M, Np, dim) C-order == W x Path(dim,Np,0:2Nb); xnew/xold: (n, dim)."""
        s = sys.posys()
        Paths, xnew, xold, VT, WF = f64(Paths), f64(xnew), f64(xold), f64(VT), f64(WF)
        walker = np.ascontiguousarray(walker, np.int32)
        ip = np.ascontiguousarray(ip, np.int32)
        ib = np.ascontiguousarray(ib, np.int32)
        n = walker.size
        out = np.empty(n)
        self.L.po_delta_action_batch(C.byref(s), _d(WF), _d(VT), _d(Paths), n, _i(walker), _i(ip),
                                     _i(ib), _d(xnew), _d(xold),
                                     sys.dt if dt is None else dt, _d(out))
        return out

    def potential_energy(self, sys, VT, R, want_f2):
        s = sys.posys()
        p, f = C.c_double(), C.c_double()
        R, VT = f64(R), f64(VT)
        self.L.po_potential_energy(C.byref(s), _d(VT), _d(R), C.byref(p),
                                   C.byref(f) if want_f2 else None)
        return p.value, (f.value if want_f2 else 0.0)

    def local_energy(self, sys, WF, VT, R):
        s = sys.posys()
        e, k, p = C.c_double(), C.c_double(), C.c_double()
        R, VT, WF = f64(R), f64(VT), f64(WF)
        self.L.po_local_energy(C.byref(s), _d(WF), _d(VT), _d(R), C.byref(e), C.byref(k), C.byref(p))
        return e.value, k.value, p.value

    def therm_energy(self, sys, VT, Path, dt=None):
        s = sys.posys()
        e, k, p = C.c_double(), C.c_double(), C.c_double()
        Path, VT = f64(Path), f64(VT)
        self.L.po_therm_energy(C.byref(s), _d(VT), _d(Path), sys.dt if dt is None else dt,
                               C.byref(e), C.byref(k), C.byref(p))
        return e.value, k.value, p.value

    def pair_correlation(self, sys, R):
        s = sys.posys()
        gr = np.zeros(sys.Nbin)
        R = f64(R)
        self.L.po_pair_correlation(C.byref(s), sys.Nbin, sys.rbin, _d(R), _d(gr))
        return gr

    def structure_factor(self, sys, Nk, R):
        s = sys.posys()
        Sk = np.zeros((Nk, sys.dim))
        R = f64(R)
        self.L.po_structure_factor(C.byref(s), Nk, _d(R), _d(Sk))
        return Sk

    def obdm(self, sys, xend):
        s = sys.posys()
        nrho = np.zeros((sys.Nbin, sys.Npw + 1))
        xend = f64(xend)
        self.L.po_obdm(C.byref(s), sys.Nbin, sys.Npw, sys.rbin, _d(xend), _d(nrho))
        return nrho

    # -- RNG
    def rng(self, seed):
        g = PoRng()
        self.L.po_sgrnd(C.byref(g), seed)
        return g

    def grnd(self, g):
        return self.L.po_grnd(C.byref(g))

    def rangauss(self, g, sigma, mu):
        a, b = C.c_double(), C.c_double()
        self.L.po_rangauss(C.byref(g), sigma, mu, C.byref(a), C.byref(b))
        return a.value, b.value

    def init_path(self, sys, seed):
        """vpi_mod.f90:189,232-248: uniform random R, every bead of a particle at the same point."""
        g = self.rng(seed)
        R = np.empty((sys.Np, sys.dim))
        for ip in range(sys.Np):
            for k in range(sys.dim):
                if sys.trap:
                    R[ip, k] = 2.0 * sys.a_ho[k] * (self.grnd(g) - 0.5)
                else:
                    R[ip, k] = sys.Lbox[k] * (self.grnd(g) - 0.5)
        Path = np.broadcast_to(R, (sys.M, sys.Np, sys.dim)).copy()
        return Path, g


class Ref:
    """The unmodified reference, through ref_probe.f90.  Holds GLOBAL module state: one System
    at a time (set_system)."""

    @staticmethod
    def available():
        return os.path.exists(REF_SO)

    def __init__(self, path=REF_SO):
        L = self.L = C.CDLL(path)
        self.sys = None
        L.ref_box_length.restype = C.c_double
        L.ref_box_length.argtypes = [C.c_int, C.c_int, C.c_double]
        L.ref_get_dr.restype = C.c_double
        L.ref_set_dr.argtypes = [C.c_double]
        L.ref_set_globals.argtypes = [C.c_int] * 4 + [_dp, C.c_double, C.c_double, _dp] + \
            [C.c_int] * 4 + [C.c_double]
        L.ref_jastrow_table.argtypes = [C.c_double, _dp]
        L.ref_potential_table.argtypes = [C.c_double, _dp]
        L.ref_potential.restype = C.c_double
        L.ref_potential.argtypes = [C.c_double]
        L.ref_logpsi.restype = C.c_double
        L.ref_logpsi.argtypes = [C.c_int, C.c_double, C.c_double]
        L.ref_interpolate.restype = C.c_double
        L.ref_interpolate.argtypes = [C.c_int, C.c_int, C.c_double, _dp, C.c_double]
        L.ref_minimum_image.argtypes = [_dp, _dp]
        L.ref_green_function.restype = C.c_double
        L.ref_green_function.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double]
        L.ref_update_action.argtypes = [C.c_int, _dp, _dp, _dp, C.c_int, C.c_int, _dp, _dp,
                                        C.c_double, _dp]
        L.ref_update_pot.argtypes = [C.c_int, _dp, C.c_int, _dp, _dp, _dp, _dp, _dp, C.c_int]
        L.ref_update_wf.argtypes = [C.c_int, _dp, C.c_int, _dp, _dp, _dp, _dp]
        L.ref_potential_energy.argtypes = [C.c_int, _dp, _dp, _dp, _dp, C.c_int]
        L.ref_local_energy.argtypes = [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp]
        L.ref_therm_energy.argtypes = [C.c_int, _dp, _dp, C.c_double, _dp, _dp, _dp]
        L.ref_pair_correlation.argtypes = [_dp, _dp]
        L.ref_structure_factor.argtypes = [C.c_int, _dp, _dp]
        L.ref_obdm.argtypes = [_dp, _dp]
        L.ref_sgrnd.argtypes = [C.c_int]
        L.ref_grnd.restype = C.c_double
        L.ref_rangauss.argtypes = [C.c_double, C.c_double, _dp, _dp]
        L.ref_rng_get_state.argtypes = [_ip, _ip]
        L.ref_rng_set_state.argtypes = [C.c_int, _ip]
        L.ref_init.argtypes = [C.c_int, C.c_int, _dp, _dp]
        L.ref_translate_chain.argtypes = [C.c_int, C.c_double, _dp, _dp, C.c_double, C.c_int, _dp, _ip]
        L.ref_diag_move.argtypes = [C.c_int, C.c_int, _dp, _dp, C.c_double, C.c_int, C.c_int, _dp, _ip]
        L.ref_half_move.argtypes = [C.c_int, C.c_int, C.c_int, C.c_double, _dp, _dp, C.c_double,
                                    C.c_int, C.c_int, _dp, _dp, _ip]
        L.ref_open_chain.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_double, C.c_int, C.c_int,
                                     _dp, _dp, _ip, _ip]
        L.ref_close_chain.argtypes = L.ref_open_chain.argtypes
        L.ref_swap.argtypes = [C.c_int, _dp, _dp, C.c_double, C.c_int, _ip, _dp, _dp, _ip, _ip, _ip]

    def set_system(self, sys):
        Lb = f64(sys.Lbox[:sys.dim])
        ah = f64(sys.a_ho[:sys.dim])
        self.L.ref_set_globals(sys.dim, sys.Np, sys.Nb, sys.Nmax, _d(Lb), sys.rcut, sys.Rm, _d(ah),
                               int(sys.wf_table), int(sys.v_table), sys.Nbin, sys.Npw, sys.CWorm)
        self.L.ref_set_dr(sys.dr)
        self.sys = sys

    def tables(self, sys):
        """Tables exactly as the reference builds them (vpi_mod.f90:84-145); the builders also
        write jastrow.out / potential.out into the cwd, so run them in a scratch directory."""
        self.set_system(sys)
        VT = np.zeros(sys.Nmax + 2)
        WF = np.zeros(sys.Nmax + 2)
        cwd = os.getcwd()
        with tempfile.TemporaryDirectory() as td:
            os.chdir(td)
            try:
                self.L.ref_potential_table(sys.rcut, _d(VT))
                self.L.ref_jastrow_table(sys.rcut, _d(WF))
            finally:
                os.chdir(cwd)
        assert self.L.ref_get_dr() == sys.dr, (self.L.ref_get_dr(), sys.dr)
        return VT, WF

    def box_length(self, Np, dim, density):
        return self.L.ref_box_length(Np, dim, density)

    def interpolate(self, opt, N, dx, F, x):
        F = f64(F)
        return self.L.ref_interpolate(opt, N, dx, _d(F), x)

    def green_function(self, opt, ib, dt, Pot, F2):
        return self.L.ref_green_function(opt, ib, dt, Pot, F2)

    def minimum_image(self, xij):
        x = f64(xij).copy()
        r2 = C.c_double()
        self.L.ref_minimum_image(_d(x), C.byref(r2))
        return x, r2.value

    def update_pot(self, VT, ip, R, xnew, xold, want_f2):
        dp, df = C.c_double(), C.c_double()
        R, xnew, xold, VT = f64(R), f64(xnew), f64(xold), f64(VT)
        self.L.ref_update_pot(int(self.sys.trap), _d(VT), ip, _d(R), _d(xnew), _d(xold),
                              C.byref(dp), C.byref(df), int(want_f2))
        return dp.value, df.value

    def update_wf(self, WF, ip, R, xnew, xold):
        d = C.c_double()
        R, xnew, xold, WF = f64(R), f64(xnew), f64(xold), f64(WF)
        self.L.ref_update_wf(int(self.sys.trap), _d(WF), ip, _d(R), _d(xnew), _d(xold), C.byref(d))
        return d.value

    def update_action(self, WF, VT, Path, ip, ib, xnew, xold, dt=None):
        d = C.c_double()
        Path, xnew, xold, VT, WF = f64(Path), f64(xnew), f64(xold), f64(VT), f64(WF)
        self.L.ref_update_action(int(self.sys.trap), _d(WF), _d(VT), _d(Path), ip, ib, _d(xnew),
                                 _d(xold), self.sys.dt if dt is None else dt, C.byref(d))
        return d.value

    def potential_energy(self, VT, R, want_f2):
        p, f = C.c_double(), C.c_double()
        R, VT = f64(R), f64(VT)
        self.L.ref_potential_energy(int(self.sys.trap), _d(VT), _d(R), C.byref(p), C.byref(f),
                                    int(want_f2))
        return p.value, f.value

    def local_energy(self, WF, VT, R):
        e, k, p = C.c_double(), C.c_double(), C.c_double()
        R, VT, WF = f64(R), f64(VT), f64(WF)
        self.L.ref_local_energy(int(self.sys.trap), _d(WF), _d(VT), _d(R), C.byref(e), C.byref(k),
                                C.byref(p))
        return e.value, k.value, p.value

    def therm_energy(self, VT, Path, dt=None):
        e, k, p = C.c_double(), C.c_double(), C.c_double()
        Path, VT = f64(Path), f64(VT)
        self.L.ref_therm_energy(int(self.sys.trap), _d(VT), _d(Path),
                                self.sys.dt if dt is None else dt, C.byref(e), C.byref(k), C.byref(p))
        return e.value, k.value, p.value

    def pair_correlation(self, R):
        gr = np.zeros(self.sys.Nbin)
        R = f64(R)
        self.L.ref_pair_correlation(_d(R), _d(gr))
        return gr

    def structure_factor(self, Nk, R):
        Sk = np.zeros((Nk, self.sys.dim))
        R = f64(R)
        self.L.ref_structure_factor(Nk, _d(R), _d(Sk))
        return Sk

    def obdm(self, xend):
        nrho = np.zeros((self.sys.Nbin, self.sys.Npw + 1))
        xend = f64(xend)
        self.L.ref_obdm(_d(xend), _d(nrho))
        return nrho

    # RNG / init
    def sgrnd(self, seed):
        self.L.ref_sgrnd(seed)

    def grnd(self):
        return self.L.ref_grnd()

    def rangauss(self, sigma, mu):
        a, b = C.c_double(), C.c_double()
        self.L.ref_rangauss(sigma, mu, C.byref(a), C.byref(b))
        return a.value, b.value

    def rng_get_state(self):
        mti = C.c_int32()
        mt = np.zeros(624, np.int32)
        self.L.ref_rng_get_state(C.byref(mti), _i(mt))
        return mti.value, mt.view(np.uint32).copy()

    def rng_set_state(self, mti, mt):
        mt = np.ascontiguousarray(mt, np.uint32).view(np.int32)
        self.L.ref_rng_set_state(mti, _i(mt))

    def init(self, seed):
        s = self.sys
        Path = np.zeros((s.M, s.Np, s.dim))
        xend = np.zeros((2, s.dim))
        self.L.ref_init(int(s.trap), seed, _d(Path), _d(xend))
        return Path, xend

    # move set (callers of the hot path)
    def translate_chain(self, delta, WF, VT, ip, Path, acc=0, dt=None):
        a = C.c_int32(acc)
        self.L.ref_translate_chain(int(self.sys.trap), delta, _d(WF), _d(VT),
                                   self.sys.dt if dt is None else dt, ip, _d(Path), C.byref(a))
        return a.value

    DIAG = {"Staging": 1, "MoveHead": 2, "MoveTail": 3, "Bisection": 4,
            "MoveHeadBisection": 5, "MoveTailBisection": 6}
    HALF = {"TranslateHalfChain": 1, "StagingHalfChain": 2, "MoveHeadHalfChain": 3,
            "MoveTailHalfChain": 4}

    def diag_move(self, name, WF, VT, par, ip, Path, acc=0, dt=None):
        a = C.c_int32(acc)
        self.L.ref_diag_move(self.DIAG[name], int(self.sys.trap), _d(WF), _d(VT),
                             self.sys.dt if dt is None else dt, par, ip, _d(Path), C.byref(a))
        return a.value

    def half_move(self, name, half, delta, WF, VT, Lstag, ip, Path, xend, acc=0, dt=None):
        a = C.c_int32(acc)
        self.L.ref_half_move(self.HALF[name], int(self.sys.trap), half, delta, _d(WF), _d(VT),
                             self.sys.dt if dt is None else dt, Lstag, ip, _d(Path), _d(xend),
                             C.byref(a))
        return a.value

    def open_chain(self, WF, VT, Lstag, ip, Path, xend, isopen, acc=0, dt=None):
        a, o = C.c_int32(acc), C.c_int32(int(isopen))
        self.L.ref_open_chain(int(self.sys.trap), _d(WF), _d(VT), self.sys.density,
                              self.sys.dt if dt is None else dt, Lstag, ip, _d(Path), _d(xend),
                              C.byref(o), C.byref(a))
        return bool(o.value), a.value

    def close_chain(self, WF, VT, Lstag, ip, Path, xend, isopen, acc=0, dt=None):
        a, o = C.c_int32(acc), C.c_int32(int(isopen))
        self.L.ref_close_chain(int(self.sys.trap), _d(WF), _d(VT), self.sys.density,
                               self.sys.dt if dt is None else dt, Lstag, ip, _d(Path), _d(xend),
                               C.byref(o), C.byref(a))
        return bool(o.value), a.value

    def swap(self, WF, VT, Lstag, iw, Path, xend, acc=0, dt=None):
        a, w, k, sw = C.c_int32(acc), C.c_int32(iw), C.c_int32(0), C.c_int32(0)
        self.L.ref_swap(int(self.sys.trap), _d(WF), _d(VT), self.sys.dt if dt is None else dt,
                        Lstag, C.byref(w), _d(Path), _d(xend), C.byref(a), C.byref(k), C.byref(sw))
        return w.value, k.value, bool(sw.value), a.value
