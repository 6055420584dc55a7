"""Host-side mirror of the reference's procedure interface over the C ABI (include/pigs_hip.h).

The reference boundary is the Fortran call interface: ``UpdateAction``
(reference vpi_mod.f90:2491), ``PotentialEnergy`` / ``LocalEnergy`` / ``ThermEnergy``
(reference sample_mod.f90:13,154,323).  :class:`PigsContext` exposes the same operations with
the same argument meaning (ip 1-based, ib 0-based, arrays in the reference's column-major
layout, i.e. numpy C-order ``(M, Np, dim)`` for ``Path(dim,Np,0:2*Nb)``), batched over walkers.
All compute happens in ``libpigs_hip.so``; if that library or a GPU is missing every call
raises :class:`PigsError` -- there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from .system import SystemConfig

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpigs_hip.so")

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)

# every symbol include/pigs_hip.h declares
ABI_SYMBOLS = [
    "pigs_ctx_create", "pigs_ctx_destroy", "pigs_last_error", "pigs_abi_version",
    "pigs_device_count", "pigs_sync", "pigs_stream", "pigs_build_tables",
    "pigs_path_upload", "pigs_path_download", "pigs_path_upload_all", "pigs_path_download_all",
    "pigs_delta_action_batch", "pigs_delta_action_batch_dev", "pigs_delta_action_parts",
    "pigs_commit_beads", "pigs_swap_tails", "pigs_potential_energy_slice",
    "pigs_therm_energy_batch", "pigs_local_energy_batch", "pigs_comm_unique_id",
    "pigs_comm_init_rank", "pigs_comm_init_all", "pigs_estimators_allreduce",
    "pigs_set_tuning", "pigs_selftest_fastmath", "pigs_selftest_stream_read", "pigs_selftest_log",
    "pigs_stage_reserve", "pigs_delta_action_staged", "pigs_commit_reserve", "pigs_commit_staged",
    "pigs_sampler_init", "pigs_sampler_seed", "pigs_sampler_set_rng", "pigs_sampler_get_rng", "pigs_sampler_step",
    "pigs_sampler_counters", "pigs_sampler_counters16", "pigs_sampler_get_worm", "pigs_sampler_set_worm",
    "pigs_sampler_events", "pigs_sampler_event_ints", "pigs_sampler_nrho", "pigs_slice_download", "pigs_build_tables_kind", "pigs_structure_batch",
    "pigs_diagonal_estimators", "pigs_diagonal_estimators_begin", "pigs_diagonal_estimators_end",
]


class PigsError(RuntimeError):
    pass


class PigsSweepParams(C.Structure):
    _fields_ = [("Nlev", C.c_int32), ("Nstag", C.c_int32), ("CMFreq", C.c_int32), ("Lstag", C.c_int32),
                ("delta_cm", C.c_double), ("CWorm", C.c_double), ("density", C.c_double), ("rbin", C.c_double),
                ("swapping", C.c_int32), ("Nobdm", C.c_int32), ("Nbin", C.c_int32), ("Npw", C.c_int32),
                ("sampling", C.c_int32), ("reserved", C.c_int32)]


class PigsParams(C.Structure):
    _fields_ = [("dim", C.c_int32), ("Np", C.c_int32), ("Nb", C.c_int32), ("Nmax", C.c_int32),
                ("trap", C.c_int32), ("wf_table", C.c_int32), ("v_table", C.c_int32),
                ("reserved", C.c_int32),
                ("dr", C.c_double), ("rcut2", C.c_double), ("dt", C.c_double), ("Rm", C.c_double),
                ("Lbox", C.c_double * 3), ("a_ho", C.c_double * 3)]


_lib = None


def load_library(path=LIB_PATH):
    """dlopen libpigs_hip.so (built in-tree by pathintegralgroundstate_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("PIGS_LIB", path)           # experiment builds (scripts/: e.g. the -DPIGS_SWEEP_TIMING library)
    if not os.path.exists(path):
        raise PigsError(
            f"{path} is missing: build it with `python -m pathintegralgroundstate_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback for the product path.")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.pigs_last_error.restype = C.c_char_p
    L.pigs_ctx_create.argtypes = [C.POINTER(PigsParams), _dp, _dp, C.c_int32, C.c_int32, C.POINTER(vp)]
    L.pigs_ctx_destroy.argtypes = [vp]
    L.pigs_device_count.argtypes = [_ip]
    L.pigs_sync.argtypes = [vp]
    L.pigs_stream.argtypes = [vp, C.POINTER(vp)]
    L.pigs_build_tables.argtypes = [C.c_int32, C.c_double, C.c_double, _dp, _dp, _dp]
    L.pigs_build_tables_kind.argtypes = [C.c_int32, C.c_int32, C.c_double, C.c_double, _dp, _dp, _dp]
    L.pigs_path_upload.argtypes = [vp, C.c_int32, _dp]
    L.pigs_path_download.argtypes = [vp, C.c_int32, _dp]
    L.pigs_path_upload_all.argtypes = [vp, _dp]
    L.pigs_path_download_all.argtypes = [vp, _dp]
    L.pigs_delta_action_batch.argtypes = [vp, C.c_int64, _ip, _ip, _ip, _dp, _dp, _dp]
    L.pigs_delta_action_parts.argtypes = [vp, C.c_int64, _ip, _ip, _ip, _dp, _dp, _dp]
    L.pigs_delta_action_batch_dev.argtypes = [vp, C.c_int64, vp, vp, vp, vp, vp, vp]
    L.pigs_commit_beads.argtypes = [vp, C.c_int64, _ip, _ip, _ip, _dp]
    L.pigs_swap_tails.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32]
    L.pigs_potential_energy_slice.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, _dp, _dp]
    L.pigs_therm_energy_batch.argtypes = [vp, C.c_int32, _ip, _dp, _dp, _dp]
    L.pigs_local_energy_batch.argtypes = [vp, C.c_int32, _ip, C.c_int32, _dp, _dp, _dp]
    L.pigs_comm_unique_id.argtypes = [C.c_char_p]
    L.pigs_comm_init_rank.argtypes = [vp, C.c_int32, C.c_int32, C.c_char_p]
    L.pigs_comm_init_all.argtypes = [C.POINTER(vp), C.c_int32]
    L.pigs_estimators_allreduce.argtypes = [vp, _dp, C.c_int32]
    L.pigs_stage_reserve.argtypes = [vp, C.c_int64, C.c_int64] + [C.POINTER(_ip)] * 3 + [C.POINTER(_dp)] * 3
    L.pigs_delta_action_staged.argtypes = [vp, C.c_int64]
    L.pigs_commit_reserve.argtypes = [vp, C.c_int64, C.c_int64] + [C.POINTER(_ip)] * 3 + [C.POINTER(_dp)]
    L.pigs_commit_staged.argtypes = [vp, C.c_int64]
    L.pigs_sampler_init.argtypes = [vp, C.POINTER(PigsSweepParams)]
    L.pigs_sampler_seed.argtypes = [vp, C.c_int32, C.c_int32]
    L.pigs_sampler_set_rng.argtypes = [vp, C.c_int32, C.c_int32, _ip]
    L.pigs_sampler_get_rng.argtypes = [vp, C.c_int32, _ip, _ip]
    L.pigs_sampler_step.argtypes = [vp, C.c_int32]
    L.pigs_sampler_counters.argtypes = [vp, C.POINTER(C.c_int64)]
    L.pigs_sampler_counters16.argtypes = [vp, C.POINTER(C.c_int64)]
    L.pigs_sampler_get_worm.argtypes = [vp, _ip, _ip, _dp]
    L.pigs_sampler_set_worm.argtypes = [vp, _ip, _ip, _dp]
    L.pigs_sampler_events.argtypes = [vp, _ip]
    L.pigs_sampler_event_ints.argtypes = [vp, C.POINTER(C.c_int32)]
    L.pigs_sampler_nrho.argtypes = [vp, _dp, _ip]
    L.pigs_slice_download.argtypes = [vp, C.c_int32, _dp]
    L.pigs_structure_batch.argtypes = [vp, C.c_int32, _ip, C.c_int32, C.c_int32, C.c_double, C.c_int32, _dp, _dp]
    L.pigs_diagonal_estimators.argtypes = [vp, C.c_int32, _ip, C.c_int32, C.c_double, C.c_int32, _dp, _dp, _dp]
    L.pigs_diagonal_estimators_begin.argtypes = [vp, C.c_int32, _ip, C.c_int32, C.c_double, C.c_int32, C.c_int32]
    L.pigs_diagonal_estimators_end.argtypes = [vp, _dp, _dp, _dp]
    L.pigs_set_tuning.argtypes = [vp, C.c_char_p, C.c_int32]
    L.pigs_selftest_fastmath.argtypes = [vp, C.c_int32, C.c_int32, C.POINTER(C.c_uint64)]
    L.pigs_selftest_stream_read.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    L.pigs_selftest_log.argtypes = [vp, C.c_int64, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    for name in ABI_SYMBOLS:
        fn = getattr(L, name)
        if name != "pigs_last_error":
            fn.restype = C.c_int
    _lib = L
    return L


def _chk(L, rc, what):
    if rc != 0:
        raise PigsError(f"{what} failed (status {rc}): {L.pigs_last_error().decode(errors='replace')}")


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def _d(a):
    return a.ctypes.data_as(_dp)


def _i(a):
    return a.ctypes.data_as(_ip)


def device_count():
    L = load_library()
    n = C.c_int32(0)
    rc = L.pigs_device_count(C.byref(n))
    return n.value if rc == 0 else 0


POTENTIALS = {"aziz2": 0, "lj": 1, "dipolar": 2}


def build_tables(cfg: SystemConfig, potential="aziz2"):
    """PotentialTable / JastrowTable (reference vpi_mod.f90:84-145) on the host."""
    L = load_library()
    VT = np.zeros(cfg.Nmax + 2)
    WF = np.zeros(cfg.Nmax + 2)
    dr = C.c_double()
    _chk(L, L.pigs_build_tables_kind(POTENTIALS[potential], cfg.Nmax, cfg.Rm, cfg.rcut, _d(VT), _d(WF),
                                     C.byref(dr)), "pigs_build_tables_kind")
    assert dr.value == cfg.dr
    return VT, WF


class PigsContext:
    """W resident walkers on one MI355X + the batched hot-path operations."""

    def __init__(self, cfg: SystemConfig, VTable, LogWF, n_walkers=1, device_id=0):
        self.L = load_library()
        self.cfg = cfg
        self.n_walkers = int(n_walkers)
        p = PigsParams()
        p.dim, p.Np, p.Nb, p.Nmax = cfg.dim, cfg.Np, cfg.Nb, cfg.Nmax
        p.trap, p.wf_table, p.v_table = int(cfg.trap), int(cfg.wf_table), int(cfg.v_table)
        p.dr, p.rcut2, p.dt, p.Rm = cfg.dr, cfg.rcut2, cfg.dt, cfg.Rm
        for k in range(3):
            p.Lbox[k] = cfg.Lbox[k]
            p.a_ho[k] = cfg.a_ho[k]
        self._VT = _f64(VTable)
        # wf_table = F (the reference's default): the trial function is evaluated analytically, LogWF may be None
        self._WF = _f64(LogWF) if LogWF is not None else np.zeros(cfg.Nmax + 2)
        if LogWF is None and cfg.wf_table:
            raise ValueError("wf_table = T needs a LogWF table")
        if self._VT.size != cfg.Nmax + 2 or self._WF.size != cfg.Nmax + 2:
            raise PigsError("tables must hold Nmax+2 doubles (F(0:Nmax+1))")
        h = C.c_void_p()
        _chk(self.L, self.L.pigs_ctx_create(C.byref(p), _d(self._VT), _d(self._WF), self.n_walkers,
                                            int(device_id), C.byref(h)), "pigs_ctx_create")
        self.h = h

    def close(self):
        if getattr(self, "h", None):
            self.L.pigs_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- residency
    def upload(self, walker, Path):
        Path = _f64(Path)
        assert Path.shape == self.cfg.path_shape, (Path.shape, self.cfg.path_shape)
        _chk(self.L, self.L.pigs_path_upload(self.h, int(walker), _d(Path)), "pigs_path_upload")

    def download(self, walker):
        Path = np.empty(self.cfg.path_shape)
        _chk(self.L, self.L.pigs_path_download(self.h, int(walker), _d(Path)), "pigs_path_download")
        return Path

    def upload_all(self, Paths):
        Paths = _f64(Paths)
        assert Paths.shape == (self.n_walkers,) + self.cfg.path_shape
        _chk(self.L, self.L.pigs_path_upload_all(self.h, _d(Paths)), "pigs_path_upload_all")

    def download_all(self):
        Paths = np.empty((self.n_walkers,) + self.cfg.path_shape)
        _chk(self.L, self.L.pigs_path_download_all(self.h, _d(Paths)), "pigs_path_download_all")
        return Paths

    def sync(self):
        _chk(self.L, self.L.pigs_sync(self.h), "pigs_sync")

    def stream(self):
        s = C.c_void_p()
        _chk(self.L, self.L.pigs_stream(self.h, C.byref(s)), "pigs_stream")
        return s.value

    def set_tuning(self, key, value):
        _chk(self.L, self.L.pigs_set_tuning(self.h, key.encode(), int(value)), "pigs_set_tuning")

    def selftest_fastmath(self, blocks=1024, iters=256):
        bad = (C.c_uint64 * 4)()
        _chk(self.L, self.L.pigs_selftest_fastmath(self.h, blocks, iters, bad), "pigs_selftest_fastmath")
        return list(bad), blocks * 256 * iters

    def selftest_log(self, n=1 << 30, seed=0x5eed):
        """(mismatches, one differing argument) of the sampler's device log against this host's libm on n arguments."""
        bad, x = C.c_uint64(), C.c_double()
        _chk(self.L, self.L.pigs_selftest_log(self.h, int(n), int(seed), C.byref(bad), C.byref(x)), "pigs_selftest_log")
        return bad.value, x.value

    def stream_read(self, reps=50):
        """(bytes, seconds) per plain streaming read of the resident worldlines (measurement aid)."""
        b, t = C.c_double(), C.c_double()
        _chk(self.L, self.L.pigs_selftest_stream_read(self.h, int(reps), C.byref(b), C.byref(t)), "pigs_selftest_stream_read")
        return b.value, t.value

    # ---- K1
    def delta_action_batch(self, walker, ip, ib, xnew, xold):
        """DeltaS[i] of moving bead ib[i] of particle ip[i] (1-based) of walker[i] from xold[i]
        to xnew[i]: UpdateAction's output (reference vpi_mod.f90:2491-2530) per item."""
        walker, ip, ib = _i32(walker).ravel(), _i32(ip).ravel(), _i32(ib).ravel()
        n = walker.size
        xnew, xold = _f64(xnew).reshape(n, self.cfg.dim), _f64(xold).reshape(n, self.cfg.dim)
        out = np.empty(n)
        _chk(self.L, self.L.pigs_delta_action_batch(self.h, n, _i(walker), _i(ip), _i(ib), _d(xnew),
                                                    _d(xold), _d(out)), "pigs_delta_action_batch")
        return out

    def delta_action_staged(self, walker, ip, ib, xnew, xold):
        """Same through the pinned staging arrays (the sampler's low-latency path)."""
        walker, ip, ib = _i32(walker).ravel(), _i32(ip).ravel(), _i32(ib).ravel()
        n, d = walker.size, self.cfg.dim
        pw, pp, pb = _ip(), _ip(), _ip()
        pxn, pxo, pds = _dp(), _dp(), _dp()
        _chk(self.L, self.L.pigs_stage_reserve(self.h, max(n, 1), 0, C.byref(pw), C.byref(pp), C.byref(pb),
                                               C.byref(pxn), C.byref(pxo), C.byref(pds)), "pigs_stage_reserve")
        np.ctypeslib.as_array(pw, (max(n, 1),))[:n] = walker
        np.ctypeslib.as_array(pp, (max(n, 1),))[:n] = ip
        np.ctypeslib.as_array(pb, (max(n, 1),))[:n] = ib
        np.ctypeslib.as_array(pxn, (max(n, 1) * d,))[:n * d] = _f64(xnew).ravel()
        np.ctypeslib.as_array(pxo, (max(n, 1) * d,))[:n * d] = _f64(xold).ravel()
        _chk(self.L, self.L.pigs_delta_action_staged(self.h, n), "pigs_delta_action_staged")
        return np.ctypeslib.as_array(pds, (max(n, 1),))[:n].copy()

    def commit_staged(self, walker, ip, ib, x):
        walker, ip, ib = _i32(walker).ravel(), _i32(ip).ravel(), _i32(ib).ravel()
        n, d = walker.size, self.cfg.dim
        pw, pp, pb, px = _ip(), _ip(), _ip(), _dp()
        _chk(self.L, self.L.pigs_commit_reserve(self.h, max(n, 1), 0, C.byref(pw), C.byref(pp), C.byref(pb),
                                                C.byref(px)), "pigs_commit_reserve")
        np.ctypeslib.as_array(pw, (max(n, 1),))[:n] = walker
        np.ctypeslib.as_array(pp, (max(n, 1),))[:n] = ip
        np.ctypeslib.as_array(pb, (max(n, 1),))[:n] = ib
        np.ctypeslib.as_array(px, (max(n, 1) * d,))[:n * d] = _f64(x).ravel()
        _chk(self.L, self.L.pigs_commit_staged(self.h, n), "pigs_commit_staged")
        self.sync()

    def delta_action_parts(self, walker, ip, ib, xnew, xold):
        """(DeltaPot, DeltaF2, DeltaLogPsi) per item: UpdatePot / UpdateWf outputs."""
        walker, ip, ib = _i32(walker).ravel(), _i32(ip).ravel(), _i32(ib).ravel()
        n = walker.size
        xnew, xold = _f64(xnew).reshape(n, self.cfg.dim), _f64(xold).reshape(n, self.cfg.dim)
        out = np.empty((n, 3))
        _chk(self.L, self.L.pigs_delta_action_parts(self.h, n, _i(walker), _i(ip), _i(ib), _d(xnew),
                                                    _d(xold), _d(out)), "pigs_delta_action_parts")
        return out

    def delta_action_batch_dev(self, n, d_walker, d_ip, d_ib, d_xnew, d_xold, d_out):
        """Device-pointer form (ints = raw device addresses), asynchronous on the context stream."""
        _chk(self.L, self.L.pigs_delta_action_batch_dev(self.h, int(n), d_walker, d_ip, d_ib, d_xnew,
                                                        d_xold, d_out), "pigs_delta_action_batch_dev")

    # ---- K6: device-resident sampler
    def sampler_init(self, Nlev=None, Nstag=None, CMFreq=None, Lstag=None, delta_cm=None, CWorm=0.0,
                     swapping=False, Nobdm=0, Nbin=None, Npw=0, sampling="bis"):
        c = self.cfg
        self._nbin = c.Nbin if Nbin is None else Nbin
        self._npw = Npw
        sp = PigsSweepParams(c.Nlev if Nlev is None else Nlev, c.Nstag if Nstag is None else Nstag,
                             c.CMFreq if CMFreq is None else CMFreq, c.Lstag if Lstag is None else Lstag,
                             c.delta_cm_eff if delta_cm is None else delta_cm, CWorm, c.density,
                             c.rcut / float(np.float32(self._nbin)), int(swapping), Nobdm, self._nbin, Npw,
                             {"bis": 0, "sta": 1}[sampling], 0)
        _chk(self.L, self.L.pigs_sampler_init(self.h, C.byref(sp)), "pigs_sampler_init")

    def sampler_seed(self, walker, seed):
        _chk(self.L, self.L.pigs_sampler_seed(self.h, int(walker), int(seed)), "pigs_sampler_seed")

    def sampler_set_rng(self, walker, mti, mt):
        mt = np.ascontiguousarray(mt, np.uint32).view(np.int32)
        _chk(self.L, self.L.pigs_sampler_set_rng(self.h, int(walker), int(mti), _i(mt)), "pigs_sampler_set_rng")

    def sampler_get_rng(self, walker):
        mti = C.c_int32()
        mt = np.zeros(624, np.int32)
        _chk(self.L, self.L.pigs_sampler_get_rng(self.h, int(walker), C.byref(mti), _i(mt)), "pigs_sampler_get_rng")
        return mti.value, mt.view(np.uint32).copy()

    def sampler_step(self, istep):
        _chk(self.L, self.L.pigs_sampler_step(self.h, int(istep)), "pigs_sampler_step")

    def sampler_counters(self):
        acc = np.zeros((self.n_walkers, 4), np.int64)
        _chk(self.L, self.L.pigs_sampler_counters(self.h, acc.ctypes.data_as(C.POINTER(C.c_int64))),
             "pigs_sampler_counters")
        return acc

    def sampler_counters16(self):
        acc = np.zeros((self.n_walkers, 16), np.int64)
        _chk(self.L, self.L.pigs_sampler_counters16(self.h, acc.ctypes.data_as(C.POINTER(C.c_int64))),
             "pigs_sampler_counters16")
        return acc

    def sampler_get_worm(self):
        W, d = self.n_walkers, self.cfg.dim
        isopen, iworm, xend = np.zeros(W, np.int32), np.zeros(W, np.int32), np.zeros((W, 2, d))
        _chk(self.L, self.L.pigs_sampler_get_worm(self.h, _i(isopen), _i(iworm), _d(xend)), "pigs_sampler_get_worm")
        return isopen.astype(bool), iworm, xend

    def sampler_set_worm(self, isopen, iworm, xend):
        isopen, iworm, xend = _i32(isopen), _i32(iworm), _f64(xend)
        _chk(self.L, self.L.pigs_sampler_set_worm(self.h, _i(isopen), _i(iworm), _d(xend)), "pigs_sampler_set_worm")

    def sampler_events(self):
        n = C.c_int32()
        _chk(self.L, self.L.pigs_sampler_event_ints(self.h, C.byref(n)), "pigs_sampler_event_ints")
        ev = np.zeros((self.n_walkers, n.value), np.int32)
        _chk(self.L, self.L.pigs_sampler_events(self.h, _i(ev)), "pigs_sampler_events")
        return ev

    def sampler_nrho(self, reset=None):
        """nrho[w, ibin, l]; reset: None, True (all walkers) or a per-walker mask"""
        out = np.zeros((self.n_walkers, self._nbin, self._npw + 1))
        if reset is None or reset is False:
            keep, mask = None, None
        else:
            keep = np.ones(self.n_walkers, np.int32) if reset is True else _i32(reset)
            mask = _i(keep)
        _chk(self.L, self.L.pigs_sampler_nrho(self.h, _d(out), mask), "pigs_sampler_nrho")
        return out

    def slice_download(self, ib):
        R = np.empty((self.n_walkers, self.cfg.Np, self.cfg.dim))
        _chk(self.L, self.L.pigs_slice_download(self.h, int(ib), _d(R)), "pigs_slice_download")
        return R

    # ---- K5
    def commit_beads(self, walker, ip, ib, x):
        walker, ip, ib = _i32(walker).ravel(), _i32(ip).ravel(), _i32(ib).ravel()
        n = walker.size
        x = _f64(x).reshape(n, self.cfg.dim)
        _chk(self.L, self.L.pigs_commit_beads(self.h, n, _i(walker), _i(ip), _i(ib), _d(x)),
             "pigs_commit_beads")

    def swap_tails(self, walker, iw, ik):
        _chk(self.L, self.L.pigs_swap_tails(self.h, int(walker), int(iw), int(ik)), "pigs_swap_tails")

    # ---- K2..K4
    def potential_energy_slice(self, walker, ib, want_F2=False):
        p, f = C.c_double(), C.c_double()
        _chk(self.L, self.L.pigs_potential_energy_slice(self.h, int(walker), int(ib), int(want_F2),
                                                        C.byref(p), C.byref(f)),
             "pigs_potential_energy_slice")
        return p.value, f.value

    def therm_energy_batch(self, walkers=None):
        w = None if walkers is None else _i32(walkers).ravel()
        n = self.n_walkers if w is None else w.size
        E, Ec, Ep = np.empty(n), np.empty(n), np.empty(n)
        _chk(self.L, self.L.pigs_therm_energy_batch(self.h, n, None if w is None else _i(w), _d(E),
                                                    _d(Ec), _d(Ep)), "pigs_therm_energy_batch")
        return E, Ec, Ep

    def local_energy_batch(self, ib, walkers=None):
        w = None if walkers is None else _i32(walkers).ravel()
        n = self.n_walkers if w is None else w.size
        E, K, P = np.empty(n), np.empty(n), np.empty(n)
        _chk(self.L, self.L.pigs_local_energy_batch(self.h, n, None if w is None else _i(w), int(ib),
                                                    _d(E), _d(K), _d(P)), "pigs_local_energy_batch")
        return E, K, P

    # ---- K7
    def structure_batch(self, ib, Nbin, rbin, Nk, walkers=None):
        """(gr increments (n,Nbin), Sk increments (n,Nk,dim)) of slice ib: PairCorrelation and
        StructureFactor (reference sample_mod.f90:392-473)."""
        w = None if walkers is None else _i32(walkers).ravel()
        n = self.n_walkers if w is None else w.size
        gr = np.empty((n, Nbin))
        Sk = np.empty((n, Nk, self.cfg.dim))
        _chk(self.L, self.L.pigs_structure_batch(self.h, n, None if w is None else _i(w), int(ib), int(Nbin),
                                                 float(rbin), int(Nk), _d(gr), _d(Sk)), "pigs_structure_batch")
        return gr, Sk

    def diagonal_estimators(self, Nbin=0, rbin=0.0, Nk=0, walkers=None, structure=True):
        """All estimators of a diagonal MC step (vpi.f90:443-469) in one call: returns a dict with
        E1,K1,V1 (LocalEnergy slice 0), E2,K2,V2 (slice 2Nb), Et,Kt,Vt (ThermEnergy) -- arrays of n -- and, for PBC
        runs with structure=True, gr (n,Nbin), Sk (n,Nk,dim)."""
        w = None if walkers is None else _i32(walkers).ravel()
        n = self.n_walkers if w is None else w.size
        en = np.empty((n, 9))
        gr = Sk = None
        if structure and not self.cfg.trap:
            gr = np.empty((n, Nbin))
            Sk = np.empty((n, Nk, self.cfg.dim))
        _chk(self.L, self.L.pigs_diagonal_estimators(self.h, n, None if w is None else _i(w), int(Nbin), float(rbin), int(Nk),
                                                     _d(en), None if gr is None else _d(gr), None if Sk is None else _d(Sk)),
             "pigs_diagonal_estimators")
        out = {k: en[:, i] for i, k in enumerate(("E1", "K1", "V1", "E2", "K2", "V2", "Et", "Kt", "Vt"))}
        out["gr"], out["Sk"] = gr, Sk
        return out

    def diagonal_estimators_begin(self, Nbin=0, rbin=0.0, Nk=0, walkers=None, structure=True):
        """Asynchronous form: snapshot of the worldlines + estimator kernels on the context's second stream; go on (e.g.
        with the next sampler_step) and collect with diagonal_estimators_end()."""
        w = None if walkers is None else _i32(walkers).ravel()
        n = self.n_walkers if w is None else w.size
        st = bool(structure and not self.cfg.trap)
        self._est_pending = (n, int(Nbin), int(Nk), st)
        _chk(self.L, self.L.pigs_diagonal_estimators_begin(self.h, n, None if w is None else _i(w), int(Nbin), float(rbin),
                                                           int(Nk), int(st)), "pigs_diagonal_estimators_begin")

    def diagonal_estimators_end(self):
        n, Nbin, Nk, st = self._est_pending
        en = np.empty((n, 9))
        gr = np.empty((n, Nbin)) if st else None
        Sk = np.empty((n, Nk, self.cfg.dim)) if st else None
        _chk(self.L, self.L.pigs_diagonal_estimators_end(self.h, _d(en), None if gr is None else _d(gr),
                                                         None if Sk is None else _d(Sk)), "pigs_diagonal_estimators_end")
        out = {k: en[:, i] for i, k in enumerate(("E1", "K1", "V1", "E2", "K2", "V2", "Et", "Kt", "Vt"))}
        out["gr"], out["Sk"] = gr, Sk
        return out

    # ---- multi-GPU
    def comm_init_rank(self, nranks, rank, unique_id: bytes):
        _chk(self.L, self.L.pigs_comm_init_rank(self.h, nranks, rank, unique_id), "pigs_comm_init_rank")

    def estimators_allreduce(self, vec):
        vec = _f64(vec).copy()
        _chk(self.L, self.L.pigs_estimators_allreduce(self.h, _d(vec), vec.size),
             "pigs_estimators_allreduce")
        return vec

    # ---- reference-named single-call mirrors (argument meaning as in the reference) ----------
    def UpdateAction(self, walker, ip, ib, xnew, xold):
        """reference vpi_mod.f90:2491: DeltaS for one bead move."""
        return float(self.delta_action_batch([walker], [ip], [ib], [xnew], [xold])[0])

    def PotentialEnergy(self, walker, ib, want_F2=False):
        """reference sample_mod.f90:13: (Pot, F2) of slice ib."""
        return self.potential_energy_slice(walker, ib, want_F2)

    def LocalEnergy(self, walker, ib):
        """reference sample_mod.f90:154: (E, Kin, Pot) of slice ib."""
        E, K, P = self.local_energy_batch(ib, [walker])
        return float(E[0]), float(K[0]), float(P[0])

    def ThermEnergy(self, walker):
        """reference sample_mod.f90:323: (E, Ec, Ep) of one worldline."""
        E, Ec, Ep = self.therm_energy_batch([walker])
        return float(E[0]), float(Ec[0]), float(Ep[0])


def comm_unique_id():
    L = load_library()
    buf = C.create_string_buffer(128)
    _chk(L, L.pigs_comm_unique_id(buf), "pigs_comm_unique_id")
    return buf.raw
